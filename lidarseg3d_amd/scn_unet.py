"""UNetSCN3D — the sparse-conv UNet voxel encoder (det3d/models/backbones/scn_unet.py:72-249), same registry
name, constructor signature, attribute names (=> state_dict keys) and batch_dict contract; forward runs on the
libls3d gather-GEMM with BatchNorm(eval)/ReLU/residual/channel-reduction fused into each conv's epilogue."""
from functools import partial

import numpy as np
import torch
from torch import nn

from . import ops
from . import spconv
from .registry import BACKBONES
from .spconv import conv_bn_act


def post_act_block(in_channels, out_channels, kernel_size, indice_key=None, stride=1, padding=0, conv_type="subm",
                   norm_fn=None):
    """(conv, BN, ReLU) triple — scn_unet.py:11-30"""
    if conv_type == "subm":
        conv = spconv.SubMConv3d(in_channels, out_channels, kernel_size, bias=False, indice_key=indice_key)
    elif conv_type == "spconv":
        conv = spconv.SparseConv3d(in_channels, out_channels, kernel_size, stride=stride, padding=padding, bias=False,
                                   indice_key=indice_key)
    elif conv_type == "inverseconv":
        conv = spconv.SparseInverseConv3d(in_channels, out_channels, kernel_size, indice_key=indice_key, bias=False)
    else:
        raise NotImplementedError
    return spconv.SparseSequential(conv, norm_fn(out_channels), nn.ReLU())


class SparseBasicBlock(spconv.SparseModule):
    """scn_unet.py:34-69: relu(bn2(conv2(relu(bn1(conv1(x))))) + x), two kernel launches.  bias=True: the biased variant of
    det3d/models/backbones/scn.py:37-80 (SpMiddleResNetFHD; the bias folds into the epilogue's shift)."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, indice_key=None, norm_fn=None, bias=False):
        super().__init__()
        self.conv1 = spconv.SubMConv3d(inplanes, planes, kernel_size=3, stride=stride, padding=1, bias=bias,
                                       indice_key=indice_key)
        self.bn1 = norm_fn(planes)
        self.relu = nn.ReLU()
        self.conv2 = spconv.SubMConv3d(planes, planes, kernel_size=3, stride=1, padding=1, bias=bias,
                                       indice_key=indice_key)
        self.bn2 = norm_fn(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        assert x.features.dim() == 2, "x.features.dim()=%d" % x.features.dim()
        identity = x.features if self.downsample is None else self.downsample(x)
        if self.training or (torch.is_grad_enabled() and x.features.requires_grad):  # scn_unet.py:51-69 as written, differentiable
            out = self.conv1(x)
            y = ops.batch_norm_train(self.bn1, out.features, relu=True)  # BatchNorm + ReLU (+ residual) fused, forward and backward (csrc/norm.hip)
            out.features = y if y is not None else torch.relu(self.bn1(out.features))
            out = self.conv2(out)
            y = ops.batch_norm_train(self.bn2, out.features, res=identity, relu=True)
            out.features = y if y is not None else torch.relu(self.bn2(out.features) + identity)
            return out
        out = conv_bn_act(self.conv1, self.bn1, x, relu=True)
        return conv_bn_act(self.conv2, self.bn2, out, relu=True, res_pre=identity)


_SIDE_STREAMS = {}
import os as _os
# side streams of the capacity-mode geometry: 2 = {strided-rulebook chain | per-level rulebooks, plans, orders, devoxelization search}
# (measured on MI355X, one hipGraph per frame: 7.40 ms; 3 streams 7.63 ms; everything on one chain 7.84 ms)
_N_SIDE = 2


# The lateral SparseBasicBlock of a decoder level (conv_up_t<l>, scn_unet.py:163-165) reads the ENCODER output of its level only: it does not
# depend on anything the deeper levels compute.  It runs on its own stream from the moment the encoder leaves the level, beside the
# deeper levels - whose launches do not fill the chip (level 4 of a 120k-point frame: 215 tiles for 512 workgroup slots) and whose tails
# leave CUs idle - and the decoder picks it up with one event.  Same kernels on the same inputs: bit-identical.  LS3D_LATERAL_STREAM=0: inline.
_LATERAL = True
# capacity mode: `encoded_spconv_tensor` (scn_unet.py:218-222: conv_out of the deepest level) feeds no segmentation head - the key holds a proxy that
# runs the convolution (and builds its rulebook) when something reads it, instead of one more rulebook, mask sort and launch beside every frame
# (lidarseg3d_amd.set_reference_outputs(True) / LS3D_REFERENCE_OUTPUTS=1: computed eagerly, as the reference does)
_LAZY_ENCODED = _os.environ.get("LS3D_REFERENCE_OUTPUTS", "0") == "0"


def set_lazy_encoded(on):
    """capacity mode: batch_dict["encoded_spconv_tensor"] as an on-demand proxy (default) or computed with every frame (the reference's forward)"""
    global _LAZY_ENCODED
    _LAZY_ENCODED = bool(on)


class _LazyEncoded(spconv.SparseConvTensor):
    """batch_dict["encoded_spconv_tensor"] in capacity mode: a SparseConvTensor (isinstance holds) whose convolution - conv_out on the
    deepest level and its rulebook - runs when the first of its attributes is read, on the stream that is current then.  It reads the
    frame's level-4 tensor, so it has to be read (or materialize()d) before the next frame overwrites that tensor.  Nothing refreshes a
    proxy that was already read: graph.FrameGraph replays never see the batch_dict (the captured forward hands out predict()'s list only),
    so under a FrameGraph the tensor is available through set_reference_outputs(True) - computed inside every replay - and not through
    this proxy; invalidate() is for callers that keep a batch_dict across frames themselves."""

    def __init__(self, fn):  # no SparseConvTensor.__init__: the attributes do not exist until they are asked for
        self.__dict__["_fn"] = fn

    def materialize(self):
        if "features" not in self.__dict__:
            with torch.no_grad():
                value = self.__dict__["_fn"]()
            self.__dict__.update(value.__dict__)
        return self

    def invalidate(self):
        """drop the computed value (the inputs have been overwritten by a new frame): the next read computes it again"""
        fn = self.__dict__["_fn"]
        self.__dict__.clear()
        self.__dict__["_fn"] = fn

    def __getattr__(self, name):  # only called for attributes that are not there yet
        if name.startswith("__"):
            raise AttributeError(name)
        self.materialize()
        try:
            return self.__dict__[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self.materialize()
        self.__dict__[name] = value
_LATERAL_STREAMS = {}


def _lateral_stream(dev):
    st = _LATERAL_STREAMS.get(dev)
    if st is None:
        st = _LATERAL_STREAMS[dev] = torch.cuda.Stream(dev)
    return st


class _Chain(object):
    """Consecutive SubM layers on ONE rulebook (the SparseBasicBlocks of a UNet level, scn_unet.py:34-69, its lateral block and conv_m,
    :163-171) collected first and launched together: ONE persistent launch of the tile kernel (ops.tile_conv_chain, include/ls3d.h:
    ls3d_tile_conv_chain) in which the tiles of a layer start as soon as the tiles that own their halo rows have finished the previous layer -
    a layer's tail (677 tiles on 512 workgroup slots) is filled with the next layer's tiles.  Where the library declines (other precisions,
    layers too narrow for the tile kernel, LS3D_TILE_CHAIN=0) the layers run one by one through SparseConvolution.conv: same arithmetic,
    bit-identical results."""

    def __init__(self, x, key):
        self.x, self.rb, self.recs = x, x.find_indice_pair(key), []
        assert self.rb is not None and self.rb.kind == "subm", key

    def add(self, conv, bn, feats, relu=True, res_pre=None, pair=None, out=None):
        """queue conv + BN(eval) [+ res_pre] [+ ReLU] [+ channel-pair sum of `pair`] on feats -> the output tensor (written when run() is called)"""
        assert conv.subm and conv.indice_key is not None and self.x.find_indice_pair(conv.indice_key) is self.rb
        if out is None:
            out = ops.empty_rows(self.rb.tbl.shape[0], conv.out_channels, feats.device)
        scale, shift = spconv.cached_bn_scale_shift(conv, bn)
        self.recs.append((conv, feats, scale, shift, relu, res_pre, pair, out))
        return out

    def block(self, blk, feats, out=None):
        """SparseBasicBlock (scn_unet.py:51-69): relu(bn2(conv2(relu(bn1(conv1(x))))) + x)"""
        mid = self.add(blk.conv1, blk.bn1, feats)
        return self.add(blk.conv2, blk.bn2, mid, res_pre=feats, out=out)

    def _chain_layers(self):
        """ops.ChainLayer per queued layer, or None when one of them does not take the tile kernel in its chained form"""
        if not ops.tile_chain_enabled() or len(self.recs) < 2 or self.rb.tbl.shape[0] == 0:
            return None
        layers, k = [], self.rb.tbl.shape[1]
        for conv, feats, scale, shift, relu, res_pre, pair, out in self.recs:
            pk = conv.packed()
            if k not in pk:
                with torch.no_grad():
                    pk[k] = spconv.pack_spconv(conv._weight_for(self.rb))
            W, _, _, cout = pk[k]
            if conv.bias is not None or not ops.use_tile("subm", k, W.shape[1], cout) or cout > 128 or cout % 32:
                return None
            if not ops.tile_chain_pays(self.rb.tbl.shape[0], cout):
                return None
            if feats.shape[1] != W.shape[1]:  # e.g. 13 input channels feeding a 16-wide K chunk (a layer fed from outside the chain)
                feats = torch.nn.functional.pad(feats, (0, W.shape[1] - feats.shape[1]))
            if feats.stride(1) != 1 or feats.stride(0) % 4 or out.stride(1) != 1 or out.stride(0) % 32 or out.data_ptr() % 128:
                return None
            layers.append(ops.ChainLayer(feats, W, out, cout=cout, scale=scale, shift=shift, res_pre=res_pre, relu=relu, pair=pair))
        nt = set(1 if l.cout <= 32 else 2 if l.cout <= 64 else 4 for l in layers)
        return layers if len(nt) == 1 else None

    def run(self):
        layers = self._chain_layers()
        recs, self.recs = self.recs, []
        if layers is not None:
            plan = self.rb.tile_plan(False, layers[0].cout)
            done = 0
            while done < len(layers):
                part = layers[done:done + ops.TILE_CHAIN_MAX]
                if len(part) == 1:  # a single layer left over: the plain launch (most-expensive-first dispatch)
                    l = part[0]
                    ops.tile_conv(l.x, l.w, plan, cout=l.cout, scale=l.scale, shift=l.shift, res_pre=l.res_pre, relu=l.relu, pair=l.pair, out=l.out,
                                  in_ld=l.x.stride(0))
                elif not ops.tile_conv_chain(part, plan):
                    break
                done += len(part)
            if done == len(layers):
                return
            recs = recs[done:]
        for conv, feats, scale, shift, relu, res_pre, pair, out in recs:
            conv.conv(feats, self.rb, scale=scale, shift=shift, relu=relu, res_pre=res_pre, pair=pair, out=out, out_ld=out.stride(0))


class _GeometryStream(object):
    """`with _GeometryStream(t):` runs the body on a per-device side stream that starts after the current stream's work so far;
    on exit the current stream waits for it.  hand_over() tells the caching allocator that tensors created inside are used
    on the main stream from now on.  A no-op for CPU tensors (tests/hipsim) and with LS3D_OVERLAP=0."""

    def __init__(self, like, ready=None, join=True, index=0, after=()):
        """index: which of the device's side streams (capacity mode builds the levels' geometry on several, side by side);
        after: events of other side streams this block depends on"""
        import os
        self.ready = ready
        self.released = not join  # join=False: the main stream does not wait at the end of the block (pick up finish_event())
        self.on = like.is_cuda and os.environ.get("LS3D_OVERLAP", "1") != "0"
        self.dev, self.index, self.after = like.device, index, after

    def __enter__(self):
        if self.on:
            self.main = torch.cuda.current_stream(self.dev)
            idx = min(self.index, _N_SIDE - 1) if _N_SIDE < 3 else self.index  # fewer side streams: 2 -> {strided chain | rest}, 1 -> one chain
            idx = 0 if (_N_SIDE == 2 and self.index == 2) else idx
            self.side = _SIDE_STREAMS.get((self.dev, idx))
            if self.side is None:
                self.side = _SIDE_STREAMS[(self.dev, idx)] = torch.cuda.Stream(self.dev)
            if self.ready is not None:
                self.side.wait_event(self.ready)  # only the coordinates, not the reader that was launched after them
            else:
                self.side.wait_stream(self.main)
            for ev in self.after:
                if ev is not None and getattr(ev, "_ls3d_stream", None) is not self.side:  # in-order on its own stream anyway
                    self.side.wait_event(ev)
            self.ctx = torch.cuda.stream(self.side)
            self.ctx.__enter__()
        return self

    def hand_over(self, rulebooks):
        if not self.on:
            return
        for rb in rulebooks:
            # everything allocated on the side stream that main-stream kernels read - in capacity mode also the device-side row
            # counts, which the convolutions of the whole frame dereference long after the geometry stream has moved on
            for name in ("tbl", "tbl_inv", "in_indices", "out_indices", "n_in_dev", "n_out_dev", "overflow_dev"):
                t = getattr(rb, name, None)
                if t is not None and t.is_cuda:
                    t.record_stream(self.main)
            for o in (getattr(rb, "_orders", None) or {}).values():
                if o is not None:
                    o.record_stream(self.main)
            for pl in (getattr(rb, "_plans", None) or {}).values():
                pl.record_stream(self.main)

    def release(self):
        """everything enqueued on the side stream so far must be done before the main stream continues; what follows inside
        the block keeps running beside the main stream (pick it up with finish_event())"""
        if self.on:
            ev = torch.cuda.Event()
            ev.record(self.side)
            self.main.wait_event(ev)
            self.released = True

    def finish_event(self):
        if not self.on:
            return None
        ev = torch.cuda.Event()
        ev.record(self.side)
        ev._ls3d_stream = self.side
        return ev

    def keep(self, *tensors):
        if self.on:
            for t in tensors:
                t.record_stream(self.main)

    def __exit__(self, *exc):
        if self.on:
            self.ctx.__exit__(*exc)
            if not self.released:
                self.main.wait_stream(self.side)
        return False


@BACKBONES.register_module
class UNetSCN3D(nn.Module):
    def __init__(self, num_input_features=128, name="UNetSCN3D", voxel_size=[], point_cloud_range=[], model_cfg={},
                 **kwargs):
        super().__init__()
        self.model_cfg, self.voxel_size, self.point_cloud_range = model_cfg, voxel_size, point_cloud_range
        norm_fn = partial(nn.BatchNorm1d, eps=1e-3, momentum=0.01)
        r = model_cfg.get("SCALING_RATIO", 1)
        c1, c2, c3, c4 = 16 * r, 32 * r, 64 * r, 64 * r
        block = post_act_block
        self.conv_input = spconv.SparseSequential(
            spconv.SubMConv3d(num_input_features, c1, 3, padding=1, bias=False, indice_key="subm1"), norm_fn(c1), nn.ReLU())
        self.conv1 = spconv.SparseSequential(SparseBasicBlock(c1, c1, norm_fn=norm_fn, indice_key="subm1"),
                                             SparseBasicBlock(c1, c1, norm_fn=norm_fn, indice_key="subm1"))
        self.conv2 = spconv.SparseSequential(
            block(c1, c2, 3, norm_fn=norm_fn, stride=2, padding=1, indice_key="spconv2", conv_type="spconv"),
            SparseBasicBlock(c2, c2, norm_fn=norm_fn, indice_key="subm2"),
            SparseBasicBlock(c2, c2, norm_fn=norm_fn, indice_key="subm2"))
        self.conv3 = spconv.SparseSequential(
            block(c2, c3, 3, norm_fn=norm_fn, stride=2, padding=1, indice_key="spconv3", conv_type="spconv"),
            SparseBasicBlock(c3, c3, norm_fn=norm_fn, indice_key="subm3"),
            SparseBasicBlock(c3, c3, norm_fn=norm_fn, indice_key="subm3"))
        self.conv4 = spconv.SparseSequential(
            block(c3, c4, 3, norm_fn=norm_fn, stride=2, padding=(0, 1, 1), indice_key="spconv4", conv_type="spconv"),
            SparseBasicBlock(c4, c4, norm_fn=norm_fn, indice_key="subm4"),
            SparseBasicBlock(c4, c4, norm_fn=norm_fn, indice_key="subm4"))
        if self.model_cfg.get("RETURN_ENCODED_TENSOR", True):
            last_pad = self.model_cfg.get("last_pad", 0)
            self.conv_out = spconv.SparseSequential(
                spconv.SparseConv3d(c4, 128, (3, 1, 1), stride=(2, 1, 1), padding=last_pad, bias=False,
                                    indice_key="spconv_down2"), norm_fn(128), nn.ReLU())
        else:
            self.conv_out = None
        # decoder (scn_unet.py:138-160)
        self.conv_up_t4 = SparseBasicBlock(c4, c4, indice_key="subm4", norm_fn=norm_fn)
        self.conv_up_m4 = block(2 * c4, c4, 3, norm_fn=norm_fn, padding=1, indice_key="subm4")
        self.inv_conv4 = block(c4, c3, 3, norm_fn=norm_fn, indice_key="spconv4", conv_type="inverseconv")
        self.conv_up_t3 = SparseBasicBlock(c3, c3, indice_key="subm3", norm_fn=norm_fn)
        self.conv_up_m3 = block(2 * c3, c3, 3, norm_fn=norm_fn, padding=1, indice_key="subm3")
        self.inv_conv3 = block(c3, c2, 3, norm_fn=norm_fn, indice_key="spconv3", conv_type="inverseconv")
        self.conv_up_t2 = SparseBasicBlock(c2, c2, indice_key="subm2", norm_fn=norm_fn)
        self.conv_up_m2 = block(2 * c2, c2, 3, norm_fn=norm_fn, indice_key="subm2")
        self.inv_conv2 = block(c2, c1, 3, norm_fn=norm_fn, indice_key="spconv2", conv_type="inverseconv")
        self.conv_up_t1 = SparseBasicBlock(c1, c1, indice_key="subm1", norm_fn=norm_fn)
        self.conv_up_m1 = block(2 * c1, c1, 3, norm_fn=norm_fn, indice_key="subm1")
        self.conv5 = spconv.SparseSequential(block(c1, c1, 3, norm_fn=norm_fn, padding=1, indice_key="subm1"))
        self.num_point_features = c1

    @staticmethod
    def _lateral_block(x_lateral, conv_t, cat):
        """conv_t's two convolutions; the second writes straight into the right half of the level's concat buffer"""
        c = x_lateral.features.shape[1]
        mid = conv_bn_act(conv_t.conv1, conv_t.bn1, x_lateral, relu=True)
        rb = conv_t.conv2.rulebook(mid)
        s, t = spconv.cached_bn_scale_shift(conv_t.conv2, conv_t.bn2)
        conv_t.conv2.conv(mid, rb, scale=s, shift=t, relu=True, res_pre=x_lateral.features, out=cat[:, c:], out_ld=2 * c)

    def _lateral_launch(self, x_lateral, conv_t, cat):
        """-> event behind conv_t's block on the lateral stream (None: not launched, UR_block_forward runs it inline)"""
        f = x_lateral.features
        if not (_LATERAL and f.is_cuda and _os.environ.get("LS3D_OVERLAP", "1") != "0"):
            return None
        main, lat = torch.cuda.current_stream(f.device), _lateral_stream(f.device)
        lat.wait_stream(main)
        with torch.cuda.stream(lat):
            self._lateral_block(x_lateral, conv_t, cat)
            done = torch.cuda.Event()
            done.record(lat)
        f.record_stream(lat)
        cat.record_stream(lat)
        return done

    @staticmethod
    def _chained(features):
        """the chained form of the inference forward (one persistent tile-kernel launch per level segment) applies: device tensors (or the
        host emulation of the tests), 6-product arithmetic, not switched off"""
        return ops.tile_chain_enabled() and (features.is_cuda or ops.sim_mode())

    def _level_chain(self, x_s, key, blocks, lateral, first=None, conv_m=None):
        """One UNet level's SubM layers on the level's rulebook `key` as ONE chained launch (_Chain): [first = (conv, bn) in front of the blocks:
        conv_input] + the encoder's SparseBasicBlocks + the decoder's lateral block conv_up_t of the same level (it reads the encoder output
        only, scn_unet.py:163-165) [+ conv_m where the level is the deepest: its concat buffer is complete then].  The last encoder layer
        writes the LEFT half of the level's concat buffer when conv_m follows, the lateral block's second layer always the RIGHT half.
        -> (encoder output tensor, concat buffer, conv_m's output tensor or None)"""
        ch = _Chain(x_s, key)
        f = x_s.features
        if first is not None:
            f = ch.add(first[0], first[1], f)
        c = blocks[-1].conv2.out_channels
        cat = ops.empty_rows(f.shape[0], 2 * c, f.device)
        for i, blk in enumerate(blocks):
            f = ch.block(blk, f, out=cat[:, :c] if (conv_m is not None and i == len(blocks) - 1) else None)
        x_enc = x_s._like(f)
        mid = ch.add(lateral.conv1, lateral.bn1, f)
        ch.add(lateral.conv2, lateral.bn2, mid, res_pre=f, out=cat[:, c:])
        x_m = None
        if conv_m is not None:
            x_m = x_s._like(ch.add(conv_m[0], conv_m[1], cat, pair=cat))
        ch.run()
        return x_enc, cat, x_m

    def _decoder_chained(self, x_conv1, x_conv2, x_conv3, x_m4, cats):
        """the decoder behind the chained encoder levels: conv_m4 ran with level 4's chain (x_m4), the lateral blocks with their levels' chains
        (the right halves of cats = [level 3, level 2, level 1] are written); what is left per level is the inverse convolution into the next
        level's left half, conv_m of that level, and at the top conv_m1 + conv5 as one more chain"""
        x_up4 = self._inverse_into(x_m4, self.inv_conv4, cats[0])
        x_up3 = self.UR_block_forward(x_conv3, x_up4, self.conv_up_t3, self.conv_up_m3, self.inv_conv3, cat=cats[0], next_cat=cats[1], lateral=True)
        x_up2 = self.UR_block_forward(x_conv2, x_up3, self.conv_up_t2, self.conv_up_m2, self.inv_conv2, cat=cats[1], next_cat=cats[2], lateral=True)
        ch = _Chain(x_conv1, "subm1")
        fm = ch.add(self.conv_up_m1[0], self.conv_up_m1[1], cats[2], pair=cats[2])
        f5 = ch.add(self.conv5[0][0], self.conv5[0][1], fm)
        ch.run()
        return x_conv1._like(f5), x_up2, x_up3, x_up4

    @staticmethod
    def _inverse_into(x, conv_inv, next_cat):
        """the inverse convolution of a decoder level, written straight into the left half of the next level's concat buffer"""
        inv, bn = conv_inv[0], conv_inv[1]
        rbi = inv.rulebook(x)
        s, t = spconv.cached_bn_scale_shift(inv, bn)
        cout = inv.out_channels
        inv.conv(x, rbi, scale=s, shift=t, relu=True, out=next_cat[:, :cout], out_ld=next_cat.shape[1])
        return x._like(next_cat[:, :cout], rbi.in_indices, rbi.in_shape, n_dev=rbi.rows_dev(True))

    def UR_block_forward(self, x_lateral, x_bottom, conv_t, conv_m, conv_inv, cat=None, next_cat=None, lateral=None):
        """scn_unet.py:163-171 with the data movement fused away:
          * `cat` [V, 2C] is the concat buffer; its left half already holds x_bottom when the previous UR block's
            inverse conv wrote there (`next_cat`), otherwise it is copied in;
          * the lateral block's second conv writes straight into the right half;
          * conv_m's epilogue adds the channel-pair sums of `cat` (channel_reduction + add);
          * the inverse conv writes into the left half of the NEXT level's concat buffer.
        cat / view-sum / add / copies never run as separate passes."""
        n, c = x_lateral.features.shape
        if cat is None:
            cat = torch.empty((n, 2 * c), dtype=torch.float32, device=x_lateral.features.device)
            cat[:, :c].copy_(x_bottom.features)
        if lateral is None:
            self._lateral_block(x_lateral, conv_t, cat)
        elif lateral is not True:
            torch.cuda.current_stream(cat.device).wait_event(lateral)  # the block ran beside the deeper levels (_lateral_launch)
        x = x_lateral._like(cat)
        x = conv_bn_act(conv_m[0], conv_m[1], x, relu=True, pair=cat)
        if next_cat is None:
            return conv_inv(x)
        return self._inverse_into(x, conv_inv, next_cat)

    def UR_block_forward_train(self, x_lateral, x_bottom, conv_t, conv_m, conv_inv):
        """scn_unet.py:163-171 as written (training: every step is a differentiable op)"""
        x_trans = conv_t(x_lateral)
        x = x_trans._like(torch.cat([x_bottom.features, x_trans.features], dim=1))
        x_m = conv_m(x)
        n, c = x_m.features.shape
        x_m.features = x_m.features + x.features.view(n, c, -1).sum(dim=2)  # channel_reduction + add
        return conv_inv(x_m)

    @staticmethod
    def channel_reduction(x, out_channels):
        """scn_unet.py:173-187 (kept for API parity; the forward fuses it into conv_m's epilogue)"""
        n, cin = x.features.shape
        assert cin % out_channels == 0 and cin >= out_channels
        x.features = x.features.view(n, out_channels, -1).sum(dim=2)
        return x

    def forward(self, batch_dict):
        voxel_features, voxel_coords = batch_dict["voxel_features"], batch_dict["voxel_coords"]
        batch_size = batch_dict["batch_size"]
        sparse_shape = np.array(batch_dict["input_shape"][::-1]) + [1, 0, 0]
        vc = voxel_coords.int().contiguous()
        n_dev = batch_dict.get("num_active_voxels_dev")  # capacity mode (detectors.py): rows beyond this device count are spare
        if n_dev is not None:
            return self._forward_capacity(batch_dict, voxel_features, vc, sparse_shape, batch_size, n_dev)
        x = spconv.SparseConvTensor(voxel_features, vc, sparse_shape, batch_size)
        # the "coordinates ready" event covers the tensor the reader's caller recorded it for; a converted copy (int64 / strided
        # coordinates from a custom loader) is written by a kernel enqueued AFTER that event: wait for the main stream instead
        ready = batch_dict.get("voxel_coords_ready") if vc.data_ptr() == voxel_coords.data_ptr() else None
        # The geometry of the frame depends on the voxel COORDINATES only, so it is built on a side stream while the main stream is
        # still busy with the reader that produces voxel_features (k_transvfe: one LDS-bound workgroup per CU, the small
        # latency-bound rulebook kernels fit beside it) - in two stages, so that the main stream never waits for more than it needs:
        # stage 1 = the level-1 SubM rulebook and its tile plan (sizes known on the host: no sync) ...
        with _GeometryStream(x.indices, ready) as gs:
            x.indice_dict["subm1"] = spconv.subm_rulebook(x.indices, x.spatial_shape, 3, x.batch_size)
            spconv.prebuild_orders(x, self.modules())
            gs.hand_over(x.indice_dict.values())
            gs.release()
        infer = not (self.training or (torch.is_grad_enabled() and voxel_features.requires_grad))
        chained = infer and self._chained(voxel_features)
        ev0 = self._stack_event()
        if chained:  # conv_input + conv1's blocks + the level's lateral block: one launch
            x_conv1, cat1, _ = self._level_chain(x, "subm1", [self.conv1[0], self.conv1[1]], self.conv_up_t1, first=(self.conv_input[0], self.conv_input[1]))
        else:
            x = self.conv_input(x)  # ... the five level-1 launches are queued behind it ...
            x_conv1 = self.conv1(x)
        ev1 = self._stack_event(ev0)
        # ... stage 2 = the strided rulebooks of the encoder (the first one now, the other three chained on device counts behind the
        # level-2 convolutions: two host syncs for their sizes, both while the main stream is busy) ...
        with _GeometryStream(x.indices, ready, join=False) as gs:
            # the first strided rulebook on its own (one host sync for its size): level 2's geometry and convolutions do not wait for the
            # other three, which follow - chained on device counts, one more sync - while the level-2 convolutions run
            spconv.prebuild_conv_rulebooks(x, [self.conv2[0][0]])
        # ... then level by level: SubM rulebook, tile plan and the mask-sorted row order of the strided layer that enters the level
        # on the side stream, one event, the level's convolutions behind that event on the main stream.  The HOST alternates between
        # the two streams: submitting a level's ~30 small geometry launches takes longer than running them, so the main stream
        # gets level k's convolutions (a millisecond of GPU work) before the host turns to level k+1's geometry (round-2 trace with
        # all geometry submitted first: 0.6 ms of idle main stream per frame, all of it host submission time).
        ev0, x_enc = None, x_conv1
        cats, lats = [None, None, (cat1 if chained else self._new_cat(x_conv1)) if infer else None], [None, None, None]  # levels 3, 2, 1 (the order the decoder takes them)
        x_m4 = None
        if infer and not chained:
            lats[2] = self._lateral_launch(x_conv1, self.conv_up_t1, cats[2])
        for lvl, (key, src, stage) in enumerate((("subm2", "spconv2", self.conv2), ("subm3", "spconv3", self.conv3), ("subm4", "spconv4", self.conv4))):
            with _GeometryStream(x.indices, ready, join=False) as gs:
                rb = x.find_indice_pair(src)
                x.indice_dict[key] = spconv.subm_rulebook(rb.out_indices, rb.out_shape, 3, x.batch_size)
                spconv.prebuild_orders(x, stage.modules() if lvl < 2 else self.modules())  # the last event covers the decoder's orders
                gs.hand_over(x.indice_dict.values())
                level_ready = gs.finish_event()
            self._wait(x, level_ready)
            if lvl == 0:
                ev0 = self._stack_event() if ev1 is not None else None
            if chained:  # the strided convolution, then the level's blocks + lateral block (+ conv_m on the deepest level) as one launch
                x_enc, cat_l, x_m4 = self._level_chain(stage[0](x_enc), key, [stage[1], stage[2]], (self.conv_up_t2, self.conv_up_t3, self.conv_up_t4)[lvl],
                                                       conv_m=self.conv_up_m4 if lvl == 2 else None)
                if lvl < 2:
                    cats[1 - lvl] = cat_l
            else:
                x_enc = stage(x_enc)
            if lvl == 0:
                x_conv2 = x_enc
                with _GeometryStream(x.indices, ready, join=False) as gs:
                    rb2 = x.find_indice_pair("spconv2")
                    rest = [self.conv3[0][0], self.conv4[0][0]] + ([self.conv_out[0]] if self.conv_out is not None else [])
                    spconv.prebuild_conv_rulebooks(x, rest, coords=rb2.out_indices, shape=rb2.out_shape)
            elif lvl == 1:
                x_conv3 = x_enc
            if infer and not chained and lvl < 2:  # the level's lateral block starts beside the deeper levels
                cats[1 - lvl] = self._new_cat(x_enc)
                lats[1 - lvl] = self._lateral_launch(x_enc, self.conv_up_t2 if lvl == 0 else self.conv_up_t3, cats[1 - lvl])
        x_conv4 = x_enc
        # the neighbour search of the devoxelization (points -> 3 nearest voxel centres + weights) is geometry as well: it runs on the
        # side stream beside the decoder; the point head only interpolates (point_heads._devoxelize)
        with _GeometryStream(x.indices, ready, join=False) as gs2:
            self._start_devox_search(batch_dict, x, gs2)
        conv_out_done = self._conv_out(batch_dict, x_conv4, beside=infer)
        if self.training or (torch.is_grad_enabled() and voxel_features.requires_grad):
            x_up4 = self.UR_block_forward_train(x_conv4, x_conv4, self.conv_up_t4, self.conv_up_m4, self.inv_conv4)
            x_up3 = self.UR_block_forward_train(x_conv3, x_up4, self.conv_up_t3, self.conv_up_m3, self.inv_conv3)
            x_up2 = self.UR_block_forward_train(x_conv2, x_up3, self.conv_up_t2, self.conv_up_m2, self.inv_conv2)
            x_up1 = self.UR_block_forward_train(x_conv1, x_up2, self.conv_up_t1, self.conv_up_m1, self.conv5)
            self._stack_event(ev0)
            return self._outputs(batch_dict, x_up1, x_up2, x_up3, x_up4, x_conv4)
        if chained:
            x_up1, x_up2, x_up3, x_up4 = self._decoder_chained(x_conv1, x_conv2, x_conv3, x_m4, cats)
            self._stack_event(ev0)
            self._wait(x, conv_out_done)
            return self._outputs(batch_dict, x_up1, x_up2, x_up3, x_up4, x_conv4)
        x_up4 = self.UR_block_forward(x_conv4, x_conv4, self.conv_up_t4, self.conv_up_m4, self.inv_conv4, next_cat=cats[0])
        x_up3 = self.UR_block_forward(x_conv3, x_up4, self.conv_up_t3, self.conv_up_m3, self.inv_conv3, cat=cats[0], next_cat=cats[1], lateral=lats[0])
        x_up2 = self.UR_block_forward(x_conv2, x_up3, self.conv_up_t2, self.conv_up_m2, self.inv_conv2, cat=cats[1], next_cat=cats[2], lateral=lats[1])
        x_up1 = self.UR_block_forward(x_conv1, x_up2, self.conv_up_t1, self.conv_up_m1, self.conv5, cat=cats[2], lateral=lats[2])
        self._stack_event(ev0)
        self._wait(x, conv_out_done)
        return self._outputs(batch_dict, x_up1, x_up2, x_up3, x_up4, x_conv4)

    def _conv_out(self, batch_dict, x_conv4, beside=True):
        """`encoded_spconv_tensor` (scn_unet.py:218-222) feeds no segmentation head: the key stays in batch_dict, its (3,1,1) convolution
        runs beside the decoder on the lateral stream.  -> event to wait for at the end of the forward (None: it ran inline)"""
        if self.conv_out is None:
            return None
        f = x_conv4.features
        if not (beside and _LATERAL and f.is_cuda and _os.environ.get("LS3D_OVERLAP", "1") != "0"):
            batch_dict["encoded_spconv_tensor"] = self.conv_out(x_conv4)
            batch_dict["encoded_spconv_tensor_stride"] = 8
            return None
        main, lat = torch.cuda.current_stream(f.device), _lateral_stream(f.device)
        lat.wait_stream(main)
        with torch.cuda.stream(lat):
            enc = self.conv_out(x_conv4)
            done = torch.cuda.Event()
            done.record(lat)
        f.record_stream(lat)
        enc.features.record_stream(main)
        batch_dict["encoded_spconv_tensor"], batch_dict["encoded_spconv_tensor_stride"] = enc, 8
        return done

    @staticmethod
    def _new_cat(x):
        """concat buffer [rows, 2C] of a decoder level: left half <- the inverse convolution from below, right half <- the lateral block"""
        n, c = x.features.shape
        return torch.empty((n, 2 * c), dtype=torch.float32, device=x.features.device)

    # ---------------------------------------------------------------------------------------------- capacity mode (no host syncs)
    _caps = None  # {strided layer key: [capacity of its output sites, largest count seen]} - adapted from the frames seen so far

    def _strided_chain(self):
        """the strided convolutions whose rulebooks a capacity-mode frame builds up front (conv_out's only when its output is computed eagerly)"""
        return [self.conv2[0][0], self.conv3[0][0], self.conv4[0][0]] + ([self.conv_out[0]] if (self.conv_out is not None and not _LAZY_ENCODED) else [])

    def _capacities(self, n_in_cap, batch_size, shape):
        """output capacities of the encoder's strided convolutions for an input capacity of n_in_cap rows.  Worst case (every input
        site feeds 8 outputs, bounded by the output grid) until counts have been seen; then 1.3x the largest count seen so far for
        this input capacity, rounded up to 4096 - a frame that overflows is run again with the worst case (geometry_check)."""
        worst, n, sh = [], n_in_cap, list(shape)
        for c in self._strided_chain():
            osh = ops.conv_out_shape(sh, c.kernel_size, c.stride, c.padding)
            per_in = 1
            for a in range(3):
                per_in *= -(-c.kernel_size[a] // c.stride[a])
            n = max(1, min(n * per_in, batch_size * osh[0] * osh[1] * osh[2]))
            worst.append(n)
            sh = osh
        if self._caps is None:
            self._caps = {}
        seen = self._caps.get((n_in_cap, batch_size, len(worst)))  # (the chain has one more rulebook when conv_out is computed eagerly)
        if seen is None:
            return worst
        return [min(w, max(4096, -(-int(1.3 * m) // 4096) * 4096)) for w, m in zip(worst, seen)]

    def geometry_check(self, batch_dict):
        """capacity mode: wait (host) for this frame's strided-rulebook counts - they are produced early in the frame, on the
        geometry stream, so this returns while the GPU is still busy with the frame's convolutions: no bubble, and the host can
        submit the next frame meanwhile.  Updates the capacities from the counts; returns False when a table overflowed its capacity
        (the caller runs the frame again: the capacities are back at the worst case then)."""
        rec = batch_dict.pop("geometry_record", None)
        if rec is None:
            return True
        host, ev, key = rec
        if host.is_pinned() and torch.cuda.is_current_stream_capturing():
            self.__dict__["_captured_record"] = (host, key)  # graph.FrameGraph reads the counts after each replay
            return True
        if ev is not None:
            ev.synchronize()
        return self.apply_counts(host.tolist(), key)

    def apply_counts(self, cnt, key):
        """cnt = [[n_out, overflow], ...] of the strided rulebooks of a frame: learn the capacities, or forget them after an overflow"""
        if any(o for _, o in cnt):
            self._caps.pop(key, None)
            return False
        seen = self._caps.get(key)
        self._caps[key] = [n for n, _ in cnt] if seen is None else [max(a, n) for a, (n, _) in zip(seen, cnt)]
        return True

    def _forward_capacity(self, batch_dict, voxel_features, vc, sparse_shape, batch_size, n_dev):
        """forward() with device-side row counts: the same launches in the same order on the same two streams, tensors sized by
        capacities, every kernel bounded by its device count - no host synchronisation inside the frame ("avoid host
        syncs for V").  Results are bit-identical to forward(): spare rows sort behind the valid ones in every order, so tiles, row
        orders and summation orders of the valid rows are the same."""
        if self.training or (torch.is_grad_enabled() and voxel_features.requires_grad):
            raise ops.CapacityModeUnsupported("capacity mode is an inference path")
        x = spconv.SparseConvTensor(voxel_features, vc, sparse_shape, batch_size, n_dev=n_dev)
        ready = batch_dict.get("voxel_coords_ready")
        chain = self._strided_chain()
        caps = self._capacities(vc.shape[0], batch_size, x.spatial_shape)
        # Nothing here needs a count on the host, so the WHOLE geometry of the frame is submitted up front, on three side streams that
        # run side by side behind the "coordinates ready" event (a chain of ~100 dependent 2-80 us kernels on ONE stream takes
        # ~1.2 ms of GPU time however early it is submitted - the level-2 convolutions used to wait 0.6 ms for it,
        # profiles/round3_timeline_eager.txt):
        #   stream 1: all strided rulebooks of the encoder, chained on device counts; their counts and overflow flags go to pinned
        #             host memory right behind them (read in geometry_check);
        #   streams 0 / 2, alternating: per level the SubM rulebook, the tile plan and the row orders, each behind the event of the
        #             strided rulebook that creates the level's sites; the devoxelization's neighbour search at the end.
        # The main stream picks the levels up one event at a time.
        with _GeometryStream(x.indices, ready, join=False, index=1) as gs:
            # (an early mask sort of the first strided layer's table - the layer 114 -> 51 us alone - was measured in round 4: its sort kernels
            # slow the reader beside them and the frame gains nothing; profiles/round4_ab_gather_knobs.txt.  Removed in round 5.)
            rb_events = []
            spconv.prebuild_conv_rulebooks(x, chain, nosync=True, caps=caps, after_each=lambda: rb_events.append(gs.finish_event()))
            cnts = torch.stack([torch.cat([x.indice_dict[c.indice_key].n_out_dev, x.indice_dict[c.indice_key].overflow_dev]) for c in chain])
            vflag = batch_dict.get("voxel_overflow_dev")
            if vflag is not None:  # a frame of the batch exceeded the voxelizer's per-frame cap (detectors._voxel_inputs): reported as an overflow
                cnts[0, 1:2] += vflag.to(cnts.dtype)
            if cnts.is_cuda:
                # one pinned buffer per model: a frame's counts are read (geometry_check) before the next frame is submitted, and a
                # captured frame (graph.FrameGraph) needs the same host address on every replay
                host = self.__dict__.get("_pinned_counts")
                if host is None or host.shape != cnts.shape:
                    host = self.__dict__["_pinned_counts"] = ops.registered_host(cnts.shape, cnts.dtype)
                host.copy_(cnts, non_blocking=True)
                batch_dict["geometry_record"] = (host, gs.finish_event(), (vc.shape[0], batch_size, len(chain)))
                gs.keep(cnts)
            else:
                batch_dict["geometry_record"] = (cnts, None, (vc.shape[0], batch_size, len(chain)))
            counts_copied = gs.finish_event()
        level_ready = []
        with _GeometryStream(x.indices, ready, join=False, index=0) as gs:
            x.indice_dict["subm1"] = spconv.subm_rulebook(x.indices, x.spatial_shape, 3, x.batch_size, n_dev=n_dev)
            spconv.prebuild_orders(x, list(self.conv_input.modules()) + list(self.conv1.modules()))
            gs.hand_over(x.indice_dict.values())
            level_ready.append(gs.finish_event())
        for lvl, (key, src, stage) in enumerate((("subm2", "spconv2", self.conv2), ("subm3", "spconv3", self.conv3), ("subm4", "spconv4", self.conv4))):
            # the last level's block also builds what is left - the decoder's row orders (the inverse tables of EVERY strided rulebook of the
            # chain; conv_out's only when it is computed eagerly, set_lazy_encoded(False)) - so it waits for the whole chain
            deps = rb_events[lvl:lvl + 1] + (rb_events[-1:] if lvl == 2 else [])
            with _GeometryStream(x.indices, ready, join=False, index=(2 if lvl % 2 == 0 else 0), after=deps) as gs:
                rb = x.find_indice_pair(src)
                x.indice_dict[key] = spconv.subm_rulebook(rb.out_indices, rb.out_shape, 3, x.batch_size, n_dev=rb.n_out_dev)
                spconv.prebuild_orders(x, stage.modules() if lvl < 2 else self.modules())
                gs.hand_over(x.indice_dict.values())
                level_ready.append(gs.finish_event())
        with _GeometryStream(x.indices, ready, join=False, index=0) as gs2:
            self._start_devox_search(batch_dict, x, gs2)
        chained = self._chained(voxel_features)
        ev0 = self._stack_event()
        self._wait(x, level_ready[0])
        cats, lats, x_m4 = [None, None, None], [None, None, None], None  # levels 3, 2, 1 (the order the decoder takes them)
        if chained:  # conv_input + conv1's blocks + the level's lateral block: one launch
            x_conv1, cats[2], _ = self._level_chain(x, "subm1", [self.conv1[0], self.conv1[1]], self.conv_up_t1, first=(self.conv_input[0], self.conv_input[1]))
        else:
            x = self.conv_input(x)
            x_conv1 = self.conv1(x)
        ev1 = self._stack_event(ev0)
        ev0, x_enc = None, x_conv1
        if not chained:
            cats[2] = self._new_cat(x_conv1)
            lats[2] = self._lateral_launch(x_conv1, self.conv_up_t1, cats[2])
        for lvl, (key, stage) in enumerate((("subm2", self.conv2), ("subm3", self.conv3), ("subm4", self.conv4))):
            self._wait(x, level_ready[lvl + 1])
            if lvl == 0:
                ev0 = self._stack_event() if ev1 is not None else None
            if chained:  # the strided convolution, then the level's blocks + lateral block (+ conv_m on the deepest level) as one launch
                x_enc, cat_l, x_m4 = self._level_chain(stage[0](x_enc), key, [stage[1], stage[2]], (self.conv_up_t2, self.conv_up_t3, self.conv_up_t4)[lvl],
                                                       conv_m=self.conv_up_m4 if lvl == 2 else None)
                if lvl < 2:
                    cats[1 - lvl] = cat_l
            else:
                x_enc = stage(x_enc)
            if lvl == 0:
                x_conv2 = x_enc
            elif lvl == 1:
                x_conv3 = x_enc
            if not chained and lvl < 2:  # the level's lateral block starts beside the deeper levels
                cats[1 - lvl] = self._new_cat(x_enc)
                lats[1 - lvl] = self._lateral_launch(x_enc, self.conv_up_t2 if lvl == 0 else self.conv_up_t3, cats[1 - lvl])
        x_conv4 = x_enc
        self._wait(x, counts_copied)  # joins stream 1 (matters for a captured frame: no unjoined work at the end of the capture)
        if _LAZY_ENCODED and self.conv_out is not None:
            batch_dict["encoded_spconv_tensor"], batch_dict["encoded_spconv_tensor_stride"] = _LazyEncoded(lambda t=x_conv4: self.conv_out(t)), 8
            conv_out_done = None
        else:
            conv_out_done = self._conv_out(batch_dict, x_conv4)
        if chained:
            x_up1, x_up2, x_up3, x_up4 = self._decoder_chained(x_conv1, x_conv2, x_conv3, x_m4, cats)
        else:
            x_up4 = self.UR_block_forward(x_conv4, x_conv4, self.conv_up_t4, self.conv_up_m4, self.inv_conv4, next_cat=cats[0])
            x_up3 = self.UR_block_forward(x_conv3, x_up4, self.conv_up_t3, self.conv_up_m3, self.inv_conv3, cat=cats[0], next_cat=cats[1], lateral=lats[0])
            x_up2 = self.UR_block_forward(x_conv2, x_up3, self.conv_up_t2, self.conv_up_m2, self.inv_conv2, cat=cats[1], next_cat=cats[2], lateral=lats[1])
            x_up1 = self.UR_block_forward(x_conv1, x_up2, self.conv_up_t1, self.conv_up_m1, self.conv5, cat=cats[2], lateral=lats[2])
        self._stack_event(ev0)
        self._wait(x, conv_out_done)
        batch_dict["num_active_voxels_dev"] = x_up1.n_dev
        return self._outputs(batch_dict, x_up1, x_up2, x_up3, x_up4, x_conv4)

    @staticmethod
    def _wait(x, ev):
        if ev is not None:
            torch.cuda.current_stream(x.indices.device).wait_event(ev)

    conv_stack_events = None  # measurement hook (bench.py): a list that receives the (start, end) HIP-event pairs of a forward

    def _stack_event(self, start=None):
        """brackets of the sparse-conv stack: its 37 launches are contiguous on the main stream except for one wait for the second
        geometry stage, so two (start, end) pairs per frame (level 1 | the rest) instead of one pair per launch, which cost
        ~10 us of idle GPU each"""
        if self.conv_stack_events is None:
            return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        if start is not None:
            self.conv_stack_events.append((start, ev))
        return ev

    def _start_devox_search(self, batch_dict, x, gs):
        pts = batch_dict.get("points")
        if pts is None or pts.dim() != 2 or pts.shape[1] < 4 or pts.device != x.indices.device or not pts.is_contiguous():
            return
        bs = int(batch_dict["batch_size"])
        centers = ops.voxel_centers(x.indices, self.voxel_size, self.point_cloud_range, n_dev=x.n_dev)
        pt_off, vx_off = ops.frame_offsets(pts, bs), ops.frame_offsets(centers, bs, n_dev=x.n_dev)
        idx, w = ops.devoxelize_grid(pts, pt_off, x.indices, centers, vx_off, bs, list(self.voxel_size), list(self.point_cloud_range), None,
                                     n_dev=x.n_dev)
        gs.keep(centers, pt_off, vx_off, idx, w)
        batch_dict["devox_search"] = dict(points=pts, indices=x.indices, centers=centers, pt_off=pt_off, vx_off=vx_off, idx=idx, weight=w,
                                          event=gs.finish_event())

    def _outputs(self, batch_dict, x_up1, x_up2, x_up3, x_up4, x_conv4):
        batch_dict["multi_scale_3d_features"] = dict(x_conv1=x_up2, x_conv2=x_up3, x_conv3=x_up4, x_conv4=x_conv4)
        batch_dict["conv_point_features"] = x_up1.features
        ds = batch_dict.get("devox_search")
        if ds is not None and ds["indices"] is x_up1.indices:  # the output sites are the input sites: centres already computed
            if ds["event"] is not None:
                torch.cuda.current_stream(x_up1.indices.device).wait_event(ds["event"])
            batch_dict["conv_point_coords"] = ds["centers"]
        else:
            batch_dict.pop("devox_search", None)
            batch_dict["conv_point_coords"] = ops.voxel_centers(x_up1.indices, self.voxel_size, self.point_cloud_range, n_dev=x_up1.n_dev)
        # extra keys (not in the reference): integer lattice coordinates + geometry of the output voxels, which let
        # the point heads use the grid-accelerated exact 3-NN instead of the O(N*V) scan
        batch_dict["conv_point_indices"] = x_up1.indices
        batch_dict["voxel_geometry"] = (list(self.voxel_size), list(self.point_cloud_range))
        return batch_dict
