"""A `spconv`-v1-shaped namespace (SparseConvTensor, SubMConv3d, SparseConv3d, SparseInverseConv3d,
SparseSequential, SparseModule) backed by libls3d.

The reference builds its backbone from third-party spconv v1.x @ fad3000 (det3d/models/backbones/scn_unet.py:3,
15-24,34,205; docs/INSTALL.md:88-99).  Module constructor signatures, the (kD,kH,kW,Cin,Cout) weight layout
and the `indice_key` rulebook sharing are kept so that model code written against spconv v1 and its
checkpoints work unchanged; the arithmetic is the output-stationary gather-GEMM of csrc/spconv.hip over the
output-major rulebooks of csrc/rulebook.hip (semantics: SURVEY.md §2.3)."""
import math
import os

import numpy as np
import torch
from torch import nn

from . import ops
from .packing import PackedModule, pack_spconv


def _triple(v):
    return tuple(int(x) for x in v) if isinstance(v, (tuple, list)) else (int(v),) * 3


# mask-sorted row order (ops.rulebook_order) for the gather-GEMM layers with cin * cout >= this: only where the matrix work repays the
# sort (>= 64 x 64 channels).  tools/bench_layers.py: sorted, the 32 -> 64 / 64 -> 32 strided layers of the 120k frame take 61 / 44 us
# instead of 124 / 87 (~4 offsets per tile instead of ~27), but their two extra 4-pass sorts sit on the geometry stream in front of
# the level-2 plan and the frame does not get shorter (LS3D_ORDER_MIN_CC=0: 6.08 vs 6.08 ms of convolutions) - left off.
ORDER_MIN_CC = 4096
# weight gradients of the layers that share a table run on one set of pair lists (False: each layer builds its own)
CACHE_PAIRS = True


class SparseConvTensor(object):
    """features [V,C] f32, indices [V,4] int32 (batch,z,y,x), spatial_shape (Z,Y,X), batch_size.
    `indice_dict` is shared by reference by every tensor derived from this one (as in spconv)."""

    def __init__(self, features, indices, spatial_shape, batch_size, grid=None, n_dev=None):
        self.features = features
        self.indices = indices if indices.dtype == torch.int32 else indices.int()
        self.spatial_shape = [int(v) for v in spatial_shape]
        self.batch_size = int(batch_size)
        self.indice_dict = {}
        self.grid = grid
        # capacity mode (not in spconv): n_dev = device int32 with the number of VALID rows; features / indices then have a
        # capacity's worth of rows and every kernel stops at the count - no host synchronisation for tensor shapes
        self.n_dev = n_dev

    def _like(self, features, indices=None, spatial_shape=None, n_dev="same"):
        t = SparseConvTensor(features, self.indices if indices is None else indices,
                             self.spatial_shape if spatial_shape is None else spatial_shape, self.batch_size,
                             n_dev=self.n_dev if isinstance(n_dev, str) else n_dev)
        t.indice_dict = self.indice_dict
        return t

    def find_indice_pair(self, key):
        return self.indice_dict.get(key) if key is not None else None

    def dense(self, channels_first=True):
        """scatter to a dense [B,C,Z,Y,X] (or [B,Z,Y,X,C]) tensor — torch indexing, for inspection only"""
        b, (z, y, x) = self.batch_size, self.spatial_shape
        out = torch.zeros((b, z, y, x, self.features.shape[1]), dtype=self.features.dtype, device=self.features.device)
        i = self.indices.long()
        out[i[:, 0], i[:, 1], i[:, 2], i[:, 3]] = self.features
        return out.permute(0, 4, 1, 2, 3).contiguous() if channels_first else out


class SparseModule(nn.Module):
    """marker base class: SparseSequential hands these the SparseConvTensor, anything else the feature matrix"""


class _Rulebook(object):
    __slots__ = ("kind", "tbl", "tbl_inv", "in_indices", "in_shape", "out_indices", "out_shape", "index", "_orders", "_plans", "_pairs", "batch_size",
                 "n_in_dev", "n_out_dev", "overflow_dev", "conv_geom")

    def rows_dev(self, inverse):
        """device count of the rows of the (inverse) table = the output rows of a launch on it (None: the table has no spare rows)"""
        return getattr(self, "n_in_dev", None) if inverse else getattr(self, "n_out_dev", None)

    def tile_plan(self, inverse, width=0):
        """tile-halo plan (ops.tile_plan) of the (inverse) table, built once per rulebook.  width: output channels of the layer that asks first -
        with ops._TILE_COLOR = 1 the SubM tables of the >= 64-channel levels get the coloured halo layout (ops.tile_plan; the 16 / 32-channel
        level keeps the neighbour-mask order, whose per-wave offset skipping is worth more there)"""
        if getattr(self, "_plans", None) is None:
            self._plans = {}
        if inverse not in self._plans:
            tbl = self.tbl_inv if inverse else self.tbl
            sites, shape = (self.in_indices, self.in_shape) if inverse else (self.out_indices, self.out_shape)
            self._plans[inverse] = ops.tile_plan(tbl, sites[:tbl.shape[0]], shape, getattr(self, "batch_size", None) or 256,
                                                 n_dev=self.rows_dev(inverse), color=(self.kind == "subm" and not inverse and width >= 64))
        return self._plans[inverse]

    def parity_geom(self, inverse):
        """(ksize, stride, padding) when the rows of the (inverse) table can be ordered by the residue class of their coordinates instead of
        their 27-bit masks (ops.rulebook_parity_orders: the transposed table of a strided convolution), else None"""
        g = getattr(self, "conv_geom", None)
        return g if (inverse and self.kind == "conv" and g is not None and ops._PARITY_ORDER) else None

    def order(self, inverse):
        """processing order of the (inverse) table - rows that share their empty offsets side by side, densest first - built once per rulebook"""
        if getattr(self, "_orders", None) is None:
            self._orders = {}
        if inverse not in self._orders:
            o = None
            if self.parity_geom(inverse) is not None:
                o = ops.rulebook_parity_orders([self.in_indices[:self.tbl_inv.shape[0]]], [self.conv_geom], [self.rows_dev(True)])[0]
            if o is None:
                o = ops.rulebook_order(self.tbl_inv if inverse else self.tbl, self.in_indices if inverse else self.out_indices,
                                       n_dev=self.rows_dev(inverse))
            self._orders[inverse] = o
        return self._orders[inverse]

    def order_for(self, inverse, cin_cout):
        """the order a layer of cin x cout channels processes the (inverse) table in: one that exists already (prebuild_orders, an earlier
        layer), the cheap coordinate-class order of a transposed strided table, or - where the matrix work repays four sort passes - the mask order"""
        o = (getattr(self, "_orders", None) or {}).get(inverse)
        if o is not None:
            return o
        return self.order(inverse) if (cin_cout >= ORDER_MIN_CC or self.parity_geom(inverse) is not None) else None

    def pairs(self, inverse, ordered):
        """compacted pair lists of the (inverse) table for the weight gradients (ops.spconv_pairs), built once per rulebook: every
        layer of an indice_key shares them.  Dropped with the rulebook, so a training step holds them only until its backward ends."""
        if getattr(self, "_pairs", None) is None:
            self._pairs = {}
        key = (inverse, ordered)
        if key not in self._pairs:
            self._pairs[key] = ops.spconv_pairs(self.tbl_inv if inverse else self.tbl, self.order(inverse) if ordered else None)
        return self._pairs[key]


class _SparseConvFn(torch.autograd.Function):
    """Differentiable sparse convolution (the training path; inference uses the packed, epilogue-fused launches).
    forward: out = gather_gemm(feats, W, tbl) (+ bias); backward: grad_feats = the same gather-GEMM on the transposed table
    with W^T (SubM: mirrored offsets), grad_W = ops.spconv_wgrad (include/ls3d.h: "Backward of the sparse convolutions")."""

    @staticmethod
    def forward(ctx, feats, weight, bias, rb, inverse, subm):
        cin, cout = weight.shape[-2], weight.shape[-1]
        W = pack_spconv(weight)[0]
        tbl = rb.tbl_inv if inverse else rb.tbl
        order = rb.order_for(inverse, cin * cout)
        x = feats.detach().contiguous()
        if x.shape[1] != W.shape[1]:
            x = torch.nn.functional.pad(x, (0, W.shape[1] - x.shape[1]))
        shift = None if bias is None else bias.detach()
        if subm and ops.use_tile("subm", tbl.shape[1], W.shape[1], cout):  # the f32-grade 3-plane modes: SubM layers on the tile-halo kernel
            out = ops.tile_conv(x, W, rb.tile_plan(False, cout), cout=cout, shift=shift)
        else:
            out = ops.gather_gemm(x, W, tbl=tbl, order=order, cout=cout, shift=shift)
        ctx.save_for_backward(feats, weight)
        ctx.rb, ctx.inverse, ctx.subm, ctx.has_bias = rb, inverse, subm, bias is not None
        return out

    @staticmethod
    def backward(ctx, gout):
        from .packing import PackedWeight, _pad16
        feats, weight = ctx.saved_tensors
        rb, inverse, subm = ctx.rb, ctx.inverse, ctx.subm
        cin, cout = weight.shape[-2], weight.shape[-1]
        kvol = weight.numel() // (cin * cout)
        gout = gout.contiguous()
        gin = gw = gb = None
        if ctx.needs_input_grad[0]:
            wd = weight.detach().reshape(kvol, cin, cout).transpose(1, 2)  # [kvol, cout, cin]: grad_out -> grad_in
            if subm:
                wd = wd.flip(0)  # input i sees output o through the mirrored offset
            Wd = PackedWeight(wd.contiguous(), kvol, cout, _pad16(cout), cin)
            if subm:
                tbl_t, order_t = rb.tbl, rb.order_for(False, cin * cout)
            else:
                tbl_t = rb.tbl if inverse else rb.tbl_inv
                order_t = rb.order_for(not inverse, cin * cout)
            g = gout
            if g.shape[1] != Wd.shape[1]:
                g = torch.nn.functional.pad(g, (0, Wd.shape[1] - g.shape[1]))
            if subm and ops.use_tile("subm", kvol, Wd.shape[1], cin):  # same table, same plan as the forward
                gin = ops.tile_conv(g, Wd, rb.tile_plan(False, cin), cout=cin)
            else:
                gin = ops.gather_gemm(g, Wd, tbl=tbl_t, order=order_t, cout=cin)
        if ctx.needs_input_grad[1]:
            tbl = rb.tbl_inv if inverse else rb.tbl
            order = rb.order_for(inverse, cin * cout)
            pairs = rb.pairs(inverse, order is not None) if CACHE_PAIRS else None
            gw = ops.spconv_wgrad(feats.detach().contiguous(), gout, tbl, order, cin, cout, pairs=pairs).reshape(weight.shape)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = ops.column_sums(gout)
        return gin, gw, gb, None, None, None


class SparseConvolution(PackedModule, SparseModule):
    def __init__(self, ndim, in_channels, out_channels, kernel_size=3, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, subm=False, output_padding=0, transposed=False, inverse=False, indice_key=None,
                 fused_bn=False):
        super().__init__()
        assert ndim == 3 and groups == 1 and _triple(dilation) == (1, 1, 1), "3-D, groups=1, dilation=1 only"
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = _triple(kernel_size), _triple(stride), _triple(padding)
        self.subm, self.inverse, self.indice_key = subm, inverse, indice_key
        self.weight = nn.Parameter(torch.empty(*self.kernel_size, in_channels, out_channels))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in = self.weight.numel() // self.out_channels
            nn.init.uniform_(self.bias, -1 / math.sqrt(fan_in), 1 / math.sqrt(fan_in))

    def _pack(self):
        return {self._kvol(): pack_spconv(self.weight)}

    def _kvol(self):
        return self.kernel_size[0] * self.kernel_size[1] * self.kernel_size[2]

    def _weight_for(self, rb):
        """spconv v1 semantics when layers of DIFFERENT kernel shapes share an indice_key (Cylinder3D's blocks do: a (1,3,3), a
        (3,1,3) and a 3x3x3 SubMConv3d under one key): the pairs stored by the first layer are used as they are, with the first
        `stored kernel volume` filters of this layer's weight viewed as [-1, Cin, Cout] (spconv v1.2.1 ops.indice_conv: the kernel
        volume is the pair table's, the filters are `weight.view(-1, Cin, Cout)`).  None = this layer's own weight fits."""
        k = (rb.tbl_inv if self.inverse else rb.tbl).shape[1]
        if k == self._kvol():
            return None
        if k > self._kvol():
            raise ValueError("indice_key %r holds a rulebook of %d kernel offsets; this layer has only %d filters"
                             % (self.indice_key, k, self._kvol()))
        return self.weight.reshape(self._kvol(), self.in_channels, self.out_channels)[:k].reshape(k, 1, 1, self.in_channels, self.out_channels)

    # ---- rulebooks (shared through indice_key exactly like spconv's indice_dict)
    def rulebook(self, x):
        rb = x.find_indice_pair(self.indice_key)
        if self.inverse:
            assert rb is not None and rb.kind == "conv", "SparseInverseConv3d needs the rulebook of the SparseConv3d " \
                "with indice_key=%r" % (self.indice_key,)
            return rb
        if rb is not None and (self.subm or rb.kind == "conv"):
            return rb  # spconv semantics: layers with one indice_key share the pairs (also prebuild_conv_rulebooks below)
        if self.subm:
            rb = subm_rulebook(x.indices, x.spatial_shape, self.kernel_size, x.batch_size, n_dev=x.n_dev)
        elif x.n_dev is not None:  # capacity mode: no host sync, worst-case output capacity (cannot overflow)
            assert self.indice_key is not None, "capacity mode needs an indice_key on strided convolutions"
            prebuild_conv_rulebooks(x, [self], nosync=True)
            return x.find_indice_pair(self.indice_key)
        else:
            rb = _Rulebook()
            rb._orders = rb._plans = rb._pairs = None
            rb.batch_size = x.batch_size
            rb.n_in_dev = rb.n_out_dev = rb.overflow_dev = None
            rb.in_indices, rb.in_shape = x.indices, list(x.spatial_shape)
            rb.kind = "conv"
            oc, cnt, nbr_out, nbr_inv, oshape = ops.rulebook_conv(x.indices, x.batch_size, x.spatial_shape,
                                                                   self.kernel_size, self.stride, self.padding)
            n_out, overflow = (int(v) for v in cnt.tolist())  # host sync: tensor shapes need the count
            assert not overflow
            rb.out_indices, rb.out_shape = oc[:n_out], oshape
            rb.tbl, rb.tbl_inv = nbr_out[:n_out], nbr_inv
            rb.conv_geom = (_triple(self.kernel_size), _triple(self.stride), _triple(self.padding))
        if self.indice_key is not None:
            x.indice_dict[self.indice_key] = rb
        return rb

    def conv(self, x, rb, scale=None, shift=None, relu=False, res_pre=None, pair=None, out=None, out_ld=None):
        """the gather-GEMM with a fused epilogue; x: SparseConvTensor or a feature matrix on rb's input sites"""
        feats = x.features if isinstance(x, SparseConvTensor) else x
        pk, k = self.packed(), (rb.tbl_inv if self.inverse else rb.tbl).shape[1]
        if k not in pk:
            with torch.no_grad():
                pk[k] = pack_spconv(self._weight_for(rb))
        W, _, _, cout = pk[k]
        if self.bias is not None:
            b = self.bias.detach()
            shift = b if shift is None else shift + (b * (scale if scale is not None else 1.0))
        tbl = rb.tbl_inv if self.inverse else rb.tbl
        if feats.shape[1] != W.shape[1]:  # e.g. 13 input channels feeding a 16-wide K chunk
            feats = torch.nn.functional.pad(feats, (0, W.shape[1] - feats.shape[1]))
        if ops.use_tile("inverse" if self.inverse else rb.kind, k, W.shape[1], cout) and tbl.shape[0] > 0:
            return ops.tile_conv(feats.contiguous(), W, rb.tile_plan(bool(self.inverse), cout), cout=cout, scale=scale, shift=shift, relu=relu,
                                 res_pre=res_pre, pair=pair, out=out, out_ld=out_ld)
        # mask-sorted processing order only for the layers whose matrix work can repay the sort (>= 64x64 channels) - or whose order exists / is cheap
        order = rb.order_for(bool(self.inverse), self.in_channels * self.out_channels)
        return ops.gather_gemm(feats.contiguous(), W, tbl=tbl, order=order, cout=cout, scale=scale, shift=shift, relu=relu,
                               res_pre=res_pre, pair=pair, out=out, out_ld=out_ld, n_dev=rb.rows_dev(bool(self.inverse)))

    def forward(self, x):
        rb = self.rulebook(x)
        if needs_grad(self, x.features):
            w = self._weight_for(rb)
            f = _SparseConvFn.apply(x.features, self.weight if w is None else w, self.bias, rb, bool(self.inverse), bool(self.subm))
        else:
            f = self.conv(x, rb)
        if self.inverse:
            return x._like(f, rb.in_indices, rb.in_shape, n_dev=rb.rows_dev(True))
        return x._like(f, rb.out_indices, rb.out_shape, n_dev=rb.rows_dev(False))


def subm_rulebook(indices, spatial_shape, kernel_size, batch_size=None, n_dev=None):
    rb = _Rulebook()
    rb._orders = rb._plans = rb._pairs = None
    rb.batch_size = batch_size
    rb.kind = "subm"
    rb.n_in_dev = rb.n_out_dev = n_dev
    rb.overflow_dev = None
    rb.in_indices, rb.in_shape = indices, list(spatial_shape)
    rb.tbl = ops.rulebook_subm(indices, spatial_shape, _triple(kernel_size), n_dev=n_dev)
    rb.out_indices, rb.out_shape, rb.tbl_inv = indices, list(spatial_shape), None
    return rb


def prebuild_orders(x, layers):
    """mask-sorted processing orders of every rulebook table the given layers will ask for (same criterion as
    SparseConvolution.conv), from ONE batched sort instead of one sort per table; tile-halo plans for the layers that
    take that path"""
    want, want_parity = [], []
    for m in layers:
        if not isinstance(m, SparseConvolution):
            continue
        rb = x.find_indice_pair(m.indice_key)
        if rb is None:
            continue
        if ops.use_tile("inverse" if m.inverse else rb.kind, (rb.tbl_inv if m.inverse else rb.tbl).shape[1], (m.in_channels + 15) // 16 * 16,
                        m.out_channels):
            rb.tile_plan(bool(m.inverse), m.out_channels)
            continue
        inv = bool(m.inverse)
        cheap = rb.parity_geom(inv) is not None
        if m.in_channels * m.out_channels < ORDER_MIN_CC and not cheap:
            continue
        if rb._orders is None:
            rb._orders = {}
        if inv in rb._orders or any(r is rb and i == inv for r, i in want + want_parity):
            continue
        (want_parity if cheap else want).append((rb, inv))
    if want_parity:  # transposed strided tables: ordered by the residue class of their input coordinates, one radix pass for all of them
        got = ops.rulebook_parity_orders([rb.in_indices[:rb.tbl_inv.shape[0]] for rb, _ in want_parity], [rb.conv_geom for rb, _ in want_parity],
                                         [rb.rows_dev(True) for rb, _ in want_parity])
        for (rb, inv), o in zip(want_parity, got):
            if o is None:
                want.append((rb, inv))
            else:
                rb._orders[inv] = o
    if want:
        for (rb, inv), o in zip(want, ops.rulebook_orders([rb.tbl_inv if inv else rb.tbl for rb, inv in want], [rb.rows_dev(inv) for rb, inv in want])):
            rb._orders[inv] = o


def prebuild_conv_rulebooks(x, convs, coords=None, shape=None, nosync=False, caps=None, n_dev=None, after_each=None):
    """Rulebooks of a chain of strided SparseConv3d layers (each one's output sites are the next one's input sites, as in a
    UNet encoder) built back to back on device-side site counts, with ONE host synchronisation for all their sizes instead
    of one per layer.  Intermediate tables are allocated for the worst case (min(8 x inputs, grid cells)) and sliced once
    the counts are known.  The rulebooks are registered under the layers' indice_keys.  coords / shape: the input sites of the first layer
    when the chain does not start at x's own sites (a chain continued after an earlier call)."""
    if coords is None:
        coords, n_dev = x.indices, (x.n_dev if n_dev is None else n_dev)
    first_coords = coords
    pend, shape = [], list(x.spatial_shape if shape is None else shape)
    for li, c in enumerate(convs):
        assert not c.subm and not c.inverse and c.indice_key is not None
        oc, cnt, nbr_out, nbr_inv, oshape = ops.rulebook_conv(coords, x.batch_size, shape, c.kernel_size, c.stride, c.padding,
                                                               out_cap=(caps[li] if caps else None), n_dev=n_dev)
        pend.append((c, coords, shape, oc, cnt, nbr_out, nbr_inv, oshape, n_dev))
        coords, n_dev, shape = oc, cnt[0:1], oshape  # cnt = [n_out, overflow]
        if after_each is not None:
            after_each()  # e.g. an event per layer: consumers of this layer's sites need not wait for the rest of the chain
    if nosync:
        # capacity mode: every table keeps its capacity's worth of rows, the counts stay on the device ([n_out, overflow] per layer);
        # the caller checks the overflow flags once per frame (UNetSCN3D.geometry_record)
        for c, icoords, ishape, oc, cnt, nbr_out, nbr_inv, oshape, n_in_dev in pend:
            rb = _Rulebook()
            rb._orders = rb._plans = rb._pairs = None
            rb.batch_size = x.batch_size
            rb.kind, rb.in_indices, rb.in_shape = "conv", icoords, list(ishape)
            rb.out_indices, rb.out_shape = oc, oshape
            rb.tbl, rb.tbl_inv = nbr_out, nbr_inv
            rb.n_in_dev, rb.n_out_dev, rb.overflow_dev = n_in_dev, cnt[0:1], cnt[1:2]
            rb.conv_geom = (_triple(c.kernel_size), _triple(c.stride), _triple(c.padding))
            x.indice_dict[c.indice_key] = rb
        return
    pend = [p[:8] for p in pend]
    counts = torch.stack([p[4] for p in pend]).tolist()  # host sync: tensor shapes need the counts
    n_in = first_coords.shape[0]
    for (c, icoords, ishape, oc, cnt, nbr_out, nbr_inv, oshape), (n_out, overflow) in zip(pend, counts):
        assert not overflow
        rb = _Rulebook()
        rb._orders = rb._plans = rb._pairs = None
        rb.batch_size = x.batch_size
        rb.n_in_dev = rb.n_out_dev = rb.overflow_dev = None
        rb.kind, rb.in_indices, rb.in_shape = "conv", (icoords if n_in == icoords.shape[0] else icoords[:n_in]), list(ishape)
        rb.out_indices, rb.out_shape = oc[:n_out], oshape
        rb.tbl, rb.tbl_inv = nbr_out[:n_out], nbr_inv[:n_in]
        rb.conv_geom = (_triple(c.kernel_size), _triple(c.stride), _triple(c.padding))
        x.indice_dict[c.indice_key] = rb
        n_in = n_out


class SubMConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None, **kw):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, subm=True,
                         indice_key=indice_key)


class SparseConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None, **kw):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias,
                         indice_key=indice_key)


class SparseInverseConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, indice_key=None, bias=True, **kw):
        super().__init__(3, in_channels, out_channels, kernel_size, bias=bias, inverse=True, indice_key=indice_key)


class SparseSequential(SparseModule):
    """spconv.SparseSequential: sparse modules get the tensor, plain modules get `.features` (only when the
    tensor is non-empty).  The pattern (sparse conv, BatchNorm1d[, ReLU]) in eval mode is fused into the conv's
    epilogue instead of running BN/ReLU as separate passes."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        if len(args) == 1 and isinstance(args[0], dict):
            for k, m in args[0].items():
                self.add_module(k, m)
        else:
            for i, m in enumerate(args):
                self.add_module(str(i), m)
        for k, m in kwargs.items():
            self.add_module(k, m)

    def __getitem__(self, i):
        return list(self._modules.values())[i]

    def __len__(self):
        return len(self._modules)

    def forward(self, x):
        mods = list(self._modules.values())
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, SparseConvolution) and i + 1 < len(mods) and isinstance(mods[i + 1], nn.BatchNorm1d) \
                    and not mods[i + 1].training:
                relu = i + 2 < len(mods) and isinstance(mods[i + 2], nn.ReLU)
                x = conv_bn_act(m, mods[i + 1], x, relu=relu)
                i += 3 if relu else 2
            elif isinstance(m, nn.BatchNorm1d) and m.training and i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU) and x.indices.shape[0] != 0:
                # training: BatchNorm (batch statistics) + ReLU as one pair of kernels forward and backward (csrc/norm.hip)
                y = ops.batch_norm_train(m, x.features, relu=True) if (x.features.is_cuda or ops.sim_mode()) else None
                x.features = y if y is not None else mods[i + 1](m(x.features))
                i += 2
            elif isinstance(m, SparseModule):
                x = m(x)
                i += 1
            else:
                if x.indices.shape[0] != 0:
                    x.features = m(x.features)
                i += 1
        return x


def needs_grad(conv, feats):
    """the epilogue-fused launches are not differentiable: while autograd records and either the input carries a gradient (input
    gradients in eval mode) or the layer is being trained (frozen-BN fine-tuning: bn.eval() inside a model in train mode), the
    caller takes the differentiable composition (conv through _SparseConvFn, then torch BatchNorm / ReLU)"""
    return torch.is_grad_enabled() and (feats.requires_grad or (conv.training and conv.weight.requires_grad))


def bn_scale_shift(bn):
    s = bn.weight.detach().double() / torch.sqrt(bn.running_var.detach().double() + bn.eps)
    t = bn.bias.detach().double() - bn.running_mean.detach().double() * s
    return s.float().contiguous(), t.float().contiguous()


def cached_bn_scale_shift(conv, bn):
    """folded eval-BN of the BatchNorm that follows `conv`, recomputed only when the BN tensors change"""
    cache = conv.__dict__.setdefault("_bn_cache", {})
    key = (id(bn), bn.weight._version, bn.bias._version, bn.running_mean._version, bn.running_var._version,
           bn.weight.data_ptr())
    if cache.get("key") != key:
        cache["key"], cache["ss"] = key, bn_scale_shift(bn)
    return cache["ss"]


def conv_bn_act(conv, bn, x, relu=True, res_pre=None, pair=None):
    """SparseSequential(conv, BN(eval), ReLU) as ONE kernel launch (or, when a gradient must flow, its differentiable composition)"""
    if needs_grad(conv, x.features):
        y = conv(x)
        f = bn(y.features)
        if res_pre is not None:
            f = f + res_pre
        if relu:
            f = torch.relu(f)
        if pair is not None:
            f = f + pair.view(pair.shape[0], f.shape[1], -1).sum(dim=2)
        y.features = f
        return y
    scale, shift = cached_bn_scale_shift(conv, bn)
    rb = conv.rulebook(x)
    f = conv.conv(x, rb, scale=scale, shift=shift, relu=relu, res_pre=res_pre, pair=pair)
    if conv.inverse:
        return x._like(f, rb.in_indices, rb.in_shape, n_dev=rb.rows_dev(True))
    return x._like(f, rb.out_indices, rb.out_shape, n_dev=rb.rows_dev(False))
