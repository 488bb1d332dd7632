"""Weight packing for the HIP kernels: reference-layout parameters (nn.Linear (out,in), spconv
(kD,kH,kW,Cin,Cout), BatchNorm running stats) -> GEMM-ready device buffers.

ls3d_gather_gemm wants W[kvol][cin_pad][roundup(cout,32)] (input-major, zero padded) and a per-column
scale/shift epilogue; eval-mode BatchNorm and biases fold into that epilogue:
    BN(x W + b) = x W * s + ((b - mean) * s + beta),   s = gamma / sqrt(var + eps)
Packing happens once per weight version (see PackedModule), never per forward.
"""
import torch
from torch import nn


def _pad32(n):
    return (n + 31) // 32 * 32


def _pad16(n):
    return (n + 15) // 16 * 16


def fold_bn(bias, bn, cout, dev):
    """-> (scale, shift) float32 [cout] or (None, bias)"""
    if bn is None:
        return None, (bias.detach().float().contiguous() if bias is not None else None)
    s = bn.weight.detach().double() / torch.sqrt(bn.running_var.detach().double() + bn.eps)
    b = bias.detach().double() if bias is not None else torch.zeros(cout, dtype=torch.float64, device=dev)
    t = (b - bn.running_mean.detach().double()) * s + bn.bias.detach().double()
    return s.float().contiguous(), t.float().contiguous()


class PackedWeight(object):
    """kernel-layout weights (ls3d_gather_gemm_pack).  The layout depends on the column-block count `nt` the launch
    uses (chosen from the row count, ops.choose_nt), so packed copies are made lazily per nt and cached."""
    __slots__ = ("plain", "kvol", "cin_src", "cin", "cout", "_by_nt")

    def __init__(self, plain, kvol, cin_src, cin_pad, cout):
        self.plain, self.kvol, self.cin_src, self.cin, self.cout = plain, kvol, cin_src, cin_pad, cout
        self._by_nt = {}

    def for_nt(self, nt, precision=0):
        d = self._by_nt.get((nt, precision))
        if d is None:
            from . import ops
            d = self._by_nt[(nt, precision)] = ops.gather_gemm_pack(self.plain, self.kvol, self.cin_src, self.cin, self.cout, nt,
                                                                     precision)
        return d

    def for_tile(self, bf16=False):
        """layout of ls3d_tile_conv (16-channel chunks): three bf16 planes, or bf16=True the head plane only (products = 1)"""
        key = "tile_bf16" if bf16 else "tile"
        d = self._by_nt.get(key)
        if d is None:
            from . import ops
            d = self._by_nt[key] = ops.tile_conv_pack(self.plain, self.kvol, self.cin_src, self.cin, self.cout, bf16=bf16)
        return d

    @property
    def shape(self):  # (kvol, cin_pad, cout_pad) — what callers size their inputs against
        return (self.kvol, self.cin, _pad32(self.cout))


def _pack(plain, kvol, cin, cin_pad, cout):
    return PackedWeight(plain.contiguous(), kvol, cin, cin_pad, cout)


def pack_linear(weight, bias=None, bn=None, cin_pad=None):
    """nn.Linear / Conv1d(k=1) weight (out,in[,1]) -> (PackedWeight, scale, shift, cout)"""
    w = weight.detach().float()
    if w.dim() == 3:
        w = w.squeeze(-1)
    cout, cin = w.shape
    cin_pad = cin_pad or _pad16(cin)
    scale, shift = fold_bn(bias, bn, cout, w.device)
    return _pack(w.t(), 1, cin, cin_pad, cout), scale, shift, cout


def pack_spconv(weight, bn=None, cin_pad=None):
    """spconv weight (kD,kH,kW,Cin,Cout) -> (W[kvol,cin_pad,cout_pad], scale, shift, cout)"""
    w = weight.detach().float()
    cin, cout = w.shape[-2], w.shape[-1]
    kvol = w.numel() // (cin * cout)
    cin_pad = cin_pad or _pad16(cin)
    scale, shift = fold_bn(None, bn, cout, w.device)
    return _pack(w.reshape(kvol, cin, cout), kvol, cin, cin_pad, cout), scale, shift, cout


class PackedModule(nn.Module):
    """nn.Module whose kernel-ready buffers are rebuilt lazily after anything that can change the weights:
    load_state_dict, .to()/.cuda() (via _apply), train()/eval()."""

    def __init__(self):
        super().__init__()
        self._packed = None
        self.register_load_state_dict_post_hook(lambda m, keys: m.invalidate_packed())

    def invalidate_packed(self):
        for m in self.modules():
            if isinstance(m, PackedModule):
                m._packed = None

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._packed = None
        return r

    def train(self, mode=True):
        self._packed = None
        return super().train(mode)

    def packed(self):
        if self._packed is None:
            with torch.no_grad():
                self._packed = self._pack()
        return self._packed

    def _pack(self):
        raise NotImplementedError

    def _require_eval(self):
        if self.training:
            raise NotImplementedError(
                "%s: the HIP path implements the inference forward (eval-mode BatchNorm); the training step is the "
                "next row of the scope table (SURVEY.md §8f rank 1). Call .eval()." % type(self).__name__)
