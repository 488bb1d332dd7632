"""Thin torch-tensor front end of the C ABI (include/ls3d.h).

torch is used for device memory and streams only: every function below allocates its outputs with torch,
passes raw device pointers + the current HIP stream to libls3d.so, and returns torch tensors.  There is no
computation in Python and no fallback: tensors must live on the GPU (the only exception is the test hook
``_lib.use_library_for_testing`` which swaps in tests/hipsim's host build of the same kernels and then
requires CPU tensors).
"""
import os as _os
import sys as _sys
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import Epilogue, Grid, PointsLayout, check

_i32 = torch.int32
_SIM = False


def _L():
    return _lib.load()


class CapacityModeUnsupported(RuntimeError):
    """raised by a stage that cannot run on device-side row counts; the detector then runs the frame with host-side counts"""


def set_sim(flag):
    """tests only — see _lib.use_library_for_testing"""
    global _SIM
    _SIM = bool(flag)


def _ptr(t):
    if t is None:
        return None
    if _SIM:
        if t.is_cuda:
            raise RuntimeError("hipsim test hook active: CPU tensors only")
    elif not t.is_cuda:
        raise RuntimeError("lidarseg3d_amd ops need tensors on the MI355X (got a %s tensor); "
                           "there is no CPU fallback" % t.device)
    if not t.is_contiguous():
        raise RuntimeError("non-contiguous tensor passed to libls3d")
    return ctypes.c_void_p(t.data_ptr())


def _stream(t):
    if t.is_cuda:
        return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
    return ctypes.c_void_p(0)


def _i3(v):
    return (ctypes.c_int32 * 3)(*[int(x) for x in v])


def _f3(v):
    return (ctypes.c_float * 3)(*[float(np.float32(x)) for x in v])


def make_grid(voxel_size, pc_range):
    """grid = round((hi-lo)/vs) in f32, as point_cloud_ops.py:26-29 / voxelization_cpu.cpp:118-121."""
    vs = np.asarray(voxel_size, np.float32)
    r = np.asarray(pc_range, np.float32)
    g = np.round((r[3:] - r[:3]) / vs).astype(np.int64)
    return Grid(_f3(vs), _f3(r[:3]), _i3(g)), [int(x) for x in g]


_REGISTERED_HOST = "hip"   # measurement hook (LS3D_EXPERIMENT): "heap" = hipHostRegister on the tensor's own malloc'ed bytes (round 6's first form: a long-lived
#                            userptr mapping of a heap page, profiles/round6_experiments.md 5), "" = the caching host allocator's pinned memory
_HOST_SLABS = []           # [tensor over a whole hipHostMalloc'ed slab, bytes handed out]; kept until the process ends
_HOST_SLAB_BYTES = 1 << 16
_HIPRT = None


def _hip_runtime():
    """the HIP runtime this process already runs on (the one libtorch_hip.so brought in), for the two calls torch does not expose"""
    global _HIPRT
    if _HIPRT is None:
        path = None
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64.so" in line:
                    path = line.split()[-1]
                    break
        if path is None:
            raise RuntimeError("the HIP runtime is not loaded in this process")
        _HIPRT = ctypes.CDLL(path)
        _HIPRT.hipHostMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
        _HIPRT.hipHostMalloc.restype = ctypes.c_int
    return _HIPRT


def registered_host(shape, dtype):
    """page-locked host tensor that PyTorch's caching host allocator does not own: a 256-byte slice of a slab from hipHostMalloc that lives until the
    process ends (a count buffer is a few dozen bytes).  For the count buffers a CAPTURED frame copies into (scn_unet.UNetSCN3D, graph.FrameGraph):
    a non_blocking copy into a torch.empty(pin_memory=True) tensor makes the caching host allocator record one of its pooled events on the stream -
    inside a capture that becomes an event-record node of the hipGraph whose event outlives the graph.  Driver-allocated host memory is mapped for
    the copy engine once and for good: no user-pointer mapping the kernel would have to revalidate whenever it touches the page (what registering a
    malloc'ed tensor with hipHostRegister created; with it the -m gpu suite died of a GPU memory fault inside one of torch's pageable H2D copies
    in 3 of 16 runs, never before or after: profiles/round6_experiments.md 5)."""
    if not torch.cuda.is_available():
        return torch.empty(shape, dtype=dtype)
    if not _REGISTERED_HOST:
        return torch.empty(shape, dtype=dtype, pin_memory=True)
    n = 1
    for d in (shape if isinstance(shape, (tuple, list, torch.Size)) else (shape,)):
        n *= int(d)
    nbytes = max(n * torch.empty((), dtype=dtype).element_size(), 1)
    if _REGISTERED_HOST == "heap":
        import weakref
        rt = torch.cuda.cudart()
        t = torch.empty(shape, dtype=dtype)
        err = rt.cudaHostRegister(t.data_ptr(), nbytes, 0)
        if int(err) != 0:
            raise RuntimeError("hipHostRegister failed: %r" % (err,))
        weakref.finalize(t.untyped_storage(), rt.cudaHostUnregister, t.data_ptr())
        return t
    need = (nbytes + 255) & ~255
    if not _HOST_SLABS or _HOST_SLABS[-1][1] + need > _HOST_SLABS[-1][0].numel():
        torch.cuda.current_device()   # the runtime is initialised and this thread has its device
        size = max(_HOST_SLAB_BYTES, (need + 4095) & ~4095)
        ptr = ctypes.c_void_p()
        check(_hip_runtime().hipHostMalloc(ctypes.byref(ptr), size, 0), "hipHostMalloc")
        whole = torch.frombuffer((ctypes.c_uint8 * size).from_address(ptr.value), dtype=torch.uint8)
        whole.zero_()
        _HOST_SLABS.append([whole, 0])
    slab = _HOST_SLABS[-1]
    t = slab[0][slab[1]:slab[1] + nbytes].view(dtype).view(shape)
    slab[1] += need
    return t


def stamp(buf, i):
    """diagnostics: buf[i] (uint64 viewed as int64, on the GPU) = the 100 MHz wall clock when the current stream gets here (ls3d_stamp)"""
    check(_L().ls3d_stamp(ctypes.c_void_p(buf.data_ptr() + 8 * int(i)), _stream(buf)), "ls3d_stamp")


def _ws(nbytes, like):
    return torch.empty((max(int(nbytes), 256),), dtype=torch.uint8, device=like.device)


# ---------------------------------------------------------------------------------------------- voxelization
def voxelize_dynamic(points, voxel_size, pc_range, xyz_col=0):
    n = points.shape[0]
    grid, _ = make_grid(voxel_size, pc_range)
    lay = PointsLayout(points.shape[1], xyz_col, -1, xyz_col, points.shape[1] - xyz_col)
    coors = torch.empty((n, 3), dtype=_i32, device=points.device)
    check(_L().ls3d_voxelize_dynamic(_ptr(points), n, ctypes.byref(lay), ctypes.byref(grid), _ptr(coors),
                                     _stream(points)), "ls3d_voxelize_dynamic")
    return coors


def voxelize_hard(points, voxel_size, pc_range, max_points, max_voxels, overflow="numba", batched=False):
    """points [N,C] (one frame) or, batched=True, [N,1+C] with the batch index in column 0.
    -> voxels[cap,max_points,C], coors[cap,3|4], num_points[cap], num_voxels (1-elt device tensor).
    Rows beyond num_voxels are undefined; slice after reading the count (one host sync)."""
    n = points.shape[0]
    grid, _ = make_grid(voxel_size, pc_range)
    off = 1 if batched else 0
    c = points.shape[1] - off
    lay = PointsLayout(points.shape[1], off, 0 if batched else -1, off, c)
    cap = max(min(n, int(max_voxels)), 1)
    dev = points.device
    voxels = torch.empty((cap, max_points, c), dtype=torch.float32, device=dev)
    cols = 4 if batched else 3
    coors = torch.empty((cap, cols), dtype=_i32, device=dev)
    num = torch.empty((cap,), dtype=_i32, device=dev)
    nv = torch.zeros((1,), dtype=_i32, device=dev)
    L = _L()
    ws = _ws(L.ls3d_voxelize_hard_workspace_bytes(n, max_points, int(max_voxels)), points)
    check(L.ls3d_voxelize_hard(_ptr(points), n, ctypes.byref(lay), ctypes.byref(grid), int(max_points), int(max_voxels),
                               1 if overflow == "break" else 0, _ptr(ws), ctypes.c_size_t(ws.numel()), _ptr(voxels),
                               _ptr(coors), cols, _ptr(num), _ptr(nv), _stream(points)), "ls3d_voxelize_hard")
    return voxels, coors, num, nv


def dynamic_scatter(feats, coors, shape_zyx, mode="mean"):
    n, c = feats.shape
    cols = coors.shape[1]
    dev = feats.device
    out = torch.empty((max(n, 1), c), dtype=torch.float32, device=dev)
    vc = torch.empty((max(n, 1), cols), dtype=_i32, device=dev)
    p2v = torch.empty((max(n, 1),), dtype=_i32, device=dev)
    nv = torch.zeros((1,), dtype=_i32, device=dev)
    L = _L()
    ws = _ws(L.ls3d_dynamic_scatter_workspace_bytes(n), feats)
    check(L.ls3d_dynamic_scatter(_ptr(feats), n, c, _ptr(coors), cols, _i3(shape_zyx), {"mean": 0, "max": 1, "sum": 2}[mode],
                                 _ptr(ws), ctypes.c_size_t(ws.numel()), _ptr(out), _ptr(vc), _ptr(p2v), _ptr(nv),
                                 _stream(feats)), "ls3d_dynamic_scatter")
    return out, vc, p2v, nv


def dynamic_scatter_backward(grad_voxels, p2v, feats_in, feats_out, mode="mean"):
    """grad of dynamic_scatter's `out` rows w.r.t. the points (include/ls3d.h: ls3d_dynamic_scatter_backward)"""
    n, c = feats_in.shape
    gp = torch.empty((n, c), dtype=torch.float32, device=feats_in.device)
    L = _L()
    ws = _ws(L.ls3d_dynamic_scatter_backward_workspace_bytes(n, c), feats_in)
    check(L.ls3d_dynamic_scatter_backward(_ptr(grad_voxels), _ptr(p2v), n, c, 0 if mode == "mean" else 1, _ptr(feats_in), _ptr(feats_out),
                                          _ptr(ws), ctypes.c_size_t(ws.numel()), _ptr(gp), _stream(feats_in)),
          "ls3d_dynamic_scatter_backward")
    return gp


def dynamic_point_to_voxel_index(voxel_mapping, shape_zyx):
    """voxel_mapping [n, 3 or 4] int32 -> point_to_voxelidx [n], coor_to_voxelidx [n], num_points_per_voxel [n], voxel_coors [n, cols],
    counts [2] = (voxel_num, max_points) on the device (include/ls3d.h: ls3d_dynamic_point_to_voxel_index)"""
    n, cols = voxel_mapping.shape
    dev = voxel_mapping.device
    p2v, c2v, num = (torch.empty((max(n, 1),), dtype=_i32, device=dev) for _ in range(3))
    vc = torch.empty((max(n, 1), cols), dtype=_i32, device=dev)
    counts = torch.zeros((2,), dtype=_i32, device=dev)
    L = _L()
    ws = _ws(L.ls3d_dynamic_point_to_voxel_workspace_bytes(n), voxel_mapping)
    check(L.ls3d_dynamic_point_to_voxel_index(_ptr(voxel_mapping), n, cols, _i3(shape_zyx), _ptr(ws), ctypes.c_size_t(ws.numel()), _ptr(p2v), _ptr(c2v),
                                              _ptr(num), _ptr(vc), _ptr(counts), _stream(voxel_mapping)), "ls3d_dynamic_point_to_voxel_index")
    return p2v[:n], c2v[:n], num, vc, counts


def dynamic_point_to_voxel_forward(points, p2v, c2v, voxel_num, max_points):
    """-> voxels [voxel_num, max_points, C], zero padded (ls3d_dynamic_point_to_voxel_forward)"""
    n, c = points.shape
    voxels = torch.empty((int(voxel_num), int(max_points), c), dtype=torch.float32, device=points.device)
    check(_L().ls3d_dynamic_point_to_voxel_forward(_ptr(points), n, c, _ptr(p2v), _ptr(c2v), int(voxel_num), int(max_points), _ptr(voxels), _stream(points)),
          "ls3d_dynamic_point_to_voxel_forward")
    return voxels


def dynamic_point_to_voxel_backward(grad_points, grad_voxels, p2v, c2v):
    """in place: grad_points[i] = grad_voxels[c2v[i], p2v[i]] for the points inside a voxel (ls3d_dynamic_point_to_voxel_backward)"""
    n, c = grad_points.shape
    check(_L().ls3d_dynamic_point_to_voxel_backward(_ptr(grad_points), _ptr(grad_voxels), _ptr(p2v), _ptr(c2v), n, c, int(grad_voxels.shape[1]),
                                                   _stream(grad_points)), "ls3d_dynamic_point_to_voxel_backward")
    return grad_points


# ---------------------------------------------------------------------------------------------- dynamic readers, Cylinder3D tails (csrc/dynreader.hip)
_ACT = {None: 0, "none": 0, "relu": 1, "leaky": 2, "sigmoid": 3}


def act_affine(x, pre=None, post=None, slope=0.01, scale=None, shift=None, add=None, mul=None, out=None, n_dev=None):
    """out = post(pre(x) * scale + shift) [+ add] [* mul] row-wise (ls3d_act_affine); x / add / mul / out may be column views of wider buffers"""
    n, c = x.shape
    if out is None:
        out = torch.empty((n, c), dtype=torch.float32, device=x.device)
    for t in (x, add, mul, out):
        assert t is None or (t.stride(1) == 1 and t.shape == (n, c))
    check(_L().ls3d_act_affine(_vp_any(x), _ld(x), n, _ndev(n_dev), c, _ACT[pre], _ACT[post], ctypes.c_float(slope), _vp(scale), _vp(shift),
                               _vp_any(add) if add is not None else None, _ld(add) if add is not None else 0,
                               _vp_any(mul) if mul is not None else None, _ld(mul) if mul is not None else 0, _vp_any(out), _ld(out), _stream(x)),
          "ls3d_act_affine")
    return out


def tta_merge(logits, first_rows, n, want_probs=False):
    """mean over the variants (frames of n rows starting at first_rows[t]) of softmax(logits), argmax -> labels [n] int64 (, probs [n, C])  (ls3d_tta_merge)"""
    c = logits.shape[1]
    labels = torch.empty((n,), dtype=torch.int64, device=logits.device)
    probs = torch.empty((n, c), dtype=torch.float32, device=logits.device) if want_probs else None
    rows = (ctypes.c_int32 * len(first_rows))(*[int(v) for v in first_rows])
    check(_L().ls3d_tta_merge(_vp_any(logits), _ld(logits), c, int(n), rows, len(first_rows), _vp(probs), _ptr(labels), _stream(logits)), "ls3d_tta_merge")
    return (labels, probs) if want_probs else labels


def cyl_grid(grid_size, pc_range):
    """the readers' own cell size (range / grid in double, used as f32: voxel_encoder.py:319-323,549-553) as an ls3d_grid_t"""
    vs = [(float(pc_range[3 + i]) - float(pc_range[i])) / float(grid_size[i]) for i in range(3)]
    return Grid(_f3(vs), _f3(pc_range[:3]), _i3(grid_size)), vs


def cyl_voxelize(points, grid_size, pc_range, reverse, collapse_last, batch_size):
    """points [n, 1 + 3 + f] (batch, x, y, z, ...) -> cyl5 [n, 5] f32, vcoors [n, 4] int64, keys [n] (ls3d_cyl_voxelize)"""
    n = points.shape[0]
    dev = points.device
    grid, _ = cyl_grid(grid_size, pc_range)
    cyl5 = torch.empty((n, 5), dtype=torch.float32, device=dev)
    vcoors = torch.empty((n, 4), dtype=torch.int64, device=dev)
    keys = torch.empty((n,), dtype=_i32, device=dev)
    check(_L().ls3d_cyl_voxelize(_ptr(points), n, points.shape[1], ctypes.byref(grid), 1 if reverse else 0, 1 if collapse_last else 0, int(batch_size),
                                 _ptr(cyl5), _ptr(vcoors), _ptr(keys), _stream(points)), "ls3d_cyl_voxelize")
    return cyl5, vcoors, keys


def unique_rows(keys, dims, batch_size):
    """torch.unique(rows, return_inverse=True, return_counts=True, dim=0) of the rows the uint32 keys linearise (dims = sizes of the last three
    columns): in-library radix sort + ls3d_unique_sorted -> unique [V, 4] int64 (sorted), inverse [n] int64, counts [V] int64.  One host
    synchronisation: V sizes the result (torch.unique has the same one)."""
    n = keys.shape[0]
    dev = keys.device
    L = _L()
    bits = max(1, int(int(batch_size) * int(dims[0]) * int(dims[1]) * int(dims[2]) - 1).bit_length())
    skeys = torch.empty((n,), dtype=_i32, device=dev)
    perm = torch.empty((n,), dtype=_i32, device=dev)
    ws = _ws(L.ls3d_radix_sort_workspace_bytes(n), keys)
    check(L.ls3d_radix_sort(_ptr(keys), None, n, None, min(32, bits), _ptr(skeys), _ptr(perm), _ptr(ws), ctypes.c_size_t(ws.numel()), _stream(keys)),
          "ls3d_radix_sort")
    inverse = torch.empty((n,), dtype=torch.int64, device=dev)
    rows = torch.empty((max(n, 1), 4), dtype=torch.int64, device=dev)
    counts = torch.empty((max(n, 1),), dtype=torch.int64, device=dev)
    nu = torch.zeros((1,), dtype=_i32, device=dev)
    ws2 = _ws(L.ls3d_unique_sorted_workspace_bytes(n), keys)
    check(L.ls3d_unique_sorted(_ptr(skeys), _ptr(perm), n, _i3(dims), _ptr(ws2), ctypes.c_size_t(ws2.numel()), _ptr(inverse), _ptr(rows), _ptr(counts),
                               _ptr(nu), _stream(keys)), "ls3d_unique_sorted")
    v = int(nu.item())
    return rows[:v], inverse, counts[:v]


def dyn_point_features(points, cyl5, vcoors, inverse, mean5, grid_size, pc_range, scale=None, shift=None, ld=None):
    """[n, ld] input rows of the dynamic readers' point MLP (ls3d_dyn_point_features), ld >= C + 9 columns, zero padded"""
    n, stride = points.shape
    ld = ld or (stride + 9 + 15) // 16 * 16
    grid, _ = cyl_grid(grid_size, pc_range)
    out = torch.empty((n, ld), dtype=torch.float32, device=points.device)
    check(_L().ls3d_dyn_point_features(_ptr(points), n, stride, _ptr(cyl5), _ptr(vcoors), _ptr(inverse), _ptr(mean5), ctypes.byref(grid), _vp(scale),
                                       _vp(shift), _ptr(out), ld, _stream(points)), "ls3d_dyn_point_features")
    return out


def segment_reduce(src, index, n_seg, mode="mean", want_arg=False):
    """src [n,C] f32, index [n] int64 -> out [n_seg,C] (+ arg [n_seg,C] int64 for mode "max")  (ls3d_segment_reduce)"""
    n, c = src.shape
    dev = src.device
    out = torch.empty((n_seg, c), dtype=torch.float32, device=dev)
    arg = torch.empty((n_seg, c), dtype=torch.int64, device=dev) if (want_arg and mode == "max") else None
    L = _L()
    ws = _ws(L.ls3d_segment_reduce_workspace_bytes(n, n_seg), src)
    check(L.ls3d_segment_reduce(_ptr(src), _ptr(index), n, c, n_seg, 0 if mode == "mean" else 1, _ptr(ws), ctypes.c_size_t(ws.numel()),
                                _ptr(out), _ptr(arg), _stream(src)), "ls3d_segment_reduce")
    return (out, arg) if arg is not None else out


# ---------------------------------------------------------------------------------------------- readers
def _ndev(nd):
    return _ptr(nd) if nd is not None else None


def vfe_mean(voxels, num_points, n_dev=None):
    n, p, c = voxels.shape
    out = torch.empty((n, c), dtype=torch.float32, device=voxels.device)
    check(_L().ls3d_vfe_mean(_ptr(voxels), _ptr(num_points), n, _ndev(n_dev), p, c, _ptr(out), c, _stream(voxels)),
          "ls3d_vfe_mean")
    return out


def vfe_improved_mean(voxels, num_points, out_ld=None, n_dev=None):
    n, p, c = voxels.shape
    ld = out_ld or (c + 8)
    out = torch.empty((n, ld), dtype=torch.float32, device=voxels.device)
    check(_L().ls3d_vfe_improved_mean(_ptr(voxels), _ptr(num_points), n, _ndev(n_dev), p, c, _ptr(out), ld,
                                      _stream(voxels)), "ls3d_vfe_improved_mean")
    return out


def vfe_tokens(voxels, num_points, tok_ld, n_dev=None):
    n, p, c = voxels.shape
    out = torch.empty((n * p, tok_ld), dtype=torch.float32, device=voxels.device)
    check(_L().ls3d_vfe_tokens(_ptr(voxels), _ptr(num_points), n, _ndev(n_dev), p, c, _ptr(out), tok_ld,
                               _stream(voxels)), "ls3d_vfe_tokens")
    return out


class TransVFEModel(object):
    """kernel-side description of a TransformerVoxelFeatureExtractor (keeps the packed tensors alive)"""

    def __init__(self, embed, layers, compress, num_embed, num_head, ffn, token_ld, planes=0):
        """embed: (w_packed_nt2, bias); layers: dicts with wqkv,bqkv,wo,bo,w1,b1,w2,b2 (packed nt=2 / biases), n1/n2 =
        (gamma, beta, eps); compress: (plain weight [out, in], bias) or None.  planes = 6 | 8: the GEMM weights are the plane-packed
        buffers of ls3d_transvfe_pack_planes (for_planes builds that variant from an f32 model)."""
        self._keep = [embed, layers, compress]
        self._args = (num_embed, num_head, ffn, token_ld)
        self._variants = {}
        arr = (_lib.TransVFELayer * max(len(layers), 1))()
        for i, l in enumerate(layers):
            for k in ("wqkv", "bqkv", "wo", "bo", "w1", "b1", "w2", "b2"):
                setattr(arr[i], k, l[k].data_ptr())
            arr[i].n1_gamma, arr[i].n1_beta, arr[i].n1_eps = l["n1"][0].data_ptr(), l["n1"][1].data_ptr(), float(l["n1"][2])
            arr[i].n2_gamma, arr[i].n2_beta, arr[i].n2_eps = l["n2"][0].data_ptr(), l["n2"][1].data_ptr(), float(l["n2"][2])
        self._arr = arr
        self.num_out = compress[0].shape[0] if compress is not None else num_embed
        self.c = _lib.TransVFE(embed[0].data_ptr(), embed[1].data_ptr(), compress[0].data_ptr() if compress is not None else None,
                               compress[1].data_ptr() if compress is not None else None, arr, len(layers),
                               compress[0].shape[0] if compress is not None else 0, num_embed, num_head, ffn, token_ld, int(planes), 0)

    def for_planes(self, products):
        """the same reader on the exact 3-plane bf16 split with `products` (6 | 8) plane products per f32 product (built once)"""
        v = self._variants.get(products)
        if v is None:
            embed, layers, compress = self._keep
            num_embed, num_head, ffn, token_ld = self._args
            if token_ld % 32 or num_embed % 64 or ffn % 64:  # not a shape of the fused kernel: ls3d_transvfe answers UNSUPPORTED either way
                self._variants[products] = self
                return self

            def conv(w, K, N):
                L = _L()
                out = torch.empty((int(L.ls3d_transvfe_planes_bytes(K, N)),), dtype=torch.uint8, device=w.device)
                check(L.ls3d_transvfe_pack_planes(_ptr(w), K, N, _ptr(out), _stream(w)), "ls3d_transvfe_pack_planes")
                return out
            e2 = (conv(embed[0], token_ld, num_embed), embed[1])
            l2 = [dict(l, wqkv=conv(l["wqkv"], num_embed, 3 * num_embed), wo=conv(l["wo"], num_embed, num_embed), w1=conv(l["w1"], num_embed, ffn),
                       w2=conv(l["w2"], ffn, num_embed)) for l in layers]
            v = self._variants[products] = TransVFEModel(e2, l2, compress, num_embed, num_head, ffn, token_ld, planes=products)
        return v


_TRANSVFE_PLANES = True
_FAST_LAYERNORM = True  # training: nn.LayerNorm over >= 4096 rows on ls3d_layer_norm_* (fast_linear_backward)
_GATHER_X6 = True  # bf16x6: strided / inverse layers on the 6-product gather-GEMM (split accumulators); False: exact f32


def transvfe(voxels, num_points, model, n_dev=None):
    """the whole TransformerVoxelFeatureExtractor in one kernel; returns None if the configuration is not the one the fused
    kernel is specialised for (the caller then composes the layer from the individual ops).  In the 3-plane modes of
    ops.set_precision the reader's GEMMs run on the same exact bf16 split as the SubM convolutions."""
    n, p, c = voxels.shape
    _ptr(voxels)  # device / contiguity checks
    products = tile_products() if _TRANSVFE_PLANES else 0
    products = 6 if products == 1 else products  # plain-bf16 mode: the reader stays on its f32-grade planes
    m = model.for_planes(products) if products else model
    out = torch.empty((n, model.num_out), dtype=torch.float32, device=voxels.device)
    m.c.flags = (1 if _TRANSVFE_DIRECT else 0) | (0 if _TRANSVFE_DEDUP else 2)
    L = _L()
    ws = _ws(L.ls3d_transvfe_workspace_bytes(n, p), voxels) if (_TRANSVFE_DEDUP and n > 0) else None
    rc = L.ls3d_transvfe(_ptr(voxels), _ptr(num_points), n, _ndev(n_dev), p, c, ctypes.byref(m.c), _ptr(out), model.num_out,
                         _vp(ws), ctypes.c_size_t(ws.numel() if ws is not None else 0), _stream(voxels))
    if rc == _lib.ERR_UNSUPPORTED:
        return None
    check(rc, "ls3d_transvfe")
    return out


def mha_core(qkv, groups, seq, embed, heads):
    out = torch.empty((groups * seq, embed), dtype=torch.float32, device=qkv.device)
    check(_L().ls3d_mha_core(_ptr(qkv), groups, None, seq, embed, heads, _ptr(out), _stream(qkv)), "ls3d_mha_core")
    return out


def group_max(x, groups, seq):
    c = x.shape[1]
    out = torch.empty((groups, c), dtype=torch.float32, device=x.device)
    check(_L().ls3d_group_max(_ptr(x), groups, None, seq, c, _ptr(out), _stream(x)), "ls3d_group_max")
    return out


def layernorm(x, gamma, beta, eps=1e-5, res=None):
    rows, c = x.shape
    y = torch.empty_like(x)
    check(_L().ls3d_layernorm(_ptr(x), _ptr(res), _ptr(gamma), _ptr(beta), ctypes.c_float(eps), rows, None, c, _ptr(y),
                              _stream(x)), "ls3d_layernorm")
    return y


# ---------------------------------------------------------------------------------------------- sparse conv
def hash_capacity(n):
    c = 1024
    while c < 2 * n:
        c *= 2
    return c


def index_build(coords, shape_zyx, n_dev=None):
    """n_dev (here and in the functions below): optional device int32 holding the number of valid rows; coords.shape[0] is then a
    capacity and rows beyond the count are ignored / left unwritten (include/ls3d.h: producer -> consumer chains without host syncs)"""
    n = coords.shape[0]
    cap = hash_capacity(n)
    keys = torch.empty((cap,), dtype=torch.int64, device=coords.device)
    vals = torch.empty((cap,), dtype=_i32, device=coords.device)
    check(_L().ls3d_index_build(_ptr(coords), n, _ndev(n_dev), _i3(shape_zyx), _ptr(keys), _ptr(vals), cap, _stream(coords)),
          "ls3d_index_build")
    return keys, vals


def rulebook_subm(coords, shape_zyx, ksize, index=None, n_dev=None):
    n = coords.shape[0]
    keys, vals = index if index is not None else index_build(coords, shape_zyx, n_dev)
    kvol = int(ksize[0] * ksize[1] * ksize[2])
    nbr = torch.empty((n, kvol), dtype=_i32, device=coords.device)
    check(_L().ls3d_rulebook_subm(_ptr(coords), n, _ndev(n_dev), _i3(shape_zyx), _i3(ksize), _ptr(keys), _ptr(vals),
                                  keys.numel(), _ptr(nbr), _stream(coords)), "ls3d_rulebook_subm")
    return nbr


def conv_out_shape(shape_zyx, ksize, stride, pad):
    return [(int(shape_zyx[a]) + 2 * int(pad[a]) - int(ksize[a])) // int(stride[a]) + 1 for a in range(3)]


def rulebook_conv(coords, batch, shape_zyx, ksize, stride, pad, out_cap=None, n_dev=None):
    """-> out_coords[cap,4], cnt = [n_out, overflow] (device), nbr_out[cap,kvol], nbr_inv[n_in,kvol], out_shape.
    n_dev: optional device int32 holding the number of valid rows of `coords` (<= coords.shape[0], which then is a
    capacity): lets several rulebooks be chained without a host round trip for each size."""
    n = coords.shape[0]
    kvol = int(ksize[0] * ksize[1] * ksize[2])
    oshape = conv_out_shape(shape_zyx, ksize, stride, pad)
    per_in = 1
    for a in range(3):  # outputs one input can feed along an axis
        per_in *= -(-int(ksize[a]) // int(stride[a]))
    cells = batch * oshape[0] * oshape[1] * oshape[2]
    cap = int(out_cap) if out_cap else max(1, min(n * per_in, cells))
    dev = coords.device
    oc = torch.empty((cap, 4), dtype=_i32, device=dev)
    nbr_out = torch.empty((cap, kvol), dtype=_i32, device=dev)
    nbr_inv = torch.empty((max(n, 1), kvol), dtype=_i32, device=dev)
    cnt = torch.zeros((2,), dtype=_i32, device=dev)  # [n_out, overflow]
    L = _L()
    ws = _ws(L.ls3d_rulebook_conv_workspace_bytes(batch, _i3(oshape)), coords)
    check(L.ls3d_rulebook_conv(_ptr(coords), n, _ptr(n_dev), batch, _i3(shape_zyx), _i3(ksize), _i3(stride), _i3(pad), _ptr(ws),
                               ctypes.c_size_t(ws.numel()), _ptr(oc), cap, _ptr(cnt), _ptr(nbr_out), _ptr(nbr_inv),
                               ctypes.c_void_p(cnt.data_ptr() + 4), _stream(coords)), "ls3d_rulebook_conv")
    return oc, cnt, nbr_out, nbr_inv, oshape


_ROW_ORDER = "mask"


def set_row_order(kind):
    """"mask": gather-GEMM tiles take rows sorted by neighbour bitmask (skips zero MFMA work); "none": natural row order
    (keeps the spatial locality of sorted coordinates)"""
    global _ROW_ORDER
    assert kind in ("mask", "none")
    _ROW_ORDER = kind


def rulebook_order(tbl, coords=None, n_dev=None):
    """processing order of a rulebook table's rows: sorted by neighbour bitmask, densest rows first (kernel
    ls3d_rulebook_masks; the sort itself is torch.argsort - plumbing; rulebook_orders below does several tables with one sort).
    `coords` is unused (a region-blocked order was measured slower, profiles/round1_experiments.md)."""
    n, kvol = tbl.shape
    if n == 0 or kvol > 31 or _ROW_ORDER == "none":
        return None
    return rulebook_orders([tbl], [n_dev])[0]


F32, BF16X3, BF16X6, BF16X8, BF16 = 0, 1, 2, 3, 4
_PREC_NAMES = {"f32": F32, "bf16x3": BF16X3, "bf16x6": BF16X6, "bf16x8": BF16X8, "bf16": BF16}
# Default since round 3: "bf16x6", the f32-grade arithmetic whose end-to-end error against float64 is BELOW the exact-f32 MFMA path's own
# on every frame measured (asserted by tests/test_gpu_parity.py::test_sdseg3d_every_arithmetic_vs_float64_... and
# ..._are_f32_grade_on_other_frames) at 1.5x its speed.  LS3D_PRECISION=f32 (or set_precision("f32")) selects exact-f32 MFMA everywhere.
_PRECISION = _PREC_NAMES[_os.environ.get("LS3D_PRECISION", "bf16x6")]


def set_precision(name):
    """arithmetic of the sparse convolutions: "bf16x6" (default) / "bf16x8": the f32-grade 3-plane modes (exact 3-way bf16 split of both
    operands with round-to-nearest planes; the 6 plane products of weight >= 2^-16 / every product except tail x tail, head x head in
    its own accumulator; SubM layers on the tile-halo kernel ls3d_tile_conv, strided and inverse layers on the 6-product gather-GEMM /
    exact f32; measured end-to-end logit error against float64: 0.4x / 0.6x of the exact-f32 path's rms), "f32" (exact f32 MFMA
    everywhere), "bf16x3" (2-way split, 3 products on the gather-GEMM: ~1e-5 relative error per layer, NOT f32-grade) or "bf16"
    (BASELINE configs[4]: SubM layers with plain bf16 operands - one MFMA per product, f32 accumulation - strided / inverse layers
    on the bf16x3 gather-GEMM, reader on its 6-product planes; ~2^-9 relative error per layer: a throughput mode with a stated
    tolerance, tests/test_gpu_parity.py::test_bf16_mode_tolerance_vs_oracle)."""
    global _PRECISION
    _PRECISION = _PREC_NAMES[name]


def get_precision():
    return {v: k for k, v in _PREC_NAMES.items()}[_PRECISION]


def gather_gemm_pack(w_plain, kvol, cin, cin_pad, cout, nt=0, precision=F32):
    """plain [kvol,cin,cout] -> kernel layout (flat tensor) for column-block count nt (0 = default)"""
    L = _L()
    out = torch.empty((L.ls3d_gather_gemm_packed_floats(kvol, cin_pad, cout),), dtype=torch.float32, device=w_plain.device)
    check(L.ls3d_gather_gemm_pack(_ptr(w_plain), kvol, cin, cin_pad, cout, nt, precision, _ptr(out), _stream(w_plain)),
          "ls3d_gather_gemm_pack")
    return out


_TARGET_BLOCKS = 0


def choose_geometry(cout, n_rows, target_blocks=None):
    """(nt, wc): 32-column blocks per wave and waves along the columns (workgroup = 32*(4/wc) rows x 32*nt*wc columns).
    Among the geometries that give the 256 CUs at least ~3 workgroups each, take the one with the fewest column slabs
    (every extra slab re-gathers the input rows) and then the tallest tile; if none does, take the one with the most
    workgroups (fewest slabs on ties)."""
    if target_blocks is None:
        # measured on MI355X (120k-pt frame): the f32 path is matrix-pipe bound and wants many small workgroups
        # (60.5 fps at >=1500 vs 53 at 256); the split-bf16 path is bound by re-gathering the input rows once per
        # column slab and wants few, wide workgroups (84.6 fps at 128-384 vs 67.5 at 1500)
        # bf16x6 (round 4, profiles/round4_ab_gather_knobs.txt): 256 instead of 512 puts conv4.0 / conv_out on two column blocks per wave (half the
        # re-gathers): slower alone (142 -> 166 us), faster inside the frame (stack 5.80 -> 5.75 ms, 147.3 -> 148.4 frames/s)
        target_blocks = _TARGET_BLOCKS or (192 if _PRECISION in (BF16X3, BF16) else 256 if _PRECISION == BF16X6 else 2000)
    total = (cout + 31) // 32
    cands = []
    # measured on MI355X (profiles/): sharing gathered rows between waves (wc > 1) is slower than re-gathering them
    # per column slab (L3: 779 vs 749 us, L4: 414 vs 299 us) — the wide geometries stay available but are not chosen
    for nt, wc in ((4, 1), (3, 1), (2, 1), (1, 1)):
        if total % (nt * wc):
            continue
        slabs = total // (nt * wc)
        blocks = -(-n_rows // (32 * (4 // wc))) * slabs
        cands.append((blocks, slabs, wc, nt))
    ok = [c for c in cands if c[0] >= target_blocks]
    if ok:
        blocks, slabs, wc, nt = min(ok, key=lambda c: (c[1], c[2]))
    else:
        blocks, slabs, wc, nt = max(cands, key=lambda c: (c[0], -c[1]))
    return nt, wc


def radix_argsort(keys, bits=32, n_dev=None):
    """stable ascending argsort of int32/uint32 keys by their low `bits` bits -> int32 permutation (ls3d_radix_sort); n_dev: only
    the first *n_dev keys are sorted (the rest of the permutation is unspecified)"""
    n = keys.shape[0]
    L = _L()
    perm = torch.empty((n,), dtype=_i32, device=keys.device)
    ws = _ws(L.ls3d_radix_sort_workspace_bytes(n), keys)
    check(L.ls3d_radix_sort(_ptr(keys), None, n, _ndev(n_dev), int(bits), None, _ptr(perm), _ptr(ws), ctypes.c_size_t(ws.numel()), _stream(keys)),
          "ls3d_radix_sort")
    return perm


def rulebook_orders(tbls, n_devs=None):
    """rulebook_order for several tables at once: one batched sort (keys carry the table number in their top bits) instead
    of one sort per table - the sorts are launch-bound (~10 small kernels each).  -> list of int32 orders (None where a
    table has no rows / too many offsets / row order is disabled)."""
    descending = True
    out = [None] * len(tbls)
    sel = [i for i, t in enumerate(tbls) if t.shape[0] > 0 and t.shape[1] <= 27 and _ROW_ORDER != "none"]
    for lo in range(0, len(sel), 16):
        grp = sel[lo:lo + 16]
        offs = [0]
        for i in grp:
            offs.append(offs[-1] + tbls[i].shape[0])
        dev = tbls[grp[0]].device
        keys = torch.empty((offs[-1],), dtype=_i32, device=dev)
        L = _L()
        for s, i in enumerate(grp):
            t = tbls[i]
            check(L.ls3d_rulebook_sort_keys(_ptr(t), t.shape[0], _ndev(n_devs[i]) if n_devs is not None else None, t.shape[1], s, 1 if descending else 0,
                                            ctypes.c_void_p(keys.data_ptr() + 4 * offs[s]), _stream(t)), "ls3d_rulebook_sort_keys")
        perm = radix_argsort(keys, 27 + max(len(grp) - 1, 1).bit_length())  # in-library stable radix sort (csrc/sort.hip)
        local = torch.empty((offs[-1],), dtype=_i32, device=dev)
        check(L.ls3d_segment_local_index32(_ptr(perm), offs[-1], (ctypes.c_int32 * len(offs))(*offs), len(grp), _ptr(local), _stream(perm)),
              "ls3d_segment_local_index32")
        for s, i in enumerate(grp):
            out[i] = local[offs[s]:offs[s + 1]]
    return out


_PARITY_ORDER = True  # A/B: transposed strided tables ordered by coordinate residue class (one radix pass)


def rulebook_parity_orders(coords, geoms, n_devs=None):
    """processing orders of the TRANSPOSED tables of strided convolutions (SparseInverseConv3d / dgrad of SparseConv3d) from the input
    coordinates alone (include/ls3d.h: ls3d_rulebook_parity_keys): coords[i] [n_i, 4] int32 input sites, geoms[i] = (ksize, stride, padding)
    triples.  One batched sort with ONE radix pass.  -> list of int32 orders, None where the geometry is not supported (the caller sorts
    those tables by their masks: rulebook_orders) or row order is disabled."""
    out = [None] * len(coords)
    if _ROW_ORDER == "none" or not _PARITY_ORDER:
        return out
    L = _L()
    sel = [i for i, c in enumerate(coords) if c.shape[0] > 0 and all(1 <= int(v) <= 8 for v in geoms[i][1])
           and int(geoms[i][1][0]) * int(geoms[i][1][1]) * int(geoms[i][1][2]) < 15]
    for lo in range(0, len(sel), 16):
        grp = sel[lo:lo + 16]
        offs = [0]
        for i in grp:
            offs.append(offs[-1] + coords[i].shape[0])
        dev = coords[grp[0]].device
        keys = torch.empty((offs[-1],), dtype=_i32, device=dev)
        for s_, i in enumerate(grp):
            c = coords[i]
            assert c.dtype == _i32 and c.is_contiguous() and c.shape[1] == 4
            k3, s3, p3 = ((ctypes.c_int32 * 3)(*[int(v) for v in g]) for g in geoms[i])
            check(L.ls3d_rulebook_parity_keys(_ptr(c), c.shape[0], _ndev(n_devs[i]) if n_devs is not None else None, k3, s3, p3, s_, 4,
                                              ctypes.c_void_p(keys.data_ptr() + 4 * offs[s_]), _stream(c)), "ls3d_rulebook_parity_keys")
        perm = radix_argsort(keys, 4 + max(len(grp) - 1, 1).bit_length())
        local = torch.empty((offs[-1],), dtype=_i32, device=dev)
        check(L.ls3d_segment_local_index32(_ptr(perm), offs[-1], (ctypes.c_int32 * len(offs))(*offs), len(grp), _ptr(local), _stream(perm)),
              "ls3d_segment_local_index32")
        for s_, i in enumerate(grp):
            out[i] = local[offs[s_]:offs[s_ + 1]]
    return out


_TRANSVFE_DIRECT = False
_TRANSVFE_DEDUP = True  # (set_transvfe_dedup: A/B in the tests) identical padding tokens of a voxel computed once (ls3d_transvfe)
# per-call flags of ls3d_gather_gemm (include/ls3d.h): bits 0-1 workgroup -> tile mapping, bit 2 the one-stage pipeline of the sparse 6-product kernel (A/B)
_GEMM_FLAGS = 0  # set_gemm_flags: the `flags` of ls3d_gather_gemm (workgroup -> tile mapping, one-stage pipeline) for A/B runs


def set_transvfe_direct(on):
    """fused TransVFE reader with its weights read straight from L2 (no LDS staging, no workgroup barriers): experimental, measured
    slower (descriptor flag of ls3d_transvfe; env LS3D_TRANSVFE_DIRECT=1 sets it at start-up)"""
    global _TRANSVFE_DIRECT
    _TRANSVFE_DIRECT = bool(on)


def set_transvfe_dedup(on):
    """token deduplication of the fused TransVFE reader (include/ls3d.h: ls3d_transvfe's workspace) on / off (A/B, tests)"""
    global _TRANSVFE_DEDUP
    _TRANSVFE_DEDUP = bool(on)


def set_gemm_flags(flags):
    """per-call flags of ls3d_gather_gemm (include/ls3d.h): workgroup -> (tile, slab) mapping (bits 0-1), one-stage pipeline of the sparse
    6-product kernel (bit 2) for A/B runs; results identical"""
    global _GEMM_FLAGS
    _GEMM_FLAGS = int(flags) & 7


def gather_gemm(x, w, tbl=None, order=None, n_rows=None, cout=None, scale=None, shift=None, res_pre=None, relu=False, pair=None,
                out=None, out_ld=None, in_ld=None, cin=None, ln=None, n_dev=None):
    """out[r, :cout] = epilogue(sum_k W[k]^T x[tbl[r,k]]).  w: packing.PackedWeight."""
    kvol, wcin, wld = w.shape
    cin = cin or wcin
    assert wcin == cin
    in_ld = in_ld or x.shape[1]
    cout = cout or w.cout
    assert cout == w.cout, "packed layout depends on cout"
    rows_hint = tbl.shape[0] if (tbl is not None and n_rows is None) else (n_rows if n_rows is not None else x.shape[0])
    # split-bf16 only where it pays and where its error budget is spent wisely: the sparse convolutions (matrix-pipe
    # bound).  Dense Linear layers (TransVFE, heads, SF-Phase) are memory-bound and stay in exact f32.
    prec = _PRECISION if (_PRECISION != F32 and cin % 32 == 0 and tbl is not None) else F32
    if prec == BF16:
        prec = BF16X3  # the layers the tile kernel does not take (strided / inverse) in the plain-bf16 mode
    if prec == BF16X8 or (prec == BF16X6 and _TILE and not _GATHER_X6):
        # the 3-plane modes = tile-halo kernel (ls3d_tile_conv, 8 / 6 plane products, head x head in its own accumulator) for the layers
        # that take it.  The other sparse layers (strided / inverse convolutions: 1.6 pairs per output row, mask-sorted gathers win):
        # "bf16x8" runs them exact f32; "bf16x6" on the 6-product gather-GEMM, which since round 2 also keeps head x head in its own
        # accumulator and splits with round-to-nearest planes (measured: end-to-end error 0.38x of the exact-f32 path's, 0.53x with
        # these layers in exact f32; LS3D_GATHER_X6=0 restores that).  (With the tile path switched off, "bf16x6" is that gather-GEMM
        # for every layer.)
        prec = F32
    nt, wc = choose_geometry(cout, rows_hint)
    if ln is not None:  # LayerNorm epilogue: the whole row must sit in one workgroup slab
        assert cout <= 128, "LayerNorm epilogue supports up to 128 columns"
        nt, wc = (cout + 31) // 32, 1
    if prec != F32:
        wc = 1
    wdata = w.for_nt(nt, prec)
    if tbl is not None:
        n_rows = tbl.shape[0] if n_rows is None else n_rows
        assert tbl.shape[1] == kvol
    else:
        n_rows = x.shape[0] if n_rows is None else n_rows
    if out is None:
        out_ld = out_ld or cout
        out = torch.empty((n_rows, out_ld), dtype=torch.float32, device=x.device)
        out_view = out
    else:
        out_view = out
        out_ld = out_ld or _ld(out)
    epi = Epilogue(_vp(scale), _vp(shift), _vp_any(res_pre) if res_pre is not None else ctypes.c_void_p(0), _ld(res_pre) if res_pre is not None else 0,
                   _vp_any(pair) if pair is not None else ctypes.c_void_p(0), _ld(pair) if pair is not None else 0, 1 if relu else 0,
                   _vp(ln[0]) if ln is not None else ctypes.c_void_p(0), _vp(ln[1]) if ln is not None else ctypes.c_void_p(0),
                   float(ln[2]) if ln is not None else 0.0)
    check(_L().ls3d_gather_gemm(_ptr(x), in_ld, _ptr(tbl), _ptr(order), kvol, _ptr(wdata), nt, wc, prec, cin, cout, n_rows, _ndev(n_dev), ctypes.byref(epi),
                                _vp_any(out_view), out_ld, _GEMM_FLAGS, _stream(x)), "ls3d_gather_gemm")
    return out


# ---------------------------------------------------------------------------------------------- tile-halo convolution
_TILE = True          # 3-plane modes: SubM layers on ls3d_tile_conv
_TILE_KINDS = "subm"  # which rulebook kinds take the tile path: subm[,conv][,inverse]
_TILE_MIN_CC = 512  # cin*cout below which the gather-GEMM stays


def set_tile(on, kinds=None, min_cc=None):
    global _TILE, _TILE_KINDS, _TILE_MIN_CC
    _TILE = bool(on)
    if kinds is not None:
        _TILE_KINDS = kinds
    if min_cc is not None:
        _TILE_MIN_CC = int(min_cc)


_TILE_FLAGS = 0       # per-call flags of ls3d_tile_conv (include/ls3d.h); A/B runs set it through experiments.py
_TILE_PLAN_FLAGS = 0  # ... of ls3d_tile_plan / ls3d_tile_build


def set_tile_flags(conv=None, plan=None):
    """A/B knobs: the `flags` argument that tile_conv / tile_plan pass to ls3d_tile_conv / ls3d_tile_plan (include/ls3d.h)"""
    global _TILE_FLAGS, _TILE_PLAN_FLAGS
    if conv is not None:
        _TILE_FLAGS = int(conv)
    if plan is not None:
        _TILE_PLAN_FLAGS = int(plan)


def tile_products():
    """plane products per f32 product on the tile path, or 0 when the current precision does not use it"""
    if not _TILE:
        return 0
    return 8 if _PRECISION == BF16X8 else 6 if _PRECISION == BF16X6 else 1 if _PRECISION == BF16 else 0


def use_tile(kind, kvol, cin, cout):
    return tile_products() != 0 and kind in _TILE_KINDS.split(",") and kvol <= 32 and cin % 16 == 0 and cin * cout >= _TILE_MIN_CC


class TilePlan(object):
    """device-resident plan of ls3d_tile_build for one rulebook table (n_dev: device count of the table's valid rows, or None)"""
    __slots__ = ("buf", "n_rows", "kvol", "order", "tbl", "n_dev")

    def record_stream(self, s):
        self.buf.record_stream(s)
        if self.order is not None and self.order.is_cuda:
            self.order.record_stream(s)


def tile_keys(coords, shape_zyx, batch, n_dev=None):
    n = coords.shape[0]
    keys = torch.empty((n,), dtype=_i32, device=coords.device)
    check(_L().ls3d_tile_keys(_ptr(coords), n, _ndev(n_dev), _i3(shape_zyx), int(batch), _ptr(keys), _stream(coords)), "ls3d_tile_keys")
    return keys


# 0: neighbour-mask slot order everywhere (default: the coloured layout removes 70 % of the tile kernel's LDS bank conflicts and changes neither its
# cycles nor its time - profiles/round5_experiments.md 1b - while its plan costs 18 us more per level); 1: coloured halo layout where the caller
# asks for it (the >= 64-channel levels); 2: on every 3x3x3 SubM table
_TILE_COLOR = 0


def tile_plan(tbl, coords, shape_zyx, batch, order=None, n_dev=None, color=False):
    """plan for table tbl[n, kvol] whose output sites are coords[n, 4] (b, z, y, x).  `order`: a precomputed spatial row
    order (int32 permutation); default = stable sort of ls3d_tile_keys (torch.sort: plumbing).  color: the coloured halo layout
    (include/ls3d.h: ls3d_tile_plan flag bit 1; SubM 3x3x3 tables), honoured when _TILE_COLOR >= 1."""
    n, kvol = tbl.shape
    L = _L()
    p = TilePlan()
    p.n_rows, p.kvol, p.tbl, p.n_dev = n, kvol, tbl, n_dev
    p.buf = torch.empty((max(int(L.ls3d_tile_plan_bytes(n, kvol)), 256),), dtype=torch.uint8, device=tbl.device)
    if order is None:  # keys -> in-library radix sort -> plan: one C call (ls3d_tile_plan)
        ws = _ws(L.ls3d_tile_plan_workspace_bytes(n), tbl)
        check(L.ls3d_tile_plan(_ptr(tbl), _ptr(coords), n, _ndev(n_dev), kvol, _i3(shape_zyx), int(batch), _ptr(ws), ctypes.c_size_t(ws.numel()), _ptr(p.buf),
                               ctypes.c_size_t(p.buf.numel()), _TILE_PLAN_FLAGS | (2 if (_TILE_COLOR == 2 or (color and _TILE_COLOR == 1)) else 0), _stream(tbl)),
              "ls3d_tile_plan")
        p.order = None  # the spatial order lives in the call's workspace and is not needed once the plan is built
        return p
    p.order = order
    check(L.ls3d_tile_build(_ptr(tbl), n, _ndev(n_dev), kvol, _ptr(order), _ptr(p.buf), ctypes.c_size_t(p.buf.numel()), _TILE_PLAN_FLAGS, _stream(tbl)),
          "ls3d_tile_build")
    return p


def tile_conv_pack(w_plain, kvol, cin, cin_pad, cout, bf16=False):
    """packed weights of ls3d_tile_conv: the exact 3-plane split (products 6 / 8) or, bf16=True, the head plane only (products 1)"""
    L = _L()
    nbytes, pack = (L.ls3d_tile_conv_packed_bytes_bf16, L.ls3d_tile_conv_pack_bf16) if bf16 else (L.ls3d_tile_conv_packed_bytes, L.ls3d_tile_conv_pack)
    out = torch.empty((int(nbytes(kvol, cin_pad, cout)),), dtype=torch.uint8, device=w_plain.device)
    check(pack(_ptr(w_plain), kvol, cin, cin_pad, cout, _ptr(out), _stream(w_plain)), "ls3d_tile_conv_pack")
    return out


def tile_conv(x, w, plan, cout=None, products=None, scale=None, shift=None, res_pre=None, relu=False, pair=None, out=None, out_ld=None,
              in_ld=None, ln=None):
    """out[r, :cout] = epilogue(sum_k W[k]^T x[tbl[r,k]]) on the tile plan of tbl.  w: packing.PackedWeight."""
    kvol, cin, _ = w.shape
    assert kvol == plan.kvol
    cout = cout or w.cout
    in_ld = in_ld or x.shape[1]
    products = products or tile_products() or 8
    if out is None:
        out_ld = out_ld or cout
        out = torch.empty((plan.n_rows, out_ld), dtype=torch.float32, device=x.device)
    else:
        out_ld = out_ld or _ld(out)
    # residual / pair operands may be column views of a wider buffer (a level's concat buffer): their row stride is the leading dimension
    epi = Epilogue(_vp(scale), _vp(shift), _vp_any(res_pre) if res_pre is not None else ctypes.c_void_p(0), _ld(res_pre) if res_pre is not None else 0,
                   _vp_any(pair) if pair is not None else ctypes.c_void_p(0), _ld(pair) if pair is not None else 0, 1 if relu else 0,
                   _vp(ln[0]) if ln is not None else ctypes.c_void_p(0), _vp(ln[1]) if ln is not None else ctypes.c_void_p(0),
                   float(ln[2]) if ln is not None else 0.0)
    split = _TILE_KSPLIT and cin >= 64
    flags = _TILE_FLAGS
    nbytes = int(_L().ls3d_tile_conv_workspace_bytes(plan.n_rows, cout)) if split else 0
    if _TILE_TRACE is not None and products == 6:  # tools/trace_tile.py: the tracing build writes its records behind the partial sums
        flags |= 32
        off = int(_L().ls3d_tile_conv_workspace_bytes(plan.n_rows, cout))
        nbytes = off + int(_L().ls3d_tile_conv_trace_bytes(plan.n_rows))
    ws = _tile_ws(nbytes, x) if nbytes else None
    check(_L().ls3d_tile_conv(_vp_any(x), in_ld, _ptr(plan.buf), plan.n_rows, kvol, _ptr(w.for_tile(bf16=(products == 1))), cin, cout, products, ctypes.byref(epi),
                              _vp_any(out), out_ld, _vp(ws), ctypes.c_size_t(ws.numel() if ws is not None else 0),
                              _vp(_tile_counters(x) if split else None), flags, _stream(x)), "ls3d_tile_conv")
    if flags & 32:
        _TILE_TRACE.append(dict(rows=plan.n_rows, kvol=kvol, cin=cin, cout=cout, records=ws[off:].view(torch.int32).view(-1, 4, 16).clone()))
    return out


class ChainLayer(object):
    """one layer of tile_conv_chain: the arguments of tile_conv (x: [rows, >= cin] features, possibly a column view of a wider buffer - its row
    stride is taken from the tensor; out: preallocated [rows, >= cout] or a column view)"""
    __slots__ = ("x", "w", "cout", "scale", "shift", "res_pre", "relu", "pair", "out")

    def __init__(self, x, w, out, cout=None, scale=None, shift=None, res_pre=None, relu=False, pair=None):
        self.x, self.w, self.out, self.cout, self.scale, self.shift, self.res_pre, self.relu, self.pair = x, w, out, cout or w.cout, scale, shift, res_pre, relu, pair


_TILE_CHAIN = True  # A/B: consecutive SubM layers of a UNet level as ONE persistent launch (ls3d_tile_conv_chain)
TILE_CHAIN_MAX = 8
_CHAIN_STATES = None  # tests: a list that collects the state buffers of the chained launches (state[1] != 0: a wait ran into its watchdog)


def collect_chain_states(on=True):
    """-> the list that receives the int32 state buffer of every chained launch from now on (None: off)"""
    global _CHAIN_STATES
    _CHAIN_STATES = [] if on else None
    return _CHAIN_STATES


# Where chaining is used (measured on the 120k-point frame, profiles/round5_experiments.md): a level whose tiles exceed the chip's 512 workgroup
# slots (levels 2 and 3; on level 4 - 216 tiles - the layers of a chain run as a wavefront of dependent tiles: 1.11 ms against 0.92 ms of
# launches) and units long enough to carry a ticket, a poll of the producers' counters and a written-through epilogue (64+ output channels; the
# 32-channel units of level 1 last 10 - 17 us: 0.32 ms chained against 0.15 ms).  What it buys there is small (+0.5 - 1 % frames/s: fewer
# launches; the conv stack itself does not get shorter): the tails the chain fills were not idle POWER - the chip is power-limited in these
# layers, a workgroup alone on its CU runs 122 us per tile against 213 us for two sharing one, and filling the tails trades clock for occupancy.
_CHAIN_ABLATE = 0  # (& 62) timing experiments of the chained kernel (ls3d_tile_conv_chain flags bits 1-3)
_CHAIN_MIN_TILES = 600
_CHAIN_MIN_COUT = 64


def set_tile_chain(on, min_tiles=None, min_cout=None):
    """A/B switch of the chained launches (tests lower the two thresholds to chain small frames / narrow layers)"""
    global _TILE_CHAIN, _CHAIN_MIN_TILES, _CHAIN_MIN_COUT
    _TILE_CHAIN = bool(on)
    if min_tiles is not None:
        _CHAIN_MIN_TILES = int(min_tiles)
    if min_cout is not None:
        _CHAIN_MIN_COUT = int(min_cout)


def empty_rows(rows, cols, device):
    """[rows, cols] f32 whose rows are whole 128-byte lines when cols % 32 == 0 (what a chained layer's output needs): the device allocator's
    blocks are 512-byte aligned; the CPU allocator of the host emulation (tests) gives 64 bytes, so there the view starts at an aligned offset"""
    if device.type == "cuda" or rows * cols == 0:
        return torch.empty((rows, cols), dtype=torch.float32, device=device)
    flat = torch.empty((rows * cols + 32,), dtype=torch.float32, device=device)
    off = (-flat.data_ptr() // 4) % 32
    return flat[off:off + rows * cols].view(rows, cols)


def tile_chain_pays(n_rows, cout):
    return (n_rows + 127) // 128 >= _CHAIN_MIN_TILES and cout >= _CHAIN_MIN_COUT


def tile_chain_enabled():
    return _TILE_CHAIN and tile_products() == 6


def _ld(t):
    return t.stride(0) if t.dim() == 2 else t.shape[-1]


def tile_conv_chain(layers, plan):
    """the layers (ChainLayer, <= TILE_CHAIN_MAX, each reading what earlier ones wrote on the plan's rows) in ONE persistent launch of the tile
    kernel (include/ls3d.h: ls3d_tile_conv_chain); -> True, or False when the library declines the combination (the caller launches the
    layers one by one: same results)"""
    L = _L()
    n = len(layers)
    arr = (_lib.TileChainLayer * n)()
    keep = []
    split = False
    for a, l in zip(arr, layers):
        kvol, cin, _ = l.w.shape
        assert kvol == plan.kvol and l.x.stride(1) == 1 and l.out.stride(1) == 1
        a.in_, a.in_ld, a.cin, a.cout = _vp_any(l.x).value, _ld(l.x), cin, l.cout
        wp = l.w.for_tile(bf16=False)
        keep.append(wp)
        a.w_packed = _ptr(wp).value
        a.epi = Epilogue(_vp(l.scale), _vp(l.shift), _vp_any(l.res_pre) if l.res_pre is not None else ctypes.c_void_p(0),
                         _ld(l.res_pre) if l.res_pre is not None else 0, _vp_any(l.pair) if l.pair is not None else ctypes.c_void_p(0),
                         _ld(l.pair) if l.pair is not None else 0, 1 if l.relu else 0, ctypes.c_void_p(0), ctypes.c_void_p(0), 0.0)
        a.out, a.out_ld = _vp_any(l.out).value, _ld(l.out)
        split = split or (_TILE_KSPLIT and cin >= 64)
    x0 = layers[0].x
    state = _ws(int(L.ls3d_tile_chain_state_bytes(plan.n_rows)), x0)
    nbytes = max(int(L.ls3d_tile_conv_workspace_bytes(plan.n_rows, l.cout)) for l in layers) if split else 0
    ws = _tile_ws(nbytes, x0) if nbytes else None
    rc = L.ls3d_tile_conv_chain(_ptr(plan.buf), plan.n_rows, plan.kvol, arr, n, 6, _ptr(state), ctypes.c_size_t(state.numel()), _vp(ws),
                                ctypes.c_size_t(ws.numel() if ws is not None else 0), _vp(_tile_counters(x0) if split else None), (_TILE_FLAGS & ~63) | _CHAIN_ABLATE,
                                _stream(x0))
    if rc == _lib.ERR_UNSUPPORTED:
        return False
    check(rc, "ls3d_tile_conv_chain")
    if _CHAIN_STATES is not None:
        _CHAIN_STATES.append(state.view(torch.int32))
    return True


_TILE_TRACE = None


def trace_tile_convs(on=True):
    """tools/trace_tile.py: make every 6-product tile_conv call run the tracing build (flags bit 5 of ls3d_tile_conv) and collect its
    per-wave records ([unit][wave][16] int32, include/ls3d.h) - returns the list they are appended to (None switches it off)"""
    global _TILE_TRACE
    _TILE_TRACE = [] if on else None
    return _TILE_TRACE


_TILE_KSPLIT = True  # hand ls3d_tile_conv the workspace for its split over the input channels
_TILE_COUNTERS = {}


def _tile_counters(like):
    """arrival counters of ls3d_tile_conv's channel split: one zeroed-once array per (device, stream) - the launches of a stream
    run one after the other, which is all the counters need (include/ls3d.h).  A recycled stream handle finds an all-even array:
    harmless."""
    key = (like.device, torch.cuda.current_stream(like.device).cuda_stream) if like.is_cuda else "host-emulation"
    buf = _TILE_COUNTERS.get(key)
    if buf is None:
        buf = _TILE_COUNTERS[key] = torch.zeros((int(_L().ls3d_tile_conv_counter_bytes()) // 4,), dtype=_i32, device=like.device)
    return buf


def _tile_ws(nbytes, like):
    """per-call scratch from torch's stream-ordered caching allocator: no host synchronisation, the block is reused by the next
    launch on the stream (round 2 kept a grow-only buffer keyed by the raw stream handle, which a destroyed stream can hand on)"""
    return _ws(nbytes, like)


_WGRAD_PLANES = True


def spconv_pairs(tbl, order=None):
    """compacted (input row, output row) lists per kernel offset of one (table, row order), for spconv_wgrad(pairs=...): built once for
    all the layers that share the table (ls3d_spconv_pairs)"""
    n, kvol = tbl.shape
    L = _L()
    pairs = torch.empty((int(L.ls3d_spconv_pairs_bytes(kvol, n)),), dtype=torch.uint8, device=tbl.device)
    check(L.ls3d_spconv_pairs(_ptr(tbl), _ptr(order), n, None, kvol, _ptr(pairs), ctypes.c_size_t(pairs.numel()), _stream(tbl)),
          "ls3d_spconv_pairs")
    return pairs


def spconv_wgrad(x, grad_out, tbl, order, cin, cout, products=None, pairs=None):
    """grad_w[kvol, cin, cout] of a sparse convolution: x = the forward input features (rows indexed by tbl), grad_out on the
    forward output rows, tbl/order = the table and row order of the forward launch.  products: 0 = exact-f32 MFMA kernel, 6 / 8 = the exact
    3-plane bf16 split (f32-grade; the library uses it for layers with >= 4 output blocks of 32 x 32 and the exact-f32 kernel below that);
    None = ops.set_precision's product count (0 in "f32" / "bf16x3").  pairs = spconv_pairs(tbl, order)
    when several layers share the table (same result, the lists are not rebuilt)."""
    n, kvol = tbl.shape
    gw = torch.empty((kvol, cin, cout), dtype=torch.float32, device=x.device)
    L = _L()
    ws = _ws(L.ls3d_spconv_wgrad_workspace_bytes(kvol, cin, cout, n), x)
    if products is None:
        products = tile_products() if _WGRAD_PLANES else 0
        products = 0 if products == 1 else products  # no plain-bf16 weight gradient: exact f32
    if pairs is not None:
        check(L.ls3d_spconv_wgrad_on_pairs(_ptr(x), x.shape[1], _ptr(grad_out), grad_out.shape[1], _ptr(pairs), kvol, cin, cout, n,
                                           int(products), _ptr(ws), ctypes.c_size_t(ws.numel()), _ptr(gw), _stream(x)),
              "ls3d_spconv_wgrad_on_pairs")
        return gw
    check(L.ls3d_spconv_wgrad(_ptr(x), x.shape[1], _ptr(grad_out), grad_out.shape[1], _ptr(tbl), _ptr(order), kvol, cin, cout, n, None,
                              int(products), _ptr(ws), ctypes.c_size_t(ws.numel()), _ptr(gw), _stream(x)), "ls3d_spconv_wgrad")
    return gw


_IDENTITY_PAIRS = {}
_IDENTITY_STEP = 1 << 16


def _identity(device, n):
    """(row capacity, pair lists) of the identity table for n rows: ONE buffer per capacity (n rounded up to 65 536 rows), built once - a new row
    count only sets the number of valid rows (one tiny launch, and only when it differs from the last one used: every training step has its own
    point / voxel counts, every Linear layer of a step the same ones).  Round 4 rebuilt an arange + three pair kernels per new row count."""
    cap = max(_IDENTITY_STEP, -(-n // _IDENTITY_STEP) * _IDENTITY_STEP)
    stream = torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0
    key = (device, cap, stream)  # per stream: the valid-row count is set in stream order
    ent = _IDENTITY_PAIRS.get(key)
    L = _L()
    if ent is None:
        if len(_IDENTITY_PAIRS) > 16:
            _IDENTITY_PAIRS.clear()
        buf = torch.empty((int(L.ls3d_spconv_pairs_bytes(1, cap)),), dtype=torch.uint8, device=device)
        ent = _IDENTITY_PAIRS[key] = [buf, -1]
    if ent[1] != n:
        check(L.ls3d_spconv_identity_pairs(cap, n, 1 if ent[1] < 0 else 0, _ptr(ent[0]), ctypes.c_size_t(ent[0].numel()), ctypes.c_void_p(stream)),
              "ls3d_spconv_identity_pairs")
        ent[1] = n
    return cap, ent[0]


def linear_wgrad(x, gy, products=None):
    """grad_W [cout, cin] = gy^T x of a Linear layer over many rows (x [n, cin], gy [n, cout], both row-major): the tall-skinny GEMM
    (n = 10^5..10^6 rows reduced into a <= 256 x 256 matrix) that hipBLASLt serves with 32 x 32 macro tiles at ~10 TFLOP/s - here it is
    ls3d_spconv_wgrad on the identity table (one kernel offset), whose row-pair MFMA reduction is built for exactly this shape.
    Output columns beyond 128 are done in slices of 128 (the kernel's limit).  products: 0 = exact f32, 6 / 8 = the exact bf16 planes
    (f32-grade); None = what is faster (measured, tools/probe_linear_wgrad.py: planes from 128 input channels on - 192 -> 96 on 360k
    rows: torch 0.76 ms, exact f32 0.63, planes 0.43; 96 -> 96: 0.67 / 0.35 / 0.42)."""
    n, cin = x.shape
    cout = gy.shape[1]
    if products is None:
        products = 6 if cin >= 128 else 0
    cap, pairs = _identity(x.device, n)
    L = _L()
    gw = torch.empty((cout, cin), dtype=torch.float32, device=x.device)
    for c0 in range(0, cout, 128):
        c1 = min(c0 + 128, cout)
        part = torch.empty((1, cin, c1 - c0), dtype=torch.float32, device=x.device)
        ws = _ws(L.ls3d_spconv_wgrad_workspace_bytes(1, cin, c1 - c0, cap), x)
        check(L.ls3d_spconv_wgrad_on_pairs(_ptr(x), x.shape[1], ctypes.c_void_p(gy.data_ptr() + 4 * c0), gy.shape[1], _ptr(pairs), 1, cin, c1 - c0, cap,
                                           int(products), _ptr(ws), ctypes.c_size_t(ws.numel()), _ptr(part), _stream(x)),
              "ls3d_spconv_wgrad_on_pairs")
        gw[c0:c1] = part[0].t()
    return gw


class _LinearFn(torch.autograd.Function):
    """y = x W^T + b with the weight gradient on ls3d_spconv_wgrad (linear_wgrad); everything else is the library GEMM"""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        cout, cin = weight.shape
        if _LINEAR_DGRAD and cin % 16 == 0 and cout % 4 == 0 and cout <= 256 and (x.is_cuda or _SIM) and x.is_contiguous():
            # y = x W^T + b on the dense exact-f32 gather-GEMM as well (hipBLASLt: ~6 ms of a Waymo step in 32-row macro tiles for these shapes)
            from .packing import PackedWeight
            wt = weight.detach().t().contiguous().reshape(1, cin, cout)
            shift = bias.detach().contiguous() if bias is not None else None
            if shift is not None and shift.data_ptr() % 16:  # a view into a flattened parameter buffer: the epilogue reads scale / shift with float4 loads
                shift = shift.clone()
            return gather_gemm(x.detach(), PackedWeight(wt, 1, cin, cin, cout), cout=cout, shift=shift)
        return torch.nn.functional.linear(x, weight, bias) if _ORIG_LINEAR is None else _ORIG_LINEAR(x, weight, bias)

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gy = gy.contiguous()
        gx = None
        if ctx.needs_input_grad[0]:
            cout, cin = weight.shape
            if _LINEAR_DGRAD and cout % 16 == 0 and cin % 4 == 0 and cin <= 256 and (gy.is_cuda or _SIM):
                # grad_x = grad_y W on the dense exact-f32 gather-GEMM (W [out, in] IS the [K][N] operand): hipBLASLt serves these
                # [10^5..10^6 rows] x [<= 256 x 256] products with 32 x 32 macro tiles (~10 ms of a Waymo step, profiles/round4_train_step_*)
                from .packing import PackedWeight
                gx = gather_gemm(gy, PackedWeight(weight.detach().reshape(1, cout, cin), 1, cout, cout, cin), cout=cin)
            else:
                gx = gy @ weight
        gw = linear_wgrad(x.detach().contiguous(), gy) if ctx.needs_input_grad[1] else None
        gb = column_sums(gy) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return gx, gw, gb


_ORIG_LINEAR = None
_LINEAR_DGRAD = True  # A/B: the input gradient of the tall-skinny Linear layers on ls3d_gather_gemm instead of hipBLASLt
_FAST_LINEAR_DEBUG = False  # print the Linear shapes the context does not take (once each)
_FAST_LINEAR_SEEN = set()
_FAST_LINEAR_MIN_ROWS = 32768


class fast_linear_backward(object):
    """`with ops.fast_linear_backward():` - inside, torch.nn.functional.linear (hence nn.Linear, nn.MultiheadAttention's projections)
    records _LinearFn for the tall-skinny case (>= 32768 rows on the device, both dimensions <= 256, a gradient wanted) and
    torch.nn.functional.layer_norm records _LayerNormFn (csrc/norm.hip) for [>= 32768, c <= 256] f32 rows.  The training forward of the
    detectors runs under it (the backward then runs the recorded functions wherever it is called); LS3D_FAST_LAYERNORM=0 switches the
    LayerNorm part off."""

    def __init__(self, model=None):
        """model: only the BatchNorm1d modules of THIS model take the HIP kernels while the context is active (they are marked once); without a
        model every training-mode BatchNorm1d over >= 4096 rows in the process does (tools, tests)"""
        self.model = model
        if model is not None and not model.__dict__.get("_ls3d_bn_marked"):
            for m in model.modules():
                if isinstance(m, torch.nn.BatchNorm1d):
                    m.__dict__["_ls3d_bn_kernels"] = True
            model.__dict__["_ls3d_bn_marked"] = True

    def __enter__(self):
        global _ORIG_LINEAR
        self.on = _ORIG_LINEAR is None
        scoped = self.model is not None
        if self.on:
            _ORIG_LINEAR = torch.nn.functional.linear
            orig = _ORIG_LINEAR

            def linear(input, weight, bias=None):
                rows = input.numel() // max(input.shape[-1], 1)
                # more than 128 output columns run as two slices that both read x: only worth it from ~96 input channels on (64 -> 192 on
                # 1.2M rows: torch 1.79 ms, here 2.17; 96 -> 192 on 360k rows: 0.76 / 0.71)
                if (input.is_cuda and input.is_contiguous() and rows >= _FAST_LINEAR_MIN_ROWS and weight.shape[0] <= 256 and 16 <= weight.shape[1] <= 256
                        and (weight.shape[0] <= 128 or weight.shape[1] >= 96)
                        and torch.is_grad_enabled() and weight.requires_grad and input.dtype == torch.float32):
                    y = _LinearFn.apply(input.reshape(rows, input.shape[-1]), weight, bias)
                    return y.reshape(*input.shape[:-1], weight.shape[0])
                if _FAST_LINEAR_DEBUG and rows >= _FAST_LINEAR_MIN_ROWS and weight.requires_grad and torch.is_grad_enabled():
                    key = (tuple(input.shape), tuple(weight.shape), input.is_contiguous(), str(input.dtype))
                    if key not in _FAST_LINEAR_SEEN:
                        _FAST_LINEAR_SEEN.add(key)
                        print("fast_linear_backward: not taken:", key, file=_sys.stderr)
                return orig(input, weight, bias)
            torch.nn.functional.linear = linear
            orig_bn = self.orig_bn = torch.nn.BatchNorm1d.forward

            def bn_forward(mod, input):
                # BatchNorm1d over [rows, c] in training mode: statistics, normalisation and their backward on csrc/norm.hip (the sites with a
                # ReLU / residual behind the BatchNorm call ops.batch_norm_train themselves: spconv.SparseSequential, scn_unet.SparseBasicBlock)
                if (mod.training and input.dim() == 2 and input.is_cuda and input.shape[0] >= 4096 and torch.is_grad_enabled()
                        and (not scoped or mod.__dict__.get("_ls3d_bn_kernels"))):
                    y = batch_norm_train(mod, input)
                    if y is not None:
                        return y
                return orig_bn(mod, input)
            torch.nn.BatchNorm1d.forward = bn_forward
            orig_ln = self.orig_ln = torch.nn.functional.layer_norm

            def layer_norm(input, normalized_shape, weight=None, bias=None, eps=1e-5):
                # the LayerNorms of the reader and the SF-Phase decoder over 10^5 - 10^6 token rows: csrc/norm.hip (forward 0.36 -> ~0.1 ms,
                # backward 0.59 -> ~0.15 ms on 360 000 x 96)
                c = input.shape[-1]
                rows = input.numel() // max(c, 1)
                if (input.is_cuda and input.is_contiguous() and input.dtype == torch.float32 and len(normalized_shape) == 1 and normalized_shape[0] == c
                        and weight is not None and bias is not None and c % 4 == 0 and 4 <= c <= 256 and rows >= _FAST_LINEAR_MIN_ROWS
                        and torch.is_grad_enabled() and _FAST_LAYERNORM):
                    return _LayerNormFn.apply(input.reshape(rows, c), weight, bias, float(eps)).reshape(input.shape)
                return orig_ln(input, normalized_shape, weight, bias, eps)
            torch.nn.functional.layer_norm = layer_norm
        return self

    def __exit__(self, *exc):
        global _ORIG_LINEAR
        if self.on:
            torch.nn.functional.linear = _ORIG_LINEAR
            torch.nn.functional.layer_norm = self.orig_ln
            torch.nn.BatchNorm1d.forward = self.orig_bn
            _ORIG_LINEAR = None
        return False


def _vp(t):
    p = _ptr(t)
    return p if p is not None else ctypes.c_void_p(0)


def _vp_any(t):
    """pointer of a possibly non-contiguous VIEW (column slice of a wider buffer): rows stay out_ld apart."""
    if _SIM and t.is_cuda:
        raise RuntimeError("hipsim test hook active: CPU tensors only")
    if not _SIM and not t.is_cuda:
        raise RuntimeError("lidarseg3d_amd ops need tensors on the MI355X; there is no CPU fallback")
    return ctypes.c_void_p(t.data_ptr())


# ---------------------------------------------------------------------------------------------- devoxelization
def voxel_centers(coords, voxel_size, pc_range, n_dev=None):
    n = coords.shape[0]
    out = torch.empty((n, 4), dtype=torch.float32, device=coords.device)
    check(_L().ls3d_voxel_centers(_ptr(coords), n, _ndev(n_dev), _f3(voxel_size), _f3(pc_range[:3]), _ptr(out), _stream(coords)),
          "ls3d_voxel_centers")
    return out


def three_nn(unknown, known):
    """unknown (B,N,3), known (B,M,3) -> dist2 (B,N,3) SQUARED, idx (B,N,3)"""
    b, n, _ = unknown.shape
    m = known.shape[1]
    d2 = torch.empty((b, n, 3), dtype=torch.float32, device=unknown.device)
    idx = torch.empty((b, n, 3), dtype=_i32, device=unknown.device)
    check(_L().ls3d_three_nn(b, n, m, _ptr(unknown), _ptr(known), _ptr(d2), _ptr(idx), _stream(unknown)), "ls3d_three_nn")
    return d2, idx


def three_interpolate(features, idx, weight):
    b, c, m = features.shape
    n = idx.shape[1]
    out = torch.empty((b, c, n), dtype=torch.float32, device=features.device)
    check(_L().ls3d_three_interpolate(b, c, m, n, _ptr(features), _ptr(idx), _ptr(weight), _ptr(out), _stream(features)),
          "ls3d_three_interpolate")
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    b, c, n = grad_out.shape
    g = torch.zeros((b, c, m), dtype=torch.float32, device=grad_out.device)
    check(_L().ls3d_three_interpolate_grad(b, c, n, m, _ptr(grad_out), _ptr(idx), _ptr(weight), _ptr(g),
                                           _stream(grad_out)), "ls3d_three_interpolate_grad")
    return g


def frame_offsets(table, batch_size, col=0, n_dev=None):
    """[B+1] int32 device offsets of the frames of a frame-sorted table (f32 points or int32 coordinates, batch index in
    column `col`); a 1-D tensor is taken as the batch column itself.  One tiny kernel, no host sync."""
    if table.dim() == 1:
        table = table.unsqueeze(1)
    if not table.is_contiguous():
        table = table.contiguous()
    is_float = 1 if table.dtype == torch.float32 else 0
    if not is_float and table.dtype != _i32:
        table = table.to(_i32)
    # one entry more than the caller sees: off[B + 1] = the end of the rows with batch index B, an EMPTY extra frame when the table is a voxel
    # table (no voxel carries index B) - so that a kernel that looks up the frame of a PADDING point (batch index B, graph.BucketedFrameGraph)
    # finds a frame of zero voxels instead of reading past the array
    off = torch.empty((batch_size + 2,), dtype=_i32, device=table.device)
    check(_L().ls3d_frame_offsets(_ptr(table), is_float, table.shape[1], col, table.shape[0], _ndev(n_dev), batch_size + 1, _ptr(off), _stream(table)),
          "ls3d_frame_offsets")
    return off[:batch_size + 1]


def devoxelize(points, pt_off, centers, vx_off, batch, max_frame_points, feat, c=None, return_idx=False):
    n = points.shape[0]
    c = c or feat.shape[1]
    out = torch.empty((n, c), dtype=torch.float32, device=points.device)
    idx = torch.empty((n, 3), dtype=_i32, device=points.device) if return_idx else None
    check(_L().ls3d_devoxelize(_ptr(points), points.shape[1], n, _ptr(pt_off), _ptr(centers), _ptr(vx_off), batch,
                               int(max_frame_points), _ptr(feat), feat.shape[1], c, _ptr(out), c, _ptr(idx),
                               _stream(points)), "ls3d_devoxelize")
    return (out, idx) if return_idx else out


def devoxelize_grid(points, pt_off, coords, centers, vx_off, batch, voxel_size, pc_range, feat, c=None, return_idx=False, n_dev=None):
    """grid-accelerated exact 3-NN devoxelization (known points = voxel centres on the voxel lattice).
    feat=None: the neighbour search only -> (idx [n,3] int32 frame-local, weight [n,3]); finish with interpolate_rows."""
    n = points.shape[0]
    search_only = feat is None
    _, grid = make_grid(voxel_size, pc_range)
    dev = points.device
    if search_only:
        out, c, feat_ld = None, 0, 0
        idx = torch.empty((n, 3), dtype=_i32, device=dev)
        w = torch.empty((n, 3), dtype=torch.float32, device=dev)
    else:
        c = c or feat.shape[1]
        feat_ld = feat.shape[1]
        out = torch.empty((n, c), dtype=torch.float32, device=dev)
        idx = torch.empty((n, 3), dtype=_i32, device=dev) if return_idx else None
        w = None
    L = _L()
    V = coords.shape[0]
    ws = _ws(L.ls3d_devoxelize_grid_workspace_bytes(n, V, batch, _i3(grid)), points)
    check(L.ls3d_devoxelize_grid(_ptr(points), points.shape[1], n, _ptr(pt_off), n, _ptr(coords), _ptr(centers), V, _ndev(n_dev), _ptr(vx_off), batch,
                                 _f3(voxel_size), _f3(pc_range[:3]), _i3(grid), _ptr(feat), feat_ld, c, _ptr(out), c,
                                 _ptr(idx), _ptr(w), _ptr(ws), ctypes.c_size_t(ws.numel()), _stream(points)), "ls3d_devoxelize_grid")
    if search_only:
        return idx, w
    return (out, idx) if return_idx else out


class PointMlp(object):
    """plain layers of a per-point Linear (+ BatchNorm(eval) + ReLU) chain for ls3d_point_mlp: [(W [cin, cout] f32, scale | None, shift | None, relu)]"""

    def __init__(self, layers):
        from ._lib import PointMlpLayer
        self.keep = [(w.contiguous(), None if sc is None else sc.contiguous(), None if sh is None else sh.contiguous(), bool(r)) for w, sc, sh, r in layers]
        self.c = (PointMlpLayer * len(self.keep))()
        for i, (w, sc, sh, r) in enumerate(self.keep):
            self.c[i] = PointMlpLayer(w.data_ptr(), 0 if sc is None else sc.data_ptr(), 0 if sh is None else sh.data_ptr(), int(w.shape[0]), int(w.shape[1]),
                                      1 if r else 0)
        self.c_in, self.c_out = int(self.keep[0][0].shape[0]), int(self.keep[-1][0].shape[1])

    def supported(self):
        widths = [w.shape[1] for w, _, _, _ in self.keep]
        return (self.c_in in (32, 64) and len(self.keep) <= 6 and all(c in (32, 64) for c in widths[:-1]) and widths[-1] <= 64
                and sum(w.shape[0] * 72 + 128 for w, _, _, _ in self.keep) * 4 <= 80 * 1024)


_POINT_MLP = True  # A/B: the per-point tail of PointSegBatchlossHead in one launch


def point_mlp(feat, model, idx=None, weight=None, points=None, vx_off=None, n=None, want_labels=True):
    """include/ls3d.h: ls3d_point_mlp.  idx / weight / points / vx_off: the 3-NN interpolation of `feat` in front of the chain (None: the chain runs on
    the rows of feat).  -> (out [n, c_out] f32, labels [n] int64 | None)"""
    n = (idx.shape[0] if idx is not None else feat.shape[0]) if n is None else n
    out = torch.empty((n, model.c_out), dtype=torch.float32, device=feat.device)
    labels = torch.empty((n,), dtype=torch.int64, device=feat.device) if want_labels else None
    check(_L().ls3d_point_mlp(_ptr(feat), feat.shape[1], model.c_in, _ptr(idx), _ptr(weight), _ptr(points), points.shape[1] if points is not None else 1,
                              _ptr(vx_off), n, len(model.keep), model.c, _ptr(out), model.c_out, _ptr(labels), _stream(feat)), "ls3d_point_mlp")
    return out, labels


def interpolate_rows(feat, idx, weight, points, vx_off, c=None):
    """second half of a split devoxelize_grid: out[p] = sum_j weight[p,j] * feat[vx_off[frame(p)] + idx[p,j]]"""
    n = points.shape[0]
    c = c or feat.shape[1]
    out = torch.empty((n, c), dtype=torch.float32, device=feat.device)
    check(_L().ls3d_interpolate_rows(_ptr(feat), feat.shape[1], c, _ptr(idx), _ptr(weight), _ptr(points), points.shape[1], _ptr(vx_off), n,
                                     _ptr(out), c, _stream(feat)), "ls3d_interpolate_rows")
    return out


def interpolate_rows_backward(grad_out, idx, weight, points, vx_off, n_voxels):
    """d interpolate_rows / d feat: deterministic (entries sorted by voxel row, summed in entry order; include/ls3d.h)"""
    n, c = grad_out.shape
    gf = torch.empty((n_voxels, c), dtype=torch.float32, device=grad_out.device)
    L = _L()
    ws = _ws(L.ls3d_interpolate_rows_backward_workspace_bytes(n, n_voxels), grad_out)
    check(L.ls3d_interpolate_rows_backward(_ptr(grad_out), c, c, _ptr(idx), _ptr(weight), _ptr(points), points.shape[1], _ptr(vx_off), n, n_voxels,
                                           _ptr(ws), ctypes.c_size_t(ws.numel()), _ptr(gf), c, _stream(grad_out)), "ls3d_interpolate_rows_backward")
    return gf


class _InterpolateRowsFn(torch.autograd.Function):
    """the devoxelization of the training forward (point_utils.py:8-52 with the neighbour search done): forward ls3d_interpolate_rows, backward
    ls3d_interpolate_rows_backward - instead of a torch gather whose backward scatters with atomics (run-to-run different gradients)"""

    @staticmethod
    def forward(ctx, feat, idx, weight, points, vx_off):
        ctx.save_for_backward(idx, weight, points, vx_off)
        ctx.nv = feat.shape[0]
        return interpolate_rows(feat.contiguous(), idx, weight, points, vx_off)

    @staticmethod
    def backward(ctx, gout):
        idx, weight, points, vx_off = ctx.saved_tensors
        return interpolate_rows_backward(gout.contiguous(), idx, weight, points, vx_off, ctx.nv), None, None, None, None


def interpolate_rows_autograd(feat, idx, weight, points, vx_off):
    return _InterpolateRowsFn.apply(feat, idx.contiguous(), weight.contiguous(), points, vx_off)


# ---------------------------------------------------------------------------------------------- fusion
def grid_gather(image_features, points_cuv, points):
    b, ncam, c, h, w = image_features.shape
    n = points_cuv.shape[0]
    out = torch.empty((n, c), dtype=torch.float32, device=points.device)
    L = _L()
    # one transpose of the camera feature maps to channels-last (44 MB for 6 x 48 x 160 x 240), then coalesced corner reads
    nhwc = torch.empty_like(image_features)
    check(L.ls3d_nchw_to_nhwc(_ptr(image_features), b * ncam, c, h * w, _ptr(nhwc), _stream(points)), "ls3d_nchw_to_nhwc")
    check(L.ls3d_grid_gather(_ptr(nhwc), b, ncam, c, h, w, 1, _ptr(points_cuv), _ptr(points), points.shape[1], n,
                             _ptr(out), c, _stream(points)), "ls3d_grid_gather")
    return out


def complete_concat(lidar, camera, pseudo, points_cuv):
    n, cl = lidar.shape
    cc = camera.shape[1]
    out = torch.empty((n, cl + cc), dtype=torch.float32, device=lidar.device)
    check(_L().ls3d_complete_concat(_ptr(lidar), cl, _ptr(camera), _ptr(pseudo), cc, _ptr(points_cuv), n, _ptr(out),
                                    _stream(lidar)), "ls3d_complete_concat")
    return out


def sfam(feats, logits, vx_off, batch, max_frame_voxels, c=None):
    c = c or feats.shape[1]
    cls = logits.shape[1]
    ws = torch.empty((2 * batch * cls,), dtype=torch.float32, device=feats.device)
    emb = torch.empty((batch, cls, c), dtype=torch.float32, device=feats.device)
    check(_L().ls3d_sfam(_ptr(feats), feats.shape[1], c, _ptr(logits), cls, _ptr(vx_off), batch, int(max_frame_voxels),
                         _ptr(ws), _ptr(emb), _stream(feats)), "ls3d_sfam")
    return emb


def sim_mode():
    return _SIM


def nchw_to_nhwc(x):
    """[n, c, h, w] -> channels-last rows [n * h * w, c] (ls3d_nchw_to_nhwc)"""
    n, c, h, w = x.shape
    out = torch.empty((n * h * w, c), dtype=torch.float32, device=x.device)
    check(_L().ls3d_nchw_to_nhwc(_ptr(x), n, c, h * w, _ptr(out), _stream(x)), "ls3d_nchw_to_nhwc")
    return out


def camera_sfam(feats, probs, batch_size):
    """CameraSemanticFeatureAggregationModule (fcn_mseg3d_head.py:23-51): feats [B*ncam, C, h, w], probs [B*ncam, cls, h, w]
    -> semantic embeddings [B, C, cls, 1].  The maps go channels-last once; then it is the LiDAR SFAM kernels with
    "voxels" = the ncam*h*w pixels of a frame."""
    bn, c, h, w = feats.shape
    cls = probs.shape[1]
    rows = (bn // batch_size) * h * w
    L = _L()
    f = torch.empty((bn * h * w, c), dtype=torch.float32, device=feats.device)
    p = torch.empty((bn * h * w, cls), dtype=torch.float32, device=feats.device)
    check(L.ls3d_nchw_to_nhwc(_ptr(feats), bn, c, h * w, _ptr(f), _stream(feats)), "ls3d_nchw_to_nhwc")
    check(L.ls3d_nchw_to_nhwc(_ptr(probs), bn, cls, h * w, _ptr(p), _stream(feats)), "ls3d_nchw_to_nhwc")
    off = torch.arange(0, (batch_size + 1) * rows, rows, dtype=_i32, device=feats.device)
    emb = sfam(f, p, off, batch_size, rows)  # [B, cls, C]
    return emb.permute(0, 2, 1).contiguous().unsqueeze(3)


class SffmModel(object):
    """device weights + the host-side descriptor of ls3d_sffm_decoder.  `layers`: dicts of packing.PackedWeight / tensors"""

    def __init__(self, w_in, b_in, layers, norm, d_in, d_model, heads, ffn):
        from ._lib import Sffm, SffmLayer
        F = lambda pw: pw.for_nt(3, F32)
        self.keep = [F(w_in), b_in]
        arr = (SffmLayer * max(len(layers), 1))()
        for i, l in enumerate(layers):
            t = dict(wq=F(l["wq"]), bq=l["bq"], wo=F(l["wo"]), bo=l["bo"], w1a=F(l["w1a"]), w1b=F(l["w1b"]), b1=l["b1"], w2a=F(l["w2a"]),
                     w2b=F(l["w2b"]), b2=l["b2"], n2_gamma=l["n2"][0], n2_beta=l["n2"][1], n3_gamma=l["n3"][0], n3_beta=l["n3"][1])
            self.keep.extend(t.values())
            for k, v in t.items():
                setattr(arr[i], k, v.data_ptr())
            arr[i].n2_eps, arr[i].n3_eps = float(l["n2"][2]), float(l["n3"][2])
            # the same matrices as three exact bf16 planes for the register-resident (transposed) form of the decoder: input channels of every
            # 16-block in the order the MFMA's C layout hands them on (include/ls3d.h: ls3d_sffm_layer_t)
            for k in ("wq", "wo", "w1a", "w1b", "w2a", "w2b"):
                t = self._transposed_planes(l[k])
                self.keep.append(t)
                setattr(arr[i], k + "_planes", t.data_ptr())
        self.layers = arr
        self.keep.extend([norm[0], norm[1]] if norm is not None else [])
        w_in_planes = self._transposed_planes(w_in)
        self.keep.append(w_in_planes)
        self.c = Sffm(self.keep[0].data_ptr(), b_in.data_ptr(), arr, len(layers), int(d_in), int(d_model), int(heads), int(ffn),
                      norm[0].data_ptr() if norm is not None else None, norm[1].data_ptr() if norm is not None else None,
                      float(norm[2]) if norm is not None else 0.0, 0, w_in_planes.data_ptr(), None, 0)

    @staticmethod
    def _transposed_planes(pw):
        """PackedWeight of a Linear ([in][out] plain) -> ls3d_tile_conv_pack of the matrix with packed input row 16 c + 8 kk + q taken from
        plain row 16 c + 8 (q // 4) + 4 kk + q % 4"""
        cin = pw.cin
        k = torch.arange(cin, device=pw.plain.device)
        c, kk, q = k // 16, (k % 16) // 8, k % 8
        src = 16 * c + 8 * (q // 4) + 4 * kk + q % 4
        plain = pw.plain.reshape(-1, pw.plain.shape[-1])
        if plain.shape[0] < cin:  # rows beyond the module's input width are zero padding
            plain = torch.cat([plain, plain.new_zeros((cin - plain.shape[0], plain.shape[1]))])
        return tile_conv_pack(plain[src].contiguous(), 1, cin, cin, pw.cout)


_SFFM_ATTENTION = 0
_SFFM_ABLATE = 0  # measurement only (tools/bench_decoder.py): parts of k_sffm_decoder_rt left out, see SfParams::ablate
_SFFM_PLANES = True  # A/B: the decoder's GEMMs on the 3-plane bf16 split in the 3-plane precisions


def set_sffm_attention(mode):
    """"f32" (default: exact-f32 MFMA), "bf16" / "fp8" (MFMA operands rounded to bf16 / OCP e4m3, f32 accumulation and softmax) or
    "valu" (vector pipe): the `attention` field of the ls3d_sffm_t descriptor that sffm_decoder passes with each call"""
    global _SFFM_ATTENTION
    _SFFM_ATTENTION = {"f32": 0, "bf16": 1, "valu": 2, "fp8": 3}[mode]


def sffm_decoder(x, points, kv, L, batch, model, pt_off=None):
    """fused point side of the SF-Phase decoder (ls3d_sffm_decoder); returns None when the shape is not supported.  In the 3-plane precisions
    (and with exact-f32 attention, L <= 64) the register-resident form runs: GEMMs on the 3-plane bf16 split, f32-grade like the convolutions'
    (pt_off: the frames' first rows, computed here when the caller has none)"""
    n = x.shape[0]
    out = torch.empty((n, model.c.d_model), dtype=torch.float32, device=x.device)
    model.c.attention = _SFFM_ATTENTION
    planes = _PRECISION in (BF16X6, BF16X8, BF16) and _SFFM_PLANES and _SFFM_ATTENTION == 0 and L <= 64
    model.c.gemm_products = (6 | (_SFFM_ABLATE << 8)) if planes else 0
    if planes:
        if pt_off is None:
            pt_off = frame_offsets(points if points.dim() == 2 else points.unsqueeze(1).contiguous(), batch)
        model.c.pt_off = pt_off.data_ptr()
    rc = _L().ls3d_sffm_decoder(_ptr(x), x.shape[1], n, _ptr(points), points.shape[1] if points.dim() == 2 else 1, _ptr(kv), int(L), int(batch),
                                ctypes.byref(model.c), _ptr(out), out.shape[1], _stream(x))
    if rc == _lib.ERR_UNSUPPORTED:
        return None
    check(rc, "ls3d_sffm_decoder")
    return out


class SffmMemoryModel(object):
    """device weights (transposed to [in][out]) + the host-side layer descriptors of ls3d_sffm_memory.  `layers`: dicts with wqkv_t, bqkv, wo_t,
    bo, n1 = (gamma, beta, eps), wk_t, bk, wv_t, bv"""

    def __init__(self, layers, embed, heads):
        from ._lib import SffmMemoryLayer
        self.embed, self.heads, self.keep = int(embed), int(heads), []
        arr = (SffmMemoryLayer * max(len(layers), 1))()
        for i, l in enumerate(layers):
            t = dict(wqkv_t=l["wqkv_t"], bqkv=l["bqkv"], wo_t=l["wo_t"], bo=l["bo"], n1_gamma=l["n1"][0], n1_beta=l["n1"][1], wk_t=l["wk_t"], bk=l["bk"],
                     wv_t=l["wv_t"], bv=l["bv"])
            self.keep.extend(t.values())
            for k, v in t.items():
                _ptr(v)  # device / contiguity check
                setattr(arr[i], k, v.data_ptr())
            arr[i].n1_eps = float(l["n1"][2])
        self.layers, self.num_layers = arr, len(layers)


def sffm_memory(mem, batch, L, model, return_memory=False):
    """class-embedding side of the SF-Phase decoder for all layers in one launch (ls3d_sffm_memory): mem [batch * L, 96] -> kv
    [2 * layers, batch, 96, L] for sffm_decoder (and the memory after the last layer); None when the shape is not supported"""
    kv = torch.empty((2 * model.num_layers, batch, model.embed, L), dtype=torch.float32, device=mem.device)
    out = torch.empty_like(mem) if return_memory else None
    rc = _L().ls3d_sffm_memory(_ptr(mem), int(batch), int(L), model.embed, model.heads, model.num_layers, model.layers, _ptr(kv), _ptr(out), _stream(mem))
    if rc == _lib.ERR_UNSUPPORTED:
        return None
    check(rc, "ls3d_sffm_memory")
    return (kv, out) if return_memory else kv


def cross_attn(q, k, v, batch, heads, points):
    n, e = q.shape
    L = k.numel() // (batch * e)
    out = torch.empty((n, e), dtype=torch.float32, device=q.device)
    check(_L().ls3d_cross_attn(_ptr(q), _ptr(k), _ptr(v), batch, heads, e, L, _ptr(points), points.shape[1], n, _ptr(out),
                               _stream(q)), "ls3d_cross_attn")
    return out


# ---------------------------------------------------------------------------------------------- camera-branch input step
def points_cp(points, ref_to_global, cams_from_global, intrinsics, im_shape=(900, 1600), xyz_col=0):
    """loading.py:384-413 on the GPU: points [N, >=3] (xyz at xyz_col), ref_to_global [4,4], cams_from_global [ncam,4,4],
    intrinsics [ncam,3,3] (numpy float64, host) -> points_cp [N,3] = (cam_id + 1, u, v) or -100s"""
    n = points.shape[0]
    r2g = np.ascontiguousarray(ref_to_global, np.float64)
    c2g = np.ascontiguousarray(cams_from_global, np.float64)
    K = np.ascontiguousarray(intrinsics, np.float64)
    ncam = c2g.shape[0]
    out = torch.empty((n, 3), dtype=torch.float32, device=points.device)
    dp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    check(_L().ls3d_points_cp(_ptr(points), points.shape[1], xyz_col, n, dp(r2g), dp(c2g), dp(K), ncam, int(im_shape[0]), int(im_shape[1]),
                              _ptr(out), _stream(points)), "ls3d_points_cp")
    return out


def points_cuv(points_cp_, ncam, res_shape):
    """segpreprocess.py:649-671: points_cp [N,3] in feature-map-input pixel coordinates -> points_cuv [N,4]"""
    n = points_cp_.shape[0]
    out = torch.empty((n, 4), dtype=torch.float32, device=points_cp_.device)
    check(_L().ls3d_points_cuv(_ptr(points_cp_), n, int(ncam), int(res_shape[0]), int(res_shape[1]), _ptr(out), _stream(points_cp_)),
          "ls3d_points_cuv")
    return out


# ---------------------------------------------------------------------------------------------- segmentation loss
def seg_loss_forward(logits, labels, ignore):
    """(out2 = [cross entropy, Lovasz-Softmax] on the device, workspace for seg_loss_backward) of flat [P, C] f32 logits and int32 labels
    (ls3d_seg_loss_forward)"""
    P, C = logits.shape
    ws = _ws(_L().ls3d_seg_loss_workspace_bytes(P, C), logits)
    out = torch.empty((2,), dtype=torch.float32, device=logits.device)
    check(_L().ls3d_seg_loss_forward(_ptr(logits), logits.shape[1], _ptr(labels), P, C, int(ignore), _ptr(ws), ctypes.c_size_t(ws.numel()), _ptr(out),
                                     _stream(logits)), "ls3d_seg_loss_forward")
    # the backward reads the head of the workspace only (softmax, Lovasz gradient, counts): keep a copy of that prefix for autograd and let the
    # sort's five [P, C] arrays and histograms go back to the allocator now (Waymo, 2 frames: ~165 MB per prediction level, two levels per step)
    saved = _L().ls3d_seg_loss_saved_bytes(P, C)
    return out, (ws[:saved].clone() if 0 < saved < ws.numel() else ws)


def seg_loss_backward(labels, shape, ignore, ws, grad_ce, grad_lv):
    """d (grad_ce * ce + grad_lv * lovasz) / d logits from the forward's workspace; grad_ce / grad_lv: 1-element device tensors or None (= 1)"""
    P, C = shape
    grad = torch.empty((P, C), dtype=torch.float32, device=labels.device)
    check(_L().ls3d_seg_loss_backward(_ptr(labels), P, C, int(ignore), _ptr(ws), ctypes.c_size_t(ws.numel()), _vp(grad_ce), _vp(grad_lv), _ptr(grad), C,
                                      _stream(labels)), "ls3d_seg_loss_backward")
    return grad


# ---------------------------------------------------------------------------------------------- LayerNorm (training step)
def layer_norm_forward(x, gamma, beta, eps, want_stats=True):
    n, c = x.shape
    y = torch.empty_like(x)
    stats = torch.empty((n, 2), dtype=torch.float32, device=x.device) if want_stats else None
    check(_L().ls3d_layer_norm_forward(_ptr(x), n, c, _ptr(gamma), _ptr(beta), ctypes.c_float(float(eps)), _ptr(y), _vp(stats), _stream(x)),
          "ls3d_layer_norm_forward")
    return y, stats


def layer_norm_backward(x, dy, gamma, stats):
    n, c = x.shape
    dx = torch.empty_like(x)
    dg = torch.empty((c,), dtype=torch.float32, device=x.device)
    db = torch.empty((c,), dtype=torch.float32, device=x.device)
    ws = _ws(_L().ls3d_layer_norm_workspace_bytes(n, c), x)
    check(_L().ls3d_layer_norm_backward(_ptr(x), _ptr(dy), _ptr(gamma), _ptr(stats), n, c, _ptr(dx), _ptr(dg), _ptr(db), _ptr(ws), ctypes.c_size_t(ws.numel()),
                                        _stream(x)), "ls3d_layer_norm_backward")
    return dx, dg, db


class _LayerNormFn(torch.autograd.Function):
    """nn.LayerNorm over the last dimension of [n, c] rows on ls3d_layer_norm_forward / _backward (csrc/norm.hip)"""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        y, stats = layer_norm_forward(x, weight.detach().contiguous(), bias.detach().contiguous(), eps)
        ctx.save_for_backward(x, weight, stats)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, stats = ctx.saved_tensors
        dx, dg, db = layer_norm_backward(x, gy.contiguous(), weight.detach().contiguous(), stats)
        return dx, dg, db, None


# ---------------------------------------------------------------------------------------------- BatchNorm1d, training mode
def batch_norm_supported(x):
    """shapes ls3d_batch_norm_* take: [n, c] f32 rows on the device, c % 4 == 0, c <= 256, 256 % (c / 4) == 0"""
    return (x.dim() == 2 and x.dtype == torch.float32 and (x.is_cuda or _SIM) and x.shape[1] % 4 == 0 and 4 <= x.shape[1] <= 256
            and 256 % (x.shape[1] // 4) == 0 and x.stride(1) == 1 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0)


def batch_norm_stats(x):
    """-> [2 c + 1]: per-column mean and sum of squared deviations over the rows of x, then the row count as a float (ls3d_batch_norm_stats): the
    triple a data-parallel step gathers over its ranks"""
    n, c = x.shape
    out = torch.empty((2 * c + 1,), dtype=torch.float32, device=x.device)
    ws = _ws(_L().ls3d_batch_norm_workspace_bytes(n, c), x)
    check(_L().ls3d_batch_norm_stats(_vp_any(x), x.stride(0), n, c, _ptr(ws), ctypes.c_size_t(ws.numel()), _ptr(out), _stream(x)), "ls3d_batch_norm_stats")
    return out


def token_attention_supported(q, k):
    """shapes ls3d_token_attention_* take: q [n, H, 24] f32 rows on the device, k [H, 24, L] with L in {34, 38, 40, 46}"""
    return ((q.is_cuda or _SIM) and q.dtype == torch.float32 and q.dim() == 3 and q.shape[2] == 24 and k.dim() == 3 and k.shape[2] in (34, 38, 40, 46)
            and q.is_contiguous() and q.data_ptr() % 16 == 0)


def token_attention_forward(q, k, v, scale, out=None):
    """softmax(scale q_h K_h) V_h per point and head: q [n, H, hd], k / v [H, hd, L] -> [n, H, hd] (ls3d_token_attention_forward); out: a row slice
    of a caller's buffer (the frames of a batch write one output)"""
    n, H, hd = q.shape
    if out is None:
        out = torch.empty_like(q)
    check(_L().ls3d_token_attention_forward(_vp_any(q), n, H, hd, _ptr(k.contiguous()), _ptr(v.contiguous()), k.shape[2], ctypes.c_float(scale), _vp_any(out),
                                            _stream(q)), "ls3d_token_attention_forward")
    return out


def token_attention_backward(q, dout, k, v, scale, dq=None):
    """-> (dq [n, H, hd], dk [H, hd, L], dv [H, hd, L]) of token_attention_forward, the probabilities recomputed from q (ls3d_token_attention_backward);
    q / dout / dq may be row slices of the batch's buffers"""
    n, H, hd = q.shape
    L = k.shape[2]
    if dq is None:
        dq = torch.empty_like(q)
    dk = torch.empty((H, hd, L), dtype=torch.float32, device=q.device)
    dv = torch.empty((H, hd, L), dtype=torch.float32, device=q.device)
    ws = _ws(_L().ls3d_token_attention_workspace_bytes(n, H, L), q)
    check(_L().ls3d_token_attention_backward(_vp_any(q), _vp_any(dout), n, H, hd, _ptr(k.contiguous()), _ptr(v.contiguous()), L, ctypes.c_float(scale), _vp_any(dq),
                                             _ptr(dk), _ptr(dv), _ptr(ws), ctypes.c_size_t(ws.numel()), _stream(q)), "ls3d_token_attention_backward")
    return dq, dk, dv


def column_sums(x):
    """x[n, c].sum(0) on ls3d_column_sums (deterministic row blocks + fixed tree), or torch's reduction where the shape is not covered"""
    n, c = x.shape
    if not ((x.is_cuda or _SIM) and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1 and 1 <= c <= 256 and n >= 4096):
        return x.sum(0)
    out = torch.empty((c,), dtype=torch.float32, device=x.device)
    ws = _ws(_L().ls3d_column_sums_workspace_bytes(n, c), x)
    check(_L().ls3d_column_sums(_vp_any(x), x.stride(0), n, c, _ptr(ws), ctypes.c_size_t(ws.numel()), _ptr(out), _stream(x)), "ls3d_column_sums")
    return out


def batch_norm_finalize(parts, c, eps, bn=None):
    """the ranks' triples parts [world, 2 c + 1] (or one [2 c + 1]) -> (mean [c], var [c] biased, rstd [c], count [1] on the device) in ONE launch, and
    - bn given (an nn.BatchNorm1d with running statistics and a momentum) - its running_mean / running_var / num_batches_tracked update"""
    world = parts.shape[0] if parts.dim() == 2 else 1
    dev = parts.device
    out = torch.empty((3 * c + 1,), dtype=torch.float32, device=dev)
    mean, var, rstd, count = out[:c], out[c:2 * c], out[2 * c:3 * c], out[3 * c:]
    rm = rv = nbt = None
    mom = 0.0
    if bn is not None:
        rm, rv, nbt, mom = bn.running_mean, bn.running_var, bn.num_batches_tracked, float(bn.momentum)
    check(_L().ls3d_batch_norm_finalize(_ptr(parts), world, c, -1, ctypes.c_float(eps), ctypes.c_float(mom), _ptr(rm), _ptr(rv), _ptr(nbt), _vp_any(mean), _vp_any(var),
                                        _vp_any(rstd), _vp_any(count), _stream(parts)), "ls3d_batch_norm_finalize")
    return mean, var, rstd, count


def batch_norm_apply(x, mean, rstd, gamma, beta, res=None, relu=False):
    n, c = x.shape
    y = torch.empty((n, c), dtype=torch.float32, device=x.device)
    check(_L().ls3d_batch_norm_apply(_vp_any(x), x.stride(0), n, c, _vp_any(mean), _vp_any(rstd), _ptr(gamma), _ptr(beta), _vp_any(res) if res is not None else None,
                                     res.stride(0) if res is not None else 0, 1 if relu else 0, _ptr(y), c, _stream(x)), "ls3d_batch_norm_apply")
    return y


def batch_norm_backward_sums(x, dy, y, mean, rstd):
    n, c = x.shape
    sums = torch.empty((2 * c,), dtype=torch.float32, device=x.device)
    ws = _ws(_L().ls3d_batch_norm_workspace_bytes(n, c), x)
    check(_L().ls3d_batch_norm_backward_sums(_vp_any(x), x.stride(0), _ptr(dy), _vp(y), n, c, _vp_any(mean), _vp_any(rstd), _ptr(ws), ctypes.c_size_t(ws.numel()),
                                             _ptr(sums), _stream(x)), "ls3d_batch_norm_backward_sums")
    return sums


def batch_norm_backward_apply(x, dy, y, mean, rstd, gamma, sums, count, want_dres):
    """count: a host number, or the device scalar of batch_norm_finalize (no host synchronisation)"""
    n, c = x.shape
    dx = torch.empty((n, c), dtype=torch.float32, device=x.device)
    dres = torch.empty((n, c), dtype=torch.float32, device=x.device) if want_dres else None
    on_dev = torch.is_tensor(count)
    check(_L().ls3d_batch_norm_backward_apply(_vp_any(x), x.stride(0), _ptr(dy), _vp(y), n, c, _vp_any(mean), _vp_any(rstd), _ptr(gamma), _ptr(sums),
                                              ctypes.c_float(0.0 if on_dev else 1.0 / max(float(count), 1.0)), _vp_any(count) if on_dev else None,
                                              _ptr(dx), _vp(dres), _stream(x)), "ls3d_batch_norm_backward_apply")
    return dx, dres


class _BatchNormTrainFn(torch.autograd.Function):
    """[relu](BatchNorm1d(x) [+ res]) with batch statistics on ls3d_batch_norm_* (csrc/norm.hip): statistics (2 launches), the merge of the ranks'
    triples + rstd + running statistics (ONE launch, ls3d_batch_norm_finalize), apply.  `gather`: None, or a callable [2 c + 1] -> [world, 2 c + 1]
    that all-gathers the ranks' (mean, M2, n) triples (syncbn.py) - its partner `reduce` all-reduces the backward's two column sums.  `bn`: the
    module whose running statistics are updated in the same launch (None: no update).  Returns y and the batch statistics."""

    @staticmethod
    def forward(ctx, x, weight, bias, res, eps, relu, gather, reduce, bn):
        n, c = x.shape
        st = batch_norm_stats(x)
        parts = st if gather is None else gather(st)
        mean, var, rstd, count = batch_norm_finalize(parts, c, eps, bn)
        y = batch_norm_apply(x, mean, rstd, weight.detach().contiguous(), bias.detach().contiguous(), res, relu)
        ctx.save_for_backward(x, y if relu else None, mean, rstd, weight, count)
        ctx.reduce, ctx.has_res = reduce, res is not None and res.requires_grad
        ctx.mark_non_differentiable(mean, var, count)
        return y, mean, var, count

    @staticmethod
    def backward(ctx, gy, _gm, _gv, _gc):
        x, y, mean, rstd, weight, count = ctx.saved_tensors
        c = x.shape[1]
        gy = gy.contiguous()
        sums = batch_norm_backward_sums(x, gy, y, mean, rstd)
        gb, gw = sums[:c], sums[c:]  # local sums: the parameter gradients (DDP averages them over the ranks)
        if ctx.reduce is not None:
            sums = ctx.reduce(sums)  # (a copy: gb / gw keep the local sums)
        dx, dres = batch_norm_backward_apply(x, gy, y, mean, rstd, weight.detach().contiguous(), sums, count, ctx.has_res)
        return dx, gw, gb, dres, None, None, None, None, None


_BN_KERNELS = True
_TORCH_RELU = torch.relu


def batch_norm_train(bn, x, res=None, relu=False):
    """bn (nn.BatchNorm1d in training mode, or its count-weighted SyncBN variant) applied to the rows x with batch statistics, fused with the residual
    add and the ReLU that follow it; updates the running statistics as nn.BatchNorm1d does.  -> y, or None when the shape / module is not covered
    (the caller composes it from the torch modules)"""
    if not (_BN_KERNELS and bn.training and bn.affine and batch_norm_supported(x) and (res is None or batch_norm_supported(res)) and x.shape[0] > 1):
        return None
    if bn.weight.data_ptr() % 16 or bn.bias.data_ptr() % 16:  # the kernels read gamma / beta as float4 (flattened-parameter views may not be aligned)
        return None
    gather = reduce = None
    sync = getattr(bn, "_ls3d_sync", None)
    if sync is not None:
        gather, reduce = sync()  # (None, None) without an active process group
    fuse_relu = relu and torch.relu is _TORCH_RELU  # an instrumented torch.relu (tests pin / count ReLU gates by patching it) still sees the ReLU
    in_kernel = bn.track_running_stats and bn.momentum is not None  # the running statistics in ls3d_batch_norm_finalize's launch
    y, mean, var, count = _BatchNormTrainFn.apply(x, bn.weight, bn.bias, res, bn.eps, fuse_relu, gather, reduce, bn if in_kernel else None)
    if relu and not fuse_relu:
        y = torch.relu(y)
    if bn.track_running_stats and not in_kernel:  # momentum None: the cumulative average needs the batch counter on the host
        with torch.no_grad():
            bn.num_batches_tracked += 1
            m = 1.0 / float(bn.num_batches_tracked)
            bn.running_mean.mul_(1 - m).add_(mean, alpha=m)
            bn.running_var.mul_(1 - m).add_(var * (count / (count - 1.0).clamp_min(1.0)), alpha=m)  # unbiased, as nn.BatchNorm does
    return y
