"""Kernel LOGIC tests without a GPU: the unmodified HIP sources are compiled for the host against tests/hipsim (a
fiber-based emulation of blocks / waves / LDS / MFMA — test infrastructure, see tests/hipsim/hip/hip_runtime.h) and driven
through the same C ABI and the same Python host layer as on the MI355X.  They catch indexing / protocol bugs before a
GPU run; numerical parity proper is the job of tests/test_gpu_parity.py on the real device."""
import os
import sys

import numpy as np
import pytest
import torch

from lidarseg3d_amd import _lib, ops, point_heads, readers, scn_unet, synth
from lidarseg3d_amd.packing import PackedWeight
from oracle import ref as orc
from tests.util import golden, seeded_sd

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module", autouse=True)
def sim_library():
    sys.path.insert(0, os.path.join(HERE, "hipsim"))
    import build_sim
    path = build_sim.build()
    _lib.use_library_for_testing(path)
    ops.set_sim(True)
    yield
    ops.set_sim(False)
    _lib.use_library_for_testing(None)


@pytest.mark.parametrize("tag", ["nusc", "nusc_cap", "kitti"])
def test_voxelize_bit_exact_vs_reference(tag):
    g = golden("voxelize_%s.npz" % tag)
    pts, mv = torch.from_numpy(g["points"]), int(g["max_voxels"])
    for mode, pre in (("numba", "numba"), ("break", "cpp_hard")):
        v, c, n, nv = ops.voxelize_hard(pts, g["voxel_size"], g["pc_range"], 5, mv, overflow=mode)
        V = int(nv)
        assert V == g[pre + "_coors"].shape[0]
        assert np.array_equal(c[:V].numpy(), g[pre + "_coors"]) and np.array_equal(n[:V].numpy(), g[pre + "_num"])
        assert np.array_equal(v[:V].numpy(), g[pre + "_voxels"])
    assert np.array_equal(ops.voxelize_dynamic(pts, g["voxel_size"], g["pc_range"]).numpy(), g["cpp_dyn_coors"])


@pytest.mark.parametrize("n_in,n_out,cin,cout,dens", [(500, 700, 32, 64, 0.06), (300, 333, 64, 32, 0.12), (400, 260, 128, 128, 0.35), (600, 601, 64, 128, 0.2)])
def test_sparse_six_product_gather_pipelines_are_bit_identical(n_in, n_out, cin, cout, dens):
    """k_gather_gemm_x6 (strided / inverse layers of the 3-plane modes: rows gathered two stages ahead, neighbour indices of the tile in LDS)
    against the one-stage kernel it replaces (flags bit 2 of ls3d_gather_gemm): the same products in the same order - bit-identical, with
    the fused epilogue, mask-sorted rows and tables as sparse as a stride-2 layer's (1.6 pairs per row)"""
    from lidarseg3d_amd.packing import PackedWeight
    rng = np.random.default_rng(n_in + cout)
    tbl = rng.integers(0, n_in, size=(n_out, 27)).astype(np.int32)
    tbl[rng.uniform(size=tbl.shape) > dens] = -1
    tbl[5] = -1  # a row without neighbours
    x = torch.from_numpy(rng.normal(size=(n_in, cin)).astype(np.float32))
    w = torch.from_numpy((rng.normal(size=(27, cin, cout)) * 0.1).astype(np.float32))
    pw, t = PackedWeight(w, 27, cin, cin, cout), torch.from_numpy(tbl)
    sc, sh = torch.rand(cout) + 0.5, torch.randn(cout)
    outs = {}
    try:
        ops.set_precision("bf16x6")
        for order in (ops.rulebook_order(t), None):
            for fl in (0, 4):
                ops.set_gemm_flags(fl)
                outs[fl] = ops.gather_gemm(x, pw, tbl=t, order=order, cout=cout, scale=sc, shift=sh, relu=True)
            assert torch.equal(outs[0], outs[4])
    finally:
        ops.set_gemm_flags(0)
        ops.set_precision("f32")
    ref = torch.zeros(n_out, cout, dtype=torch.float64)
    for k in range(27):
        o = np.nonzero(tbl[:, k] >= 0)[0]
        ref[o] += x[tbl[o, k]].double() @ w[k].double()
    ref = torch.relu(ref * sc.double() + sh.double())
    assert float((outs[0].double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())


def test_voxel_ops_modules_vs_reference_cpp():
    """Voxelization (hard and max_num_points = -1), HardSimpleVFE, DynamicSimpleVFE, DynamicScatterWithDistance: tests/voxel_cases.py"""
    from tests import voxel_cases
    voxel_cases.run("cpu")


@pytest.mark.parametrize("L,n", [(34, 300), (46, 257), (40, 70), (38, 64)])
def test_token_attention_kernels_vs_torch_autograd(L, n):
    """ls3d_token_attention_forward / _backward (the SF-Phase decoder's point -> class-token attention in training, csrc/tokenattn.hip) through
    point_heads._TokenAttention against torch autograd of the same einsum / softmax algebra in float64: output, d q, d k, d v; ragged last block;
    an unsupported token count takes the torch branch of the same Function"""
    from lidarseg3d_amd.point_heads import _TokenAttention
    rng = np.random.default_rng(L + n)
    H, hd = 4, 24
    q = torch.from_numpy(rng.normal(size=(n, H, hd)).astype(np.float32)).requires_grad_()
    k = torch.from_numpy(rng.normal(size=(H, hd, L)).astype(np.float32)).requires_grad_()
    v = torch.from_numpy(rng.normal(size=(H, hd, L)).astype(np.float32)).requires_grad_()
    g = torch.from_numpy(rng.normal(size=(n, H, hd)).astype(np.float32))
    assert ops.token_attention_supported(q, k)
    out = _TokenAttention.apply(q, k[None], v[None], hd ** -0.5, [0, n])
    out.backward(g)
    qd, kd, vd = (t.detach().double().requires_grad_() for t in (q, k, v))
    ref = torch.einsum("nhl,hdl->nhd", torch.softmax(torch.einsum("nhd,hdl->nhl", qd, kd) * hd ** -0.5, dim=-1), vd)
    ref.backward(g.double())
    for name, got, want in (("out", out, ref), ("dq", q.grad, qd.grad), ("dk", k.grad, kd.grad), ("dv", v.grad, vd.grad)):
        err = float((got.detach().double() - want.detach()).abs().max()) / float(want.detach().abs().max())
        assert err <= 2e-6, (name, err)
    # a batch of two frames (ragged, each with its own tokens): one output / one d q, the frames' slices handed to the kernels
    off = [0, n // 3, n]
    kb = torch.from_numpy(rng.normal(size=(2, H, hd, L)).astype(np.float32)).requires_grad_()
    vb = torch.from_numpy(rng.normal(size=(2, H, hd, L)).astype(np.float32)).requires_grad_()
    q2 = q.detach().clone().requires_grad_()
    ob = _TokenAttention.apply(q2, kb, vb, hd ** -0.5, off)
    ob.backward(g)
    qd, kd, vd = (t.detach().double().requires_grad_() for t in (q2, kb, vb))
    refb = torch.cat([torch.einsum("nhl,hdl->nhd", torch.softmax(torch.einsum("nhd,hdl->nhl", qd[off[b]:off[b + 1]], kd[b]) * hd ** -0.5, dim=-1), vd[b])
                      for b in range(2)], 0)
    refb.backward(g.double())
    for name, got, want in (("out", ob, refb), ("dq", q2.grad, qd.grad), ("dk", kb.grad, kd.grad), ("dv", vb.grad, vd.grad)):
        err = float((got.detach().double() - want.detach()).abs().max()) / float(want.detach().abs().max())
        assert err <= 2e-6, ("batch", name, err)
    # 50 tokens: not a compiled token count -> the torch branch
    k2 = torch.from_numpy(rng.normal(size=(H, hd, 50)).astype(np.float32))
    assert not ops.token_attention_supported(q, k2)
    o2 = _TokenAttention.apply(q.detach(), k2[None], k2[None], hd ** -0.5, [0, n])
    r2 = torch.einsum("nhl,hdl->nhd", torch.softmax(torch.einsum("nhd,hdl->nhl", q.detach(), k2) * hd ** -0.5, dim=-1), k2)
    assert float((o2 - r2).abs().max()) <= 1e-5


def test_column_sums_kernel_is_the_float64_sum_and_deterministic():
    """ls3d_column_sums (a Linear layer's bias gradient): column slices with a row stride, ragged row counts around the 512-row blocks, against
    float64; the same bits on every call; shapes it does not take fall back to torch"""
    rng = np.random.default_rng(0)
    for n, c, ld in ((4096, 64, 64), (5000, 32, 96), (9217, 128, 128), (4097, 4, 8), (4500, 96, 96), (4100, 192, 256), (4096, 20, 24), (4300, 23, 23), (4200, 17, 40), (4096, 1, 3)):
        wide = torch.from_numpy((rng.normal(size=(n, ld)) * np.exp(rng.normal(size=(n, ld)))).astype(np.float32))
        x = wide[:, ld - c:]
        got = ops.column_sums(x)
        want = x.double().sum(0)
        assert float((got.double() - want).abs().max()) <= 1e-6 * float(x.double().abs().sum(0).max())
        assert torch.equal(ops.column_sums(x), got)
    small = torch.from_numpy(rng.normal(size=(100, 23)).astype(np.float32))  # few rows: torch
    assert torch.equal(ops.column_sums(small), small.sum(0))


@pytest.mark.parametrize("n,c,relu,res", [(1500, 64, True, True), (777, 16, True, False), (2100, 128, False, False), (5, 32, True, True), (1030, 32, False, True)])
def test_batch_norm_train_kernels_vs_torch(n, c, relu, res):
    """ls3d_batch_norm_* (training-mode BatchNorm1d with the ReLU / residual add fused in, csrc/norm.hip) against nn.BatchNorm1d + add + relu under
    torch autograd: output, dx, dres, dgamma, dbeta and the running statistics; row counts that are no multiple of the 512-row blocks"""
    torch.manual_seed(n + c)
    x = (torch.randn(n, c) * 2 + 1.5).requires_grad_(True)
    r = torch.randn(n, c).requires_grad_(True) if res else None
    bn = torch.nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_()
    ref = torch.nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).train()
    ref.load_state_dict(bn.state_dict())
    y = ops.batch_norm_train(bn, x, res=r, relu=relu)
    assert y is not None
    x2 = x.detach().clone().requires_grad_(True)
    r2 = r.detach().clone().requires_grad_(True) if res else None
    z = ref(x2)
    z = z + r2 if res else z
    z = torch.relu(z) if relu else z
    g = torch.randn(n, c)
    y.backward(g)
    z.backward(g)
    np.testing.assert_allclose(y.detach().numpy(), z.detach().numpy(), rtol=0, atol=3e-6)
    np.testing.assert_allclose(x.grad.numpy(), x2.grad.numpy(), rtol=0, atol=2e-6)
    if res:
        assert torch.equal(r.grad, r2.grad)
    for a, b in ((bn.weight.grad, ref.weight.grad), (bn.bias.grad, ref.bias.grad)):
        assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max()) * max(1.0, n ** 0.5 / 8)
    np.testing.assert_allclose(bn.running_mean.numpy(), ref.running_mean.numpy(), rtol=0, atol=1e-7)
    np.testing.assert_allclose(bn.running_var.numpy(), ref.running_var.numpy(), rtol=1e-6, atol=1e-7)
    assert int(bn.num_batches_tracked) == 1
    assert ops.batch_norm_train(bn.eval(), x.detach()) is None  # eval mode: the caller's folded epilogue / torch path
    assert ops.batch_norm_train(torch.nn.BatchNorm1d(13).train(), torch.randn(40, 13)) is None  # 13 channels: not covered, torch composes it


def test_voxelize_empty_and_all_outside():
    cfg = synth.NUSC
    pts = torch.full((10, 5), 1000.0)
    v, c, n, nv = ops.voxelize_hard(pts, cfg["voxel_size"], cfg["pc_range"], 5, 100)
    assert int(nv) == 0
    assert (ops.voxelize_dynamic(pts, cfg["voxel_size"], cfg["pc_range"]) == -1).all()


def test_batched_voxelize_equals_per_frame_collate():
    cfg = synth.NUSC
    frames = [synth.lidar_frame(1500, seed=1, **cfg), synth.lidar_frame(40, seed=2, **cfg), synth.lidar_frame(900, seed=3, **cfg)]
    pts = np.concatenate([np.concatenate([np.full((f.shape[0], 1), b, np.float32), f], 1) for b, f in enumerate(frames)])
    v, c, n, nv = ops.voxelize_hard(torch.from_numpy(pts), cfg["voxel_size"], cfg["pc_range"], 5, 900000, batched=True)
    V = int(nv)
    want = orc.collate_frames(frames, cfg["voxel_size"], cfg["pc_range"], 5, 300000)
    assert torch.equal(c[:V], want["coordinates"]) and torch.equal(n[:V], want["num_points"]) and torch.equal(v[:V], want["voxels"])


def test_dynamic_scatter_vs_reference_cpp():
    g = golden("voxelize_nusc.npz")
    gs = orc.grid_size(g["voxel_size"], g["pc_range"])
    shape = [int(gs[2]), int(gs[1]), int(gs[0])]
    f, vc, p2v, nv = ops.dynamic_scatter(torch.from_numpy(g["points"]), torch.from_numpy(g["cpp_dyn_coors"]), shape, "mean")
    V = int(nv)
    assert np.array_equal(vc[:V].numpy(), g["cpp_scatter_coors"])
    want = torch.from_numpy(g["cpp_scatter_voxels"]).sum(1) / torch.from_numpy(g["cpp_scatter_num"]).float()[:, None]
    np.testing.assert_allclose(f[:V].numpy(), want.numpy(), rtol=2e-6, atol=1e-6)
    f, _, _, _ = ops.dynamic_scatter(torch.from_numpy(g["points"]), torch.from_numpy(g["cpp_dyn_coors"]), shape, "max")
    np.testing.assert_array_equal(f[:V].numpy(), g["cpp_scatter_voxels"].max(1))


@pytest.mark.parametrize("average", [True, False])
def test_dynamic_scatter_backward_vs_reference_composition(average):
    """voxel_ops.DynamicScatter under autograd == the reference's composition (padded [V,M,C] tensor, pinned here against the
    compiled reference's forward output, then torch mean / max) differentiated by torch; incl. points outside the range,
    negative-only channels (the zero padding wins the max: no gradient) and exact ties (first point wins)"""
    from lidarseg3d_amd import voxel_ops
    g = golden("voxelize_nusc.npz")
    pts = g["points"].copy()
    pts[:, 3] = -np.abs(pts[:, 3]) - 0.5     # a channel that is negative everywhere: padded voxels reduce to 0 under max
    pts[:, 4] = np.round(pts[:, 4])          # many exact ties inside a voxel
    coors = g["cpp_dyn_coors"]
    a = torch.from_numpy(pts).requires_grad_(True)
    vox, num = orc.dynamic_scatter_padded(a, coors, g["voxel_size"], g["pc_range"])
    np.testing.assert_array_equal(vox.detach().numpy()[:, :, :3], g["cpp_scatter_voxels"][:, :, :3])  # the reference's own padded tensor
    want = vox.sum(1) / num[:, None] if average else vox.max(1)[0]
    gen = torch.Generator().manual_seed(5)
    gout = torch.randn(want.shape, generator=gen)
    ga, = torch.autograd.grad(want, a, gout)
    b = torch.from_numpy(pts).requires_grad_(True)
    mod = voxel_ops.DynamicScatter(list(g["voxel_size"]), list(g["pc_range"]), average)
    f, vc = mod(b, torch.from_numpy(coors))
    assert np.array_equal(vc.numpy(), g["cpp_scatter_coors"]) and not vc.requires_grad
    np.testing.assert_allclose(f.detach().numpy(), want.detach().numpy(), rtol=2e-6, atol=1e-6)
    gb, = torch.autograd.grad(f, b, gout)
    if average:
        np.testing.assert_allclose(gb.numpy(), ga.numpy(), rtol=1e-6, atol=1e-7)
    else:
        np.testing.assert_array_equal(gb.numpy(), ga.numpy())
        assert float(gb[:, 3].abs().sum()) < float(gout[:, 3].abs().sum())  # some voxels' gradient went to the padding
    assert (gb[torch.from_numpy((coors < 0).any(1))] == 0).all()


def _segment_reference(src, index, n_seg, mode):
    """torch restatement of torch_scatter's dim-0 reductions (empty segments -> 0, arg -> n)"""
    n, c = src.shape
    if mode == "mean":
        out = torch.zeros((n_seg, c), dtype=torch.float64).index_add_(0, index, src.double())
        cnt = torch.zeros((n_seg,), dtype=torch.float64).index_add_(0, index, torch.ones(n, dtype=torch.float64))
        return (out / cnt.clamp(min=1)[:, None]).float(), None
    out = torch.full((n_seg, c), float("-inf")).scatter_reduce(0, index[:, None].expand(n, c), src, "amax", include_self=True)
    hit = src == out[index]
    rows = torch.arange(n)[:, None].expand(n, c)
    arg = torch.full((n_seg, c), n, dtype=torch.int64).scatter_reduce(0, index[:, None].expand(n, c), torch.where(hit, rows, n), "amin")
    return torch.where(torch.isinf(out), torch.zeros(()), out), arg


def test_segment_reduce_and_torch_scatter_signatures():
    from lidarseg3d_amd import scatter
    gen = torch.Generator().manual_seed(3)
    n, c, n_seg = 3000, 7, 220
    src = torch.randn(n, c, generator=gen)
    src[:, 2] = torch.round(src[:, 2])  # ties
    index = torch.randint(0, n_seg - 15, (n,), generator=gen)  # the last 15 segments stay empty
    index[index == 5] = 6                                       # and one in the middle
    want_mean, _ = _segment_reference(src, index, n_seg, "mean")
    want_max, want_arg = _segment_reference(src, index, n_seg, "max")
    got = scatter.scatter_mean(src, index, dim=0, dim_size=n_seg)
    np.testing.assert_allclose(got.numpy(), want_mean.numpy(), rtol=1e-5, atol=1e-6)
    got, arg = scatter.scatter_max(src, index, dim=0, dim_size=n_seg)
    assert torch.equal(got, want_max) and torch.equal(arg, want_arg)
    assert float(got[5].abs().max()) == 0 and bool((arg[5] == n).all()) and bool((arg[-1] == n).all())
    # dim_size inferred; 1-D int64 values (the majority-vote label path, voxel_encoder.py:419-421)
    cnt = torch.randint(1, 50, (n,), generator=gen)
    v, a = scatter.scatter_max(cnt, index)
    assert v.dtype == torch.int64 and v.shape == (int(index.max()) + 1,)
    w, wa = _segment_reference(cnt.float()[:, None], index, int(index.max()) + 1, "max")
    assert torch.equal(v, w[:, 0].long()) and torch.equal(a, wa[:, 0])
    with pytest.raises(NotImplementedError):
        scatter.scatter_mean(src, index, dim=1)


def test_rulebooks_bit_exact_vs_oracle_all_levels():
    g = golden("unet_nusc_c13.npz")
    coords = torch.from_numpy(g["coords"])
    shape = orc.spatial_shape(synth.NUSC["voxel_size"], synth.NUSC["pc_range"])
    rb = orc.UNetRulebooks(g["coords"], shape)
    assert np.array_equal(ops.rulebook_subm(coords, shape, (3, 3, 3)).numpy(), rb.subm1)
    cur, cshape = coords, list(shape)
    for want_c, want_tbl, ks, st, pd in ((rb.c2, rb.down2, (3, 3, 3), (2, 2, 2), (1, 1, 1)), (rb.c3, rb.down3, (3, 3, 3), (2, 2, 2), (1, 1, 1)),
                                         (rb.c4, rb.down4, (3, 3, 3), (2, 2, 2), (0, 1, 1)), (rb.c5, rb.down5, (3, 1, 1), (2, 1, 1), (0, 0, 0))):
        oc, cnt, nbo, nbi, osh = ops.rulebook_conv(cur, 1, cshape, ks, st, pd)
        n = int(cnt[0])
        assert int(cnt[1]) == 0 and n == want_c.shape[0]
        assert np.array_equal(oc[:n].numpy(), want_c) and np.array_equal(nbo[:n].numpy(), want_tbl)
        inv = np.full((cur.shape[0], want_tbl.shape[1]), -1, np.int32)
        o, k = np.nonzero(want_tbl >= 0)
        inv[want_tbl[o, k], k] = o
        assert np.array_equal(nbi.numpy(), inv)  # the transposed table SparseInverseConv3d uses
        cur, cshape = oc[:n].contiguous(), osh
        if ks == (3, 3, 3):
            want_subm = {tuple(rb.s2): rb.subm2, tuple(rb.s3): rb.subm3, tuple(rb.s4): rb.subm4}[tuple(osh)]
            assert np.array_equal(ops.rulebook_subm(cur, osh, (3, 3, 3)).numpy(), want_subm)
    # overflow is reported, not silently truncated
    oc, cnt, _, _, _ = ops.rulebook_conv(coords, 1, shape, (3, 3, 3), (2, 2, 2), (1, 1, 1), out_cap=100)
    assert int(cnt[0]) == 100 and int(cnt[1]) == 1


@pytest.mark.parametrize("m,k,n,nt,wc", [(70, 16, 17, 1, 1), (200, 48, 64, 2, 1), (150, 32, 96, 3, 1), (260, 64, 128, 4, 1),
                                         (129, 32, 192, 3, 1), (100, 32, 128, 2, 2), (77, 64, 128, 1, 4), (90, 16, 64, 1, 2),
                                         (300, 32, 128, 1, 2)])
def test_gather_gemm_dense_all_geometries(m, k, n, nt, wc, monkeypatch):
    """C = A B (asymmetric operands: catches transposed fragments), every (NT, WC) variant, both K-chunk sizes"""
    rng = np.random.default_rng(m)
    a, b = rng.normal(size=(m, k)).astype(np.float32), rng.normal(size=(k, n)).astype(np.float32)
    monkeypatch.setattr(ops, "choose_geometry", lambda cout, rows, target_blocks=None: (nt, wc))
    out = ops.gather_gemm(torch.from_numpy(a), PackedWeight(torch.from_numpy(b).reshape(1, k, n).contiguous(), 1, k, k, n), cout=n)
    np.testing.assert_allclose(out.numpy(), a.astype(np.float64) @ b.astype(np.float64), rtol=0, atol=1e-4)


def test_gather_gemm_workgroup_mapping_flags_do_not_change_results(monkeypatch):
    """ls3d_gather_gemm's mapping flags only permute which workgroup takes which (tile, slab): bitwise identical output (1300 rows = 11 tiles
    over 8 XCD lanes, 4 column slabs; f32 and split-bf16 kernels)"""
    rng = np.random.default_rng(11)
    m, k, n = 1300, 32, 128
    a, b = rng.normal(size=(m, k)).astype(np.float32), rng.normal(size=(k, n)).astype(np.float32)
    tbl = torch.arange(m, dtype=torch.int32).reshape(m, 1)
    monkeypatch.setattr(ops, "choose_geometry", lambda cout, rows, target_blocks=None: (1, 1))
    pw = PackedWeight(torch.from_numpy(b).reshape(1, k, n).contiguous(), 1, k, k, n)
    try:
        for prec in ("f32", "bf16x3"):
            ops.set_precision(prec)
            outs = []
            for flags in (0, 1, 2, 3):
                ops.set_gemm_flags(flags)
                outs.append(ops.gather_gemm(torch.from_numpy(a), pw, tbl=tbl, cout=n).numpy().copy())
            for o in outs[1:]:
                assert np.array_equal(o, outs[0])
            np.testing.assert_allclose(outs[0], a.astype(np.float64) @ b.astype(np.float64), rtol=0, atol=2e-3)
    finally:
        ops.set_precision("f32")
        ops.set_gemm_flags(0)


@pytest.mark.parametrize("cin,cout,wide", [(32, 32, True), (64, 64, True), (32, 128, True), (96, 128, False), (128, 16, True)])
@pytest.mark.parametrize("prec", ["f32", "bf16x3"])
def test_gather_gemm_sparse_tables_vs_float64(cin, cout, wide, prec, monkeypatch):
    """the table-driven gather-GEMM against a float64 reference: multi-chunk K, rows without neighbours, a tile with a single active
    offset, ragged last tile, natural and mask-sorted order, fused epilogue"""
    rng = np.random.default_rng(cin * 1000 + cout)
    vin, vout, kvol = 500, 333, 27
    x = rng.normal(size=(vin, cin)).astype(np.float32)
    w = (rng.normal(size=(kvol, cin, cout)) * 0.1).astype(np.float32)
    tbl = rng.integers(0, vin, size=(vout, kvol)).astype(np.int32)
    tbl[rng.uniform(size=tbl.shape) < 0.6] = -1
    tbl[7] = -1
    tbl[128:256, 1:] = -1   # a whole tile with one active offset (natural order)
    tbl[256:, 20:] = -1     # offsets no row of the last tiles uses
    scale, shift = rng.uniform(0.5, 1.5, cout).astype(np.float32), rng.normal(size=cout).astype(np.float32)
    res = rng.normal(size=(vout, cout)).astype(np.float32)
    acc = np.zeros((vout, cout), np.float64)
    for kk in range(kvol):
        o = np.nonzero(tbl[:, kk] >= 0)[0]
        acc[o] += x[tbl[o, kk]].astype(np.float64) @ w[kk].astype(np.float64)
    want = np.maximum(acc * scale + shift + res, 0)
    T = torch.from_numpy
    pw = PackedWeight(T(w), kvol, cin, cin, cout)
    if not wide:
        monkeypatch.setattr(ops, "choose_geometry", lambda c, rows, target_blocks=None: (1, 1))  # one 32-column slab per workgroup
    tol = 2e-4 if prec == "f32" else 2e-3
    try:
        ops.set_precision(prec)
        for order in (None, ops.rulebook_order(T(tbl))):
            out = ops.gather_gemm(T(x), pw, tbl=T(tbl), order=order, cout=cout, scale=T(scale), shift=T(shift), res_pre=T(res),
                                  relu=True).numpy()
            np.testing.assert_allclose(out, want, rtol=0, atol=tol)
    finally:
        ops.set_precision("f32")


def test_rulebook_orders_batched_sort():
    """one sort for several tables: every order is a permutation of its table's rows with non-increasing neighbour masks"""
    rng = np.random.default_rng(5)
    tbls = []
    for n, kvol in ((300, 27), (1, 27), (0, 27), (170, 3), (64, 27)):
        t = rng.integers(0, 50, size=(n, kvol)).astype(np.int32)
        t[rng.uniform(size=t.shape) < 0.5] = -1
        tbls.append(torch.from_numpy(t))
    orders = ops.rulebook_orders(tbls)
    assert orders[2] is None
    for t, o in zip(tbls, orders):
        if o is None:
            continue
        o = o.numpy()
        assert sorted(o.tolist()) == list(range(t.shape[0]))
        m = ((t.numpy() >= 0) * (1 << np.arange(t.shape[1]))).sum(1)
        assert np.all(np.diff(m[o]) <= 0)
        single = ops.rulebook_order(t).numpy()
        assert np.array_equal(m[single], m[o])


def test_transposed_strided_tables_are_ordered_by_coordinate_class():
    """ops.rulebook_parity_orders (ls3d_rulebook_parity_keys): the rows of a strided convolution's transposed table are grouped by the residue
    class of their input coordinate, densest class first, without reading the table - a row's offsets are the class's (a subset at the border of
    the output grid), so the grouping is as tight as the 27-bit mask order's; rows beyond the device count sort last; unsupported strides fall
    back; an inverse layer run in that order returns the same rows bit for bit"""
    rng = np.random.default_rng(17)
    shape = [9, 24, 20]
    for ksize, stride, pad in (((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 1, 1), (2, 1, 1), (0, 0, 0)), ((2, 2, 2), (2, 2, 2), (0, 0, 0)), ((3, 3, 3), (1, 2, 3), (1, 1, 1))):
        lin = rng.choice(2 * shape[0] * shape[1] * shape[2], size=900, replace=False)
        b, r = np.divmod(lin, shape[0] * shape[1] * shape[2])
        z, r2 = np.divmod(r, shape[1] * shape[2])
        y, x_ = np.divmod(r2, shape[2])
        coords = np.stack([b, z, y, x_], 1).astype(np.int32)
        coords = coords[np.lexsort((coords[:, 3], coords[:, 2], coords[:, 1], coords[:, 0]))]
        T = torch.from_numpy
        oc, cnt, nbr_out, nbr_inv, oshape = ops.rulebook_conv(T(coords), 2, shape, ksize, stride, pad)
        n_out = int(cnt[0])
        inv = nbr_inv.numpy()
        kvol = inv.shape[1]
        mask = ((inv >= 0) * (1 << np.arange(kvol))).sum(1)
        order = ops.rulebook_parity_orders([T(coords)], [(ksize, stride, pad)])[0].numpy()
        assert sorted(order.tolist()) == list(range(len(coords)))
        cls = tuple(((coords[:, 1 + d] + pad[d]) % stride[d]) for d in range(3))
        code = (cls[0] * stride[1] + cls[1]) * stride[2] + cls[2]
        full = {}
        for c in np.unique(code):
            full[c] = np.bitwise_or.reduce(mask[code == c])
            assert np.all((mask[code == c] | full[c]) == full[c])
            # interior rows carry the whole class mask: the class is the mask, up to the border
            assert (mask[code == c] == full[c]).mean() > 0.5
        seq = code[order]
        change = np.nonzero(np.diff(seq))[0]
        assert len(change) == len(np.unique(code)) - 1, "each class is one contiguous run"
        pops = [bin(int(full[c])).count("1") for c in seq[np.concatenate([[0], change + 1])]]
        assert pops == sorted(pops, reverse=True), "densest class first"
        # device count: rows beyond it are not part of the order's head
        nd = torch.tensor([600], dtype=torch.int32)
        part = ops.rulebook_parity_orders([T(coords)], [(ksize, stride, pad)], [nd])[0].numpy()
        assert sorted(part[:600].tolist()) == list(range(600))
        # an inverse layer in that order: the same output rows, bit for bit
        cin, cout = 16, 32
        w = (rng.normal(size=(kvol, cin, cout)) * 0.2).astype(np.float32)
        xin = rng.normal(size=(n_out, cin)).astype(np.float32)
        pw = PackedWeight(T(w), kvol, cin, cin, cout)
        a = ops.gather_gemm(T(xin), pw, tbl=nbr_inv, order=None, cout=cout)
        b_ = ops.gather_gemm(T(xin), pw, tbl=nbr_inv, order=T(order.astype(np.int32)), cout=cout)
        assert torch.equal(a, b_)
    # two tables in one batched sort, one of them with a stride the key does not cover
    got = ops.rulebook_parity_orders([T(coords), T(coords)], [((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (9, 1, 1), (1, 1, 1))])
    assert got[1] is None and sorted(got[0].numpy().tolist()) == list(range(len(coords)))


def test_gather_gemm_sparse_with_order_and_fused_epilogue(monkeypatch):
    rng = np.random.default_rng(3)
    vin, vout, kvol, cin, cout = 300, 170, 27, 32, 64
    x = rng.normal(size=(vin, cin)).astype(np.float32)
    w = rng.normal(size=(kvol, cin, cout)).astype(np.float32) * 0.1
    tbl = rng.integers(-1, vin, size=(vout, kvol)).astype(np.int32)
    tbl[rng.uniform(size=tbl.shape) < 0.7] = -1
    tbl[5] = -1  # a row without any neighbour
    scale, shift = rng.uniform(0.5, 1.5, cout).astype(np.float32), rng.normal(size=cout).astype(np.float32)
    res = rng.normal(size=(vout, cout)).astype(np.float32)
    pair = rng.normal(size=(vout, 2 * cout)).astype(np.float32)
    acc = np.zeros((vout, cout), np.float64)
    for kk in range(kvol):
        o = np.nonzero(tbl[:, kk] >= 0)[0]
        acc[o] += x[tbl[o, kk]].astype(np.float64) @ w[kk].astype(np.float64)
    want = np.maximum(acc * scale + shift + res, 0) + pair[:, 0::2] + pair[:, 1::2]
    T = torch.from_numpy
    pw = PackedWeight(T(w), kvol, cin, cin, cout)
    for nt, wc in ((1, 1), (2, 1), (1, 2)):
        monkeypatch.setattr(ops, "choose_geometry", lambda c, r, target_blocks=None, g=(nt, wc): g)
        for order in (None, ops.rulebook_order(T(tbl))):
            out = ops.gather_gemm(T(x), pw, tbl=T(tbl), order=order, cout=cout, scale=T(scale), shift=T(shift), res_pre=T(res),
                                  relu=True, pair=T(pair))
            np.testing.assert_allclose(out.numpy(), want, rtol=0, atol=2e-4)
    # writing into a column slice of a wider buffer (the UNet decoder's concat)
    wide = torch.zeros((vout, 2 * cout))
    ops.gather_gemm(T(x), pw, tbl=T(tbl), cout=cout, out=wide[:, cout:], out_ld=2 * cout)
    np.testing.assert_allclose(wide[:, cout:].numpy(), acc, rtol=0, atol=2e-4)
    assert float(wide[:, :cout].abs().max()) == 0.0


def test_vfe_readers_vs_reference():
    g = golden("vfe_nusc.npz")
    sel = slice(0, 96)
    vx, num = torch.from_numpy(g["voxels"][sel]), torch.from_numpy(g["num"][sel])
    np.testing.assert_allclose(readers.MeanVoxelFeatureExtractor(5)(vx, num).numpy(), g["mean"][sel], rtol=0, atol=1e-6)
    np.testing.assert_allclose(readers.ImprovedMeanVoxelFeatureExtractor(5)(vx, num).numpy(), g["improved"][sel], rtol=0, atol=1e-5)
    tv = readers.TransformerVoxelFeatureExtractor(5, 16, 64, 4, 3)
    tv.load_state_dict(seeded_sd("reader.TransformerVoxelFeatureExtractor", g["trans_seed"]), strict=True)
    fused = tv.eval()(vx, num).numpy()  # the one-kernel path (ls3d_transvfe)
    assert ops.transvfe(vx.contiguous(), num.to(torch.int32), tv.packed()["fused"]) is not None  # ... is really taken
    np.testing.assert_allclose(fused, g["trans"][sel], rtol=0, atol=1e-4)
    try:  # token deduplication (the default) vs every padding slot as a row of its own: the same function, f32 rounding of the softmax apart
        ops.set_transvfe_dedup(False)
        full = tv(vx, num).numpy()
        assert not np.array_equal(full, fused)
        np.testing.assert_allclose(full, g["trans"][sel], rtol=0, atol=1e-4)
        np.testing.assert_allclose(fused, full, rtol=0, atol=2e-5)
        # a voxel whose padding slots are NOT zero is not deduplicated: with garbage in the padding of every voxel the two paths agree bit for bit
        dirty = vx.clone()
        for i in range(dirty.shape[0]):
            dirty[i, int(num[i]):] = 0.25
        want = tv(dirty, num).numpy()
        ops.set_transvfe_dedup(True)
        assert np.array_equal(tv(dirty, num).numpy(), want)
        # classes of every size incl. full voxels and a ragged tail
        assert sorted(set(int(v) for v in num.tolist())) == [1, 2, 3, 4, 5] or len(set(num.tolist())) >= 3
    finally:
        ops.set_transvfe_dedup(True)
    try:  # experimental variant: weights straight from memory, no workgroup barriers - the same arithmetic in the same order
        ops.set_transvfe_direct(True)
        assert np.array_equal(tv(vx, num).numpy(), fused) and np.array_equal(tv(vx[:7], num[:7]).numpy(), fused[:7])
    finally:
        ops.set_transvfe_direct(False)
    try:  # the reader's GEMMs on the exact 3-plane bf16 split (what the 3-plane modes of ops.set_precision select): same golden
        for prec in ("bf16x6", "bf16x8"):
            ops.set_precision(prec)
            planes = tv(vx, num).numpy()
            assert not np.array_equal(planes, fused)  # another kernel variant ran ...
            np.testing.assert_allclose(planes, g["trans"][sel], rtol=0, atol=1e-4)
            np.testing.assert_allclose(planes, fused, rtol=0, atol=2e-5)  # ... with f32-grade results
            np.testing.assert_allclose(tv(vx[:7], num[:7]).numpy(), g["trans"][:7], rtol=0, atol=1e-4)
    finally:
        ops.set_precision("f32")
    try:  # and the layer-by-layer composition of the same module (configurations the fused kernel does not cover)
        readers._FUSED = False
        np.testing.assert_allclose(tv(vx, num).numpy(), g["trans"][sel], rtol=0, atol=1e-4)
    finally:
        readers._FUSED = True
    # ragged tail: 7 voxels = one full wave tile (6 voxels x 5 slots) + 1, workgroup partially filled
    np.testing.assert_allclose(tv(vx[:7], num[:7]).numpy(), g["trans"][:7], rtol=0, atol=1e-4)
    # no compression layer -> 64 features; 1 layer
    tv2 = readers.TransformerVoxelFeatureExtractor(5, 0, 64, 4, 1).eval()
    a = tv2(vx[:50], num[:50])
    readers._FUSED = False
    try:
        b = tv2(vx[:50], num[:50])
    finally:
        readers._FUSED = True
    assert a.shape == (50, 64)
    np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=0, atol=1e-4)
    # 4 point features (SemanticKITTI): 16-wide tokens are not a shape of the fused kernel -> the composed path, in every arithmetic
    tv3 = readers.TransformerVoxelFeatureExtractor(4, 16, 64, 4, 2).eval()
    c = tv3(vx[:20, :, :4].contiguous(), num[:20])
    try:
        ops.set_precision("bf16x6")
        d = tv3(vx[:20, :, :4].contiguous(), num[:20])
    finally:
        ops.set_precision("f32")
    assert c.shape == (20, 16) and torch.equal(c, d)


def test_unet_small_vs_oracle():
    cfg = synth.NUSC
    g = golden("unet_nusc_c13.npz")
    n = 100
    coords, feats = g["coords"][:n], torch.from_numpy(g["voxel_features"][:n])
    net = scn_unet.UNetSCN3D(num_input_features=13, voxel_size=cfg["voxel_size"], point_cloud_range=cfg["pc_range"],
                             model_cfg=dict(SCALING_RATIO=2), ds_factor=8, us_factor=8)
    sd = seeded_sd("backbone.UNetSCN3D.c13", g["seed"])
    net.load_state_dict(sd, strict=True)
    bd = net.eval()(dict(voxel_features=feats, voxel_coords=torch.from_numpy(coords), batch_size=1,
                         input_shape=np.asarray(orc.grid_size(cfg["voxel_size"], cfg["pc_range"]))))
    want, ctr, aux = orc.unet_scn3d(sd, feats, coords, orc.spatial_shape(cfg["voxel_size"], cfg["pc_range"]), cfg["voxel_size"],
                                    cfg["pc_range"], return_all=True)
    scale = float(want.abs().max())
    assert float((bd["conv_point_features"] - want).abs().max()) <= 1e-5 * scale + 1e-4
    assert torch.equal(bd["conv_point_coords"], ctr)
    assert float((bd["encoded_spconv_tensor"].features - aux["enc"]).abs().max()) <= 1e-5 * float(aux["enc"].abs().max()) + 1e-4


def test_three_nn_grid_equals_brute_force_and_oracle():
    g = golden("head_mseg3d_nusc.npz")
    pts = torch.from_numpy(g["points"][:, :4].copy())
    coords, ctr, feat = torch.from_numpy(g["coords"]), torch.from_numpy(g["conv_point_coords"]), torch.from_numpy(g["conv_point_features"])
    extra = torch.tensor([[0, 80.0, 3.0, 10.0], [0, -70.0, -60.0, -8.0], [1, 0.0, 0.0, 30.0], [1, 51.19, 51.19, 2.99], [1, -51.2, -51.2, -5.0]])
    pts = torch.cat([pts[pts[:, 0] == 0][:1500], extra[:2], pts[pts[:, 0] == 1][:1000], extra[2:]]).contiguous()
    cfg = synth.NUSC
    pt_off, vx_off = ops.frame_offsets(pts[:, 0], 2), ops.frame_offsets(ctr[:, 0], 2)
    a, ia = ops.devoxelize_grid(pts, pt_off, coords, ctr, vx_off, 2, cfg["voxel_size"], cfg["pc_range"], feat, return_idx=True)
    b, ib = ops.devoxelize(pts, pt_off, ctr, vx_off, 2, pts.shape[0], feat, return_idx=True)
    assert torch.equal(ia, ib) and torch.equal(a, b)
    want, widx = orc.three_interpolate_wrap(pts, ctr, feat, 2, return_idx=True)
    assert np.array_equal(ia.numpy(), np.concatenate(widx))
    np.testing.assert_allclose(a.numpy(), want.numpy(), rtol=0, atol=1e-5)
    # pointnet2-style API, incl. fewer than 3 known points
    d2, idx = ops.three_nn(pts[None, :200, 1:4].contiguous(), ctr[None, :2, 1:4].contiguous())
    wd2, wi = orc.three_nn(pts[:200, 1:4].numpy(), ctr[:2, 1:4].numpy())
    assert np.array_equal(idx[0].numpy(), wi) and np.array_equal(d2[0].numpy(), wd2)


def test_point_heads_small_vs_oracle():
    from lidarseg3d_amd import models_cfg
    g = golden("head_batchloss_nusc.npz")
    head = point_heads.PointSegBatchlossHead(False, 17, models_cfg.sdseg3d()["point_head"]["model_cfg"])
    sd = seeded_sd("point_head.PointSegBatchlossHead", g["seed"])
    head.load_state_dict(sd, strict=True)
    feat, ctr = torch.from_numpy(g["conv_point_features"][:500]), torch.from_numpy(g["conv_point_coords"][:500])
    pts = torch.from_numpy(g["points"][:400, :4]).contiguous()
    bd = head.eval()(dict(batch_size=1, conv_point_features=feat, conv_point_coords=ctr, points=pts), return_loss=False)
    cl, out = orc.batchloss_head(sd, feat, ctr, pts, 1)
    assert float((bd["out_logits"] - out).abs().max()) <= 1e-5 * float(out.abs().max()) + 1e-4

    g = golden("head_mseg3d_nusc.npz")
    head = point_heads.PointSegMSeg3DHead(False, 17, models_cfg.mseg3d()["point_head"]["model_cfg"])
    sd = seeded_sd("point_head.PointSegMSeg3DHead", g["seed"])
    head.load_state_dict(sd, strict=True)
    pts_all, cp_all = torch.from_numpy(g["points"]), torch.from_numpy(g["conv_point_coords"])
    h, w = (int(v) for v in g["cam_hw"])
    img, emb, cuv = synth.camera_inputs(pts_all.shape[0], seed=int(g["cam_seed"]), ncam=6, c_img=48, h=h, w=w, batch=2)
    pm = torch.cat([torch.nonzero(pts_all[:, 0] == b)[:150, 0] for b in range(2)])
    vm = torch.cat([torch.nonzero(cp_all[:, 0] == b)[:200, 0] for b in range(2)])
    pts, cuvs = pts_all[pm][:, :4].contiguous(), torch.from_numpy(cuv)[pm].contiguous()
    vf, ctr = torch.from_numpy(g["conv_point_features"])[vm].contiguous(), cp_all[vm].contiguous()
    bd = head.eval()(dict(batch_size=2, conv_point_features=vf, conv_point_coords=ctr, points=pts, image_features=torch.from_numpy(img),
                          points_cuv=cuvs, camera_semantic_embeddings=torch.from_numpy(emb)), return_loss=False)
    vl, out = orc.mseg3d_head(sd, vf, ctr, pts, cuvs, torch.from_numpy(img), torch.from_numpy(emb), 2)
    assert float((head.forward_ret_dict["voxel_logits"] - vl).abs().max()) <= 1e-4
    assert float((bd["out_logits"] - out).abs().max()) <= 1e-4


@pytest.mark.parametrize("m,k,n,nt", [(150, 32, 64, 2), (140, 64, 128, 4), (70, 96, 96, 3), (200, 32, 17, 1)])
def test_gather_gemm_bf16x3_close_to_f32(m, k, n, nt, monkeypatch):
    """split-bf16 path: within ~2e-5 of the f64 product (and much closer than plain bf16 would be)"""
    rng = np.random.default_rng(m)
    a, b = rng.normal(size=(m, k)).astype(np.float32), rng.normal(size=(k, n)).astype(np.float32)
    monkeypatch.setattr(ops, "choose_geometry", lambda cout, rows, target_blocks=None: (nt, 1))
    pw = PackedWeight(torch.from_numpy(b).reshape(1, k, n).contiguous(), 1, k, k, n)
    ops.set_precision("bf16x3")
    try:
        ident = torch.arange(m, dtype=torch.int32).unsqueeze(1).contiguous()  # identity rulebook (split-bf16 = sparse convs only)
        out = ops.gather_gemm(torch.from_numpy(a), pw, tbl=ident, cout=n)
    finally:
        ops.set_precision("f32")
    want = a.astype(np.float64) @ b.astype(np.float64)
    err = np.abs(out.numpy() - want).max()
    assert err <= 3e-5 * np.abs(a).max() * np.abs(b).max() * np.sqrt(k), err
    bf = lambda x: torch.from_numpy(x).to(torch.bfloat16).float().numpy()
    plain = np.abs(bf(a).astype(np.float64) @ bf(b).astype(np.float64) - want).max()
    assert err < plain / 20


@pytest.mark.parametrize("m,k,n,nt", [(150, 32, 64, 2), (140, 64, 128, 4), (70, 96, 96, 3), (200, 32, 17, 1)])
def test_gather_gemm_bf16x6_is_f32_grade(m, k, n, nt, monkeypatch):
    """3-plane split-bf16 (6 partial products): error vs the float64 product at the level of the exact-f32 MFMA path's own
    rounding (operands with a wide dynamic range), and the sparse path with skipped offsets"""
    rng = np.random.default_rng(m)
    a = (rng.normal(size=(m, k)) * np.exp(rng.normal(size=(m, k)) * 2)).astype(np.float32)
    b = (rng.normal(size=(k, n)) * np.exp(rng.normal(size=(k, n)) * 2)).astype(np.float32)
    monkeypatch.setattr(ops, "choose_geometry", lambda cout, rows, target_blocks=None: (nt, 1))
    pw = PackedWeight(torch.from_numpy(b).reshape(1, k, n).contiguous(), 1, k, k, n)
    ident = torch.arange(m, dtype=torch.int32).unsqueeze(1).contiguous()
    want = a.astype(np.float64) @ b.astype(np.float64)
    mag = np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64)  # the scale rounding errors live on
    errs = {}
    try:
        ops.set_tile(False)  # with the tile path off "bf16x6" is the 6-product gather-GEMM for every sparse layer
        for prec in ("f32", "bf16x6"):
            ops.set_precision(prec)
            out = ops.gather_gemm(torch.from_numpy(a), pw, tbl=ident, cout=n)
            errs[prec] = float((np.abs(out.numpy() - want) / mag).max())
    finally:
        ops.set_precision("f32")
        ops.set_tile(True)
    assert errs["bf16x6"] <= 4 * 2.0 ** -24 * np.sqrt(k) + 3 * 2.0 ** -24, errs  # f32-grade: a few ulp of the |a||b| scale
    assert errs["bf16x6"] <= 8 * errs["f32"] + 2.0 ** -22, errs


def test_sparse_conv_bf16x3_vs_f32(monkeypatch):
    rng = np.random.default_rng(4)
    vin, vout, kvol, cin, cout = 200, 150, 27, 64, 64
    x = rng.normal(size=(vin, cin)).astype(np.float32)
    w = rng.normal(size=(kvol, cin, cout)).astype(np.float32) * 0.1
    tbl = rng.integers(-1, vin, size=(vout, kvol)).astype(np.int32)
    tbl[rng.uniform(size=tbl.shape) < 0.6] = -1
    T = torch.from_numpy
    pw = PackedWeight(T(w), kvol, cin, cin, cout)
    ref = ops.gather_gemm(T(x), pw, tbl=T(tbl), cout=cout, relu=True)
    ops.set_precision("bf16x3")
    try:
        got = ops.gather_gemm(T(x), pw, tbl=T(tbl), order=ops.rulebook_order(T(tbl)), cout=cout, relu=True)
    finally:
        ops.set_precision("f32")
    assert float((got - ref).abs().max()) <= 3e-5 * float(ref.abs().max()) + 1e-6


def _spconv_ref(feats, w, tbl):
    """differentiable torch restatement of out[o] = sum_k W[k]^T in[tbl[o][k]]"""
    kvol = tbl.shape[1]
    w = w.reshape(kvol, w.shape[-2], w.shape[-1])
    out = torch.zeros((tbl.shape[0], w.shape[-1]), dtype=feats.dtype)
    for k in range(kvol):
        o = torch.nonzero(tbl[:, k] >= 0)[:, 0]
        if o.numel():
            out = out.index_add(0, o, feats[tbl[o, k].long()] @ w[k])
    return out


@pytest.mark.parametrize("cin,cout,products", [(64, 128, 6), (128, 64, 8), (32, 32, 6), (96, 48, 6), (16, 16, 8), (160, 64, 6), (32, 160, 6), (16, 288, 6)])
def test_wgrad_on_bf16_planes_is_f32_grade(cin, cout, products):
    """ls3d_spconv_wgrad with the exact 3-plane split (16 rows per bf16 MFMA, head x head in its own accumulator) against float64:
    not worse than the exact-f32 kernel (up to the emulation's per-product rounding), ragged row counts, absent neighbours"""
    rng = np.random.default_rng(cin * 7 + cout)
    n_in, n_out, kvol = 211, 173, 5
    x = (rng.normal(size=(n_in, cin)) * np.exp(rng.normal(size=(n_in, cin)))).astype(np.float32)
    go = (rng.normal(size=(n_out, cout)) * np.exp(rng.normal(size=(n_out, cout)))).astype(np.float32)
    tbl = rng.integers(0, n_in, size=(n_out, kvol)).astype(np.int32)
    tbl[rng.uniform(size=tbl.shape) < 0.4] = -1
    tbl[40:80, 3] = -1
    want = np.zeros((kvol, cin, cout))
    for k in range(kvol):
        o = np.nonzero(tbl[:, k] >= 0)[0]
        want[k] = x[tbl[o, k]].astype(np.float64).T @ go[o].astype(np.float64)
    mag = np.zeros((kvol, cin, cout))
    for k in range(kvol):
        o = np.nonzero(tbl[:, k] >= 0)[0]
        mag[k] = np.abs(x[tbl[o, k]]).astype(np.float64).T @ np.abs(go[o]).astype(np.float64)
    tx, tg, tt = torch.from_numpy(x), torch.from_numpy(go), torch.from_numpy(tbl)
    order = torch.from_numpy(rng.permutation(n_out).astype(np.int32))
    err = {}
    for name, pr, od in (("f32", 0, None), ("planes", products, None), ("planes_ordered", products, order)):
        got = ops.spconv_wgrad(tx, tg, tt, od, cin, cout, products=pr).numpy()
        err[name] = float((np.abs(got - want) / np.maximum(mag, 1e-30)).max())
        # the two-step form (pair lists built once, shared by the layers of a table) is the same computation
        shared = ops.spconv_wgrad(tx, tg, tt, od, cin, cout, products=pr, pairs=ops.spconv_pairs(tt, od)).numpy()
        assert np.array_equal(shared, got), name
    assert err["planes"] <= 4 * err["f32"] + 2.0 ** -22 and err["planes_ordered"] <= 4 * err["f32"] + 2.0 ** -22, err


@pytest.mark.parametrize("cin,cout", [(64, 32), (32, 160), (160, 32)])
def test_subm_training_path_on_the_tile_kernel(cin, cout):
    """training forward and dgrad of a SubM layer in the 3-plane mode (tile-halo kernel, same plan for both) == the exact-f32
    gather-GEMM path to f32 rounding; also with more than 128 channels on either side (slabs of 128 columns in the forward, the
    dgrad and the weight gradient: SCALING_RATIO > 2 of the reference's UNet)"""
    from lidarseg3d_amd import spconv
    rng = np.random.default_rng(3)
    shape = [9, 24, 24]
    cells = rng.choice(shape[0] * shape[1] * shape[2], size=300, replace=False)
    coords = torch.from_numpy(np.stack([np.zeros_like(cells), cells // (24 * 24), (cells // 24) % 24, cells % 24], 1).astype(np.int32))
    conv = spconv.SubMConv3d(cin, cout, 3, padding=1, bias=True, indice_key="s").train()
    feats0 = torch.from_numpy(rng.normal(size=(300, cin)).astype(np.float32))
    res = {}
    try:
        for prec in ("f32", "bf16x6"):
            ops.set_precision(prec)
            f = feats0.clone().requires_grad_(True)
            conv.weight.grad = None
            y = conv(spconv.SparseConvTensor(f, coords, shape, 1)).features
            (y * torch.from_numpy(rng.normal(size=tuple(y.shape)).astype(np.float32) * 0 + 1.0) * y).sum().backward()
            res[prec] = (y.detach().clone(), f.grad.clone(), conv.weight.grad.clone())
    finally:
        ops.set_precision("f32")
    for a, b in zip(res["f32"], res["bf16x6"]):
        assert not torch.equal(a, b) or a is res["f32"][2]  # a different kernel ran (the weight gradient kernel is shared)
        assert float((a - b).abs().max()) <= 2e-5 * float(a.abs().max())


@pytest.mark.parametrize("n,cin,cout", [(300, 96, 192), (257, 64, 23), (1000, 32, 130)])
def test_linear_weight_gradient_on_the_wgrad_kernel(n, cin, cout):
    """ops.linear_wgrad / ops._LinearFn: the tall-skinny weight gradient of nn.Linear (gy^T x over many rows) on ls3d_spconv_wgrad with the
    identity table, output columns beyond 128 in slices; everything against torch autograd of F.linear"""
    rng = np.random.default_rng(n + cin + cout)
    x = torch.from_numpy(rng.normal(size=(n, cin)).astype(np.float32)).requires_grad_(True)
    w = torch.from_numpy((rng.normal(size=(cout, cin)) * 0.1).astype(np.float32)).requires_grad_(True)
    b = torch.from_numpy(rng.normal(size=(cout,)).astype(np.float32)).requires_grad_(True)
    r = torch.from_numpy(rng.normal(size=(n, cout)).astype(np.float32))
    (torch.nn.functional.linear(x, w, b) * r).sum().backward()
    want = [t.grad.clone() for t in (x, w, b)]
    for t in (x, w, b):
        t.grad = None
    y = ops._LinearFn.apply(x, w, b)
    ref = torch.nn.functional.linear(x, w, b)  # (round 5: eligible shapes run the forward on the dense gather-GEMM too - another summation order)
    assert float((y - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    (y * r).sum().backward()
    for g, t in zip(want, (x, w, b)):
        assert float((t.grad - g).abs().max()) <= 1e-5 * float(g.abs().max()) + 1e-6
    np.testing.assert_allclose(ops.linear_wgrad(x.detach(), r).numpy(), (r.double().t() @ x.detach().double()).numpy(), rtol=0, atol=1e-3)


@pytest.mark.parametrize("cin,cout", [(16, 32), (64, 64), (32, 128), (64, 128), (128, 64), (96, 32)])
def test_sparse_conv_backward_vs_autograd(cin, cout):
    """SubMConv3d -> SparseConv3d(stride 2) -> SparseInverseConv3d in training mode: grad of the input features and of the
    three weights (dgrad = gather-GEMM on the transposed tables, wgrad = ls3d_spconv_wgrad) vs torch autograd on a plain
    restatement of the same sums"""
    from lidarseg3d_amd import spconv
    rng = np.random.default_rng(cin + cout)
    shape = [9, 24, 24]
    cells = rng.choice(shape[0] * shape[1] * shape[2], size=420, replace=False)
    coords = np.stack([np.zeros_like(cells), cells // (24 * 24), (cells // 24) % 24, cells % 24], 1).astype(np.int32)
    coords = coords[np.lexsort((coords[:, 3], coords[:, 2], coords[:, 1]))]
    feats = torch.from_numpy(rng.normal(size=(len(coords), cin)).astype(np.float32)).requires_grad_(True)
    torch.manual_seed(0)
    c1 = spconv.SubMConv3d(cin, cout, 3, padding=1, bias=False, indice_key="s1").train()
    c2 = spconv.SparseConv3d(cout, cout, 3, stride=2, padding=1, bias=True, indice_key="d1").train()
    c3 = spconv.SparseInverseConv3d(cout, cin, 3, indice_key="d1", bias=False).train()
    x = spconv.SparseConvTensor(feats, torch.from_numpy(coords), shape, 1)
    y1 = c1(x); y2 = c2(y1); y3 = c3(y2)
    r = torch.from_numpy(rng.normal(size=tuple(y3.features.shape)).astype(np.float32))
    (y3.features * r).sum().backward()
    got = [feats.grad.clone(), c1.weight.grad.clone(), c2.weight.grad.clone(), c2.bias.grad.clone(), c3.weight.grad.clone()]
    # reference: same tables, torch autograd
    rb1, rb2 = x.find_indice_pair("s1"), x.find_indice_pair("d1")
    f2 = feats.detach().clone().requires_grad_(True)
    w1, w2, b2, w3 = (t.detach().clone().requires_grad_(True) for t in (c1.weight, c2.weight, c2.bias, c3.weight))
    z1 = _spconv_ref(f2, w1, rb1.tbl)
    z2 = _spconv_ref(z1, w2, rb2.tbl) + b2
    z3 = _spconv_ref(z2, w3, rb2.tbl_inv)
    np.testing.assert_allclose(y3.features.detach().numpy(), z3.detach().numpy(), rtol=0, atol=2e-4)
    (z3 * r).sum().backward()
    for g, w in zip(got, (f2.grad, w1.grad, w2.grad, b2.grad, w3.grad)):
        np.testing.assert_allclose(g.numpy(), w.numpy(), rtol=0, atol=2e-4 * max(1.0, float(w.abs().max())))


def test_unet_training_forward_backward_vs_autograd(monkeypatch):
    """UNetSCN3D in train() mode (batch-statistics BatchNorm, the reference's unfused composition): loss gradient w.r.t. the
    input features and the 36 convolution weights on the path to the output, HIP dgrad/wgrad vs the same graph with every sparse convolution replaced
    by the plain torch restatement"""
    from lidarseg3d_amd import spconv
    cfg = synth.NUSC
    g = golden("unet_nusc_c13.npz")
    n = 160
    coords = torch.from_numpy(g["coords"][:n])
    feats0 = torch.from_numpy(g["voxel_features"][:n])
    net = scn_unet.UNetSCN3D(num_input_features=13, voxel_size=cfg["voxel_size"], point_cloud_range=cfg["pc_range"],
                             model_cfg=dict(SCALING_RATIO=1), ds_factor=8, us_factor=8)
    torch.manual_seed(1)
    net.train()
    shape = np.asarray(orc.grid_size(cfg["voxel_size"], cfg["pc_range"]))

    def run():
        for p in net.parameters():
            p.grad = None
        for m in net.modules():  # same running-stat updates in both runs do not matter; same batch statistics do
            if isinstance(m, torch.nn.BatchNorm1d):
                m.reset_running_stats()
        f = feats0.clone().requires_grad_(True)
        bd = net(dict(voxel_features=f, voxel_coords=coords, batch_size=1, input_shape=shape))
        out = bd["conv_point_features"]
        w = torch.linspace(-1, 1, out.numel()).reshape(out.shape)
        (out * w).sum().backward()
        grads = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
        return out.detach().clone(), f.grad.clone(), grads

    out_a, gin_a, gw_a = run()

    class RefFn(object):
        @staticmethod
        def apply(feats, weight, bias, rb, inverse, subm):
            y = _spconv_ref(feats, weight, (rb.tbl_inv if inverse else rb.tbl))
            return y if bias is None else y + bias
    monkeypatch.setattr(spconv, "_SparseConvFn", RefFn)
    out_b, gin_b, gw_b = run()
    np.testing.assert_allclose(out_a.numpy(), out_b.numpy(), rtol=0, atol=1e-4 * max(1.0, float(out_b.abs().max())))
    np.testing.assert_allclose(gin_a.numpy(), gin_b.numpy(), rtol=0, atol=2e-4 * max(1.0, float(gin_b.abs().max())))
    convs = [k for k in gw_b if k.endswith("weight") and gw_b[k].dim() == 5]
    assert len(convs) == 36 and set(gw_a) == set(gw_b)  # conv_out feeds nothing the loss sees
    # a pre-activation closer to zero than the f32 difference of the two forward evaluations flips its ReLU gate between the runs:
    # rare, confined to single channels, and worth ~1e-2 of the largest entry of the layers behind it (measured in
    # test_host_logic's two-rank run) -- so: per parameter, relative to its largest entry: median deviation <= 3e-4, rms <= 1e-3, at most
    # 3 % of the entries (one flipped channel of a [27, cin, cout] weight; or one entry) beyond 1e-3, none beyond 5e-2
    for k in gw_b:
        a, b = gw_a[k].numpy().astype(np.float64), gw_b[k].numpy().astype(np.float64)
        scale = max(1.0, float(np.abs(b).max()))
        d = np.abs(a - b) / scale
        assert (np.median(d) <= 3e-4 and np.sqrt(np.mean(d ** 2)) <= 1e-3 and float(d.max()) <= 5e-2
                and np.sum(d > 1e-3) <= max(1, 0.03 * d.size)), (k, np.median(d), d.max(), np.sum(d > 1e-3))


def test_fcn_mseg3d_head_1x1_convs_vs_torch_composition():
    """fcn_mseg3d_head.py:54-200 at the shipped shape (resize_concat of 4 levels -> 2 x [1x1 conv + BN + ReLU] -> 1x1 classifier -> camera
    SFAM) on the HIP GEMM / SFAM kernels == the torch composition of the same modules; checkpoint key layout of the mmcv ConvModules"""
    from tests.fcn_head_cases import fcn_head_case, fcn_head_check
    head, inputs = fcn_head_case("cpu", ncam=3, h=8, w=12, batch=2)
    keys = set(head.state_dict())
    assert {"convs.0.conv.weight", "convs.0.bn.weight", "convs.0.bn.bias", "convs.0.bn.running_mean", "convs.0.bn.running_var",
            "convs.1.conv.weight", "convs.1.bn.running_var", "conv_seg.weight", "conv_seg.bias"} <= keys
    assert tuple(head.convs[0].conv.weight.shape) == (48, 270, 1, 1) and tuple(head.conv_seg.weight.shape) == (17, 48, 1, 1)
    fcn_head_check(head, inputs, 2)
    # the loss: weighted cross-entropy on the logits resized to the label maps
    labels = torch.randint(0, 17, (6, 1, 16, 24))
    head(dict(inputs=inputs, batch_size=2, images_sem_labels=labels), return_loss=True)
    loss, parts = head.get_loss()
    want = 0.5 * torch.nn.functional.cross_entropy(torch.nn.functional.interpolate(head.forward_ret_dict["image_logits"], size=(16, 24), mode="bilinear",
                                                                                   align_corners=False), labels.squeeze(1), ignore_index=0)
    assert abs(float(loss) - float(want)) <= 1e-6 and set(parts) == {"image_ce_loss"}


def test_camera_sfam_vs_reference():
    from lidarseg3d_amd import img_heads
    g = golden("camera_sfam.npz")
    got = img_heads.CameraSemanticFeatureAggregationModule()(torch.from_numpy(g["feats"]), torch.from_numpy(g["probs"]), int(g["batch_size"]))
    assert tuple(got.shape) == tuple(g["emb"].shape)
    np.testing.assert_allclose(got.numpy(), g["emb"], rtol=0, atol=2e-5)


def test_points_cp_and_cuv_vs_oracle():
    """GPU-side camera projection of the points (loading.py:384-413) and grid_sample normalisation (segpreprocess.py:649-671)"""
    cfg = synth.NUSC
    pts = synth.lidar_frame(6000, seed=9, **cfg)
    r2g, c2g, K = synth.camera_rig(6, seed=2)
    want = orc.points_cp(pts, r2g, c2g, K)
    got = ops.points_cp(torch.from_numpy(pts), r2g, c2g, K).numpy()
    same_cam = got[:, 0] == want[:, 0]
    assert same_cam.mean() >= 0.9995  # a pixel within 1e-9 of the 1-pixel margin may flip (different dot-product order)
    assert (want[:, 0] > 0).mean() > 0.3  # the rig sees a good part of the sweep
    np.testing.assert_allclose(got[same_cam], want[same_cam], rtol=0, atol=2e-4)
    cuv_want = orc.points_cuv(want, 6, (640, 960))
    want = np.ascontiguousarray(want)  # the oracle returns a fancy-indexed view
    cuv_got = ops.points_cuv(torch.from_numpy(want), 6, (640, 960)).numpy()
    np.testing.assert_array_equal(cuv_got, cuv_want)
    one = ops.points_cuv(torch.from_numpy(want), 1, (640, 960)).numpy()
    np.testing.assert_array_equal(one, orc.points_cuv(want, 1, (640, 960)))


def test_devoxelize_split_search_then_interpolate_equals_fused():
    g = golden("head_mseg3d_nusc.npz")
    pts = torch.from_numpy(g["points"][:, :4].copy())
    coords, ctr, feat = torch.from_numpy(g["coords"]), torch.from_numpy(g["conv_point_coords"]), torch.from_numpy(g["conv_point_features"])
    extra = torch.tensor([[0, 80.0, 3.0, 10.0], [1, 0.0, 0.0, 30.0]])
    pts = torch.cat([pts[pts[:, 0] == 0][:900], extra[:1], pts[pts[:, 0] == 1][:700], extra[1:]]).contiguous()
    cfg = synth.NUSC
    pt_off, vx_off = ops.frame_offsets(pts[:, 0], 2), ops.frame_offsets(ctr[:, 0], 2)
    fused, fi = ops.devoxelize_grid(pts, pt_off, coords, ctr, vx_off, 2, cfg["voxel_size"], cfg["pc_range"], feat, return_idx=True)
    idx, w = ops.devoxelize_grid(pts, pt_off, coords, ctr, vx_off, 2, cfg["voxel_size"], cfg["pc_range"], None)
    assert torch.equal(idx, fi)
    assert torch.equal(ops.interpolate_rows(feat, idx, w, pts, vx_off), fused)
    np.testing.assert_allclose(w.sum(1).numpy(), 1.0, atol=1e-6)


def _train_example(points_per_frame, device="cpu"):
    """drop-in style training example: voxels from the (GPU / simulated) voxelizer + random labels"""
    cfg = synth.NUSC
    frames = [synth.lidar_frame(n, seed=11 + i, **cfg) for i, n in enumerate(points_per_frame)]
    pts = torch.from_numpy(np.concatenate([np.concatenate([np.full((f.shape[0], 1), b, np.float32), f], 1) for b, f in enumerate(frames)])).to(device)
    v, c, n, nv = ops.voxelize_hard(pts, cfg["voxel_size"], cfg["pc_range"], 5, 60000 * len(frames), batched=True)
    V = int(nv)
    gen = torch.Generator().manual_seed(3)
    ex = dict(points=pts, voxels=v[:V], coordinates=c[:V], num_points=n[:V], num_voxels=[0] * len(frames),
              shape=[np.asarray(orc.grid_size(cfg["voxel_size"], cfg["pc_range"]))],
              voxel_sem_labels=torch.randint(0, 17, (V,), generator=gen).to(device),
              point_sem_labels=torch.randint(0, 17, (pts.shape[0],), generator=gen).to(device))
    return ex


def test_sdseg3d_training_step_runs(monkeypatch):
    """SegNet(return_loss=True) in train mode on a tiny batch: TransVFE (torch autograd) -> UNetSCN3D (HIP forward / dgrad / wgrad)
    -> batch-loss head (HIP 3-NN search + differentiable gather) -> CE + Lovasz.  The simulator is slow, so this only checks that
    the step is wired (finite loss, a finite gradient for every parameter on the path); the comparison against the torch
    restatement of the convolutions is test_unet_training_forward_backward_vs_autograd here and
    test_sdseg3d_training_step_gpu on the device."""
    import lidarseg3d_amd as L
    from lidarseg3d_amd import models_cfg
    torch.manual_seed(0)
    cfgm = models_cfg.sdseg3d()
    cfgm["backbone"]["model_cfg"] = dict(cfgm["backbone"].get("model_cfg", {}), SCALING_RATIO=1)  # 16..64 channels: 4x less emulated MFMA work
    cfgm["point_head"]["model_cfg"] = dict(cfgm["point_head"]["model_cfg"], CONV_IN_DIM=16)
    model = L.build_detector(cfgm, train_cfg=None, test_cfg={}).train()
    ex = _train_example([90, 50])
    out = model(dict(ex), return_loss=True)
    loss = out["loss"][0]
    loss.backward()
    assert np.isfinite(float(loss.detach())) and set(out) == {"loss", "conv_ce_loss", "conv_lovasz_loss", "out_ce_loss", "out_lovasz_loss"}
    for k, p in model.named_parameters():
        if k.startswith("backbone.conv_out"):
            assert p.grad is None, k  # feeds nothing the loss sees
        else:
            assert p.grad is not None and bool(torch.isfinite(p.grad).all()), k


def test_mseg3d_head_training_path_vs_reference_golden():
    """PointSegMSeg3DHead's autograd path (torch modules + HIP neighbour search) in eval mode reproduces the REFERENCE head's
    logits of the golden fixture (eval BatchNorm is row-wise, so the valid-subset branches agree with the inference path);
    then the loss (point_seg_mseg3d_head.py:137-196) back-propagates into every parameter and into the camera maps."""
    from lidarseg3d_amd import models_cfg
    g = golden("head_mseg3d_nusc.npz")
    head = point_heads.PointSegMSeg3DHead(False, 17, models_cfg.mseg3d()["point_head"]["model_cfg"])
    head.load_state_dict(seeded_sd("point_head.PointSegMSeg3DHead", g["seed"]), strict=True)
    pts = torch.from_numpy(g["points"][:, :4].copy())
    h, w = (int(v) for v in g["cam_hw"])
    img, emb, cuv = synth.camera_inputs(pts.shape[0], seed=int(g["cam_seed"]), ncam=6, c_img=48, h=h, w=w, batch=2)
    cfg = synth.NUSC
    gen = torch.Generator().manual_seed(1)
    vf = torch.from_numpy(g["conv_point_features"]).requires_grad_(True)
    imgt = torch.from_numpy(img).requires_grad_(True)
    bd = dict(batch_size=2, conv_point_features=vf, conv_point_coords=torch.from_numpy(g["conv_point_coords"]),
              conv_point_indices=torch.from_numpy(g["coords"]), voxel_geometry=(cfg["voxel_size"], cfg["pc_range"]),
              points=pts, image_features=imgt, points_cuv=torch.from_numpy(cuv), camera_semantic_embeddings=torch.from_numpy(emb),
              voxel_sem_labels=torch.randint(0, 17, (vf.shape[0],), generator=gen),
              point_sem_labels=torch.randint(0, 17, (pts.shape[0],), generator=gen))
    head.eval()(bd, return_loss=True)
    r = head.forward_ret_dict
    np.testing.assert_allclose(r["voxel_logits"].detach().numpy(), g["voxel_logits"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(r["out_logits"].detach().numpy(), g["out_logits"], rtol=0, atol=1e-4)
    n_valid = int((cuv[:, 0] == 1).sum())
    assert 0 < n_valid < pts.shape[0] and r["point_features_pcamera"].shape == r["point_features_camera"].shape == (n_valid, 64)
    loss, parts = head.get_loss()
    assert set(parts) == {"voxel_ce_loss", "voxel_lovasz_loss", "out_ce_loss", "out_lovasz_loss", "out_mimic_loss"}
    want = sum(float(v) for v in parts.values())
    assert abs(float(loss) - want) < 1e-5 * want
    loss.backward()
    for k, p in head.named_parameters():
        assert p.grad is not None and bool(torch.isfinite(p.grad).all()) and float(p.grad.abs().max()) > 0, k
    assert float(vf.grad.abs().max()) > 0 and float(imgt.grad.abs().max()) > 0
    # train mode (batch statistics, the camera branches normalised over the valid rows only) still runs
    head.train()(bd, return_loss=True)
    assert bool(torch.isfinite(head.get_loss()[0]))


def test_mseg3d_training_step_runs():
    """SegMSeg3DNet(return_loss=True) in train mode on a tiny two-frame batch: the wiring of the whole step (reader ->
    UNetSCN3D HIP forward / dgrad / wgrad -> GF-/SF-Phase head under autograd -> losses)"""
    import lidarseg3d_amd as L
    from lidarseg3d_amd import models_cfg
    torch.manual_seed(0)
    cfgm = models_cfg.mseg3d()
    cfgm["backbone"]["model_cfg"] = dict(cfgm["backbone"].get("model_cfg", {}), SCALING_RATIO=1)
    cfgm["point_head"]["model_cfg"] = dict(cfgm["point_head"]["model_cfg"], VOXEL_IN_DIM=16)
    model = L.build_detector(cfgm, train_cfg=None, test_cfg={}).train()
    ex = _train_example([80, 40])
    img, emb, cuv = synth.camera_inputs(ex["points"].shape[0], seed=2, ncam=6, c_img=48, h=8, w=12, batch=2)
    ex.update(image_features=torch.from_numpy(img), camera_semantic_embeddings=torch.from_numpy(emb), points_cuv=torch.from_numpy(cuv))
    out = model(dict(ex), return_loss=True)
    loss = out["loss"][0]
    loss.backward()
    assert np.isfinite(float(loss.detach()))
    assert set(out) == {"loss", "voxel_ce_loss", "voxel_lovasz_loss", "out_ce_loss", "out_lovasz_loss", "out_mimic_loss"}
    for k, p in model.named_parameters():
        if k.startswith("backbone.conv_out"):
            assert p.grad is None, k
        else:
            assert p.grad is not None and bool(torch.isfinite(p.grad).all()), k


# ------------------------------------------------------------------------------------------------ tile-halo convolution
def _plan_views(plan):
    """numpy views of an ops.TilePlan buffer (layout of csrc/tileconv.hip:tc_plan)"""
    al = lambda v: (v + 255) // 256 * 256
    t, kvol = (plan.n_rows + 127) // 128, plan.kvol
    raw = plan.buf.numpy()
    o = 0
    trow = raw[o:o + t * 128 * 4].view(np.int32).reshape(t, 128); o += al(t * 128 * 4)
    meta = raw[o:o + t * 8 * 4].view(np.int32).reshape(t, 8); o += al(t * 8 * 4)
    halo = raw[o:o + t * kvol * 128 * 4].view(np.int32).reshape(t, kvol * 128); o += al(t * kvol * 128 * 4)
    loc = raw[o:o + t * kvol * 128 * 2].view(np.uint16).reshape(t, kvol, 128)
    return trow, meta, halo, loc


def _plan_torder(plan):
    al = lambda v: (v + 255) // 256 * 256
    t, kvol = (plan.n_rows + 127) // 128, plan.kvol
    o = al(t * 128 * 4) + al(t * 8 * 4) + al(t * kvol * 128 * 4) + al(t * kvol * 128 * 2)
    return plan.buf.numpy()[o:o + t * 4].view(np.int32).copy()


def _sparse_ref(x, w, tbl):
    acc = np.zeros((tbl.shape[0], w.shape[2]), np.float64)
    for k in range(tbl.shape[1]):
        o = np.nonzero(tbl[:, k] >= 0)[0]
        acc[o] += x[tbl[o, k]].astype(np.float64) @ w[k].astype(np.float64)
    return acc


def test_tile_plan_structure_on_a_subm_rulebook():
    """the plan of a SubM table: tiles partition the rows, slots are mask-sorted, halos are the sorted unique neighbours,
    tloc points at the right halo entry, masks are the ORs they claim to be"""
    cfg = synth.NUSC
    pts = synth.lidar_frame(4000, seed=11, **cfg)
    v, c, n, nv = ops.voxelize_hard(torch.from_numpy(pts), cfg["voxel_size"], cfg["pc_range"], 5, 20000)
    V = int(nv)
    coords = torch.cat([torch.zeros((V, 1), dtype=torch.int32), c[:V]], 1).contiguous()
    shape = orc.spatial_shape(cfg["voxel_size"], cfg["pc_range"])
    tbl = ops.rulebook_subm(coords, shape, (3, 3, 3))
    plan = ops.tile_plan(tbl, coords, shape, 1)
    trow, meta, halo, loc = _plan_views(plan)
    tb = tbl.numpy()
    assert sorted(trow[trow >= 0].tolist()) == list(range(V))
    keys = ops.tile_keys(coords, shape, 1).numpy()
    order = np.argsort(keys, kind="stable")  # tiles are consecutive runs of 128 rows of the stable spatial order
    for t in range(trow.shape[0]):
        assert sorted(trow[t][trow[t] >= 0].tolist()) == sorted(order[128 * t:128 * t + 128].tolist())
    # dispatch order: a permutation of the tiles, most expensive (LDS passes x active offsets) first, ties in plan order
    torder = _plan_torder(plan)
    assert sorted(torder.tolist()) == list(range(trow.shape[0]))
    cost = np.array([((meta[t, 0] + 447) // 448) * bin(int(meta[t, 1]) & 0xFFFFFFFF).count("1") if meta[t, 6] else 0 for t in range(trow.shape[0])])
    assert all((cost[a] > cost[b]) or (cost[a] == cost[b] and a < b) for a, b in zip(torder[:-1], torder[1:]))
    for t in range(trow.shape[0]):
        rows = trow[t]
        live = rows >= 0
        assert meta[t, 6] == live.sum() and (not live.any() or live[:live.sum()].all())
        masks = np.array([sum(1 << k for k in range(27) if tb[r, k] >= 0) if r >= 0 else 0 for r in rows])
        assert (np.diff(masks[live]) <= 0).all()
        want_halo = np.unique(tb[rows[live]][tb[rows[live]] >= 0])
        H = meta[t, 0]
        assert H == want_halo.size and np.array_equal(halo[t, :H], want_halo)
        assert meta[t, 1] == np.bitwise_or.reduce(masks)
        for w in range(4):
            assert meta[t, 2 + w] == np.bitwise_or.reduce(masks[32 * w:32 * w + 32])
        for s in np.nonzero(live)[0]:
            for k in range(27):
                nb = tb[rows[s], k]
                assert (loc[t, k, s] == 0xFFFF) if nb < 0 else (halo[t, loc[t, k, s]] == nb)
    # spatial locality is the point: a tile's halo is a small multiple of its rows
    assert meta[:, 0].mean() < 3.0 * 128


_B128_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]


def _a_read_cycles(trow, meta, loc):
    """mean LDS cycles of one gathered A-fragment ds_read_b128 of k_tile_conv on a plan (MI355X_MICROARCH.md, LDS: four groups of 16 lanes, one
    cycle per group + one per extra distinct address on a busy 16-byte bank column; same address = broadcast).  Halo row l of the kernel's
    [row][32 B] planes with swapped halves on odd (l >> 3): a lane group (one k half) reads column (2 l + half) mod 16; absent -> zero row 448 + (loc & 15)"""
    tot = cnt = 0
    for t in range(trow.shape[0]):
        for w in range(4):
            if not meta[t, 2 + w]:
                continue
            for k in range(loc.shape[1]):
                if not (int(meta[t, 2 + w]) >> k) & 1:
                    continue
                raw = loc[t, k, 32 * w:32 * w + 32].astype(np.int64)
                l = np.where(raw < 448, raw, 448 + (raw & 15))
                for g in _B128_GROUPS:
                    col = {}
                    for lane in g:
                        col.setdefault((2 * int(l[lane]) + ((int(l[lane]) >> 3) & 1)) % 16, set()).add(int(l[lane]))
                    tot += 2 * max(len(v) for v in col.values())  # both k halves: the same rows, the other half-columns
                cnt += 1
    return tot / max(cnt, 1)


def test_coloured_tile_plan_places_a_lane_groups_neighbours_in_different_bank_columns(monkeypatch):
    """ls3d_tile_plan flag bit 1 on a level-3-like SubM table: the same tiles / halo sets / neighbour references as the mask-ordered plan, absent
    neighbours marked >= 0xFFF0, bit-identical convolutions (both kernel loops), and the point of it: the gathered A reads of the kernel cost far
    fewer LDS cycles (model of the ds_read_b128 groups)"""
    monkeypatch.setattr(ops, "_TILE_COLOR", 1)
    cfg = synth.NUSC
    pts = synth.lidar_frame(30000, seed=5, **cfg)
    v, c, n, nv = ops.voxelize_hard(torch.from_numpy(pts), cfg["voxel_size"], cfg["pc_range"], 5, 60000)
    c = np.unique(c[:int(nv)].numpy() // np.array([4, 4, 4]), axis=0).astype(np.int32)
    V = c.shape[0]
    coords = torch.from_numpy(np.concatenate([np.zeros((V, 1), np.int32), c], 1)).contiguous()
    shape = tuple(int(s) // 4 + 1 for s in orc.spatial_shape(cfg["voxel_size"], cfg["pc_range"]))
    tbl = ops.rulebook_subm(coords, shape, (3, 3, 3))
    tb = tbl.numpy()
    plain = _plan_views(ops.tile_plan(tbl, coords, shape, 1))
    plan = ops.tile_plan(tbl, coords, shape, 1, color=True)
    trow, meta, halo, loc = _plan_views(plan)
    assert trow.shape[0] >= 8
    for t in range(trow.shape[0]):
        rows = trow[t]
        live = rows >= 0
        assert sorted(rows[live].tolist()) == sorted(plain[0][t][plain[0][t] >= 0].tolist()) and meta[t, 6] == live.sum()
        want_halo = np.unique(tb[rows[live]][tb[rows[live]] >= 0])
        H = meta[t, 0]
        assert want_halo.size <= H <= 448 and set(halo[t, :H].tolist()) == set(want_halo.tolist())  # holes repeat a halo row
        masks = np.array([sum(1 << k for k in range(27) if tb[r, k] >= 0) if r >= 0 else 0 for r in rows])
        assert meta[t, 1] == np.bitwise_or.reduce(masks) and meta[t, 1] == plain[1][t, 1]
        for w in range(4):
            assert meta[t, 2 + w] == np.bitwise_or.reduce(masks[32 * w:32 * w + 32])
        used = {}
        for s in np.nonzero(live)[0]:
            for k in range(27):
                nb = tb[rows[s], k]
                if nb < 0:
                    assert loc[t, k, s] >= 0xFFF0
                else:
                    assert loc[t, k, s] < H and halo[t, loc[t, k, s]] == nb
                    assert used.setdefault(int(nb), int(loc[t, k, s])) == int(loc[t, k, s])  # one slot per halo row
    assert np.array_equal(_plan_deps(plan)[0][:V], _plan_deps(ops.tile_plan(tbl, coords, shape, 1))[0][:V])  # (entries beyond the rows are not written)
    cyc_plain, cyc_col = _a_read_cycles(*plain[:2], plain[3]), _a_read_cycles(trow, meta, loc)
    assert cyc_plain > 6.5 and cyc_col < 0.85 * cyc_plain and cyc_col < 6.0, (cyc_plain, cyc_col)
    rng = np.random.default_rng(3)
    for cin, cout in ((16, 32), (32, 64)):  # the plain offset loop (one column block) and the pipelined one
        x = torch.from_numpy(rng.normal(size=(V, cin)).astype(np.float32))
        pw = PackedWeight(torch.from_numpy(rng.normal(size=(27, cin, cout)).astype(np.float32) * 0.1), 27, cin, cin, cout)
        a = ops.tile_conv(x, pw, ops.tile_plan(tbl, coords, shape, 1), cout=cout, products=6)
        assert torch.equal(ops.tile_conv(x, pw, plan, cout=cout, products=6), a)


def _plan_deps(plan):
    """(rowtile [T * 128], tdep [T, 32]) of an ops.TilePlan buffer (csrc/tileconv.hip:tc_plan)"""
    al = lambda v: (v + 255) // 256 * 256
    t, kvol = (plan.n_rows + 127) // 128, plan.kvol
    o = al(t * 128 * 4) + al(t * 8 * 4) + al(t * kvol * 128 * 4) + al(t * kvol * 128 * 2) + al((t + 1) * 4)
    raw = plan.buf.numpy()
    rowtile = raw[o:o + t * 128 * 4].view(np.int32).copy(); o += al(t * 128 * 4)
    return rowtile, raw[o:o + t * 32 * 4].view(np.int32).reshape(t, 32).copy()


def _subm_frame(n_points, seed, spare=0):
    cfg = synth.NUSC
    pts = synth.lidar_frame(n_points, seed=seed, **cfg)
    v, c, n, nv = ops.voxelize_hard(torch.from_numpy(pts), cfg["voxel_size"], cfg["pc_range"], 5, 20000)
    V = int(nv)
    coords = torch.zeros((V + spare, 4), dtype=torch.int32)
    coords[:V, 1:] = c[:V]
    shape = orc.spatial_shape(cfg["voxel_size"], cfg["pc_range"])
    n_dev = torch.tensor([V], dtype=torch.int32) if spare else None
    tbl = ops.rulebook_subm(coords, shape, (3, 3, 3), n_dev=n_dev)
    return V, coords, shape, tbl, n_dev


def test_tile_plan_producer_lists_on_a_subm_rulebook():
    """the producer list of a tile (ls3d_tile_conv_chain waits on it) = the set of tiles that own its halo rows, itself included; rowtile is the
    inverse of the tiles' row lists; with spare rows behind a device count the lists of the live tiles are the same"""
    for spare in (0, 200):
        V, coords, shape, tbl, n_dev = _subm_frame(4000, 11, spare)
        plan = ops.tile_plan(tbl, coords, shape, 1, n_dev=n_dev)
        trow, meta, halo, loc = _plan_views(plan)
        rowtile, tdep = _plan_deps(plan)
        live_tiles = (V + 127) // 128
        for t in range(live_tiles):
            rows = trow[t][trow[t] >= 0]
            assert (rowtile[rows] == t).all()
            want = set(rowtile[halo[t, :meta[t, 0]]].tolist())
            assert t in want and 0 <= meta[t, 7] == len(want) <= 32
            assert set(tdep[t, :meta[t, 7]].tolist()) == want
        assert meta[:live_tiles, 7].mean() < 12  # spatial tiles: a handful of neighbours
        if spare == 0:
            ref = [set(tdep[t, :meta[t, 7]].tolist()) for t in range(live_tiles)]
        else:
            assert ref == [set(tdep[t, :meta[t, 7]].tolist()) for t in range(live_tiles)]


@pytest.mark.parametrize("c,spare", [(32, 0), (64, 0), (128, 0), (64, 300)])
def test_tile_conv_chain_equals_layer_by_layer_launches(c, spare):
    """ls3d_tile_conv_chain - the layers of a UNet level in ONE persistent launch, tiles of layer l + 1 waiting on the producer tiles of their
    halo at layer l - is bit-identical to launching the layers one by one: a first layer of another input width, two SparseBasicBlocks
    (residual from two layers back, read coherently), a lateral block whose second convolution writes the right half of a concat buffer, and
    the 2C -> C layer with the channel-pair sum of that buffer (scn_unet.py:34-69,163-171); with the split over the input channels where
    ls3d_tile_conv takes it (c >= 64, fewer tiles than workgroup slots); on tensors with spare rows behind a device count."""
    rng = np.random.default_rng(c + spare)
    V, coords, shape, tbl, n_dev = _subm_frame(2500 if c < 128 else 700, 5, spare)
    rows = V + spare
    plan = ops.tile_plan(tbl, coords, shape, 1, n_dev=n_dev)
    T = torch.from_numpy
    c0 = 16 if c == 32 else c

    def weight(cin, cout):
        return PackedWeight(T((rng.normal(size=(27, cin, cout)) * (0.3 / np.sqrt(cin))).astype(np.float32)), 27, cin, cin, cout)

    def ss(cout):
        return T(rng.uniform(0.5, 1.5, cout).astype(np.float32)), T((rng.normal(size=cout) * 0.1).astype(np.float32))
    ws = [weight(c0, c)] + [weight(c, c) for _ in range(6)] + [weight(2 * c, c)]
    sc = [ss(c) for _ in range(8)]
    x0 = T(rng.normal(size=(rows, c0)).astype(np.float32))

    def nans(r, w):  # 128-byte aligned like the device allocator's blocks (a chained layer's output rows are whole 128-byte lines)
        flat = torch.full((r * w + 32,), float("nan"))
        off = (-flat.data_ptr() // 4) % 32
        return flat[off:off + r * w].view(r, w)

    def run(chained):
        bufs = [nans(rows, c) for _ in range(5)]
        cat = nans(rows, 2 * c)
        outm = nans(rows, c)
        # conv_input | block 1 | block 2 (its output = the left half of the level's concat buffer) | lateral block -> right half | conv_m (pair)
        spec = [(x0, 0, bufs[0], None, None), (bufs[0], 1, bufs[1], None, None), (bufs[1], 2, bufs[2], bufs[0], None),
                (bufs[2], 3, bufs[3], None, None), (bufs[3], 4, cat[:, :c], bufs[2], None),
                (cat[:, :c], 5, bufs[4], None, None), (bufs[4], 6, cat[:, c:], cat[:, :c], None), (cat, 7, outm, None, cat)]
        layers = [ops.ChainLayer(x, ws[i], out, cout=c, scale=sc[i][0], shift=sc[i][1], res_pre=res, relu=True, pair=pair) for x, i, out, res, pair in spec]
        if chained:
            states = ops.collect_chain_states(True)
            try:
                assert ops.tile_conv_chain(layers, plan)
            finally:
                ops.collect_chain_states(False)
            assert len(states) == 1 and int(states[0][1]) == 0  # no wait ran into the watchdog
            total = int(states[0][2])
            assert int(states[0][0]) >= total and total >= 8 * ((V + 127) // 128)  # every ticket was taken
        else:
            for l in layers:
                ops.tile_conv(l.x, l.w, plan, cout=c, products=6, scale=l.scale, shift=l.shift, res_pre=l.res_pre, relu=True, pair=l.pair, out=l.out,
                              in_ld=l.x.stride(0))
        return bufs, cat, outm
    ops.set_precision("bf16x6")
    try:
        want, got = run(False), run(True)
    finally:
        ops.set_precision("f32")
    for a, b in zip(want[0] + [want[1], want[2]], got[0] + [got[1], got[2]]):
        assert torch.equal(a[:V], b[:V]) and bool(torch.isfinite(a[:V]).all())
    assert float(want[2][:V].abs().max()) > 0


@pytest.mark.parametrize("cin,cout,products", [(32, 64, 8), (64, 128, 8), (16, 32, 6), (48, 96, 8)])
def test_tile_conv_matches_float64_and_gather_gemm(cin, cout, products):
    """random table (halos far beyond the LDS window: several passes per tile), fused epilogue, output into a column slice"""
    rng = np.random.default_rng(cin + cout)
    vin, vout, kvol = 900, 300, 27
    x = (rng.normal(size=(vin, cin)) * np.exp(rng.normal(size=(vin, cin)))).astype(np.float32)
    w = rng.normal(size=(kvol, cin, cout)).astype(np.float32) * 0.1
    tbl = rng.integers(0, vin, size=(vout, kvol)).astype(np.int32)
    tbl[rng.uniform(size=tbl.shape) < 0.5] = -1
    tbl[7] = -1
    tbl[130:140, 3:] = -1
    scale, shift = rng.uniform(0.5, 1.5, cout).astype(np.float32), rng.normal(size=cout).astype(np.float32)
    res = rng.normal(size=(vout, cout)).astype(np.float32)
    pair = rng.normal(size=(vout, 2 * cout)).astype(np.float32)
    acc = _sparse_ref(x, w, tbl)
    mag = _sparse_ref(np.abs(x), np.abs(w), tbl) + 1e-30
    want = np.maximum(acc * scale + shift + res, 0) + pair[:, 0::2] + pair[:, 1::2]
    T = torch.from_numpy
    pw = PackedWeight(T(w), kvol, cin, cin, cout)
    coords = T(np.stack([np.zeros(vout), np.zeros(vout), rng.integers(0, 64, vout), rng.integers(0, 64, vout)], 1).astype(np.int32))
    plan = ops.tile_plan(T(tbl), coords, (1, 64, 64), 1)
    trow, meta, halo, loc = _plan_views(plan)
    assert meta[:, 0].max() > 448  # the multi-pass path is exercised
    raw = ops.tile_conv(T(x), pw, plan, cout=cout, products=products)
    err = np.abs(raw.numpy() - acc) / mag
    f32 = ops.gather_gemm(T(x), pw, tbl=T(tbl), cout=cout)
    err32 = np.abs(f32.numpy() - acc) / mag
    # the host emulation rounds after every product of every plane MFMA (8 x 16 f32 additions per 16 channels instead of 16), so
    # only the error CLASS is asserted here; the device figures are measured in tests/test_gpu_parity.py
    assert err.max() <= 4.0 * err32.max() + 2.0 ** -22, (err.max(), err32.max())
    out = ops.tile_conv(T(x), pw, plan, cout=cout, products=products, scale=T(scale), shift=T(shift), res_pre=T(res), relu=True, pair=T(pair))
    np.testing.assert_allclose(out.numpy(), want, rtol=0, atol=3e-4)
    wide = torch.zeros((vout, 2 * cout))
    ops.tile_conv(T(x), pw, plan, cout=cout, products=products, out=wide[:, cout:], out_ld=2 * cout)
    assert torch.equal(wide[:, cout:], raw) and float(wide[:, :cout].abs().max()) == 0.0
    again = ops.tile_conv(T(x), pw, plan, cout=cout, products=products)
    assert torch.equal(again, raw)


def test_round2_entry_points_on_empty_and_degenerate_inputs():
    """empty and degenerate inputs of the round-2 entry points: zero rows, one row, a table with no neighbour at all, a tile whose
    rows all lack an offset, zero points into the fused decoder, an empty segment list"""
    from lidarseg3d_amd.packing import PackedWeight
    rng = np.random.default_rng(0)
    w = torch.from_numpy((rng.normal(size=(27, 64, 64)) * 0.1).astype(np.float32))
    pw = PackedWeight(w, 27, 64, 64, 64)
    # zero rows
    tbl0 = torch.empty((0, 27), dtype=torch.int32)
    plan0 = ops.tile_plan(tbl0, torch.empty((0, 4), dtype=torch.int32), (4, 8, 8), 1)
    out0 = ops.tile_conv(torch.empty((0, 64)), pw, plan0, cout=64, products=6)
    assert tuple(out0.shape) == (0, 64)
    assert ops.radix_argsort(torch.empty((0,), dtype=torch.int32), 16).numel() == 0
    assert ops.radix_argsort(torch.tensor([5], dtype=torch.int32), 16).tolist() == [0]
    # one row that is its own (centre) neighbour, and rows with NO neighbour: bias / shift only
    x = torch.from_numpy(rng.normal(size=(3, 64)).astype(np.float32))
    tbl = torch.full((3, 27), -1, dtype=torch.int32)
    tbl[0, 13] = 0
    coords = torch.tensor([[0, 1, 2, 3], [0, 1, 2, 4], [0, 3, 7, 7]], dtype=torch.int32)
    plan = ops.tile_plan(tbl, coords, (4, 8, 8), 1)
    shift = torch.from_numpy(rng.normal(size=(64,)).astype(np.float32))
    got = ops.tile_conv(x, pw, plan, cout=64, products=8, shift=shift)
    want = torch.zeros((3, 64), dtype=torch.float64)
    want[0] = x[0].double() @ w[13].double()
    want += shift.double()
    assert float((got.double() - want).abs().max()) <= 1e-5
    # weight gradient of an empty / neighbour-free table is zero
    gw = ops.spconv_wgrad(x, torch.ones((3, 64)), torch.full((3, 27), -1, dtype=torch.int32), None, 64, 64, products=6)
    assert float(gw.abs().max()) == 0.0
    gw = ops.spconv_wgrad(torch.empty((0, 128)), torch.empty((0, 128)), torch.empty((0, 27), dtype=torch.int32), None, 128, 128, products=6)
    assert tuple(gw.shape) == (27, 128, 128) and float(gw.abs().max()) == 0.0
    # segment mean with empty segments in between
    src = torch.from_numpy(rng.normal(size=(5, 4)).astype(np.float32))
    seg = torch.tensor([3, 3, 0, 3, 6], dtype=torch.int32)
    from lidarseg3d_amd import scatter
    m = scatter.scatter_mean(src, seg.long(), dim=0, dim_size=8)
    assert torch.equal(m[0], src[2]) and float(m[[1, 2, 4, 5, 7]].abs().max()) == 0.0
    np.testing.assert_allclose(m[3].numpy(), ((src[0] + src[1]) + src[3]).numpy() / 3.0, rtol=1e-6)


@pytest.mark.parametrize("cin,cout,products", [(64, 128, 6), (64, 64, 6), (64, 32, 8), (128, 96, 6)])
def test_tile_conv_flags_and_channel_splits_are_bit_identical_where_they_must_be(cin, cout, products):
    """ls3d_tile_conv's `flags`: dispatching the tiles in plan order instead of most-expensive-first and the LDS bank swizzle change
    nothing; splitting tiles over the input channels changes only the split tiles' rows (two partial sums added at the end), is
    reproducible, and the arrival counters survive any number of launches without a reset; the fused epilogue runs on split and
    unsplit tiles.  4 tiles (one of them partial, one with rows that have no neighbour at some offsets)."""
    rng = np.random.default_rng(cin * 3 + cout)
    vin, vout, kvol = 390, 432, 27  # 4 tiles: three full, one of 48 rows
    x = rng.normal(size=(vin, cin)).astype(np.float32)
    w = rng.normal(size=(kvol, cin, cout)).astype(np.float32) * 0.1
    # neighbours from a window around the row: halos stay inside one LDS pass
    tbl = (np.arange(vout)[:, None] * vin // vout + rng.integers(-20, 20, size=(vout, kvol))).clip(0, vin - 1).astype(np.int32)
    tbl[rng.uniform(size=tbl.shape) < 0.4] = -1
    tbl[200:330, 5] = -1
    tbl[130:162, 9:] = -1
    T = torch.from_numpy
    pw = PackedWeight(T(w), kvol, cin, cin, cout)
    coords = T(np.stack([np.zeros(vout), np.zeros(vout), np.arange(vout) // 32, np.arange(vout) % 32], 1).astype(np.int32))
    scale, shift = T(rng.uniform(0.5, 1.5, cout).astype(np.float32)), T(rng.normal(size=cout).astype(np.float32))
    res = T(rng.normal(size=(vout, cout)).astype(np.float32))
    counters = torch.zeros((512,), dtype=torch.int32)  # one array for every launch of this test: never reset

    def run(conv_flags, plan_flags=0, **kw):
        ops.set_tile_flags(conv=conv_flags, plan=plan_flags)
        orig = ops._tile_counters
        ops._tile_counters = lambda like: counters
        try:
            plan = ops.tile_plan(T(tbl), coords, (1, 32, 32), 1)
            return ops.tile_conv(T(x), pw, plan, cout=cout, products=products, **kw), plan
        finally:
            ops.set_tile_flags(conv=0, plan=0)
            ops._tile_counters = orig

    NEVER, ALL = 1 << 6, 2 << 6
    base, plan = run(NEVER)
    assert _plan_views(plan)[1][:, 0].max() <= 448
    np.testing.assert_allclose(base.numpy(), _sparse_ref(x, w, tbl), rtol=0, atol=2e-4)
    assert torch.equal(run(NEVER | (1 << 30))[0], base)         # no LDS bank swizzle
    assert torch.equal(run(NEVER, plan_flags=1)[0], base)       # plan-order dispatch
    allsplit, _ = run(0)                                        # 4 tiles <= 512: every tile is split
    assert torch.equal(run(ALL)[0], allsplit) and torch.equal(run(1 << 30)[0], allsplit)
    assert torch.equal(run(0)[0], allsplit)                     # reproducible, counters left even by every launch
    assert int((counters % 2).sum()) == 0 and int(counters[:4].min()) >= 8
    np.testing.assert_allclose(allsplit.numpy(), base.numpy(), rtol=0, atol=2e-5)
    two, plan2 = run((2 + 1) << 8)                              # only the two tiles at the end of the dispatch order are split
    torder, trow = _plan_torder(plan2), _plan_views(plan2)[0]
    split_rows = np.concatenate([trow[t][trow[t] >= 0] for t in torder[-2:]])
    keep_rows = np.setdiff1d(np.arange(vout), split_rows)
    assert torch.equal(two[keep_rows], base[keep_rows]) and torch.equal(two[split_rows], allsplit[split_rows])
    if cin >= 128:
        assert not torch.equal(allsplit, base)                  # the split really happened
    pair = T(rng.normal(size=(vout, 2 * cout)).astype(np.float32))
    for fl in (0, NEVER):
        a = run(fl, scale=scale, shift=shift, res_pre=res, relu=True)[0]
        np.testing.assert_allclose(a.numpy(), np.maximum((allsplit if fl != NEVER else base).numpy() * scale.numpy() + shift.numpy() + res.numpy(), 0),
                                   rtol=0, atol=1e-5)
        # flags bit 1: the general two-pass epilogue instead of the tile kernel's single-pass one - the same arithmetic per element
        assert torch.equal(run(fl | 2, scale=scale, shift=shift, res_pre=res, relu=True)[0], a)
        b = run(fl, scale=scale, shift=shift, relu=True, pair=pair)[0]
        assert torch.equal(run(fl | 2, scale=scale, shift=shift, relu=True, pair=pair)[0], b)
        want = np.maximum((allsplit if fl != NEVER else base).numpy() * scale.numpy() + shift.numpy(), 0) + pair.numpy()[:, 0::2] + pair.numpy()[:, 1::2]
        np.testing.assert_allclose(b.numpy(), want, rtol=0, atol=1e-5)
    assert torch.equal(run(NEVER | 2)[0], base)
    # flags bit 0: the plain offset loop instead of the software-pipelined one (6 products, cout > 32): the same products in the same order
    assert torch.equal(run(NEVER | 1)[0], base) and torch.equal(run(1)[0], allsplit)


@pytest.mark.parametrize("cin,cout", [(32, 160), (16, 288)])
def test_tile_conv_more_than_128_output_columns(cin, cout):
    """cout > 128 (SCALING_RATIO > 2 of the reference's UNet): slabs of 128 columns on one plan, fused epilogue operands moving with
    the columns; == the gather-GEMM (which has no such limit) to f32 rounding, and the float64 sum"""
    rng = np.random.default_rng(cout)
    vin, vout, kvol = 150, 170, 27
    x = rng.normal(size=(vin, cin)).astype(np.float32)
    w = rng.normal(size=(kvol, cin, cout)).astype(np.float32) * 0.1
    tbl = (np.arange(vout)[:, None] * vin // vout + rng.integers(-20, 20, size=(vout, kvol))).clip(0, vin - 1).astype(np.int32)
    tbl[rng.uniform(size=tbl.shape) < 0.5] = -1
    T = torch.from_numpy
    pw = PackedWeight(T(w), kvol, cin, cin, cout)
    coords = T(np.stack([np.zeros(vout), np.zeros(vout), np.arange(vout) // 16, np.arange(vout) % 16], 1).astype(np.int32))
    scale, shift = T(rng.uniform(0.5, 1.5, cout).astype(np.float32)), T(rng.normal(size=cout).astype(np.float32))
    res, pair = T(rng.normal(size=(vout, cout)).astype(np.float32)), T(rng.normal(size=(vout, 2 * cout)).astype(np.float32))
    plan = ops.tile_plan(T(tbl), coords, (1, 16, 16), 1)
    want = _sparse_ref(x, w, tbl)
    for products in (6, 8):
        got = ops.tile_conv(T(x), pw, plan, cout=cout, products=products)
        np.testing.assert_allclose(got.numpy(), want, rtol=0, atol=2e-4)
    a = ops.tile_conv(T(x), pw, plan, cout=cout, products=6, scale=scale, shift=shift, res_pre=res, relu=True).numpy()
    np.testing.assert_allclose(a, np.maximum(got.numpy() * 0 + ops.tile_conv(T(x), pw, plan, cout=cout, products=6).numpy() * scale.numpy() + shift.numpy() + res.numpy(), 0),
                               rtol=0, atol=1e-5)
    b = ops.tile_conv(T(x), pw, plan, cout=cout, products=6, pair=pair).numpy()
    np.testing.assert_allclose(b, ops.tile_conv(T(x), pw, plan, cout=cout, products=6).numpy() + pair.numpy()[:, 0::2] + pair.numpy()[:, 1::2], rtol=0, atol=1e-5)
    gg = ops.gather_gemm(T(x), pw, tbl=T(tbl), cout=cout).numpy()
    np.testing.assert_allclose(got.numpy(), gg, rtol=0, atol=2e-5 * float(np.abs(gg).max()))


@pytest.mark.parametrize("cin,cout", [(32, 32), (48, 64), (64, 128), (16, 160)])
def test_tile_conv_plain_bf16_products_on_the_single_plane_layout(cin, cout):
    """products = 1 (BASELINE configs[4]): bf16-rounded operands, one MFMA per product, the single-plane weight layout of
    ls3d_tile_conv_pack_bf16 (12 / 6 / 3 kernel offsets per 12 KB step of the weight stream) == a float64 evaluation on operands rounded to
    bf16 (round to nearest even), to f32 accumulation noise; 27 offsets with absent neighbours, partial last steps, fused epilogue"""
    rng = np.random.default_rng(cin + cout)
    vin, vout, kvol = 300, 330, 27
    x = rng.normal(size=(vin, cin)).astype(np.float32)
    w = rng.normal(size=(kvol, cin, cout)).astype(np.float32) * 0.1
    tbl = (np.arange(vout)[:, None] * vin // vout + rng.integers(-20, 20, size=(vout, kvol))).clip(0, vin - 1).astype(np.int32)
    tbl[rng.uniform(size=tbl.shape) < 0.5] = -1
    tbl[:, 7] = -1   # an offset nobody has: the steps of the other 26 regroup
    T = torch.from_numpy

    def bf16(a):
        return T(a).to(torch.bfloat16).to(torch.float64).numpy()
    pw = PackedWeight(T(w), kvol, cin, cin, cout)
    coords = T(np.stack([np.zeros(vout), np.zeros(vout), np.arange(vout) // 32, np.arange(vout) % 32], 1).astype(np.int32))
    plan = ops.tile_plan(T(tbl), coords, (1, 32, 32), 1)
    got = ops.tile_conv(T(x), pw, plan, cout=cout, products=1)
    want = _sparse_ref(bf16(x), bf16(w), tbl)
    np.testing.assert_allclose(got.numpy(), want, rtol=0, atol=2e-5 * float(np.abs(want).max()) + 1e-5)
    assert torch.equal(ops.tile_conv(T(x), pw, plan, cout=cout, products=1), got)
    scale, shift = T(rng.uniform(0.5, 1.5, cout).astype(np.float32)), T(rng.normal(size=cout).astype(np.float32))
    a = ops.tile_conv(T(x), pw, plan, cout=cout, products=1, scale=scale, shift=shift, relu=True)
    np.testing.assert_allclose(a.numpy(), np.maximum(got.numpy() * scale.numpy() + shift.numpy(), 0), rtol=0, atol=1e-5)
    six = ops.tile_conv(T(x), pw, plan, cout=cout, products=6)   # the same PackedWeight serves both layouts
    np.testing.assert_allclose(six.numpy(), _sparse_ref(x, w, tbl), rtol=0, atol=2e-4)


def test_tile_conv_any_row_order_gives_the_same_rows():
    """tiles only group rows: with single-pass halos the per-row summation order (chunks outer, offsets inner) does not depend
    on the tiling, so two different spatial orders give bit-identical outputs"""
    rng = np.random.default_rng(9)
    vin, vout, kvol, cin, cout = 200, 260, 27, 32, 32
    x = rng.normal(size=(vin, cin)).astype(np.float32)
    w = rng.normal(size=(kvol, cin, cout)).astype(np.float32) * 0.1
    tbl = rng.integers(0, vin, size=(vout, kvol)).astype(np.int32)
    tbl[rng.uniform(size=tbl.shape) < 0.8] = -1
    T = torch.from_numpy
    pw = PackedWeight(T(w), kvol, cin, cin, cout)
    coords = T(np.zeros((vout, 4), np.int32))
    outs = []
    for order in (np.arange(vout), rng.permutation(vout)):
        plan = ops.tile_plan(T(tbl), coords, (1, 8, 8), 1, order=T(order.astype(np.int32)))
        assert _plan_views(plan)[1][:, 0].max() <= 448
        outs.append(ops.tile_conv(T(x), pw, plan, cout=cout, products=8))
    assert torch.equal(outs[0], outs[1])
    np.testing.assert_allclose(outs[0].numpy(), _sparse_ref(x, w, tbl), rtol=0, atol=1e-4)


def test_unet_bf16x8_tile_path_vs_f32(monkeypatch):
    """UNetSCN3D with the SubM layers on the tile-halo kernel (precision bf16x8) against the exact-f32 gather-GEMM path"""
    cfg = synth.NUSC
    from lidarseg3d_amd import models_cfg
    import lidarseg3d_amd as L
    pts = synth.lidar_frame(90, seed=21, **cfg)
    v, c, n, nv = ops.voxelize_hard(torch.from_numpy(pts), cfg["voxel_size"], cfg["pc_range"], 5, 20000)
    V = int(nv)
    coords = torch.cat([torch.zeros((V, 1), dtype=torch.int32), c[:V]], 1).contiguous()
    bcfg = dict(models_cfg.sdseg3d()["backbone"], model_cfg=dict(SCALING_RATIO=1))  # 16/32/64/64 channels: fewer chunks to emulate
    net = L.build_backbone(bcfg).eval()
    shapes = {k: tuple(t.shape) for k, t in net.state_dict().items()}
    net.load_state_dict({k: torch.from_numpy(a) for k, a in synth.random_state_dict(shapes, 3).items()})
    feats = torch.from_numpy(np.random.default_rng(2).normal(size=(V, 16)).astype(np.float32))
    gs = orc.grid_size(cfg["voxel_size"], cfg["pc_range"])

    def run():
        bd = dict(voxel_features=feats, voxel_coords=coords, batch_size=1, input_shape=np.array([int(gs[0]), int(gs[1]), int(gs[2])]))
        return net(bd)["conv_point_features"]

    ref = run()
    ops.set_precision("bf16x8")
    ops.set_tile(True, min_cc=256)  # also the 16-channel level of this small net
    calls = []
    orig = ops.tile_conv
    monkeypatch.setattr(ops, "tile_conv", lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
    try:
        got = run()
    finally:
        ops.set_precision("f32")
        ops.set_tile(True, min_cc=512)
    assert len(calls) >= 25
    assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 1e-6


@pytest.mark.parametrize("cin,capacity", [(13, True), (16, False)])
def test_unet_chained_levels_equal_layer_by_layer_launches(cin, capacity, monkeypatch):
    """UNetSCN3D's inference forward with every level's SubM layers as one chained launch (scn_unet._Chain -> ls3d_tile_conv_chain) is
    bit-identical to the layer-by-layer launches (ops.set_tile_chain(False): the round-4 schedule with the lateral blocks on their own stream), with
    host-side counts (one case) and in capacity mode (the other): the narrow net of the emulation (16 / 32 / 64 / 64 channels) with the chain's
    thresholds lowered - levels 2 - 4 chained (6 + 6 + 7 layers, the last with conv_m4 reading the concat buffer the chain itself fills), the
    16-channel level 1 (rows narrower than a 128-byte line: not chainable) through the same code path layer by layer; every output of the
    backbone is compared.  (First layers of another width, the 2-layer chain, 128 channels: test_tile_conv_chain_equals_...)"""
    ratio, min_cc = 1, 512
    cfg = synth.NUSC
    pts = synth.lidar_frame(150, seed=31, **cfg)
    v, c, n, nv = ops.voxelize_hard(torch.from_numpy(pts), cfg["voxel_size"], cfg["pc_range"], 5, 20000)
    V = int(nv)
    coords = torch.cat([torch.zeros((V, 1), dtype=torch.int32), c[:V]], 1).contiguous()
    net = scn_unet.UNetSCN3D(num_input_features=cin, voxel_size=cfg["voxel_size"], point_cloud_range=cfg["pc_range"],
                             model_cfg=dict(SCALING_RATIO=ratio), ds_factor=8, us_factor=8).eval()
    shapes = {k: tuple(t.shape) for k, t in net.state_dict().items()}
    net.load_state_dict({k: torch.from_numpy(a) for k, a in synth.random_state_dict(shapes, 3).items()})
    feats = torch.from_numpy(np.random.default_rng(2).normal(size=(V, cin)).astype(np.float32))
    shape = np.asarray(orc.grid_size(cfg["voxel_size"], cfg["pc_range"]))
    orig_caps = net._capacities
    monkeypatch.setattr(net, "_capacities", lambda n_, b, sh: [min(w, 6 * n_) for w in orig_caps(n_, b, sh)])
    chains = []
    orig = ops.tile_conv_chain
    monkeypatch.setattr(ops, "tile_conv_chain", lambda layers, plan: (chains.append(len(layers)), orig(layers, plan))[1])

    def run(chain, capacity):
        ops.set_tile_chain(chain, min_tiles=1, min_cout=16)  # (by default only levels of > 600 tiles and >= 64 channels are chained)
        bd = dict(voxel_features=feats, voxel_coords=coords, batch_size=1, input_shape=shape)
        if capacity:
            bd["num_active_voxels_dev"] = torch.tensor([V], dtype=torch.int32)
        with torch.no_grad():
            bd = net(bd)
        ms = bd["multi_scale_3d_features"]
        outs = [bd["conv_point_features"][:V]]
        for k in ("x_conv1", "x_conv2", "x_conv3", "x_conv4"):
            t = ms[k]
            m = int(t.n_dev) if t.n_dev is not None else t.features.shape[0]
            outs += [t.features[:m].contiguous(), t.indices[:m]]
        return outs
    ops.set_precision("bf16x6")
    ops.set_tile(True, min_cc=min_cc)
    states = ops.collect_chain_states(True)
    try:
        want = run(False, capacity)
        assert chains == []
        got = run(True, capacity)
        assert chains == [6, 6, 7]
        for a, b in zip(want, got):
            assert a.shape == b.shape and torch.equal(a, b)
        assert bool(torch.isfinite(got[0]).all()) and float(got[0].abs().max()) > 0
        assert all(int(st[1]) == 0 for st in states)
    finally:
        ops.collect_chain_states(False)
        ops.set_tile_chain(True, min_tiles=600, min_cout=64)
        ops.set_tile(True, min_cc=512)
        ops.set_precision("f32")


def test_interpolate_rows_backward_is_the_transpose_and_deterministic():
    """ls3d_interpolate_rows_backward (the devoxelization's gradient for the voxel features in the training forward): equals torch's backward of
    the gather composition, every voxel row written (also rows no point refers to), two ragged frames, bit-reproducible"""
    rng = np.random.default_rng(0)
    V0, V1, n0, n1, C = 90, 40, 170, 61, 8
    feat = torch.randn(V0 + V1, C, requires_grad=True)
    vx_off = torch.tensor([0, V0, V0 + V1], dtype=torch.int32)
    pts = torch.cat([torch.cat([torch.zeros(n0), torch.ones(n1)])[:, None], torch.randn(n0 + n1, 3)], 1).contiguous()
    idx = torch.from_numpy(np.concatenate([rng.integers(0, V0 - 10, size=(n0, 3)), rng.integers(0, V1, size=(n1, 3))]).astype(np.int32))
    w = torch.rand(n0 + n1, 3)
    out = ops.interpolate_rows_autograd(feat, idx, w, pts, vx_off)
    g = torch.randn_like(out)
    out.backward(g)
    f2 = feat.detach().clone().requires_grad_(True)
    v0 = vx_off[pts[:, 0].long()].unsqueeze(1)
    ref = (f2[(idx + v0).long()] * w.unsqueeze(-1)).sum(1)
    ref.backward(g)
    assert float((out - ref).abs().max()) <= 1e-6 and float((feat.grad - f2.grad).abs().max()) <= 2e-6 * float(f2.grad.abs().max())
    assert float(feat.grad[V0 - 10:V0].abs().max()) == 0.0  # rows nobody interpolates from: zero, not garbage
    again = ops.interpolate_rows_backward(g, idx, w, pts, vx_off, V0 + V1)
    assert torch.equal(again, feat.grad)
    assert float(ops.interpolate_rows_backward(torch.empty((0, C)), torch.empty((0, 3), dtype=torch.int32), torch.empty((0, 3)), torch.empty((0, 4)),
                                               vx_off, V0 + V1).abs().max()) == 0.0


def test_linear_function_gradients_on_the_kernels():
    """ops._LinearFn (what nn.Linear records under ops.fast_linear_backward on the device): grad_x on ls3d_gather_gemm with the [out, in] weight as
    the [K][N] operand, grad_W on ls3d_spconv_wgrad over the identity pair lists of a row capacity (two row counts share one list), grad_b"""
    torch.manual_seed(3)
    lin = torch.nn.Linear(48, 32)
    for n in (700, 333):
        x = torch.randn(n, 48).relu_().requires_grad_(True)
        gy = torch.randn(n, 32) * 0.1
        lin.zero_grad()
        y = ops._LinearFn.apply(x, lin.weight, lin.bias)
        y.backward(gy)
        xr = x.detach().clone().requires_grad_(True)
        lr = torch.nn.Linear(48, 32)
        lr.load_state_dict(lin.state_dict())
        lr(xr).backward(gy)
        assert float((x.grad - xr.grad).abs().max()) <= 2e-6 * float(xr.grad.abs().max())
        assert float((lin.weight.grad - lr.weight.grad).abs().max()) <= 2e-6 * float(lr.weight.grad.abs().max())
        assert float((lin.bias.grad - lr.bias.grad).abs().max()) <= 2e-6 * float(lr.bias.grad.abs().max())
    assert len(ops._IDENTITY_PAIRS) == 1  # one identity list for both row counts


def test_frozen_batchnorm_and_eval_mode_input_gradients_keep_the_graph():
    """with BatchNorm frozen (bn.eval() inside a model in train mode) the epilogue-fused launches must not cut the
    graph: every convolution on the path still gets its weight gradient; in eval mode an input that requires grad gets one, and
    the values equal the fused inference path's"""
    cfg = synth.NUSC
    g = golden("unet_nusc_c13.npz")
    n = 120
    coords = torch.from_numpy(g["coords"][:n])
    feats0 = torch.from_numpy(g["voxel_features"][:n])
    net = scn_unet.UNetSCN3D(num_input_features=13, voxel_size=cfg["voxel_size"], point_cloud_range=cfg["pc_range"],
                             model_cfg=dict(SCALING_RATIO=1), ds_factor=8, us_factor=8)
    shape = np.asarray(orc.grid_size(cfg["voxel_size"], cfg["pc_range"]))
    net.train()
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.eval()
    out = net(dict(voxel_features=feats0.clone(), voxel_coords=coords, batch_size=1, input_shape=shape))["conv_point_features"]
    out.sum().backward()
    missing = [k for k, p in net.named_parameters() if p.dim() == 5 and p.grad is None and not k.startswith("conv_out")]
    assert not missing, missing
    assert net.conv_input[0].weight.grad.abs().sum() > 0
    net.eval()
    with torch.no_grad():
        ref = net(dict(voxel_features=feats0.clone(), voxel_coords=coords, batch_size=1, input_shape=shape))["conv_point_features"]
    f = feats0.clone().requires_grad_(True)
    got = net(dict(voxel_features=f, voxel_coords=coords, batch_size=1, input_shape=shape))["conv_point_features"]
    got.sum().backward()
    assert f.grad is not None and float(f.grad.abs().sum()) > 0
    np.testing.assert_allclose(got.detach().numpy(), ref.numpy(), rtol=0, atol=1e-4 * max(1.0, float(ref.abs().max())))


def test_pointnet2_utils_dropin_forward_and_backward_vs_oracle():
    """ls3d_three_interpolate / _grad and pointnet2_utils.ThreeNN / ThreeInterpolate had no test"""
    from tests import pointnet2_cases
    pointnet2_cases.run("cpu")


def test_voxel_cap_applies_per_frame_like_the_dataloader():
    """max_voxel_num caps EACH frame (the dataloader voxelises per sample); a batch in which one frame overflows and the
    total does not must drop exactly the voxels the reference drops"""
    from lidarseg3d_amd import detectors
    cfg = synth.NUSC
    frames = [synth.lidar_frame(900, seed=31, **cfg), synth.lidar_frame(60, seed=32, **cfg)]
    pts = np.concatenate([np.concatenate([np.full((f.shape[0], 1), b, np.float32), f], 1) for b, f in enumerate(frames)])
    mv = 200  # frame 0 has more voxels than this, frame 1 fewer
    vcfg = dict(range=cfg["pc_range"], voxel_size=cfg["voxel_size"], max_points_in_voxel=5, max_voxel_num=[mv, mv])
    ex = dict(points=torch.from_numpy(pts), batch_size=2)
    v, c, n, bs, grid, n_dev = detectors._voxel_inputs(ex, vcfg)
    assert n_dev is None
    want = orc.collate_frames(frames, cfg["voxel_size"], cfg["pc_range"], 5, mv)
    assert int((want["coordinates"][:, 0] == 0).sum()) == mv  # the cap did bite on frame 0
    assert torch.equal(c, want["coordinates"]) and torch.equal(n, want["num_points"]) and torch.equal(v, want["voxels"])
    assert ex["num_voxels"].tolist() == [mv, int((want["coordinates"][:, 0] == 1).sum())]


@pytest.mark.parametrize("kind,precs", [("sdseg3d", ("f32", "bf16x6")), ("mseg3d", ("f32",))])
def test_capacity_mode_equals_host_count_mode_bit_for_bit(kind, precs, monkeypatch):
    """inference from raw points on device-side row counts (detectors.CAPACITY_MODE: tensors sized by capacities, no host
    synchronisation inside the frame) against the same frames with host-side counts: identical logits and labels, for exact f32 and
    for the tile-halo arithmetic; a rulebook that overflows its capacity sends the frame through the host-count path (same results)
    and makes the backbone forget the capacities it learned; two ragged frames, one of them tiny"""
    import lidarseg3d_amd as L
    from lidarseg3d_amd import detectors, models_cfg
    cfg = synth.NUSC
    mcfg = getattr(models_cfg, kind)()
    mcfg["backbone"]["model_cfg"] = dict(SCALING_RATIO=1)  # 16/32/64/64 channels: a quarter of the emulated MFMA work
    mcfg["point_head"]["model_cfg"]["CONV_IN_DIM" if kind == "sdseg3d" else "VOXEL_IN_DIM"] = 16
    model = L.build_detector(mcfg, train_cfg=None, test_cfg={}).eval()
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.random_state_dict(shapes, 7).items()})
    frames = [synth.lidar_frame(110, seed=41, **cfg), synth.lidar_frame(25, seed=42, **cfg)]
    pts = torch.from_numpy(np.concatenate([np.concatenate([np.full((f.shape[0], 1), b, np.float32), f], 1) for b, f in enumerate(frames)]))
    ex = dict(points=pts, batch_size=2)
    if kind == "mseg3d":
        img, emb, cuv = synth.camera_inputs(pts.shape[0], seed=3, ncam=6, c_img=48, h=8, w=12, batch=2)
        ex.update(points_cuv=torch.from_numpy(cuv), image_features=torch.from_numpy(img), camera_semantic_embeddings=torch.from_numpy(emb))
    bb = model.backbone
    seen = []
    orig_fc, orig_caps = bb._forward_capacity, bb._capacities
    monkeypatch.setattr(bb, "_forward_capacity", lambda *a, **k: (seen.append(1), orig_fc(*a, **k))[1])
    # the worst case (x8 per level) is 50k spare rows at the deepest level of these tiny frames: minutes on the host emulation
    roomy = lambda n, b, sh: [min(w, 6 * n) for w in orig_caps(n, b, sh)]

    def run(capacity, caps=roomy):
        monkeypatch.setattr(detectors, "CAPACITY_MODE", capacity)
        monkeypatch.setattr(bb, "_capacities", caps)
        with torch.no_grad():
            ret = model(dict(ex), return_loss=False)
        return model.point_head.forward_ret_dict["out_logits"].clone(), [r["pred_point_sem_labels"].clone() for r in ret]

    for prec in precs:
        ops.set_precision(prec)
        if prec == "bf16x6":
            ops.set_tile(True, min_cc=256)
        try:
            del seen[:]
            want, wl = run(False)
            assert not seen
            got, gl = run(True)
            assert len(seen) == 1 and torch.isfinite(got).all()
            assert torch.equal(got, want) and all(torch.equal(a, b) for a, b in zip(gl, wl))
            learned = dict(bb._caps)
            assert len(learned) == 1 and all(m > 0 for m in list(learned.values())[0])
            if prec == precs[0] and kind == "sdseg3d":
                got3, _ = run(True, caps=lambda n, b, sh: [8] * len(orig_caps(n, b, sh)))  # every strided rulebook overflows
                assert torch.equal(got3, want) and len(seen) == 2 and not bb._caps   # ran again on host counts, capacities forgotten
        finally:
            ops.set_precision("f32")
            ops.set_tile(True, min_cc=512)


def test_point_head_tail_in_one_launch_equals_layer_by_layer():
    """ls3d_point_mlp (3-NN interpolation + conv_align_layers + out_cls_layers + argmax of PointSegBatchlossHead in one kernel) against the
    layer-by-layer path (ls3d_interpolate_rows + four ls3d_gather_gemm + torch.argmax): logits to f32 rounding (another summation order), labels
    identical where the top two logits are not within that rounding; two ragged frames, a partial last tile; ties and NaN follow torch.argmax;
    chains it does not cover fall back"""
    torch.manual_seed(11)
    rng = np.random.default_rng(11)
    head = point_heads.PointSegBatchlossHead(False, 17, dict(CONV_IN_DIM=32, CONV_CLS_FC=[64], CONV_ALIGN_DIM=64, OUT_CLS_FC=[64, 64], IGNORED_LABEL=0)).eval()
    for m in head.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.normal_(0, 0.3); m.running_var.uniform_(0.5, 1.5); m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.2)
    V0, V1, n0, n1 = 90, 40, 170, 61
    feat = torch.randn(V0 + V1, 32)
    vx_off = torch.tensor([0, V0, V0 + V1], dtype=torch.int32)
    pts = torch.cat([torch.cat([torch.zeros(n0), torch.ones(n1)])[:, None], torch.randn(n0 + n1, 3)], 1).contiguous()
    idx = torch.from_numpy(np.concatenate([rng.integers(0, V0, size=(n0, 3)), rng.integers(0, V1, size=(n1, 3))]).astype(np.int32))
    w = torch.rand(n0 + n1, 3)
    w = (w / w.sum(1, keepdim=True)).contiguous()
    pk = head.packed()
    assert pk["tail"] is not None
    with torch.no_grad():
        pf = ops.interpolate_rows(feat, idx, w, pts, vx_off)
        want = point_heads._run_mlp(point_heads._run_mlp(pf, pk["align"]), pk["out_cls"])
        got, labels = ops.point_mlp(feat, pk["tail"], idx, w, pts, vx_off)
    assert got.shape == want.shape == (n0 + n1, 17)
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=0, atol=2e-5 * float(want.abs().max()))
    assert torch.equal(labels, torch.argmax(got, dim=1))  # the kernel's argmax of ITS logits is torch's
    top2 = torch.topk(want, 2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-4 * float(want.abs().max())
    assert torch.equal(labels[clear], torch.argmax(want, dim=1)[clear]) and int(clear.sum()) > 200
    # the chain on plain rows (idx = None), ties and NaN: a classifier with duplicated columns and a NaN bias entry
    lin = torch.nn.Linear(32, 17)
    with torch.no_grad():
        lin.weight[5] = lin.weight[3]; lin.bias[5] = lin.bias[3]; lin.weight[12] = lin.weight[3]; lin.bias[12] = lin.bias[3]
        lin.weight[3] *= 0; lin.weight[5] *= 0; lin.weight[12] *= 0; lin.bias[3] = lin.bias[5] = lin.bias[12] = 50.0  # three-way tie at the top
    mdl = point_heads._plain_mlp([lin])
    x = torch.randn(70, 32)
    out, lab = ops.point_mlp(x, mdl)
    np.testing.assert_allclose(out.numpy(), lin(x).detach().numpy(), rtol=0, atol=1e-4)
    assert torch.equal(lab, torch.argmax(out, dim=1)) and bool((lab == 3).all())
    with torch.no_grad():
        lin.bias[9] = float("nan")
    out, lab = ops.point_mlp(x, point_heads._plain_mlp([lin]))
    assert torch.equal(lab, torch.argmax(out, dim=1)) and bool((lab == 9).all())
    assert point_heads._plain_mlp([torch.nn.Linear(32, 48), torch.nn.ReLU(), torch.nn.Linear(48, 17)]) is None  # 48 hidden channels: composed


def test_capacity_mode_encoded_tensor_is_computed_on_demand(monkeypatch):
    """capacity mode leaves batch_dict["encoded_spconv_tensor"] (conv_out of the deepest level: no segmentation head reads it) as a proxy that runs
    the convolution - and builds its rulebook - on first access: same sites and features as the eager conv_out of the same frame, and a frame that
    never reads it builds three strided rulebooks instead of four"""
    import lidarseg3d_amd as L
    from lidarseg3d_amd import models_cfg, scn_unet
    cfg = synth.NUSC
    mcfg = models_cfg.sdseg3d()
    mcfg["backbone"]["model_cfg"] = dict(SCALING_RATIO=1)
    mcfg["point_head"]["model_cfg"]["CONV_IN_DIM"] = 16
    model = L.build_detector(mcfg, train_cfg=None, test_cfg={}).eval()
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.random_state_dict(shapes, 7).items()})
    f = synth.lidar_frame(120, seed=41, **cfg)
    pts = torch.from_numpy(np.concatenate([np.zeros((f.shape[0], 1), np.float32), f], 1))
    bb = model.backbone
    orig_caps = bb._capacities
    monkeypatch.setattr(bb, "_capacities", lambda n, b, sh: [min(w, 6 * n) for w in orig_caps(n, b, sh)])
    got = {}
    for lazy in (True, False):
        L.set_reference_outputs(not lazy)
        try:
            with torch.no_grad():
                data = model.forward_features(dict(points=pts, batch_size=1), capacity=True)
            assert len(bb._strided_chain()) == (3 if lazy else 4)
        finally:
            L.set_reference_outputs(False)
        enc = data["encoded_spconv_tensor"]
        from lidarseg3d_amd import spconv
        assert isinstance(enc, spconv.SparseConvTensor)  # the proxy IS a SparseConvTensor: isinstance checks of downstream heads hold
        assert isinstance(enc, scn_unet._LazyEncoded) == lazy
        assert ("features" in enc.__dict__) == (not lazy)  # nothing computed until it is read
        n = int(enc.n_dev) if enc.n_dev is not None else enc.features.shape[0]
        got[lazy] = (enc.indices[:n].clone(), enc.features[:n].clone())
        if lazy:  # invalidate(): the next read computes it again (a replayed graph's proxy after the inputs changed)
            enc.invalidate()
            assert "features" not in enc.__dict__ and torch.equal(enc.features[:n], got[lazy][1])
    assert torch.equal(got[True][0], got[False][0]) and torch.equal(got[True][1], got[False][1]) and got[True][0].shape[0] > 0


def test_capacity_mode_batch_beyond_one_frames_voxel_cap(monkeypatch):
    """a batch with more points than ONE frame's voxel cap (max_voxel_num): capacity mode does not read the frame sizes on the host any more - the
    per-frame cap is checked on the device and reported with the rulebooks' overflow flags.  No frame over the cap: the capacity frame stands
    (one pass, no host read); a frame over it: the frame is run again on host-side counts, which caps EACH frame as the reference's dataloader
    does - identical to the host-count mode either way"""
    import lidarseg3d_amd as L
    from lidarseg3d_amd import detectors, models_cfg
    cfg = synth.NUSC
    frames = [synth.lidar_frame(64, seed=41, **cfg), synth.lidar_frame(36, seed=42, **cfg)]
    pts = torch.from_numpy(np.concatenate([np.concatenate([np.full((f.shape[0], 1), b, np.float32), f], 1) for b, f in enumerate(frames)]))
    for cap, reruns in ((70, 0), (40, 1)):  # 100 points in the batch; frame 0 has ~60 voxels
        mcfg = models_cfg.sdseg3d()
        mcfg["backbone"]["model_cfg"] = dict(SCALING_RATIO=1)
        mcfg["point_head"]["model_cfg"]["CONV_IN_DIM"] = 16
        mcfg["voxel_generator"]["max_voxel_num"] = cap
        model = L.build_detector(mcfg, train_cfg=None, test_cfg={}).eval()
        shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.random_state_dict(shapes, 7).items()})
        bb = model.backbone
        orig_caps = bb._capacities
        monkeypatch.setattr(bb, "_capacities", lambda n, b, sh: [min(w, 6 * n) for w in orig_caps(n, b, sh)])
        seen, reads = [], []
        orig_fc = bb._forward_capacity
        monkeypatch.setattr(bb, "_forward_capacity", lambda *a, **k: (seen.append(1), orig_fc(*a, **k))[1])
        orig_vi = detectors._voxel_inputs
        monkeypatch.setattr(detectors, "_voxel_inputs", lambda e, c, capacity=False: (reads.append(capacity), orig_vi(e, c, capacity))[1])

        def run(capacity):
            monkeypatch.setattr(detectors, "CAPACITY_MODE", capacity)
            with torch.no_grad():
                ret = model(dict(points=pts, batch_size=2), return_loss=False)
            return model.point_head.forward_ret_dict["out_logits"].clone(), [r["pred_point_sem_labels"].clone() for r in ret]
        want, wl = run(False)
        del seen[:], reads[:]
        got, gl = run(True)
        assert torch.equal(got, want) and all(torch.equal(a, b) for a, b in zip(gl, wl))
        assert len(seen) == 1 and reads == [True] + [False] * reruns, (cap, reads)
        monkeypatch.undo()


@pytest.mark.parametrize("n,bits", [(1, 8), (255, 8), (2049, 13), (5000, 20), (4097, 31)])
def test_radix_sort_is_a_stable_argsort(n, bits):
    rng = np.random.default_rng(n)
    keys = rng.integers(0, 1 << bits, size=n, dtype=np.int64).astype(np.uint32)
    keys[: n // 3] = keys[0]  # many ties: stability matters
    perm = ops.radix_argsort(torch.from_numpy(keys.astype(np.int32)), bits).numpy()
    assert np.array_equal(perm, np.argsort(keys, kind="stable"))


@pytest.mark.parametrize("cls", [8, 23, 32])
def test_sffm_decoder_register_resident_form_other_token_counts(cls):
    """k_sffm_decoder_rt with L = 2 cls = 16 (one key block), 46 (Waymo: a partial second block - the P V steps beyond L are skipped) and 64
    (both blocks full) class embeddings, two ragged frames, against the layer-by-layer composition"""
    torch.manual_seed(cls)
    m = point_heads.SemanticFeatureFusionModule(64, 48, 64, d_model=96, nhead=4, num_decoder_layers=2, dim_feedforward=192).eval()
    n0, n1 = 150, 41
    x = torch.randn(n0 + n1, 64)
    e1, e2 = torch.randn(2, 48, cls, 1), torch.randn(2, 64, cls, 1)
    bidx = torch.cat([torch.zeros(n0), torch.ones(n1)])
    pts = torch.cat([bidx[:, None], torch.randn(n0 + n1, 3)], 1).contiguous()
    try:
        with torch.no_grad():
            point_heads.set_fused_sffm(False)
            ref = m(x, e1, e2, bidx, 2, points=pts)
            point_heads.set_fused_sffm(True)
            ops.set_precision("bf16x6")
            planes = m(x, e1, e2, bidx, 2, points=pts)
    finally:
        point_heads.set_fused_sffm(True)
        ops.set_precision("f32")
    np.testing.assert_allclose(planes.numpy(), ref.numpy(), rtol=0, atol=2e-5)


def test_fused_sffm_decoder_equals_layer_by_layer_and_oracle():
    """ls3d_sffm_decoder (the point side of the SF-Phase decoder as one kernel) against the layer-by-layer composition of the same
    module and against the oracle's SFFM (pinned to the reference class), two ragged frames: a 128-point tile straddles the frame
    boundary, the last tile is partial"""
    torch.manual_seed(3)
    m = point_heads.SemanticFeatureFusionModule(64, 48, 64, d_model=96, nhead=4, num_decoder_layers=3, dim_feedforward=192).eval()
    n0, n1, cls = 200, 77, 17
    x = torch.randn(n0 + n1, 64)
    e1, e2 = torch.randn(2, 48, cls, 1), torch.randn(2, 64, cls, 1)
    bidx = torch.cat([torch.zeros(n0), torch.ones(n1)])
    pts = torch.cat([bidx[:, None], torch.randn(n0 + n1, 3)], 1).contiguous()
    calls = []
    orig = ops.sffm_decoder
    ops.sffm_decoder = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        with torch.no_grad():
            fused = m(x, e1, e2, bidx, 2, points=pts)
            point_heads.set_fused_sffm(False)
            ref = m(x, e1, e2, bidx, 2, points=pts)
    finally:
        point_heads.set_fused_sffm(True)
        ops.sffm_decoder = orig
    assert len(calls) == 1
    np.testing.assert_allclose(fused.numpy(), ref.numpy(), rtol=0, atol=2e-5)
    try:  # the fused kernel's GEMMs on the exact 3-plane bf16 split (gemm_products = 6: what the 3-plane precisions select)
        ops.set_precision("bf16x6")
        with torch.no_grad():
            planes = m(x, e1, e2, bidx, 2, points=pts)
    finally:
        ops.set_precision("f32")
    assert not torch.equal(planes, fused)  # another arithmetic ran ...
    np.testing.assert_allclose(planes.numpy(), ref.numpy(), rtol=0, atol=2e-5)  # ... and it is f32-grade (LayerNorm'd outputs: unit scale)
    try:  # the other attention arithmetics of the fused kernel
        with torch.no_grad():
            ops.set_sffm_attention("valu")
            np.testing.assert_allclose(m(x, e1, e2, bidx, 2, points=pts).numpy(), ref.numpy(), rtol=0, atol=2e-5)
            ops.set_sffm_attention("bf16")
            b16 = m(x, e1, e2, bidx, 2, points=pts)
            ops.set_sffm_attention("fp8")
            f8 = m(x, e1, e2, bidx, 2, points=pts)
    finally:
        ops.set_sffm_attention("f32")
    # bf16 QK^T / PV operands (8 mantissa bits), f32 accumulation and softmax; outputs are LayerNorm'd (unit scale)
    assert float((b16 - ref).abs().max()) <= 3e-2 and float((b16 - ref).pow(2).mean().sqrt()) <= 4e-3
    # e4m3 operands (3 mantissa bits): a coarse but usable attention
    print("fp8 attention: max %.3g rms %.3g" % (float((f8 - ref).abs().max()), float((f8 - ref).pow(2).mean().sqrt())))
    assert float((f8 - ref).abs().max()) <= 0.5 and float((f8 - ref).pow(2).mean().sqrt()) <= 6e-2
    sd = {k: v.detach() for k, v in m.state_dict().items()}
    want = orc.sffm(sd, "", x, e1, e2, bidx, 2, 4)
    np.testing.assert_allclose(fused.numpy(), want.numpy(), rtol=0, atol=5e-5)


def _slot_order_mean(feats, ids, n_seg):
    """serial f32 restatement: rows of a segment added in input (= slot) order, then divided by the count"""
    out = np.zeros((n_seg, feats.shape[1]), np.float32)
    cnt = np.zeros(n_seg, np.int64)
    acc = np.zeros((n_seg, feats.shape[1]), np.float32)
    for i, s in enumerate(ids):
        if 0 <= s < n_seg:
            acc[s] = acc[s] + feats[i]
            cnt[s] += 1
    nz = cnt > 0
    out[nz] = acc[nz] / cnt[nz, None].astype(np.float32)
    return out


def test_points_in_voxel_mean_is_the_slot_order_sum_bit_for_bit():
    """DynamicScatter mean / segment mean without float atomics: bit-equal to the serial slot-order f32 sum of the
    reference's padded tensor (scatter_points.py:85-98) and bit-reproducible"""
    g = golden("voxelize_nusc.npz")
    gs = orc.grid_size(g["voxel_size"], g["pc_range"])
    shape = [int(gs[2]), int(gs[1]), int(gs[0])]
    pts, coors = torch.from_numpy(g["points"]), torch.from_numpy(g["cpp_dyn_coors"])
    f, vc, p2v, nv = ops.dynamic_scatter(pts, coors, shape, "mean")
    V = int(nv)
    want = _slot_order_mean(g["points"], p2v.numpy(), V)
    assert np.array_equal(f[:V].numpy(), want)
    f2, _, _, _ = ops.dynamic_scatter(pts, coors, shape, "mean")
    assert torch.equal(f2[:V], f[:V])
    # a voxel with many points of very different magnitudes: the order of the additions matters, and it is the slot order
    rng = np.random.default_rng(4)
    big = (rng.normal(size=(3000, 4)) * np.exp(rng.normal(size=(3000, 4)) * 6)).astype(np.float32)
    idx = rng.integers(0, 7, size=3000).astype(np.int64)
    idx[::11] = 9  # segment 7, 8 empty, 9 sparse
    got = ops.segment_reduce(torch.from_numpy(big), torch.from_numpy(idx), 10, "mean").numpy()
    assert np.array_equal(got, _slot_order_mean(big, idx, 10))


def test_other_backbones_conv_calls_replayed_on_our_spconv():
    """every sparse-conv call of the reference's Cylinder3D_Asymm_3d_spconv and SpMiddleResNetFHD (fixtures from the reference's
    files over the oracle shim): asymmetric kernels, several kernel shapes under one indice_key, stride (2,2,1), inverse, bias"""
    from tests import f4_cases
    seen, _ = f4_cases.replay("f4_cylinder3d_asymm.npz", "cpu")
    assert {(0, (1, 3, 3), (1, 1, 1)), (0, (3, 1, 3), (1, 1, 1)), (0, (3, 1, 1), (1, 1, 1)), (1, (3, 3, 3), (2, 2, 1)), (2, (3, 3, 3), (1, 1, 1))} <= seen
    seen, _ = f4_cases.replay("f4_spmiddleresnetfhd.npz", "cpu")
    assert (1, (3, 1, 1), (2, 1, 1)) in seen


@pytest.mark.parametrize("case", ["spmiddleresnetfhd", "unetcylinder3d", "cylinder3d_v2p", "cylinder3d_asymm", "reader_cylinder3d", "reader_polarnet", "tta_merge",
                                  "dynamic_point_to_voxel"])
def test_other_backbones_and_dynamic_readers_as_registered_modules(case):
    """SURVEY 8f rank 4 as components: SpMiddleResNetFHD, UNetCylinder3D, Cylinder3D_Asymm_3d_spconv(_v2p), the PolarNet / Cylinder3D dynamic
    readers built through the registry with the reference's state_dict (strict) against module-level fixtures of the reference's own files;
    the TTA merge of predict() against the reference head's; ls3d_dynamic_point_to_voxel_* against the reference's C++ (tests/f4_module_cases.py)"""
    from tests import f4_module_cases
    getattr(f4_module_cases, case)(torch.device("cpu"))


# ------------------------------------------------------------------------------------------------ fused segmentation loss, round 3
@pytest.mark.parametrize("P,C,ignore,case", [(3000, 17, 0, "mixed"), (5000, 23, 0, "absent_classes"), (700, 5, 255, "no_ignored"), (1500, 17, 0, "one_valid"),
                                              (1025, 32, 0, "mixed")])
def test_fused_seg_loss_equals_the_torch_restatement(P, C, ignore, case):
    """ls3d_seg_loss_forward / _backward (cross entropy with an ignored label + Lovasz-Softmax over the classes present, det3d/core/utils/
    loss_utils.py:217-291) against the class-by-class torch restatement of losses.py (itself pinned to the reference's loss_utils in
    test_losses_equal_reference): values, gradients for arbitrary upstream weights, ties in the errors, absent classes, bit-reproducible"""
    from lidarseg3d_amd import losses
    rng = np.random.default_rng(P + C)
    logits = torch.from_numpy((rng.normal(size=(P, C)) * 3).astype(np.float32))
    labels = torch.from_numpy(rng.integers(0, C, size=P).astype(np.int64))
    if case == "absent_classes":
        labels[labels % 3 == 1] = 2
    if case == "one_valid":
        labels[:] = ignore
        labels[77] = 3
    if case == "mixed":
        labels[rng.uniform(size=P) < 0.3] = ignore
        logits[100:140] = logits[100]          # identical rows: tied errors
        logits[200:210] = 0.0                  # uniform softmax
    res = {}
    for name, fn in (("fused", losses.seg_loss), ("torch", losses.seg_loss_torch)):
        x = logits.clone().requires_grad_(True)
        ce, lv = fn(x, labels, ignore)
        (0.7 * ce + 1.9 * lv).backward()
        res[name] = (float(ce), float(lv), x.grad.clone())
    f, t = res["fused"], res["torch"]
    assert abs(f[0] - t[0]) <= 2e-6 * max(1.0, abs(t[0])) and abs(f[1] - t[1]) <= 5e-6 * max(1.0, abs(t[1])), (f[:2], t[:2])
    scale = float(t[2].abs().max())
    tied = torch.zeros(P, dtype=torch.bool)
    if case == "mixed":
        tied[100:140] = True  # identical rows: equal errors, whose order in the sort (and with it which of them gets which Lovasz
        tied[200:210] = True  # increment) is the sort's choice - a different subgradient; their sum per class must still agree
        assert float((f[2][100:140].sum(0) - t[2][100:140].sum(0)).abs().max()) <= 2e-5 * scale
        assert float((f[2][200:210].sum(0) - t[2][200:210].sum(0)).abs().max()) <= 2e-5 * scale
    assert float((f[2] - t[2])[~tied].abs().max()) <= 2e-5 * scale + 1e-9, (float((f[2] - t[2])[~tied].abs().max()), scale)
    assert float(f[2][labels == ignore].abs().max() if (labels == ignore).any() else 0.0) == 0.0  # ignored points get no gradient
    x = logits.clone().requires_grad_(True)
    ce2, lv2 = losses.seg_loss(x, labels, ignore)
    (0.7 * ce2 + 1.9 * lv2).backward()
    assert float(ce2) == f[0] and float(lv2) == f[1] and torch.equal(x.grad, f[2])
    # only one of the two losses used downstream
    x = logits.clone().requires_grad_(True)
    losses.seg_loss(x, labels, ignore)[1].backward()
    y = logits.clone().requires_grad_(True)
    losses.seg_loss_torch(y, labels, ignore)[1].backward()
    assert float((x.grad - y.grad)[~tied].abs().max()) <= 2e-5 * float(y.grad.abs().max()) + 1e-9


def test_fused_seg_loss_vs_the_reference_fixture():
    """the fused kernels against tests/golden/seg_loss.npz: value and gradient of CE + Lovasz-Softmax as the REFERENCE's
    det3d/core/utils/loss_utils.py computes them on 700 points (fixture made by tests/golden/make_golden.py from the reference's file)"""
    from lidarseg3d_amd import losses
    from tests.util import golden
    g = golden("seg_loss.npz")
    lg = torch.from_numpy(g["logits"]).requires_grad_(True)
    ce, lv = losses.seg_loss(lg, torch.from_numpy(g["labels"]), int(g["ignore"]))
    assert isinstance(ce.grad_fn, type(losses._FusedSegLoss.apply(lg.detach().requires_grad_(True), torch.from_numpy(g["labels"]), int(g["ignore"]))[0].grad_fn))
    assert abs(float(ce) - float(g["ce"])) <= 2e-6 and abs(float(lv) - float(g["lovasz"])) <= 2e-6
    (ce + lv).backward()
    np.testing.assert_allclose(lg.grad.numpy(), g["grad"], rtol=0, atol=2e-7)


@pytest.mark.parametrize("n,c", [(1000, 96), (257, 64), (77, 192), (300, 256), (9, 4)])
def test_layer_norm_forward_backward_vs_torch(n, c):
    """ls3d_layer_norm_forward / _backward (the training step's LayerNorms) against torch.nn.functional.layer_norm and its autograd;
    the column sums d gamma / d beta are reduced in a fixed order: bit-reproducible"""
    rng = np.random.default_rng(n + c)
    x = torch.from_numpy((rng.normal(size=(n, c)) * 2 + 0.5).astype(np.float32))
    g, b = torch.from_numpy(rng.uniform(0.5, 1.5, c).astype(np.float32)), torch.from_numpy(rng.normal(size=c).astype(np.float32))
    dy = torch.from_numpy(rng.normal(size=(n, c)).astype(np.float32))
    xr, gr, br = x.clone().requires_grad_(True), g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    want = torch.nn.functional.layer_norm(xr, (c,), gr, br, 1e-5)
    want.backward(dy)
    xs, gs, bs = x.clone().requires_grad_(True), g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    got = ops._LayerNormFn.apply(xs, gs, bs, 1e-5)
    got.backward(dy)
    assert float((got - want).abs().max()) <= 2e-6 * float(want.abs().max())
    assert float((xs.grad - xr.grad).abs().max()) <= 5e-6 * float(xr.grad.abs().max())
    assert float((gs.grad - gr.grad).abs().max()) <= 2e-5 * float(gr.grad.abs().max()) and float((bs.grad - br.grad).abs().max()) <= 2e-5 * float(br.grad.abs().max())
    xs2, gs2, bs2 = x.clone().requires_grad_(True), g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ops._LayerNormFn.apply(xs2, gs2, bs2, 1e-5).backward(dy)
    assert torch.equal(xs2.grad, xs.grad) and torch.equal(gs2.grad, gs.grad) and torch.equal(bs2.grad, bs.grad)


def test_pair_list_entry_points_argument_and_workspace_errors():
    """ls3d_spconv_pairs / ls3d_spconv_wgrad_on_pairs: status codes instead of faults (include/ls3d.h: LS3D_ERR_ARG -1, LS3D_ERR_WORKSPACE -4),
    empty tables are fine, and the two-step form equals the one-call form on a table with an empty offset and a ragged tail"""
    import ctypes
    L = ops._L()
    rng = np.random.default_rng(5)
    n_in, n_out, kvol, cin, cout = 70, 53, 4, 32, 32
    tbl = torch.from_numpy(rng.integers(-1, n_in, size=(n_out, kvol)).astype(np.int32))
    tbl[:, 2] = -1  # an offset without pairs
    x, go = torch.from_numpy(rng.normal(size=(n_in, cin)).astype(np.float32)), torch.from_numpy(rng.normal(size=(n_out, cout)).astype(np.float32))
    nbytes = int(L.ls3d_spconv_pairs_bytes(kvol, n_out))
    pairs = torch.empty(nbytes, dtype=torch.uint8)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    assert L.ls3d_spconv_pairs(p(tbl), None, n_out, None, kvol, p(pairs), ctypes.c_size_t(nbytes - 1), None) == -4
    assert L.ls3d_spconv_pairs(p(tbl), None, n_out, None, kvol, ctypes.c_void_p(pairs.data_ptr() + 4), ctypes.c_size_t(nbytes), None) == -4  # 16-byte alignment
    assert L.ls3d_spconv_pairs(None, None, n_out, None, kvol, p(pairs), ctypes.c_size_t(nbytes), None) == -1
    assert L.ls3d_spconv_pairs(p(tbl), None, 0, None, kvol, None, ctypes.c_size_t(0), None) == 0  # empty table: nothing to build
    assert L.ls3d_spconv_pairs(p(tbl), None, n_out, None, kvol, p(pairs), ctypes.c_size_t(nbytes), None) == 0
    ws_bytes = int(L.ls3d_spconv_wgrad_workspace_bytes(kvol, cin, cout, n_out))
    ws, gw = torch.empty(ws_bytes, dtype=torch.uint8), torch.empty(kvol, cin, cout)
    args = lambda w, nb, prod=0: (p(x), cin, p(go), cout, p(pairs), kvol, cin, cout, n_out, prod, p(w), ctypes.c_size_t(nb), p(gw), None)
    assert L.ls3d_spconv_wgrad_on_pairs(*args(ws, 16)) == -4
    assert L.ls3d_spconv_wgrad_on_pairs(*args(ws, ws_bytes, prod=3)) == -1  # products: 0 | 6 | 8
    assert L.ls3d_spconv_wgrad_on_pairs(*args(ws, ws_bytes)) == 0
    assert torch.equal(gw, ops.spconv_wgrad(x, go, tbl, None, cin, cout, products=0))
    assert float(gw[2].abs().max()) == 0.0
    want = torch.zeros(kvol, cin, cout, dtype=torch.float64)
    for k in range(kvol):
        o = torch.nonzero(tbl[:, k] >= 0)[:, 0]
        want[k] = x[tbl[o, k].long()].double().t() @ go[o].double()
    assert float((gw.double() - want).abs().max()) < 1e-4


def test_deferred_points_pruned_scan_equals_brute_force():
    """k_devox_hard (points the shell search defers: neighbours metres away) prunes with the coarse grid - an upper bound from a sample of
    the centres, then only the cells whose box is within it.  Hundreds of such points per frame - far outside the range on every side,
    inside it but in empty space, exactly on lattice positions (equal distances: the smallest index must win) - against the brute-force
    kernel, bit for bit (indices, weights, rows)"""
    g = golden("head_mseg3d_nusc.npz")
    coords, ctr, feat = torch.from_numpy(g["coords"]), torch.from_numpy(g["conv_point_coords"]), torch.from_numpy(g["conv_point_features"])
    cfg = synth.NUSC
    lo, hi = np.float32(cfg["pc_range"][:3]), np.float32(cfg["pc_range"][3:])
    rng = np.random.default_rng(12)
    far = []
    for b in (0, 1):
        p = rng.uniform(lo - 60.0, hi + 60.0, size=(150, 3)).astype(np.float32)          # all around, mostly outside
        q = rng.uniform(lo, hi, size=(100, 3)).astype(np.float32); q[:, 2] = hi[2] - 0.05    # inside the range, above everything
        c = ctr[ctr[:, 0] == b][:: max(1, int((ctr[:, 0] == b).sum()) // 40), 1:4].numpy().copy()
        c[:, 0] += np.float32(40.0)                                                           # lattice-aligned offsets: ties between centres
        far.append(np.concatenate([np.full((len(p) + len(q) + len(c), 1), b, np.float32), np.concatenate([p, q, c])], 1))
    pts = torch.from_numpy(np.concatenate(far)).contiguous()
    pt_off, vx_off = ops.frame_offsets(pts[:, 0], 2), ops.frame_offsets(ctr[:, 0], 2)
    assert int((vx_off[1:] - vx_off[:-1]).min()) >= 2048  # the pruned path, not the scan of small frames
    a, ia = ops.devoxelize_grid(pts, pt_off, coords, ctr, vx_off, 2, cfg["voxel_size"], cfg["pc_range"], feat, return_idx=True)
    b, ib = ops.devoxelize(pts, pt_off, ctr, vx_off, 2, pts.shape[0], feat, return_idx=True)
    assert torch.equal(ia, ib) and torch.equal(a, b)
    idx, w = ops.devoxelize_grid(pts, pt_off, coords, ctr, vx_off, 2, cfg["voxel_size"], cfg["pc_range"], None)
    assert torch.equal(idx, ia)
    want, widx = orc.three_interpolate_wrap(pts, ctr, feat, 2, return_idx=True)
    assert np.array_equal(ia.numpy(), np.concatenate(widx))


@pytest.mark.parametrize("cls,layers", [(17, 3), (23, 6), (32, 1)])
def test_sffm_memory_side_in_one_launch_equals_layer_by_layer(cls, layers):
    """ls3d_sffm_memory (self-attention + norm1 of the class embeddings and every layer's k / v projection, one workgroup per frame)
    against the layer-by-layer composition on the GEMM / attention-core kernels: the kv tensor itself, then the whole SF-Phase output and
    the oracle's SFFM.  L = 2 * cls tokens: 34 (nuScenes), 46 (Waymo), 64 (the kernel's limit)"""
    torch.manual_seed(cls + layers)
    m = point_heads.SemanticFeatureFusionModule(64, 48, 64, d_model=96, nhead=4, num_decoder_layers=layers, dim_feedforward=192).eval()
    with torch.no_grad():
        for l in m.decoder.layers:  # non-trivial LayerNorm parameters and biases
            l.norm1.weight.uniform_(0.5, 1.5); l.norm1.bias.normal_(0, 0.2)
            l.self_attn.in_proj_bias.normal_(0, 0.2); l.crossocr_attn.k_proj.bias.normal_(0, 0.2)
    n0, n1, B, L, E = 150, 61, 2, 2 * cls, 96
    x = torch.randn(n0 + n1, 64)
    e1, e2 = torch.randn(B, 48, cls, 1), torch.randn(B, 64, cls, 1)
    bidx = torch.cat([torch.zeros(n0), torch.ones(n1)])
    pts = torch.cat([bidx[:, None], torch.randn(n0 + n1, 3)], 1).contiguous()
    pk = m.packed()
    assert "memory" in pk
    mem = torch.randn(B * L, E)
    kv, mem_out = ops.sffm_memory(mem, B, L, pk["memory"], return_memory=True)
    kvs, mf = [], mem
    for lp in pk["layers"]:
        att = ops.mha_core(point_heads._lin(mf, lp["sa_qkv"]), B, L, E, 4)
        mf = point_heads._lin(att, lp["sa_out"], res=mf, ln=lp["n1"])
        kvs.append(point_heads._lin(mf, lp["k"]).view(B, L, E).permute(0, 2, 1))
        kvs.append(point_heads._lin(mf, lp["v"]).view(B, L, E).permute(0, 2, 1))
    want_kv = torch.stack(kvs).contiguous()
    assert kv.shape == want_kv.shape
    np.testing.assert_allclose(kv.numpy(), want_kv.numpy(), rtol=0, atol=2e-5)
    np.testing.assert_allclose(mem_out.numpy(), mf.numpy(), rtol=0, atol=2e-5)
    try:
        with torch.no_grad():
            ref = m(x, e1, e2, bidx, B, points=pts)
            point_heads.set_fused_sffm_memory(True)
            got = m(x, e1, e2, bidx, B, points=pts)
    finally:
        point_heads.set_fused_sffm_memory(False)
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=0, atol=2e-5)
    sd = {k: v.detach() for k, v in m.state_dict().items()}
    np.testing.assert_allclose(got.numpy(), orc.sffm(sd, "", x, e1, e2, bidx, B, 4).numpy(), rtol=0, atol=5e-5)
    # other shapes are declined, not faulted
    assert ops._L().ls3d_sffm_memory(ops._ptr(mem), B, 65, E, 4, 0, None, ops._ptr(kv), None, None) == -3


def test_deferred_points_on_a_grid_whose_rows_are_not_whole_bitmap_words():
    """the same pruned scan when the coarse grid is 25 x 19 x 5 cells: occupancy words straddle rows and planes (the NUSC grid has 128 cells per
    row = 4 whole words), the last coarse cell of every axis holds more fine cells than the others"""
    rng = np.random.default_rng(3)
    vs, rng_lo = [0.1, 0.1, 0.2], [-10.0, -7.9, -3.0]
    grid = [203, 157, 21]  # fine cells x, y, z: 25 x 19 x 5 coarse cells of 8 x 8 x 4, with remainders on every axis
    pc_range = rng_lo + [rng_lo[a] + vs[a] * grid[a] for a in range(3)]
    per = 3000
    coords = []
    for b in (0, 1):
        c = np.unique(np.stack([rng.integers(0, grid[2], per), rng.integers(0, grid[1], per), rng.integers(0, grid[0], per)], 1), axis=0)
        coords.append(np.concatenate([np.full((len(c), 1), b), c], 1))
    coords = torch.from_numpy(np.concatenate(coords).astype(np.int32))
    ctr = ops.voxel_centers(coords, vs, pc_range)
    feat = torch.from_numpy(rng.normal(size=(coords.shape[0], 8)).astype(np.float32))
    lo, hi = np.float32(pc_range[:3]), np.float32(pc_range[3:])
    pts = []
    for b in (0, 1):
        p = rng.uniform(lo - 25.0, hi + 25.0, size=(200, 3)).astype(np.float32)
        q = rng.uniform(lo, hi, size=(100, 3)).astype(np.float32)
        pts.append(np.concatenate([np.full((300, 1), b, np.float32), np.concatenate([p, q])], 1))
    pts = torch.from_numpy(np.concatenate(pts)).contiguous()
    pt_off, vx_off = ops.frame_offsets(pts[:, 0], 2), ops.frame_offsets(ctr[:, 0], 2)
    assert int((vx_off[1:] - vx_off[:-1]).min()) >= 2048
    a, ia = ops.devoxelize_grid(pts, pt_off, coords, ctr, vx_off, 2, vs, pc_range, feat, return_idx=True)
    b, ib = ops.devoxelize(pts, pt_off, ctr, vx_off, 2, pts.shape[0], feat, return_idx=True)
    assert torch.equal(ia, ib) and torch.equal(a, b)
    want, widx = orc.three_interpolate_wrap(pts, ctr, feat, 2, return_idx=True)
    assert np.array_equal(ia.numpy(), np.concatenate(widx))


@pytest.mark.parametrize("kind", ["sdseg3d", "mseg3d"])
def test_padding_rows_of_a_point_bucket_belong_to_no_frame(kind):
    """graph.FrameGraph(point_keys=...) / BucketedFrameGraph pad a frame up to its bucket with rows of batch index = batch_size far outside the
    range (graph.pad_rows).  The forward of the padded batch must give the real points bit-identical logits and labels, the voxelizer must
    reject the padding, the neighbour search must never visit it (ls3d_devoxelize_grid skips rows outside [0, batch)) and the per-point
    tail must find an empty frame for it (ops.frame_offsets' hidden entry): eager capacity-mode forward on the host emulation, 2 frames"""
    from lidarseg3d_amd import graph, models_cfg
    import lidarseg3d_amd as L
    cfg = models_cfg.sdseg3d() if kind == "sdseg3d" else models_cfg.mseg3d()
    if kind == "mseg3d":  # the narrow UNet of the emulation (16 / 32 / 64 / 64 channels): what is tested here sits in front of and behind it
        cfg["backbone"]["model_cfg"] = dict(cfg["backbone"].get("model_cfg", {}), SCALING_RATIO=1)
        cfg["point_head"]["model_cfg"] = dict(cfg["point_head"]["model_cfg"], VOXEL_IN_DIM=16)
    model = L.build_detector(cfg, train_cfg=None, test_cfg={}).eval()
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.random_state_dict(shapes, 5).items()})
    frames = [synth.lidar_frame(70, seed=1, **synth.NUSC), synth.lidar_frame(66, seed=2, **synth.NUSC)]  # small: the host emulation runs ~1 min per MSeg3D forward
    pts = torch.from_numpy(np.concatenate([np.concatenate([np.full((f.shape[0], 1), b, np.float32), f], 1) for b, f in enumerate(frames)]))
    ex = dict(points=pts, batch_size=2)
    if kind == "mseg3d":
        img, emb, cuv = synth.camera_inputs(pts.shape[0], seed=3, ncam=2, c_img=48, h=8, w=12, batch=2)
        ex.update(points_cuv=torch.from_numpy(cuv), image_features=torch.from_numpy(img), camera_semantic_embeddings=torch.from_numpy(emb))
    with torch.no_grad():
        want = model(dict(ex), return_loss=False)
    want_logits = model.point_head.forward_ret_dict["out_logits"].clone()
    padded = dict(ex)
    for k in ("points", "points_cuv"):
        if k in ex:
            padded[k] = torch.cat([ex[k], graph.pad_rows(k, ex[k], 37, 2)])
    assert float(padded["points"][-1, 0]) == 2.0 and float(padded["points"][-1, 1]) == graph.PAD_COORD
    with torch.no_grad():
        got = model(dict(padded), return_loss=False)
    logits = model.point_head.forward_ret_dict["out_logits"]
    assert logits.shape[0] == padded["points"].shape[0]
    assert torch.equal(logits[:pts.shape[0]], want_logits)
    assert len(got) == len(want) == 2
    for a, b in zip(got, want):
        assert torch.equal(a["pred_point_sem_labels"], b["pred_point_sem_labels"])
