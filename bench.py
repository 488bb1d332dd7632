#!/usr/bin/env python
"""bench.py — frames/s of the SDSeg3D segmentation forward path on MI355X (BASELINE.json configs[1]).

One "step" = one full forward of one synthetic 120 000-point nuScenes-style frame per GPU:
    points (resident in HBM) -> GPU hard voxelization -> TransVFE -> UNetSCN3D (37 sparse convs + 8 rulebooks)
    -> 3-NN devoxelization -> PointSegBatchlossHead -> per-point logits [N,17] -> argmax.
Weights are random-init of the reference architecture (no checkpoints offline); data is synthetic.
Multi-GPU: frames shard one-per-GPU, no data-path collective (scaling "weak"); the only collectives are the
timing barrier and the max-over-ranks of the elapsed time.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--points 120000] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline     — the dominant kernel (k_gather_gemm, the sparse-conv gather-GEMM): algorithmic pair-model bytes
                 sum_l P_l*(Cin+Cout)*4 over the summed HIP-event durations of those launches;
  cpu_baseline — the CPU oracle (oracle/ref.py, kind "port") timed on the host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

F32_MFMA_PEAK_TFLOPS = 157.3
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s peak


def build_model(dev, seed=5, kind="sdseg3d"):
    import lidarseg3d_amd as L
    from lidarseg3d_amd import models_cfg, synth
    cfg = models_cfg.mseg3d() if kind == "mseg3d" else models_cfg.sdseg3d()
    model = L.build_detector(cfg, train_cfg=None, test_cfg={}).eval()
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = {k: torch.from_numpy(v) for k, v in synth.random_state_dict(shapes, seed).items()}
    model.load_state_dict(sd)
    return model.to(dev), sd


class ConvTimer:
    """HIP-event brackets around every sparse-conv gather-GEMM launch (installed as ops.gather_gemm wrapper)."""

    def __init__(self, ops):
        self.ops, self.orig, self.orig_tile = ops, ops.gather_gemm, ops.tile_conv
        self.events, self.algo_bytes, self.flops, self.enabled = [], 0.0, 0.0, False
        self.pairs_cache = {}

    def install(self):
        def wrapped(x, w, tbl=None, **kw):
            if not self.enabled or tbl is None:
                return self.orig(x, w, tbl=tbl, **kw)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            out = self.orig(x, w, tbl=tbl, **kw)
            b.record()
            self.events.append((a, b))
            key = (tbl.data_ptr(), tbl.shape[0])
            if key not in self.pairs_cache:
                self.pairs_cache[key] = None  # filled after the timed region (needs a sync)
            self.meta.append((key, tbl, w.shape[1], kw.get("cout") or w.cout))
            return out
        def wrapped_tile(x, w, plan, **kw):
            if not self.enabled:
                return self.orig_tile(x, w, plan, **kw)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            out = self.orig_tile(x, w, plan, **kw)
            b.record()
            self.events.append((a, b))
            self.meta.append(((plan.tbl.data_ptr(), plan.tbl.shape[0]), plan.tbl, w.shape[1], kw.get("cout") or w.cout))
            return out
        self.meta = []
        self.ops.gather_gemm = wrapped
        self.ops.tile_conv = wrapped_tile
        # modules imported `ops` as a module, so they see the replacement
        return self

    def summarize(self):
        torch.cuda.synchronize()
        total_ms = sum(a.elapsed_time(b) for a, b in self.events)
        pairs = {}
        algo = flops = 0.0
        for key, tbl, cin, cout in self.meta:
            if key not in pairs:
                pairs[key] = int((tbl >= 0).sum().item())
            p = pairs[key]
            algo += p * (cin + cout) * 4.0
            flops += 2.0 * p * cin * cout
        n = max(len(self.events), 1)
        return dict(launches=len(self.events), total_ms=total_ms, avg_us=1e3 * total_ms / n, algo_bytes=algo, flops=flops)


def cpu_baseline_worker(n_points, seed, threads):
    """runs in a child process: the CPU oracle's full SDSeg3D forward on one frame, timed"""
    from lidarseg3d_amd import synth
    from oracle import ref as orc
    torch.set_num_threads(threads)
    orc.build_c()
    import lidarseg3d_amd as L
    from lidarseg3d_amd import models_cfg
    model = L.build_detector(models_cfg.sdseg3d(), train_cfg=None, test_cfg={})
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = {k: torch.from_numpy(v) for k, v in synth.random_state_dict(shapes, 5).items()}
    frame = synth.lidar_frame(n_points, seed=seed, **synth.NUSC)
    t0 = time.time()
    orc.sdseg3d_forward(sd, [frame], synth.NUSC["voxel_size"], synth.NUSC["pc_range"])
    print(json.dumps({"seconds": time.time() - t0}))


def cpu_baseline(n_points, seed, timeout_s=420):
    """the CPU oracle (oracle/ref.py, a port: the reference has no CPU forward) on ONE frame of the same workload,
    rank 0 / N=1 only.  Thread count capped: torch-CPU index_add_/mm of the restatement does not scale past a few
    dozen threads (on a 256-thread host the uncapped run is >10x slower)."""
    import subprocess
    cores = os.cpu_count() or 1
    threads = min(cores, 32)
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads))
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", str(n_points), str(seed), str(threads)],
                           env=env, capture_output=True, text=True, timeout=timeout_s, cwd=ROOT)
        dt = json.loads(r.stdout.strip().splitlines()[-1])["seconds"]
    except Exception as e:  # never let the baseline take the bench down
        return dict(value=None, unit="frames/s", cores=threads, kind="port", sample="failed: %r" % (e,))
    return dict(value=1.0 / dt, unit="frames/s", cores=threads, kind="port",
                sample="1 frame of %d points, full SDSeg3D forward incl. CPU voxelization, %.1f s on %d of %d host threads; "
                       "oracle/ref.py (torch-CPU gather-mm-scatter spconv restatement, OpenMP C exact 3-NN)"
                       % (n_points, dt, threads, cores))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--points", type=int, default=120000)
    ap.add_argument("--cpu-points", type=int, default=None, help="points of the CPU-baseline frame (default: --points)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", choices=["f32", "bf16x8", "bf16x6", "bf16x3"], default="f32",
                    help="gather-GEMM arithmetic: f32 = exact f32 MFMA (default, the parity configuration); bf16x3 = split-bf16 "
                         "(3 bf16 MFMAs per product, ~1e-5 relative error)")
    ap.add_argument("--no-fast-mode", action="store_true", help="skip the extra bf16x3 measurement")
    ap.add_argument("--streams", type=int, default=1,
                    help="throughput mode: a step = this many frames, each a batch of one on its own HIP stream (the reader / head / "
                         "rulebook kernels of one frame run beside the conv stack of another); SDSeg3D only")
    ap.add_argument("--frames-per-step", type=int, default=1,
                    help="frames collated into one forward per GPU per step (1 = the reference's --speed_test batch size)")
    ap.add_argument("--row-order", choices=["mask", "none"], default="mask")
    ap.add_argument("--model", choices=["sdseg3d", "mseg3d"], default="sdseg3d",
                    help="sdseg3d = BASELINE configs[1] (the metric's config); mseg3d = configs[2] (LiDAR + 6-camera features)")
    ap.add_argument("--cpu-baseline-worker", nargs=3, default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_worker:
        cpu_baseline_worker(*[int(v) for v in args.cpu_baseline_worker])
        return

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("LS3D_BENCH_FORCE_DIST") == "1":  # the env knob exercises the RCCL path on a 1-GPU box
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", init_method="env://")

    from lidarseg3d_amd import ops, synth
    ops.set_precision(args.precision)
    ops.set_row_order(args.row_order)
    model, sd = build_model(dev, kind=args.model)
    B = max(1, args.frames_per_step)
    frames = [synth.lidar_frame(args.points, seed=100 + rank * B + b, **synth.NUSC) for b in range(B)]
    pts = torch.from_numpy(np.concatenate([np.concatenate([np.full((f.shape[0], 1), b, np.float32), f], 1)
                                           for b, f in enumerate(frames)])).to(dev)
    extra = {}
    if args.model == "mseg3d":  # HRNet-w18 feature maps of 6 cameras at 1/4 resolution + camera class embeddings (inputs of the path)
        img, emb, cuv = synth.camera_inputs(args.points * B, seed=100 + rank, ncam=6, c_img=48, h=160, w=240, batch=B)
        extra = dict(points_cuv=torch.from_numpy(cuv).to(dev), image_features=torch.from_numpy(img).to(dev),
                     camera_semantic_embeddings=torch.from_numpy(emb).to(dev))
    timer = ConvTimer(ops).install()

    S = max(1, args.streams)
    TP = 2 if (args.model == "sdseg3d" and B == 1 and S == 1 and world == 1 and not args.no_fast_mode) else 0  # extra throughput legs
    if S > 1 or TP:
        assert args.model == "sdseg3d" and B == 1, "--streams: SDSeg3D, one frame per stream"
        nS = max(S, TP)
        sframes = [synth.lidar_frame(args.points, seed=100 + rank * nS + i, **synth.NUSC) for i in range(nS)]
        spts = [torch.from_numpy(np.concatenate([np.zeros((f.shape[0], 1), np.float32), f], 1)).to(dev) for f in sframes]
        streams = [torch.cuda.Stream(dev) for _ in range(nS)]

    def step(n_streams=None):
        n_streams = S if n_streams is None else n_streams
        if n_streams == 1:
            ret = model(dict(points=pts, batch_size=B, **extra), return_loss=False)
            return ret[0]["pred_point_sem_labels"]
        cur = torch.cuda.current_stream(dev)
        labels = None
        for st, p in zip(streams[:n_streams], spts[:n_streams]):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                labels = model(dict(points=p, batch_size=1), return_loss=False)[0]["pred_point_sem_labels"]
        for st in streams[:n_streams]:
            cur.wait_stream(st)
        return labels

    with torch.no_grad():
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        timer.enabled = True
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            labels = step()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        timer.enabled = False
    if dist is not None:
        dist.barrier()
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    conv = timer.summarize()
    alt = {}
    if args.precision == "f32" and not args.no_fast_mode and world == 1:
        # the same step in the split-bf16 arithmetics: reported beside, never as, `value`
        ref_logits = model.point_head.forward_ret_dict["out_logits"].clone()
        for prec, label in (("bf16x8", "bf16x8 (exact 3-way bf16 split, 8 of 9 plane products: f32-grade; SubM layers on the tile-halo kernel)"),
                            ("bf16x6", "bf16x6 (exact 3-way bf16 split, 6 partial products per f32 product: f32-grade results)"),
                            ("bf16x3", "bf16x3 (split-bf16 MFMA, f32 accumulate)")):
            ops.set_precision(prec)
            timer2 = ConvTimer(ops)
            timer2.orig, timer2.orig_tile = timer.orig, timer.orig_tile
            timer2.install()
            with torch.no_grad():
                for _ in range(args.warmup):
                    step()
                torch.cuda.synchronize()
                timer2.enabled = True
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    step()
                torch.cuda.synchronize()
                el2 = time.perf_counter() - t0
                timer2.enabled = False
            c2 = timer2.summarize()
            got = model.point_head.forward_ret_dict["out_logits"]
            alt[prec] = dict(precision=label, value=B * S * args.steps / el2, ms_per_step=1e3 * el2 / args.steps,
                             sparse_conv_ms_per_frame=c2["total_ms"] / max(args.steps * S, 1),
                             roofline_frac=(c2["algo_bytes"] / (c2["total_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if c2["total_ms"] > 0 else 0.0,
                             max_rel_logit_diff_vs_f32=float((got - ref_logits).abs().max() / ref_logits.abs().max()),
                             argmax_agreement_vs_f32=float((got.argmax(1) == ref_logits.argmax(1)).float().mean()))
        ops.set_precision("f32")
    fast = alt.get("bf16x3")
    throughput = None
    if TP:
        # two frames in flight per GPU, one HIP stream each: the kernels of the two frames fill each other's idle slots (launch
        # tails, latency-bound reader / head kernels beside the matrix-bound convs).  Reported beside `value`: per-kernel event
        # timings (and with them the roofline line) are not defined while kernels of two streams overlap.
        throughput = {"streams": TP, "frames_in_flight": TP}
        with torch.no_grad():
            for prec in ("f32", "bf16x8", "bf16x6", "bf16x3"):
                ops.set_precision(prec)
                for _ in range(max(2, args.warmup // 2)):
                    step(TP)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    step(TP)
                torch.cuda.synchronize()
                throughput[prec + "_frames_per_s"] = TP * args.steps / (time.perf_counter() - t0)
        ops.set_precision("f32")
    frd = model.point_head.forward_ret_dict
    V = int((frd["conv_logits"] if "conv_logits" in frd else frd["voxel_logits"]).shape[0])
    if rank == 0:
        ms = 1e3 * elapsed / args.steps
        achieved = conv["algo_bytes"] / (conv["total_ms"] * 1e-3) / 1e9 if conv["total_ms"] > 0 else 0.0
        out = {
            "metric": "frames/sec, SDSeg3D forward, 120k-pt nuScenes-style frame",
            "value": world * B * S * args.steps / elapsed, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"f32": "f32", "bf16x8": "f32 via exact 3-way bf16 split (8 bf16 MFMAs per product, f32 accumulate)", "bf16x6": "f32 via exact 3-way bf16 split (6 bf16 MFMAs per product, f32 accumulate)", "bf16x3": "f32 via split-bf16 (bf16x3 MFMA, f32 accumulate)"}[args.precision], "data": "synthetic",
            "config": {"workload": "nuScenes LiDAR-only SDSeg3D (TransVFE->UNetSCN3D->PointSegBatchlossHead), "
                                   "%d pts/frame, voxel [0.1,0.1,0.2], range [-51.2,-51.2,-5,51.2,51.2,3], 17 classes, "
                                   "1 frame per GPU per step, GPU voxelization included" % args.points,
                       "active_voxels": V, "frames_per_gpu_per_step": B * S, "streams": S, "parallelism": "frames sharded 1/GPU (dp%d)" % world},
            "roofline": {"bound": "hbm", "kernel": "k_gather_gemm (sparse-conv gather-GEMM, %d launches/frame)"
                                  % (conv["launches"] // max(args.steps, 1)),
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None, "avg_launch_us": conv["avg_us"],
                         "algo_bytes_per_frame": conv["algo_bytes"] / max(args.steps, 1),
                         "tflops": conv["flops"] / (conv["total_ms"] * 1e-3) / 1e12 if conv["total_ms"] > 0 else 0.0,
                         "sparse_conv_ms_per_frame": conv["total_ms"] / max(args.steps * S, 1),
                         # the same launches against the matrix pipe: useful (pair) flops only, exact-f32 MFMA peak
                         # (v_mfma_f32_32x32x2_f32: 256 CUs x 4 SIMDs x 4096 flop / 64 cycles x 2.4 GHz)
                         "mfma": None if args.precision != "f32" else {"achieved_tflops": conv["flops"] / (conv["total_ms"] * 1e-3) / 1e12 if conv["total_ms"] > 0 else 0.0,
                                  "peak_tflops": F32_MFMA_PEAK_TFLOPS,
                                  "frac": (conv["flops"] / (conv["total_ms"] * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS) if conv["total_ms"] > 0 else 0.0,
                                  "counter_evidence": "profiles/round1_pmc_sq.md (SQ_VALU_MFMA_BUSY_CYCLES: 51 % of SIMD cycles)"}},
        }
        pmc = os.path.join(ROOT, "profiles", "round1_pmc.json")
        if args.model == "sdseg3d" and args.precision == "f32" and args.points == 120000 and os.path.exists(pmc):
            # HBM-side bytes per sparse-conv launch from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this
            # same command (profiles/round1_pmc.md; counters cannot be collected from inside the timed run)
            out["roofline"]["traffic"] = json.load(open(pmc))["traffic_bytes_per_launch"]
            out["roofline"]["traffic_source"] = "profiles/round1_pmc.json (2*FETCH_SIZE+WRITE_SIZE, KiB, corrected per MI355X_MICROARCH.md)"
        out["roofline"]["algo_bytes_per_launch"] = conv["algo_bytes"] / max(conv["launches"], 1)
        if fast is not None:
            out["fast_mode"] = fast
        if "bf16x6" in alt:
            out["f32_grade_mode"] = alt["bf16x6"]
        if "bf16x8" in alt:
            out["bf16x8_mode"] = alt["bf16x8"]
        if throughput is not None:
            out["throughput_mode"] = throughput
        if args.model == "mseg3d":
            out["metric"] = "frames/sec, MSeg3D forward (LiDAR + 6-cam features), 120k-pt nuScenes-style frame"
            out["config"]["workload"] = out["config"]["workload"].replace(
                "nuScenes LiDAR-only SDSeg3D (TransVFE->UNetSCN3D->PointSegBatchlossHead)",
                "nuScenes MSeg3D (ImprovedMeanVFE->UNetSCN3D->PointSegMSeg3DHead GF+SF-Phase, image_features [1,6,48,160,240])")
        if world == 1 and not args.no_cpu_baseline and args.model == "sdseg3d":
            out["cpu_baseline"] = cpu_baseline(args.cpu_points or args.points, 100)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
