#!/usr/bin/env python
"""bench.py — frames/s of the SDSeg3D segmentation forward path on MI355X (BASELINE.json configs[1]).

One "step" = one full forward of one synthetic 120 000-point nuScenes-style frame per GPU:
    points (resident in HBM) -> GPU hard voxelization -> TransVFE -> UNetSCN3D (37 sparse convs + 8 rulebooks)
    -> 3-NN devoxelization -> PointSegBatchlossHead -> per-point logits [N,17] -> argmax.
Weights are random-init of the reference architecture (no checkpoints offline); data is synthetic.
Multi-GPU: frames shard one-per-GPU, no data-path collective (scaling "weak"); the only collectives are the
timing barrier and the max-over-ranks of the elapsed time.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--points 120000] [--precision bf16x6|bf16x8|f32|...] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Arithmetic of `value` (--precision, default "bf16x6"): the SubM layers run on the tile-halo kernel with every f32 operand split
EXACTLY into three round-to-nearest bf16 planes and the 6 plane products of weight >= 2^-16 accumulated in f32 (head x head in its
own accumulator; the strided / inverse convolutions on the 6-product gather-GEMM, the TransVFE reader's GEMMs likewise); every other dense
layer runs exact-f32 MFMA.  This mode is f32-grade — its
end-to-end logit error against a float64 evaluation is BELOW the exact-f32 MFMA path's own, rms (0.4x) and max, on every frame
measured (tests/test_gpu_parity.py::test_sdseg3d_every_arithmetic_vs_float64_..., profiles/round2_accuracy_*.json) —
and the 8-product variant (`f32_grade_8_product_mode`) and the exact-f32 path (`exact_f32_mode`) are timed beside it.

`--gpus N` without a torch.distributed.run environment checks that the box has N GPUs (exit code 2 otherwise) and re-executes itself
under `python -m torch.distributed.run --nproc-per-node N` (the reference launches its test the same way: tools/dist_test.py:99-104,
docs/semanticNusc.md:72); inside such an environment WORLD_SIZE must equal N.  The record carries what RCCL saw (`rccl_world_size`,
`ranks`: per-rank device and frames/s).

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline     — the sparse-conv stack (37 launches, contiguous on the main stream, bracketed by two HIP-event pairs per frame: level 1 | the rest):
                 algorithmic pair-model bytes sum_l P_l*(Cin+Cout)*4 over its measured duration;
  stage_rooflines — every other stage SURVEY.md §8(d) lists (voxelize, reader, rulebooks + plans, devoxelization, head; MSeg3D: grid gather, SFAM,
                 SF-Phase decoder) with its algorithmic bytes / flops, duration and fraction of its bound;
  batched      — frames/s with 1 / 2 / 4 / 8 frames collated into one forward, in the arithmetic of `value`;
  cpu_baseline — the CPU oracle (oracle/ref.py, kind "port") timed on the host cores on a bounded sample;
  stages_ms    — per-stage HIP-event breakdown of a frame; latency — median / p95 of the per-step wall time.

Test hook (tests/test_host_logic.py, never set by the product): LS3D_BENCH_HIPSIM=1 runs the same code path on CPU tensors with the kernels of
tests/hipsim and the gloo backend, so that the N > 1 bookkeeping is exercised where there is no GPU.
"""
import argparse
import json
import os
import statistics
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

F32_MFMA_PEAK_TFLOPS = 157.3
BF16_MFMA_PEAK_TFLOPS = 2500.0  # /opt/skills/guides/MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA (never the 2:1-sparsity figure)
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s peak
import glob as _glob
# the newest counter record: tools/collect_profiles.sh -> profiles/summarize_pmc.py -> profiles/round<N>_pmc.json
PMC_RECORD = (sorted(_glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "round*_pmc.json"))) or [""])[-1]
PLANE_PRODUCTS = {"bf16x6": 6, "bf16x8": 8, "bf16x3": 3}

DTYPES = {
    "bf16": "bf16 operands, f32 accumulation (plain bf16 MFMA in the SubM layers, bf16x3 in the strided / inverse layers): NOT f32-grade, BASELINE configs[4]",
    "f32": "f32",
    "bf16x8": "f32 (f32-grade: SubM layers on exact 3-plane bf16 splits, 8 of 9 plane products on bf16 MFMA with f32 accumulation; "
              "everything else exact-f32 MFMA)",
    "bf16x6": "f32 (f32-grade: sparse convolutions and the reader's GEMMs on exact 3-plane bf16 splits, the 6 plane products of weight >= 2^-16 on "
              "bf16 MFMA with f32 accumulation, head x head in its own accumulator; the head's dense layers exact-f32 MFMA)",
    "bf16x3": "f32 via split-bf16 (bf16x3 MFMA, f32 accumulate)",
}


def build_model(dev, seed=5, kind="sdseg3d"):
    import lidarseg3d_amd as L
    from lidarseg3d_amd import models_cfg, synth
    kw = dict(pc_range=(-6.4, -6.4, -5.0, 6.4, 6.4, 3.0)) if SIM else {}  # test hook: a 128 x 128 x 40 grid keeps the host emulation in seconds
    cfg = models_cfg.mseg3d(**kw) if kind == "mseg3d" else models_cfg.sdseg3d(**kw)
    model = L.build_detector(cfg, train_cfg=None, test_cfg={}).eval()
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = {k: torch.from_numpy(v) for k, v in synth.random_state_dict(shapes, seed).items()}
    model.load_state_dict(sd)
    return model.to(dev), sd


class ConvCensus:
    """one untimed frame with ops.gather_gemm / ops.tile_conv wrapped: which sparse-conv launches run, their pair counts and
    channel widths -> algorithmic bytes and flops of the stack (pair model)"""

    def __init__(self, ops):
        self.ops, self.meta = ops, []

    def run(self, step):
        ops, g, t = self.ops, self.ops.gather_gemm, self.ops.tile_conv

        def wg(x, w, tbl=None, **kw):
            if tbl is not None:
                self.meta.append((tbl, w.shape[1], kw.get("cout") or w.cout, "gather", kw.get("n_dev")))
            return g(x, w, tbl=tbl, **kw)

        def wt(x, w, plan, **kw):
            self.meta.append((plan.tbl, w.shape[1], kw.get("cout") or w.cout, "tile1", plan.n_dev))
            return t(x, w, plan, **kw)
        tc = ops.tile_conv_chain

        def wc(layers, plan):  # a chained launch = its layers' launches for the pair model (one kernel launch on the device)
            ok = tc(layers, plan)
            if ok:
                self.chained_launches += 1
                for l in layers:
                    self.meta.append((plan.tbl, l.w.shape[1], l.cout, "tile", plan.n_dev))
            return ok
        self.chained_launches = 0
        ops.gather_gemm, ops.tile_conv, ops.tile_conv_chain = wg, wt, wc
        try:
            with torch.no_grad():
                step()
            torch.cuda.synchronize()
        finally:
            ops.gather_gemm, ops.tile_conv, ops.tile_conv_chain = g, t, tc
        pairs, uniq, rows, kv, algo, flops, bmin = {}, {}, {}, {}, 0.0, 0.0, 0.0
        for tbl, cin, cout, _, n_dev in self.meta:
            key = (tbl.data_ptr(), tbl.shape[0])
            if key not in pairs:
                if n_dev is not None:  # capacity mode: the table has spare rows beyond the device count
                    tbl = tbl[:int(n_dev.item())]
                pairs[key] = int((tbl >= 0).sum().item())
                uniq[key] = int(torch.unique(tbl[tbl >= 0]).numel())
                rows[key] = tbl.shape[0]
                kv[key] = int(tbl.shape[1])
            algo += pairs[key] * (cin + cout) * 4.0
            flops += 2.0 * pairs[key] * cin * cout
            # B_min: what a launch cannot avoid moving: every referenced input row once, every output row once, the weights once
            bmin += (uniq[key] * cin + rows[key] * cout + tbl.shape[1] * cin * cout) * 4.0
        first = self.meta[0] if self.meta else None
        n_in = (int(first[4].item()) if first[4] is not None else first[0].shape[0]) if first else 0
        chained_layers = sum(1 for m in self.meta if m[3] == "tile")  # ("tile1": a layer launched on its own)
        for i, m in enumerate(self.meta):
            if m[3] == "tile1":
                self.meta[i] = m[:3] + ("tile",) + m[4:]
        return dict(launches=len(self.meta), tile_launches=sum(1 for m in self.meta if m[3] == "tile"), chained_layers=chained_layers,
                    kernel_launches=len(self.meta) - chained_layers + self.chained_launches, algo_bytes=algo, flops=flops, b_min_bytes=bmin,
                    input_voxels=n_in, tables=sorted(set((rows[k], kv[k]) for k in rows), reverse=True))


class TrainCensus:
    """one untimed TRAINING step with the sparse-convolution entry points wrapped: forward and input-gradient launches (ops.gather_gemm with a table /
    ops.tile_conv: the dgrad is the same operator on the transposed table) and weight-gradient launches (ops.spconv_wgrad), each between two HIP events
    -> pair-model bytes / flops of the step (SURVEY.md 8(d)'s formulas once per direction: P (Cin + Cout) 4 bytes, 2 P Cin Cout flops) and the time
    the device spent in them"""

    def __init__(self, ops):
        self.ops, self.calls = ops, []

    def run(self, step):
        ops = self.ops
        g, t, wgr = ops.gather_gemm, ops.tile_conv, ops.spconv_wgrad

        def bracket(kind, tbl, cin, cout, fn):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            r = fn()
            b.record()
            self.calls.append((kind, tbl, int(cin), int(cout), a, b))
            return r

        def wg(x, w, tbl=None, **kw):
            if tbl is None:
                return g(x, w, tbl=tbl, **kw)
            return bracket("conv", tbl, w.shape[1], kw.get("cout") or w.cout, lambda: g(x, w, tbl=tbl, **kw))

        def wt(x, w, plan, **kw):
            return bracket("conv", plan.tbl, w.shape[1], kw.get("cout") or w.cout, lambda: t(x, w, plan, **kw))

        def ww(x, grad_out, tbl, order, cin, cout, **kw):
            return bracket("wgrad", tbl, cin, cout, lambda: wgr(x, grad_out, tbl, order, cin, cout, **kw))
        ops.gather_gemm, ops.tile_conv, ops.spconv_wgrad = wg, wt, ww
        try:
            step()
            torch.cuda.synchronize()
        finally:
            ops.gather_gemm, ops.tile_conv, ops.spconv_wgrad = g, t, wgr
        pairs, out = {}, {"conv": [0, 0.0, 0.0, 0.0], "wgrad": [0, 0.0, 0.0, 0.0]}  # launches, bytes, flops, ms
        for kind, tbl, cin, cout, a, b in self.calls:
            key = (tbl.data_ptr(), tbl.shape[0], tbl.shape[1])
            if key not in pairs:
                pairs[key] = int((tbl >= 0).sum().item())
            o = out[kind]
            o[0] += 1
            o[1] += pairs[key] * (cin + cout) * 4.0
            o[2] += 2.0 * pairs[key] * cin * cout
            o[3] += a.elapsed_time(b)
        return out


SIM = os.environ.get("LS3D_BENCH_HIPSIM") == "1"  # test hook, see the module docstring


def _sync():
    if not SIM:
        torch.cuda.synchronize()


def timed_steps(step, steps, warmup, dist=None, dev=None):
    """W untimed steps, then K steps bracketed by barrier + synchronize; per-step HIP events for median / p95"""
    with torch.no_grad():
        for _ in range(warmup):
            step()
        _sync()
        if dist is not None:
            dist.barrier()
        _sync()
        marks = [] if SIM else [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        ticks = [time.perf_counter()]
        t0 = ticks[0]
        if marks:
            marks[0].record()
        for i in range(steps):
            step()
            if marks:
                marks[i + 1].record()
            else:
                ticks.append(time.perf_counter())
        _sync()
        elapsed = time.perf_counter() - t0
    local = elapsed
    if dist is not None:
        dist.barrier()
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    per = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(steps)) if marks else sorted(1e3 * (b - a) for a, b in zip(ticks[:-1], ticks[1:]))
    return elapsed, dict(median_ms=statistics.median(per), p95_ms=per[min(len(per) - 1, int(round(0.95 * (len(per) - 1))))], min_ms=per[0],
                         max_ms=per[-1], local_elapsed_s=local)


def cpu_baseline_worker(n_points, seed, threads, repeats, dump=None, budget_s=1e9):
    """runs in a child process: the CPU oracle's full SDSeg3D forward on one frame, one untimed warm-up run of the same frame, then timed
    `repeats` times; `dump`: .npy path that receives the logits of the last run (the parity check of the frame the GPU legs time)"""
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
    orc.sdseg3d_forward(sd, [synth.lidar_frame(min(n_points, 4000), seed=1, **synth.NUSC)], synth.NUSC["voxel_size"], synth.NUSC["pc_range"])  # warm-up
    ts = []
    for _ in range(repeats):
        t0 = time.time()
        ret = orc.sdseg3d_forward(sd, [frame], synth.NUSC["voxel_size"], synth.NUSC["pc_range"])
        ts.append(time.time() - t0)
        if sum(ts) > budget_s:  # bounded sample: report the runs that fit
            break
    if dump:
        np.save(dump, ret["out_logits"].numpy())
    print(json.dumps({"seconds": ts}))


def _cpu_run(n_points, seed, threads, repeats, timeout_s, dump=None, budget_s=None, passive=False):
    import subprocess
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads))
    if passive:  # hundreds of OpenMP threads spinning at the barriers of millisecond-sized torch ops starve each other: let idle threads sleep
        env.update(OMP_WAIT_POLICY="passive", GOMP_SPINCOUNT="0", KMP_BLOCKTIME="0")
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", str(n_points), str(seed), str(threads), str(repeats),
                        dump or "-", str(budget_s or 1e9)], env=env, capture_output=True, text=True, timeout=timeout_s, cwd=ROOT)
    return json.loads(r.stdout.strip().splitlines()[-1])["seconds"]


def parity_vs_cpu(gpu_logits, cpu_logits, num_class=17):
    """the GPU logits of the timed frame against the CPU oracle's logits of the same frame and weights (BASELINE metric: frames/sec
    + per-point mIoU-parity).  Random-init logits reach several thousand, so the absolute figure is also given at the |logit|max = 10
    scale the north_star's 1e-3 tolerance is tested at (tests/test_gpu_parity.py::test_sdseg3d_120k_frame_logits_and_miou_vs_oracle)"""
    from oracle import ref as orc
    g, c = gpu_logits.double(), torch.from_numpy(cpu_logits).double()
    scale = float(c.abs().max())
    d = (g - c).abs()
    pg, pc = g.argmax(1).numpy(), c.argmax(1).numpy()
    return dict(points=int(c.shape[0]), logit_abs_max=scale, max_abs_diff=float(d.max()), max_rel_diff=float(d.max()) / scale,
                max_abs_diff_at_logit_scale_10=10.0 * float(d.max()) / scale, rms_rel_diff=float(d.pow(2).mean().sqrt()) / scale,
                argmax_agreement=float((pg == pc).mean()), miou_gpu_labels_vs_cpu_labels=orc.miou(pg, pc, num_class),
                within_1e3_at_scale_10=bool(10.0 * float(d.max()) / scale <= 1e-3))


def cpu_quota():
    """CPUs this process may use at once: the cgroup v2 quota (cpu.max = "<quota> <period>"), else None"""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        return None


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or "unknown"


def cpu_baseline(n_points, seed, gpu_logits=None):
    """the CPU oracle (oracle/ref.py, a port: the reference has no CPU forward), rank 0 / N=1 only, on a BOUNDED sample of the workload
    (all host threads AND one thread, median of 5 after a warm-up, CPU model stated):
      * `value`: median of 5 warmed runs of the FULL frame on min(cores, 32) threads - where the restatement's torch-CPU index_add_ / mm stop
        scaling (on a 256-thread host the uncapped run is > 10x slower) - and the run whose logits are compared with the GPU's;
      * `all_threads`: os.cpu_count() threads (idle OpenMP threads sleeping), median of the warmed runs (5, or what fits in 25 s) of a 1/16-size
        frame, scaled linearly; if that does not finish in 60 s, a quarter of the threads (the sweep is recorded);
      * `single_thread`: 1 thread, median of 5 warmed runs of a 1/8-size frame, scaled likewise."""
    cores = os.cpu_count() or 1
    quota = cpu_quota()
    # the thread count of `value`: where the restatement's torch-CPU index_add_ / mm stop scaling (32), and never more than the container's
    # CPU quota (a 256-thread host behind a 16-CPU cgroup quota runs 256 OpenMP threads 20x slower than 16)
    threads = max(1, min(cores, 32, int(quota + 0.999) if quota else cores))
    dump = None
    if gpu_logits is not None and gpu_logits.shape[0] == n_points:
        import tempfile
        dump = os.path.join(tempfile.gettempdir(), "ls3d_cpu_logits_%d.npy" % os.getpid())
    med = lambda v: statistics.median(v)
    try:
        ts = _cpu_run(n_points, seed, threads, 5, 400, dump)
        out = dict(value=1.0 / med(ts), unit="frames/s", cores=threads, kind="port", cpu_model=cpu_model(), host_threads=cores,
                   sample="median of 5 warmed runs of 1 frame of %d points (%s s), full SDSeg3D forward incl. CPU voxelization, %d of %d host "
                          "threads%s; oracle/ref.py (torch-CPU gather-mm-scatter spconv restatement, OpenMP C exact 3-NN)"
                          % (n_points, "/".join("%.1f" % t for t in ts), threads, cores, (" (cgroup quota: %.0f CPUs)" % quota) if quota else ""))
    except Exception as e:  # never let the baseline take the bench down
        return dict(value=None, unit="frames/s", cores=threads, kind="port", cpu_model=cpu_model(), sample="failed: %r" % (e,))
    if dump is not None:
        try:
            out["parity"] = parity_vs_cpu(gpu_logits, np.load(dump))
        except Exception as e:
            out["parity"] = dict(error=repr(e))
        finally:
            if os.path.exists(dump):
                os.remove(dump)
    try:
        out["usable_threads"] = len(os.sched_getaffinity(0))
    except (OSError, AttributeError):
        pass
    out["cpu_quota"] = quota  # CPUs' worth of time the container gets (None: no cgroup quota)

    def small_leg(thr, div, timeout_s, passive):
        small = max(n_points // div, 1000)
        try:
            t = _cpu_run(small, seed, thr, 5, timeout_s, budget_s=25.0, passive=passive)
            return dict(threads=thr, points=small, seconds=t, median_s=med(t), frames_per_s_scaled_to_full_frame=1.0 / (med(t) * n_points / small),
                        note="median of %d warmed run(s) (5, or what fits in 25 s) of a 1/%d-size frame, scaled linearly in the point count%s"
                             % (len(t), div, "; OMP_WAIT_POLICY=passive" if passive else ""))
        except Exception as e:
            return dict(threads=thr, points=small, error=repr(e))

    if cores == threads:
        out["all_threads"] = dict(threads=cores, frames_per_s=out["value"], note="the host has no more than %d threads: same run as `value`" % threads)
    else:
        # every hardware thread; when that does not finish in 60 s (256 OpenMP threads on millisecond-sized torch-CPU ops), a quarter of them:
        # the sweep shows where the port stops scaling, so that `value` is not a sandbagged baseline
        sweep, thr = [], cores
        while thr > threads and len(sweep) < 2:
            sweep.append(small_leg(thr, 16, 60, True))
            if "error" not in sweep[-1]:
                break
            thr //= 4
        out["all_threads"] = sweep[0] if len(sweep) == 1 else dict(threads=cores, sweep=sweep, error=sweep[0].get("error"))
        best = [l for l in sweep if "error" not in l]
        if best and len(sweep) > 1:
            out["all_threads"]["most_threads_that_finished"] = best[0]
    out["single_thread"] = small_leg(1, 8, 120, False)
    return out


def stage_breakdown(model, pts, B, frames=5):
    """HIP events at the stage boundaries of SegNet.forward on the main stream (geometry runs beside the reader on a side stream;
    `backbone` includes the wait for it).  Median over a few extra frames."""
    acc = {k: [] for k in ("voxelize", "reader", "backbone", "head+argmax")}
    hooks, ev = [], {}

    def mark(tag):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        ev[tag] = e

    hooks.append(model.reader.register_forward_pre_hook(lambda m, a: mark("r0")))
    hooks.append(model.reader.register_forward_hook(lambda m, a, o: mark("r1")))
    hooks.append(model.backbone.register_forward_hook(lambda m, a, o: mark("b1")))
    try:
        with torch.no_grad():
            for _ in range(frames):
                mark("s")
                model(dict(points=pts, batch_size=B), return_loss=False)
                mark("e")
                torch.cuda.synchronize()
                acc["voxelize"].append(ev["s"].elapsed_time(ev["r0"]))
                acc["reader"].append(ev["r0"].elapsed_time(ev["r1"]))
                acc["backbone"].append(ev["r1"].elapsed_time(ev["b1"]))
                acc["head+argmax"].append(ev["b1"].elapsed_time(ev["e"]))
    finally:
        for h in hooks:
            h.remove()
    return {k: statistics.median(v) for k, v in acc.items()}


class OpTimer(object):
    """HIP-event pairs around selected host-level calls of ONE eager frame that runs on a single stream (LS3D_OVERLAP=0: geometry inline): the
    per-stage durations behind `stage_rooflines`.  Each pair costs a few microseconds of idle GPU, so this leg only feeds the breakdown."""

    def __init__(self):
        self.ev, self.undo = {}, []

    def wrap(self, label, owner, name, when=None):
        fn = getattr(owner, name)

        def timed(*a, **kw):
            if when is not None and not when(*a, **kw):
                return fn(*a, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            try:
                return fn(*a, **kw)
            finally:
                e1.record()
                self.ev.setdefault(label, []).append((e0, e1))
        setattr(owner, name, timed)
        self.undo.append((owner, name, fn))

    def module(self, label, mod):
        h0 = mod.register_forward_pre_hook(lambda m, a: self._open(label))
        h1 = mod.register_forward_hook(lambda m, a, o: self._close(label))
        self.undo.append((h0, None, None))
        self.undo.append((h1, None, None))

    def _open(self, label):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self.ev.setdefault(label, []).append([e, None])

    def _close(self, label):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self.ev[label][-1][1] = e

    def restore(self):
        for owner, name, fn in reversed(self.undo):
            if name is None:
                owner.remove()
            else:
                setattr(owner, name, fn)
        self.undo = []

    def ms(self):
        return {k: sum(a.elapsed_time(b) for a, b in v) for k, v in self.ev.items()}

    def clear(self):
        self.ev = {}


def stage_rooflines(model, example, kind, census, frames=5):
    """SURVEY.md 8(d) "roofline bound per stage": duration (median over a few eager single-stream frames, HIP events around the stage's calls),
    algorithmic bytes or flops from 8(d)'s formulas, and the fraction of the bound's peak.  The sparse-conv stack has its own object (`roofline`)."""
    from lidarseg3d_amd import ops, spconv
    pts = example["points"]
    N, cp = int(pts.shape[0]), int(pts.shape[1]) - 1
    tm = OpTimer()
    prev = os.environ.get("LS3D_OVERLAP")
    os.environ["LS3D_OVERLAP"] = "0"
    shapes = {}

    def note(label, fn):  # remembers an argument-derived size of the last call
        def inner(*a, **kw):
            shapes[label] = fn(*a, **kw)
            return True
        return inner
    try:
        tm.wrap("voxelize", ops, "voxelize_hard")
        tm.module("reader", model.reader)
        tm.wrap("rulebooks", spconv, "subm_rulebook")
        tm.wrap("rulebooks", spconv, "prebuild_conv_rulebooks")
        tm.wrap("plans+orders", spconv, "prebuild_orders")
        tm.wrap("devox_search", ops, "devoxelize_grid")
        tm.wrap("interpolate", ops, "interpolate_rows", note("interp_c", lambda feat, *a, **kw: int(feat.shape[1])))
        tm.wrap("head_mlp", ops, "gather_gemm", lambda x, w, tbl=None, **kw: tbl is None and x.shape[0] >= N)
        tm.wrap("head_tail", ops, "point_mlp", note("tail", lambda feat, model_, *a, **kw: (int(model_.c_in), int(model_.c_out),
                                                                                              sum(2.0 * w.shape[0] * w.shape[1] for w, _, _, _ in model_.keep))))
        if kind == "mseg3d":
            tm.wrap("grid_gather", ops, "grid_gather", note("img_c", lambda img, *a, **kw: int(img.shape[2])))
            tm.wrap("sfam", ops, "sfam")
            tm.wrap("sffm_decoder", ops, "sffm_decoder", note("sffm", lambda x, points, kv, L, batch, m: (int(x.shape[1]), int(L), m)))
            tm.wrap("sffm_memory_side", ops, "cross_attn")
        per = []
        with torch.no_grad():
            model(dict(example), return_loss=False)
            for _ in range(frames):
                tm.clear()
                model(dict(example), return_loss=False)
                torch.cuda.synchronize()
                per.append(tm.ms())
    finally:
        tm.restore()
        if prev is None:
            os.environ.pop("LS3D_OVERLAP", None)
        else:
            os.environ["LS3D_OVERLAP"] = prev
    ms = {k: statistics.median(d.get(k, 0.0) for d in per) for k in per[0]}
    V = census.get("input_voxels", 0)
    tables = census.get("tables", [])
    out = {}

    def hbm(label, nbytes, formula):
        if label in ms and ms[label] > 0:
            gbs = nbytes / (ms[label] * 1e-3) / 1e9
            out[label] = dict(bound="hbm", ms=ms[label], algorithmic_bytes=nbytes, achieved=gbs, peak=HBM_PEAK_GBS, unit="GB/s", frac=gbs / HBM_PEAK_GBS,
                              formula=formula)

    def mfma(label, flops, peak, dtype, formula, products=1):
        if label in ms and ms[label] > 0:
            tf = flops / (ms[label] * 1e-3) / 1e12
            out[label] = dict(bound="mfma", ms=ms[label], algorithmic_flops=flops, achieved=products * tf, useful_f32_equivalent_tflops=tf, peak=peak,
                              unit="TFLOP/s", dtype=dtype, frac=products * tf / peak, formula=formula)
    hbm("voxelize", N * (1 + cp) * 4.0 + V * (5 * cp + 5) * 4.0 + 2.0 * N * 16, "N*(1+Cp)*4 read + V*(5*Cp+5)*4 written + hash 2N*16 (DESIGN.md 4)")
    c_vfe = int(model.backbone.conv_input[0].in_channels)
    if kind == "sdseg3d":
        f_tok = 2.0 * (2 * cp + 8) * 64 + 3.0 * (8 * 64 * 64 + 4 * 5 * 64 + 4 * 64 * 128)
        np_ = PLANE_PRODUCTS.get(ops.get_precision())
        mfma("reader", V * 5.0 * f_tok, BF16_MFMA_PEAK_TFLOPS if np_ else F32_MFMA_PEAK_TFLOPS, "bf16 planes" if np_ else "f32",
             "TransVFE, SURVEY.md 8(d): V*5*(2*(2Cp+8)*64 + 3*(8*64*64 + 4*5*64 + 4*64*128)) over ALL 5 slots (the kernel computes a voxel's identical "
             "padding tokens once: fewer executed flops, same algorithmic ones)", products=np_ or 1)
    else:
        hbm("reader", V * 5.0 * cp * 4 + V * 4.0 + V * c_vfe * 4.0, "ImprovedMeanVFE: V*5*Cp*4 + V*4 read + V*C_vfe*4 written")
    hbm("rulebooks", sum(r * k * 4.0 for r, k in tables) + sum(r * 16.0 * 2 for r, k in tables if k == 27),
        "sum over the 8 rulebooks of rows*kvol*4 (SURVEY.md 8(d): V*27*4 per table) + hash 2V*16 per SubM index")
    hbm("plans+orders", sum(r * k * 4.0 + r * k * 2.0 + r * 8.0 for r, k in tables if k == 27),
        "per tile plan: table read rows*27*4 + 16-bit local table written rows*27*2 + keys / order rows*8")
    hbm("devox_search", N * 16.0 + V * 16.0 + N * 24.0, "SURVEY.md 8(d) windowed search: N*12(+4 batch) + V*12(+4) read + N*(12+12) idx / weights written")
    ic = shapes.get("interp_c", 32)
    hbm("interpolate", 3.0 * N * ic * 4 + N * ic * 4.0 + N * 24.0, "3*N*C*4 gathered + N*C*4 written + N*24 idx / weights")
    if "tail" in shapes and "head_tail" in ms and ms["head_tail"] > 0:
        # ls3d_point_mlp: interpolation + the head's Linear chain + argmax in one launch.  HBM side: three gathered voxel rows, the logits and the
        # label per point; matrix side: exact-f32 MFMA
        c_in, c_out, f_pt = shapes["tail"]
        hbm("head_tail", N * (3.0 * c_in * 4 + 24 + c_out * 4 + 8), "ls3d_point_mlp: N*(3*C_in*4 gathered + 24 idx / weights + classes*4 logits + 8 label)")
        out["head_tail"]["f32_mfma_tflops"] = N * f_pt / (ms["head_tail"] * 1e-3) / 1e12
        out["head_tail"]["f32_mfma_frac"] = out["head_tail"]["f32_mfma_tflops"] / F32_MFMA_PEAK_TFLOPS
    if kind == "mseg3d":
        c_img = shapes.get("img_c", 48)
        hbm("grid_gather", 0.75 * N * 4 * c_img * 4.0 + N * c_img * 4.0, "SURVEY.md 8(d): Nv*4 taps*48*4 + N*48*4 (Nv = 0.75 N valid points)")
        hbm("sfam", V * (32 + 17) * 4.0 * 2, "softmax over voxels: V*(C + cls)*4, two passes")
        if "sffm" in shapes:
            d_in, L, m = shapes["sffm"]
            d, ffn, layers = int(m.c.d_model), int(m.c.ffn), int(m.c.num_layers)
            f_gemm = N * (2.0 * d_in * d + layers * (2.0 * d * d * 2 + 2.0 * 2 * d * ffn))
            f_attn = N * layers * (2.0 * 2 * d * L)
            formula = "SURVEY.md 8(d) SFFM: N*(2*d_in*d + layers*(2*d*d*2 + 2*2*d*L + 2*2*d*ffn)), d=%d, L=%d tokens, ffn=%d, %d layers" % (d, L, ffn, layers)
            planes = (ops.get_precision() in ("bf16x6", "bf16x8", "bf16") and getattr(ops, "_SFFM_PLANES", False) and getattr(ops, "_SFFM_ATTENTION", 0) == 0 and L <= 64)
            if planes and "sffm_decoder" in ms and ms["sffm_decoder"] > 0:
                # k_sffm_decoder_rt: the GEMMs as 6 bf16 plane products per f32 product, the attention products on the exact-f32 MFMA
                t = ms["sffm_decoder"] * 1e-3
                out["sffm_decoder"] = dict(bound="mfma", ms=ms["sffm_decoder"], algorithmic_flops=f_gemm + f_attn, achieved=6 * f_gemm / t / 1e12,
                                           useful_f32_equivalent_tflops=(f_gemm + f_attn) / t / 1e12, peak=BF16_MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                                           dtype="bf16 planes (GEMMs: 6 products per f32 product); attention QK^T / PV on the f32 MFMA beside it",
                                           frac=6 * f_gemm / t / 1e12 / BF16_MFMA_PEAK_TFLOPS, attention_f32_tflops=f_attn / t / 1e12, formula=formula,
                                           kernel="k_sffm_decoder_rt")
            else:
                mfma("sffm_decoder", f_gemm + f_attn, F32_MFMA_PEAK_TFLOPS, "f32", formula)
    out["_unattributed_ms"] = {k: v for k, v in ms.items() if k not in out}
    return out


def train_leg(args, dist, dev, rank, world, steps, warmup):
    """BASELINE configs[3] as a bench leg: ONE data-parallel training step of MSeg3D on Waymo geometry (23 classes, 5 cameras, 180k points per
    frame, 2 frames per GPU = semwaymo_..._e12.py:231) - forward + (CE + Lovasz + mimic) loss + backward + SGD update, every _BatchNorm converted
    to its count-weighted synchronised form (train.py:313-321) and the model wrapped in DistributedDataParallel over RCCL (train.py:345-352):
    one flat gradient bucket, conv_out (no gradient on the segmentation path) excluded from the reducer instead of find_unused_parameters.
    Timed like the inference legs (barrier + synchronise around K steps, MAX over ranks).  Beside the step time the record carries what the
    exchange costs: gradient bytes per step, the number of SyncBN collectives per step (counted by wrapping torch.distributed for one step),
    and the EXPOSED all-reduce time = step time - step time with the gradient all-reduce skipped (DDP.no_sync)."""
    import lidarseg3d_amd as L
    from lidarseg3d_amd import models_cfg, ops, sharding, synth, syncbn
    cfg = synth.WAYMO
    ncls, ncam, B = 23, 5, (1 if SIM else 2)
    points = args.train_points or (400 if SIM else 180000)
    kw = dict(num_class=ncls, pc_range=cfg["pc_range"], voxel_size=cfg["voxel_size"])
    if SIM:  # test hook: a small grid and narrow levels keep the host emulation in seconds
        kw.update(pc_range=(-6.4, -6.4, -2.0, 6.4, 6.4, 4.0))
    mcfg = models_cfg.mseg3d(**kw)
    if SIM:
        mcfg["backbone"]["model_cfg"] = dict(mcfg["backbone"].get("model_cfg", {}), SCALING_RATIO=1)
        mcfg["point_head"]["model_cfg"] = dict(mcfg["point_head"]["model_cfg"], VOXEL_IN_DIM=16)
    torch.manual_seed(0)
    ops.set_precision(args.precision if args.precision in ("f32", "bf16x6", "bf16x8") else "bf16x6")
    model = L.build_detector(mcfg, train_cfg=None, test_cfg={})
    model = syncbn.convert_sync_batchnorm(model).to(dev).train()
    grad_params = [(k, p) for k, p in model.named_parameters() if p.requires_grad and not k.startswith("backbone.conv_out")]
    grad_bytes = sum(p.numel() for _, p in grad_params) * 4
    net = model
    own_group = False
    if dist is None and not SIM and world == 1:
        # N = 1 without a launcher: a process group of one rank, so that the step runs the same DDP reducer and SyncBN modules as N > 1 and the
        # driver's N = 1 / 2 / 4 / 8 figures are like for like
        import socket
        import torch.distributed as dist
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
        own_group = True
    if dist is not None:
        from torch.nn.parallel import DistributedDataParallel as DDP
        # conv_out feeds nothing the segmentation loss sees (scn_unet.py:218-222): excluded from the reducer, so that DDP need not walk the
        # autograd graph for unused parameters every step (the reference passes find_unused_parameters=True, train.py:351)
        DDP._set_params_and_buffers_to_ignore_for_model(model, [k for k, _ in model.named_parameters() if k.startswith("backbone.conv_out")] +
                                                        [k for k, _ in model.named_buffers() if k.startswith("backbone.conv_out")])
        net = DDP(model, device_ids=None if SIM else [dev.index], bucket_cap_mb=128, gradient_as_bucket_view=True)
    pc = cfg["pc_range"] if not SIM else kw["pc_range"]
    fcfg = dict(cfg, pc_range=pc)
    frames = [synth.lidar_frame(points, seed=500 + rank * B + b, **fcfg) for b in range(B)]
    pts = torch.from_numpy(np.concatenate([np.concatenate([np.full((f.shape[0], 1), b, np.float32), f], 1) for b, f in enumerate(frames)])).to(dev)
    v, c, n, nv = ops.voxelize_hard(pts, cfg["voxel_size"], pc, 5, 300000 * B, batched=True)
    V = int(nv)
    gen = torch.Generator().manual_seed(7 + rank)
    ex = dict(points=pts, voxels=v[:V], coordinates=c[:V], num_points=n[:V], num_voxels=[0] * B, shape=[np.asarray(ops.make_grid(cfg["voxel_size"], pc)[1])],
              voxel_sem_labels=torch.randint(0, ncls, (V,), generator=gen).to(dev), point_sem_labels=torch.randint(0, ncls, (pts.shape[0],), generator=gen).to(dev))
    h, w = (8, 12) if SIM else (160, 240)
    img, emb, cuv = synth.camera_inputs(pts.shape[0], seed=rank, ncam=ncam, c_img=48, h=h, w=w, num_class=ncls, batch=B)
    ex.update(image_features=torch.from_numpy(img).to(dev), camera_semantic_embeddings=torch.from_numpy(emb).to(dev), points_cuv=torch.from_numpy(cuv).to(dev))
    opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9)
    losses = []

    def step(sync=True):
        opt.zero_grad(set_to_none=True)
        if dist is not None and not sync:
            with net.no_sync():
                loss = net(dict(ex), return_loss=True)["loss"][0]
                loss.backward()
        else:
            loss = net(dict(ex), return_loss=True)["loss"][0]
            loss.backward()
        opt.step()
        losses.append(loss.detach())

    def timed(k, sync=True):
        _sync()
        if dist is not None:
            dist.barrier()
        _sync()
        t0 = time.perf_counter()
        for _ in range(k):
            step(sync)
        _sync()
        el = time.perf_counter() - t0
        local = el
        if dist is not None:
            dist.barrier()
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, local
    for _ in range(warmup):
        step()
    # collectives of ONE step, by kind (the SyncBN layers' all_gather / all_reduce pairs + DDP's bucket reduction, which does not go through
    # the python API and is counted from the reducer's bucket list)
    calls = {}
    if dist is not None:
        orig = {k: getattr(dist, k) for k in ("all_gather", "all_reduce")}

        def counted(name):
            def f(*a, **kw):
                calls[name] = calls.get(name, 0) + 1
                return orig[name](*a, **kw)
            return f
        for k in orig:
            setattr(dist, k, counted(k))
        try:
            step()
        finally:
            for k, f in orig.items():
                setattr(dist, k, f)
    el, local = timed(steps)
    rec = dict(metric="frames/sec, MSeg3D TRAINING step (forward + loss + backward + SGD), Waymo geometry, %d pts/frame, %d frames per GPU (BASELINE configs[3])"
                      % (points, B), precision=ops.get_precision(), frames_per_gpu=B, points_per_frame=points, active_voxels_rank0=V, steps=steps, warmup=warmup,
               step_ms=1e3 * el / steps, value=world * B * steps / el, unit="frames/s", n_gpus=world,
               parallelism=("DistributedDataParallel (one flat %d MB bucket) + count-weighted SyncBN over %s, %d ranks" % (128, "gloo" if SIM else "RCCL", world))
               if dist is not None else "single process (host emulation test hook): plain model, local BatchNorm statistics",
               gradient_allreduce_bytes_per_step=grad_bytes if dist is not None else 0, parameters_reduced=len(grad_params),
               syncbn_layers=sum(1 for m in model.modules() if isinstance(m, (syncbn.CountSyncBatchNorm1d, syncbn.CountSyncBatchNorm2d, syncbn.CountSyncBatchNorm3d))),
               collectives_per_step=dict(calls, ddp_buckets=1) if dist is not None else {}, loss_first=float(losses[0]), loss_last=float(losses[-1]))
    if not SIM:
        # the step's roofline object: the sparse convolutions of forward, input-gradient and weight-gradient pass in the pair
        # model, over the device time between their HIP events in one instrumented step; and where the step's wall time goes (forward incl.
        # loss / backward incl. the gradient all-reduce / optimiser), each bracketed by events on the current stream
        try:
            cz = TrainCensus(ops).run(step)
            conv_b, conv_f, conv_ms = cz["conv"][1] + cz["wgrad"][1], cz["conv"][2] + cz["wgrad"][2], cz["conv"][3] + cz["wgrad"][3]
            np_ = PLANE_PRODUCTS.get(ops.get_precision())
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            opt.zero_grad(set_to_none=True)
            _sync()
            ev[0].record()
            loss = net(dict(ex), return_loss=True)["loss"][0]
            ev[1].record()
            loss.backward()
            ev[2].record()
            opt.step()
            ev[3].record()
            _sync()
            rec["stage_ms"] = dict(forward_and_loss=ev[0].elapsed_time(ev[1]), backward_and_gradient_allreduce=ev[1].elapsed_time(ev[2]),
                                   optimizer=ev[2].elapsed_time(ev[3]),
                                   note="one step bracketed by HIP events on the compute stream (the host runs ahead of the device: stage boundaries are the device's)")
            rec["roofline"] = {
                "bound": "mfma", "kernel": "sparse convolutions of the step: %d forward + input-gradient launches (ls3d_gather_gemm / ls3d_tile_conv) and %d "
                                           "weight-gradient launches (ls3d_spconv_wgrad)" % (cz["conv"][0], cz["wgrad"][0]),
                "achieved": conv_b / (conv_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": conv_b / (conv_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "traffic": None, "model": "pair model, SURVEY.md 8(d), once per direction: forward, dgrad (the same operator on the transposed table) and wgrad "
                                          "each move P (Cin + Cout) 4 bytes and do 2 P Cin Cout flops per layer",
                "algo_bytes_per_step": conv_b, "algo_flops_per_step": conv_f, "sparse_conv_ms_per_step": conv_ms,
                "share_of_step": conv_ms / rec["step_ms"], "tflops_useful": conv_f / (conv_ms * 1e-3) / 1e12,
                "forward_and_dgrad": dict(launches=cz["conv"][0], algo_bytes=cz["conv"][1], flops=cz["conv"][2], ms=cz["conv"][3],
                                          frac=cz["conv"][1] / (cz["conv"][3] * 1e-3) / 1e9 / HBM_PEAK_GBS if cz["conv"][3] else None),
                "wgrad": dict(launches=cz["wgrad"][0], algo_bytes=cz["wgrad"][1], flops=cz["wgrad"][2], ms=cz["wgrad"][3],
                              frac=cz["wgrad"][1] / (cz["wgrad"][3] * 1e-3) / 1e9 / HBM_PEAK_GBS if cz["wgrad"][3] else None),
                "mfma": (dict(unit="TFLOP/s", dtype="bf16", plane_products_per_f32_product=np_, achieved=np_ * conv_f / (conv_ms * 1e-3) / 1e12,
                              peak=BF16_MFMA_PEAK_TFLOPS, frac=np_ * conv_f / (conv_ms * 1e-3) / 1e12 / BF16_MFMA_PEAK_TFLOPS) if np_ else
                         dict(unit="TFLOP/s", dtype="f32", achieved=conv_f / (conv_ms * 1e-3) / 1e12, peak=F32_MFMA_PEAK_TFLOPS,
                              frac=conv_f / (conv_ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS)),
                "note": "device time between the HIP events around each launch of one instrumented step (launches of a stream do not overlap)"}
        except Exception as e:
            rec["roofline"] = dict(error=repr(e))
    if dist is not None:
        el0, _ = timed(max(2, steps // 2), sync=False)
        k0 = max(2, steps // 2)
        rec["step_ms_without_gradient_allreduce"] = 1e3 * el0 / k0
        rec["exposed_gradient_allreduce_ms"] = max(0.0, rec["step_ms"] - rec["step_ms_without_gradient_allreduce"])
        mine = dict(rank=rank, device=str(dev), step_ms=1e3 * local / steps, active_voxels=V)
        rec["ranks"] = sharding.gather_frame_results(mine)
    del net, model, opt, ex
    if own_group:
        dist.destroy_process_group()
    return rec


def relaunch_under_launcher(args, argv):
    """`python bench.py --gpus N` (N > 1) outside a torch.distributed.run environment: fail loudly when the box does not have N GPUs, else
    re-execute this command as one rank per GPU (tools/dist_test.py:99-104 of the reference runs under torch.distributed.launch the same way)"""
    import socket
    import subprocess
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if not SIM and have < args.gpus:
        print("bench.py --gpus %d: this box has %d GPU(s); refusing to report a %d-GPU figure from fewer devices" % (args.gpus, have, args.gpus),
              file=sys.stderr)
        raise SystemExit(2)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    raise SystemExit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--points", type=int, default=120000)
    ap.add_argument("--cpu-points", type=int, default=None, help="points of the CPU-baseline frame (default: --points)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", choices=list(DTYPES), default="bf16x6",
                    help="arithmetic of the sparse convolutions (see the module docstring); f32 = exact-f32 MFMA everywhere")
    ap.add_argument("--no-extra-modes", action="store_true", help="skip the extra legs (exact f32, bf16x3, two streams, batched, MSeg3D)")
    ap.add_argument("--no-fast-mode", action="store_true", help=argparse.SUPPRESS)  # round-1 spelling of --no-extra-modes
    ap.add_argument("--streams", type=int, default=1,
                    help="throughput mode: a step = this many frames, each a batch of one on its own HIP stream; SDSeg3D only")
    ap.add_argument("--frames-per-step", type=int, default=1,
                    help="frames collated into one forward per GPU per step (1 = the reference's --speed_test batch size)")
    ap.add_argument("--row-order", choices=["mask", "none"], default="mask")
    ap.add_argument("--no-graph", action="store_true",
                    help="time the eager path (Python submits every launch) instead of one hipGraph per frame (lidarseg3d_amd.graph.FrameGraph)")
    ap.add_argument("--model", choices=["sdseg3d", "mseg3d"], default="sdseg3d",
                    help="sdseg3d = BASELINE configs[1] (the metric's config); mseg3d = configs[2] (LiDAR + 6-camera features)")
    ap.add_argument("--no-train-leg", action="store_true", help="skip the data-parallel training step (BASELINE configs[3]) that follows the inference legs")
    ap.add_argument("--train-points", type=int, default=None, help="points per frame of the training leg (default 180000: Waymo)")
    ap.add_argument("--cpu-baseline-worker", nargs="+", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_worker:
        w = args.cpu_baseline_worker
        cpu_baseline_worker(*[int(v) for v in w[:4]], dump=(w[4] if len(w) > 4 and w[4] != "-" else None), budget_s=float(w[5]) if len(w) > 5 else 1e9)
        return
    extra_modes = not (args.no_extra_modes or args.no_fast_mode)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    launched = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if args.gpus > 1 and not launched:
        relaunch_under_launcher(args, sys.argv[1:])

    from lidarseg3d_amd import sharding
    rank, local_rank, world = sharding.env_rank_world()
    if world != args.gpus:
        print("bench.py --gpus %d inside a launcher environment with WORLD_SIZE=%d: the record would not be a %d-GPU figure" % (args.gpus, world, args.gpus),
              file=sys.stderr)
        raise SystemExit(2)
    if SIM:  # test hook: the kernels of tests/hipsim on CPU tensors, gloo instead of RCCL
        sys.path.insert(0, os.path.join(ROOT, "tests", "hipsim"))
        import build_sim
        from lidarseg3d_amd import _lib as l3lib, ops as l3ops
        l3lib.use_library_for_testing(build_sim.build())
        l3ops.set_sim(True)
        dev = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
        if torch.cuda.device_count() <= local_rank:
            print("rank %d: LOCAL_RANK %d but only %d GPU(s) visible" % (rank, local_rank, torch.cuda.device_count()), file=sys.stderr)
            raise SystemExit(2)
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("LS3D_BENCH_FORCE_DIST") == "1":  # the env knob exercises the RCCL path on a 1-GPU box
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if SIM else "nccl", init_method="env://")
        if dist.get_world_size() != args.gpus:
            raise SystemExit("process group of %d ranks for --gpus %d" % (dist.get_world_size(), args.gpus))

    from lidarseg3d_amd import detectors, ops, scn_unet, synth
    ops.set_precision(args.precision)
    ops.set_row_order(args.row_order)
    model, sd = build_model(dev, kind=args.model)
    B = max(1, args.frames_per_step)
    S = max(1, args.streams)
    single = world == 1 and dist is None
    TP = 2 if (args.model == "sdseg3d" and B == 1 and S == 1 and single and extra_modes and not SIM) else 0
    # frames shard across the ranks as independent units (sharding.shard_frames): rank r of N times frames r*B .. r*B + B - 1 of the N*B
    my_frames = sharding.shard_frames(world * B, rank, world)

    def frames_to_points(frames):
        return torch.from_numpy(np.concatenate([np.concatenate([np.full((f.shape[0], 1), b, np.float32), f], 1) for b, f in enumerate(frames)])).to(dev)

    def make_inputs(kind, n_streams, frame_ids=None, seed0=100):
        ids = my_frames if frame_ids is None else frame_ids
        pts = frames_to_points([synth.lidar_frame(args.points, seed=seed0 + i, **synth.NUSC) for i in ids])
        extra = {}
        if kind == "mseg3d":  # HRNet-w18 feature maps of 6 cameras at 1/4 resolution + camera class embeddings (inputs of the path)
            img, emb, cuv = synth.camera_inputs(args.points * len(ids), seed=seed0 + ids[0], ncam=6, c_img=48, h=160, w=240, batch=len(ids))
            extra = dict(points_cuv=torch.from_numpy(cuv).to(dev), image_features=torch.from_numpy(img).to(dev),
                         camera_semantic_embeddings=torch.from_numpy(emb).to(dev))
        spts, streams = [], []
        if n_streams > 1:
            sframes = [synth.lidar_frame(args.points, seed=seed0 + rank * n_streams + i, **synth.NUSC) for i in range(n_streams)]
            spts = [frames_to_points([f]) for f in sframes]
            streams = [torch.cuda.Stream(dev) for _ in range(n_streams)]
        return pts, extra, spts, streams

    pts, extra, spts, streams = make_inputs(args.model, max(S, TP))
    if S > 1:
        assert args.model == "sdseg3d" and B == 1, "--streams: SDSeg3D, one frame per stream"

    def make_step(m, pts_, extra_, n_streams, nb=B):
        if n_streams == 1:
            return lambda: m(dict(points=pts_, batch_size=nb, **extra_), return_loss=False)[0]["pred_point_sem_labels"]

        def step_n():
            cur = torch.cuda.current_stream(dev)
            labels = None
            for st, p in zip(streams[:n_streams], spts[:n_streams]):
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    labels = m(dict(points=p, batch_size=1), return_loss=False)[0]["pred_point_sem_labels"]
            for st in streams[:n_streams]:
                cur.wait_stream(st)
            return labels
        return step_n

    def measure(prec, steps, warmup, n_streams, with_dist):
        """one leg: W untimed + K timed steps; with one stream also the conv-stack event brackets of the timed steps"""
        ops.set_precision(prec)
        st = make_step(model, pts, extra, n_streams)
        if n_streams == 1 and not SIM:
            with torch.no_grad():  # two frames first: the census then sees the launches of the steady state (capacities learned, the chain policy of the
                for _ in range(2):  # timed frames), not a first frame on worst-case capacities
                    make_step(model, pts, extra, 1)()
        census = ConvCensus(ops).run(make_step(model, pts, extra, 1)) if (n_streams == 1 and not SIM) else None
        events = []
        scn_unet.UNetSCN3D.conv_stack_events = events if (n_streams == 1 and not SIM) else None
        try:
            with torch.no_grad():
                for _ in range(warmup):
                    st()
            _sync()
            del events[:]  # the brackets of the warm-up steps
            elapsed, lat = timed_steps(st, steps, 0, dist if with_dist else None, dev)
        finally:
            scn_unet.UNetSCN3D.conv_stack_events = None
        out = dict(precision=prec, frames_per_s=world * B * n_streams * steps / elapsed, ms_per_step=1e3 * elapsed / steps, latency=lat,
                   local_frames_per_s=B * n_streams * steps / lat["local_elapsed_s"])
        if census is not None and events:
            per = max(len(events) // steps, 1)  # (start, end) pairs per frame: the stack is bracketed in contiguous pieces
            ms = [a.elapsed_time(b) for a, b in events]
            stack_ms = sorted(sum(ms[i:i + per]) for i in range(0, len(ms) - per + 1, per))
            mean = sum(stack_ms) / len(stack_ms)
            out.update(census=census, conv_stack_ms=dict(mean=mean, median=statistics.median(stack_ms),
                                                         p95=stack_ms[min(len(stack_ms) - 1, int(round(0.95 * (len(stack_ms) - 1))))]),
                       roofline_frac=census["algo_bytes"] / (mean * 1e-3) / 1e9 / HBM_PEAK_GBS, tflops=census["flops"] / (mean * 1e-3) / 1e12)
        return out

    main_leg = measure(args.precision, args.steps, args.warmup, S, True)
    ref_logits = model.point_head.forward_ret_dict["out_logits"].clone()
    # `value`: the same frame as ONE hipGraph (capacity mode makes it capturable): the eager leg above submits ~220 launches per frame
    # from Python and its latency-bound geometry chain is host-bound; it stays in the record (`eager_mode`) and carries the HIP-event
    # brackets of the sparse-conv stack (timing events cannot be captured) - the graph replays exactly those launches
    graph_leg = None
    if not args.no_graph and S == 1 and B == 1 and detectors.CAPACITY_MODE and not SIM:
        try:
            from lidarseg3d_amd import graph as lgraph
            ops.set_precision(args.precision)
            ex = dict(points=pts, batch_size=1, **extra)
            fg = lgraph.FrameGraph(model, ex)
            gstep = lambda: fg(ex, clone=False)[0]["pred_point_sem_labels"]
            elapsed, lat = timed_steps(gstep, args.steps, args.warmup, dist, dev)
            same = bool(torch.equal(fg.logits, ref_logits))
            graph_leg = dict(frames_per_s=world * args.steps / elapsed, ms_per_step=1e3 * elapsed / args.steps, latency=lat,
                             local_frames_per_s=args.steps / lat["local_elapsed_s"],
                             logits_bit_identical_to_eager=same, fallbacks=fg.fallbacks, recaptures=fg.recaptures)
            if single:
                # the reference's `--speed_test` way of timing (tools/dist_test.py:189-230: host data in, synchronise, host results out):
                # the frame starts in pinned host memory, crosses PCIe into the graph's input buffer, the labels come back to the host
                hpts = pts.cpu().pin_memory()
                hlab = torch.empty((pts.shape[0],), dtype=torch.int64).pin_memory()

                def pstep():
                    lab = fg(dict(points=hpts, batch_size=1, **extra), clone=False)[0]["pred_point_sem_labels"]
                    hlab.copy_(lab, non_blocking=True)
                    torch.cuda.current_stream().synchronize()
                    return lab
                el3, lat3 = timed_steps(pstep, max(args.steps // 2, 5), 2)
                graph_leg["pcie_inclusive"] = dict(frames_per_s=max(args.steps // 2, 5) / el3, ms_per_step=1e3 * el3 / max(args.steps // 2, 5), latency=lat3,
                                                   bytes_in=int(hpts.numel() * 4), bytes_out=int(hlab.numel() * 8),
                                                   note="points from pinned host memory, labels back to pinned host memory, one synchronisation per frame")
            del fg
        except Exception as e:  # never let the graph path take the bench down: the eager leg is the value then
            graph_leg = dict(error=repr(e))
    head = graph_leg if (graph_leg and "error" not in graph_leg) else main_leg

    def reference_outputs_leg(m, ex, want_logits, steps, warmup, keys):
        """the same frame with EVERY output of the reference's eval forward materialised (lidarseg3d_amd.set_reference_outputs(True): the
        loss-only conv_logits / mimic features and an eager encoded_spconv_tensor) - the like-for-like figure beside `value`, whose forward
        computes only what out_logits / the labels need (config.outputs says so)"""
        import lidarseg3d_amd as L3
        from lidarseg3d_amd import graph as lgraph
        L3.set_reference_outputs(True)
        try:
            ops.set_precision(args.precision)
            with torch.no_grad():
                m(dict(ex), return_loss=False)
            have = {k: (k in m.point_head.forward_ret_dict) for k in keys}
            fgr = lgraph.FrameGraph(m, ex)
            el, lat = timed_steps(lambda: fgr(ex, clone=False)[0]["pred_point_sem_labels"], steps, warmup)
            out = dict(frames_per_s=steps / el, ms_per_step=1e3 * el / steps, latency={k: v for k, v in lat.items() if k != "local_elapsed_s"},
                       execution="one hipGraph per frame", outputs_materialised=dict(have, encoded_spconv_tensor=True),
                       logits_bit_identical_to_value_mode=(bool(torch.equal(fgr.logits, want_logits)) if want_logits is not None else None),
                       note="set_reference_outputs(True): conv_logits / point_features_pcamera evaluated and encoded_spconv_tensor (conv_out + its rulebook) "
                            "computed with every frame, as the reference's eval forward does")
            del fgr
            return out
        except Exception as e:
            return dict(error=repr(e))
        finally:
            L3.set_reference_outputs(False)

    def frame_stream_leg(m, kind, passes=3):
        """SURVEY.md 8(d) / tools/dist_test.py:189-230: a STREAM of different sweeps, not replays of one frame: seeds 0..7 with 120k +- 10 % points
        each (a sweep's point count changes from frame to frame) and the 34 720-point size of a real nuScenes key frame, every frame through
        graph.BucketedFrameGraph (one hipGraph per 16384-point bucket, the frame padded inside the graph's input buffers with rows that belong to
        no frame).  Timed frame by frame (host copy-in of device-resident points, replay, wait, overflow check) after one pass that captures
        the buckets; labels of the first frame are checked against the eager forward of the unpadded frame."""
        from lidarseg3d_amd import graph as lgraph
        ops.set_precision(args.precision)
        sizes = [int(round(args.points * (0.9 + 0.2 * float(np.random.Generator(np.random.PCG64(1000 + sd)).uniform())))) for sd in range(8)]
        exs = []
        for sd, npts in enumerate(sizes):
            pts_ = frames_to_points([synth.lidar_frame(npts, seed=sd, **synth.NUSC)])
            ex_ = dict(points=pts_, batch_size=1)
            if kind == "mseg3d":
                img, emb, cuv = synth.camera_inputs(npts, seed=sd, ncam=6, c_img=48, h=160, w=240, batch=1)
                ex_.update(points_cuv=torch.from_numpy(cuv).to(dev), image_features=torch.from_numpy(img).to(dev),
                           camera_semantic_embeddings=torch.from_numpy(emb).to(dev))
            exs.append(ex_)
        bfg = lgraph.BucketedFrameGraph(m, bucket_points=16384)
        with torch.no_grad():
            want = m(dict(exs[0]), return_loss=False)[0]["pred_point_sem_labels"].clone()
        same = bool(torch.equal(bfg(exs[0])[0]["pred_point_sem_labels"], want))
        for ex_ in exs[1:]:
            bfg(ex_, clone=False)
        _sync()
        ms = []
        t_all = time.perf_counter()
        for _ in range(passes):
            for ex_ in exs:
                t0 = time.perf_counter()
                bfg(ex_, clone=False)   # returns after the frame: finish() waits for its stream
                ms.append(1e3 * (time.perf_counter() - t0))
        _sync()
        total = time.perf_counter() - t_all
        ms_sorted = sorted(ms)
        out = dict(frames=len(ms), points_per_frame=sizes, buckets=sorted({bfg.bucket(n) for n in sizes}), graphs_captured=bfg.captures,
                   fallbacks_to_eager=bfg.fallbacks, recaptures=bfg.recaptures, labels_bit_identical_to_eager_unpadded_frame=same,
                   frames_per_s=len(ms) / total, median_frames_per_s=1e3 / statistics.median(ms), min_frames_per_s=1e3 / ms_sorted[-1],
                   max_frames_per_s=1e3 / ms_sorted[0], ms_per_frame=dict(median=statistics.median(ms), p95=ms_sorted[min(len(ms) - 1, int(round(0.95 * (len(ms) - 1))))],
                                                                          max=ms_sorted[-1]),
                   note="8 different synthetic sweeps (seeds 0..7) of 120k +- 10 %% points, %d passes, one hipGraph per 16384-point bucket" % passes)
        if kind == "sdseg3d":  # the size of a real nuScenes key frame: its own bucket
            pk = frames_to_points([synth.lidar_frame(34720, seed=0, **synth.NUSC)])
            exk = dict(points=pk, batch_size=1)
            for _ in range(3):
                bfg(exk, clone=False)
            elk, _ = timed_steps(lambda: bfg(exk, clone=False)[0]["pred_point_sem_labels"], 20, 2)
            out["keyframe_34720_points"] = dict(frames_per_s=20 / elk, ms_per_frame=1e3 * elk / 20, bucket=bfg.bucket(34720))
        del bfg
        return out

    ref_out_leg = None
    if graph_leg is not None and "error" not in graph_leg and single:
        ref_out_leg = reference_outputs_leg(model, dict(points=pts, batch_size=1, **extra), ref_logits, args.steps, args.warmup,
                                            ["conv_logits"] if args.model == "sdseg3d" else ["voxel_logits", "point_features_pcamera"])
    stream_leg = None
    if graph_leg is not None and "error" not in graph_leg and single and extra_modes and B == 1 and S == 1:
        try:
            stream_leg = frame_stream_leg(model, args.model)
        except Exception as e:
            stream_leg = dict(error=repr(e))
        torch.cuda.empty_cache()
    # what every rank measured, gathered over the process group: a record of N ranks can be checked rank by rank
    mine = dict(rank=rank, local_rank=local_rank, device=str(dev), device_name=(torch.cuda.get_device_name(dev) if not SIM else "hipsim (test hook)"),
                frame_seeds=[100 + i for i in my_frames], frames_per_s=head["local_frames_per_s"],
                ms_per_step=1e3 * head["latency"]["local_elapsed_s"] / args.steps)
    ranks = sharding.gather_frame_results(mine) if dist is not None else [mine]
    value_logits_cpu = ref_logits[:args.points].cpu() if (B == 1 and S == 1) else None  # frame 0 of rank 0 = the CPU baseline's frame
    stages = stage_breakdown(model, pts, B) if (S == 1 and args.model == "sdseg3d" and not SIM) else None
    stage_roof = None
    if S == 1 and B == 1 and single and not SIM:
        try:
            ops.set_precision(args.precision)
            stage_roof = stage_rooflines(model, dict(points=pts, batch_size=1, **extra), args.model, main_leg.get("census") or {})
        except Exception as e:
            stage_roof = dict(error=repr(e))

    legs = {}
    if extra_modes and single and S == 1 and B == 1 and not SIM:
        for prec in [p for p in ("f32", "bf16x8", "bf16x6", "bf16x3") if p != args.precision]:
            leg = measure(prec, args.steps, max(2, args.warmup), 1, False)
            got = model.point_head.forward_ret_dict["out_logits"]
            leg["max_rel_logit_diff_vs_value_mode"] = float((got - ref_logits).abs().max() / ref_logits.abs().max())
            leg["argmax_agreement_vs_value_mode"] = float((got.argmax(1) == ref_logits.argmax(1)).float().mean())
            legs[prec] = leg
    batched = None
    if extra_modes and single and S == 1 and B == 1 and not SIM:
        # "frames/s (B = 1 latency^-1 AND batched throughput)" in the arithmetic of `value`: fb frames collated into ONE forward
        # (collate.py:141-150: batch index in column 0), eager submission in capacity mode (the host's ~3 ms of launches are amortised over fb
        # frames), and - for more than one frame - the same batch as one hipGraph.  The reference trains / tests with samples_per_gpu = 2 (semwaymo_..._e12.py:231).
        batched = dict(precision=args.precision, execution="eager, capacity mode, one forward per step", legs=[])
        try:
            ops.set_precision(args.precision)
            for fb in (1, 2, 4, 8):
                pb, eb, _, _ = make_inputs(args.model, 1, frame_ids=list(range(fb)))
                stepb = make_step(model, pb, eb, 1, nb=fb)
                with torch.no_grad():
                    for _ in range(3):  # the first frames of a new batch shape run on worst-case capacities (8x the input rows per strided level)
                        stepb()
                torch.cuda.synchronize()
                torch.cuda.empty_cache()
                torch.cuda.reset_peak_memory_stats(dev)
                nb = max(args.steps // fb, 3)
                elb, latb = timed_steps(stepb, nb, 1)
                legb = dict(frames_per_step=fb, frames_per_s=fb * nb / elb, ms_per_step=1e3 * elb / nb, ms_per_frame=1e3 * elb / nb / fb,
                            steps=nb, peak_resident_GB=torch.cuda.max_memory_allocated(dev) / 1e9)
                if fb > 1 and not args.no_graph and detectors.CAPACITY_MODE:
                    # the same batch as ONE hipGraph (graph.FrameGraph over a collated batch: captured up to the labels, split by frame afterwards)
                    try:
                        exb = dict(points=pb, batch_size=fb, **eb)
                        from lidarseg3d_amd import graph as lgraph
                        fgb = lgraph.FrameGraph(model, exb, warmup=1)
                        for _ in range(2):
                            fgb(exb, clone=False)
                        elg, _ = timed_steps(lambda: fgb(exb, clone=False)[0]["pred_point_sem_labels"], nb, 1)
                        legb["graph"] = dict(frames_per_s=fb * nb / elg, ms_per_step=1e3 * elg / nb, ms_per_frame=1e3 * elg / nb / fb,
                                             fallbacks=fgb.fallbacks, recaptures=fgb.recaptures)
                        del fgb
                    except Exception as e:
                        legb["graph"] = dict(error=repr(e))
                batched["legs"].append(legb)
                del pb, eb
        except Exception as e:
            batched["error"] = repr(e)
        torch.cuda.empty_cache()
    throughput = None
    if TP:
        # two frames in flight per GPU, one HIP stream each: the host-side launch work of one frame (the start of a frame is
        # launch-bound) and its latency-bound reader / head kernels run beside the conv stack of the other
        throughput = {"streams": TP, "frames_in_flight": TP}
        for prec in sorted({args.precision, "f32"}):
            throughput[prec + "_frames_per_s"] = measure(prec, args.steps, max(2, args.warmup // 2), TP, False)["frames_per_s"]
        if not args.no_graph and detectors.CAPACITY_MODE:
            # the same with one hipGraph per frame slot, each captured on its own stream (graph.FrameGraph(stream=...)): the replays of
            # the two slots overlap on the GPU; the logits of slot 0 are checked against the eager forward of its frame
            try:
                from lidarseg3d_amd import graph as lgraph
                ops.set_precision(args.precision)
                exs = [dict(points=p, batch_size=1) for p in spts[:TP]]
                with torch.no_grad():
                    model(dict(exs[0]), return_loss=False)
                want = model.point_head.forward_ret_dict["out_logits"].clone()
                fgs = [lgraph.FrameGraph(model, ex, stream=st) for ex, st in zip(exs, streams[:TP])]

                def step_g():
                    cur = torch.cuda.current_stream(dev)
                    for st, fg, ex in zip(streams[:TP], fgs, exs):
                        st.wait_stream(cur)
                        with torch.cuda.stream(st):
                            fg.launch(ex)
                    labels = None
                    for fg, ex in zip(fgs, exs):
                        labels = fg.finish(ex, clone=False)[0]["pred_point_sem_labels"]
                    return labels
                elg, latg = timed_steps(step_g, args.steps, max(2, args.warmup // 2))
                throughput["graph_" + args.precision + "_frames_per_s"] = TP * args.steps / elg
                throughput["graph_ms_per_step_of_%d_frames" % TP] = 1e3 * elg / args.steps
                throughput["graph_logits_bit_identical_to_eager"] = bool(torch.equal(fgs[0].logits, want))
                throughput["graph_fallbacks"] = sum(fg.fallbacks for fg in fgs)
                del fgs
            except Exception as e:
                throughput["graph_error"] = repr(e)
    mseg = None
    if extra_modes and single and args.model == "sdseg3d" and S == 1 and B == 1 and not SIM:
        # BASELINE configs[2] in the same driver-timed run: MSeg3D = + 6-camera feature maps, GF-/SF-Phase head
        del model
        torch.cuda.empty_cache()
        ops.set_precision(args.precision)
        m2, _ = build_model(dev, kind="mseg3d")
        p2, e2, _, _ = make_inputs("mseg3d", 1)
        n2 = max(args.steps // 2, 5)
        census2 = ConvCensus(ops).run(make_step(m2, p2, e2, 1))
        el2, lat2 = timed_steps(make_step(m2, p2, e2, 1), n2, 3)
        mseg = dict(metric="frames/sec, MSeg3D forward (LiDAR + 6-cam HRNet-w18 features [1,6,48,160,240], GF+SF-Phase), 120k-pt frame",
                    precision=args.precision, value=n2 / el2, ms_per_step=1e3 * el2 / n2, latency=lat2, steps=n2, execution="eager")
        if not args.no_graph and detectors.CAPACITY_MODE:
            try:
                from lidarseg3d_amd import graph as lgraph
                ex2 = dict(points=p2, batch_size=1, **e2)
                fg2 = lgraph.FrameGraph(m2, ex2)
                el2g, lat2g = timed_steps(lambda: fg2(ex2, clone=False)[0]["pred_point_sem_labels"], n2, 3)
                mseg.update(eager_value=mseg["value"], eager_ms_per_step=mseg["ms_per_step"], value=n2 / el2g, ms_per_step=1e3 * el2g / n2,
                            latency=lat2g, execution="one hipGraph per frame")
                del fg2
            except Exception as e:
                mseg["graph_error"] = repr(e)
            try:
                # two frames in flight, one hipGraph per slot on its own stream: the decoder at the end of one frame (one 4-wave workgroup per CU,
                # a third of the matrix pipe) runs beside the convolution stack of the other
                sts = [torch.cuda.Stream(dev) for _ in range(2)]
                exs2 = []
                for k in range(2):
                    pk_, ek_, _, _ = make_inputs("mseg3d", 1, frame_ids=[k], seed0=300)
                    exs2.append(dict(points=pk_, batch_size=1, **ek_))
                fgs2 = [lgraph.FrameGraph(m2, ex, stream=st) for ex, st in zip(exs2, sts)]

                def step_g2():
                    cur = torch.cuda.current_stream(dev)
                    for st, fg, ex in zip(sts, fgs2, exs2):
                        st.wait_stream(cur)
                        with torch.cuda.stream(st):
                            fg.launch(ex)
                    labels = None
                    for fg, ex in zip(fgs2, exs2):
                        labels = fg.finish(ex, clone=False)[0]["pred_point_sem_labels"]
                    return labels
                el2t, _ = timed_steps(step_g2, n2, 2)
                mseg["two_graphs_in_flight"] = dict(frames_per_s=2 * n2 / el2t, ms_per_step_of_2_frames=1e3 * el2t / n2,
                                                    fallbacks=sum(fg.fallbacks for fg in fgs2))
                del fgs2, exs2
            except Exception as e:
                mseg["two_graphs_in_flight"] = dict(error=repr(e))
        if "graph_error" not in mseg and not args.no_graph and detectors.CAPACITY_MODE:
            mseg["reference_outputs_mode"] = reference_outputs_leg(m2, dict(points=p2, batch_size=1, **e2), None, n2, 3,
                                                                   ["voxel_logits", "point_features_pcamera"])
        if "graph_error" not in mseg and not args.no_graph and detectors.CAPACITY_MODE:
            try:
                mseg["frame_stream"] = frame_stream_leg(m2, "mseg3d", passes=2)
            except Exception as e:
                mseg["frame_stream"] = dict(error=repr(e))
            torch.cuda.empty_cache()
        try:  # configs[2]'s own roofline objects: the fused SF-Phase decoder (MFMA-bound) first, then the other stages of the frame
            sr2 = stage_rooflines(m2, dict(points=p2, batch_size=1, **e2), "mseg3d", census2)
            if "sffm_decoder" in sr2:
                mseg["roofline"] = dict(dict(kernel="k_sffm_decoder"), **sr2["sffm_decoder"])
                mseg["roofline"]["kernel"] += " (SF-Phase decoder, 6 layers in one launch)"
            mseg["stage_rooflines"] = sr2
        except Exception as e:
            mseg["stage_rooflines"] = dict(error=repr(e))
        # BASELINE configs[4]: MSeg3D with bf16 convolutions and fp8 (e4m3) SF-Phase attention, batches sized into the 288 GB of HBM.
        # Narrower arithmetic than the reference's fp32 (tolerance vs the oracle: tests/test_gpu_parity.py::test_bf16_mode_tolerance_vs_oracle),
        # so it never feeds `value`; frames per step are collated into one forward (host-count path: FrameGraph is single-frame).
        cfg4 = dict(metric="frames/sec, MSeg3D forward, bf16 convolutions + fp8 SF-Phase attention (BASELINE configs[4]; NOT f32-grade)", legs=[])
        try:
            ops.set_precision("bf16")
            ops.set_sffm_attention("fp8")
            last = 0.0
            for fb in (1, 4, 8, 16, 32):
                pb, eb, _, _ = make_inputs("mseg3d", 1, frame_ids=list(range(fb)), seed0=300)
                exb = dict(points=pb, batch_size=fb, **eb)
                torch.cuda.reset_peak_memory_stats(dev)
                stepb = lambda: m2(dict(exb), return_loss=False)[0]["pred_point_sem_labels"]
                nb = max(args.steps // (2 * fb), 2)
                elb, _ = timed_steps(stepb, nb, 1 if fb > 8 else 2)
                fps = fb * nb / elb
                cfg4["legs"].append(dict(frames_per_step=fb, frames_per_s=fps, ms_per_step=1e3 * elb / nb,
                                         peak_resident_GB=torch.cuda.max_memory_allocated(dev) / 1e9))
                del exb, pb, eb
                torch.cuda.empty_cache()
                if fb >= 8 and fps < 1.02 * last:  # batch sizing: stop where another doubling buys < 2 %
                    cfg4["saturated_at_frames_per_step"] = fb
                    break
                last = fps
            peak = max(l["peak_resident_GB"] for l in cfg4["legs"])
            per_frame = peak / cfg4["legs"][-1]["frames_per_step"]
            cfg4["hbm_288GB_sizing"] = dict(resident_GB_per_frame=per_frame, frames_that_fit_in_250GB=int(250.0 / per_frame),
                                            note="throughput saturates long before memory does: one frame keeps the 256 CUs busy for milliseconds")
        except Exception as e:
            cfg4["error"] = repr(e)
        finally:
            ops.set_precision(args.precision)
            ops.set_sffm_attention("f32")
        del m2
    else:
        cfg4 = None
    ops.set_precision(args.precision)
    # BASELINE configs[3] in the same run, for every N: the data-parallel training step under DDP + SyncBN (a process group of one rank when
    # N = 1 and the bench was started by a launcher; a plain model otherwise)
    train = None
    if not args.no_train_leg and args.model == "sdseg3d" and S == 1 and B == 1 and (extra_modes or world > 1 or SIM):
        try:
            model = None
            if not SIM:
                torch.cuda.empty_cache()
            train = train_leg(args, dist, dev, rank, world, steps=(1 if SIM else max(3, args.steps // 4)), warmup=(0 if SIM else 2))
        except Exception as e:
            train = dict(error=repr(e))
        if not SIM:
            torch.cuda.empty_cache()
    ops.set_precision(args.precision)

    if rank == 0:
        c = main_leg.get("census") or {}
        stack = main_leg.get("conv_stack_ms") or {}
        mean_ms = stack.get("mean", 0.0)
        achieved = c["algo_bytes"] / (mean_ms * 1e-3) / 1e9 if (mean_ms and c) else 0.0
        out = {
            "metric": "frames/sec, SDSeg3D forward, 120k-pt nuScenes-style frame",
            "value": head["frames_per_s"], "unit": "frames/s", "n_gpus": dist.get_world_size() if dist is not None else 1, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPES[args.precision], "data": "synthetic",
            "config": {"workload": "nuScenes LiDAR-only SDSeg3D (TransVFE->UNetSCN3D->PointSegBatchlossHead), "
                                   "%d pts/frame, voxel [0.1,0.1,0.2], range [-51.2,-51.2,-5,51.2,51.2,3], 17 classes, "
                                   "1 frame per GPU per step, GPU voxelization included; loss-only outputs (conv_logits, mimic) not evaluated and "
                                   "encoded_spconv_tensor computed on demand (see config.outputs / reference_outputs_mode)" % args.points,
                       "outputs": "out_logits [N,17] + pred_point_sem_labels, the outputs the metric is defined on.  Inference does NOT evaluate the "
                                  "loss-only outputs of the reference's forward (forward_ret_dict['conv_logits']; MSeg3D: the mimic features) and hands out "
                                  "batch_dict['encoded_spconv_tensor'] as an on-demand proxy (no head reads it): `reference_outputs_mode` is the same frame "
                                  "with all of them materialised (lidarseg3d_amd.set_reference_outputs(True))",
                       "precision": args.precision, "frames_per_gpu_per_step": B * S, "streams": S,
                       "host_syncs_per_frame": ("0 blocking (capacity mode: device-side row counts; one wait for the frame's rulebook counts, "
                                                "which are ready early in the frame)" if detectors.CAPACITY_MODE else "3 (host-side row counts)"),
                       "parallelism": "frames sharded 1/GPU (dp%d), no data-path collective" % world},
            "latency": {k: v for k, v in head["latency"].items() if k != "local_elapsed_s"},
            # the process group as the collective library saw it: world size from dist.get_world_size(), one entry per rank
            "rccl_world_size": dist.get_world_size() if dist is not None else 1,
            "collective_backend": (dist.get_backend() if dist is not None else None),
            "rccl_version": (list(torch.cuda.nccl.version()) if (dist is not None and not SIM) else None),
            "ranks": ranks,
        }
        if SIM:
            out["data"] = "synthetic (LS3D_BENCH_HIPSIM test hook: kernels emulated on the host - not a measurement)"
        out["config"]["execution"] = ("one hipGraph per frame (capacity mode: device-side row counts, both streams captured); the host copies the "
                                      "frame in, replays, waits for the frame and reads its overflow flags" if head is graph_leg
                                      else "eager: every launch submitted from Python")
        if graph_leg is not None:
            out["graph_mode"] = graph_leg
            out["eager_mode"] = dict(value=main_leg["frames_per_s"], ms_per_step=main_leg["ms_per_step"], latency=main_leg["latency"],
                                     note="same launches submitted one by one from Python; carries the conv-stack HIP-event brackets of `roofline`")
        if ref_out_leg is not None:
            out["reference_outputs_mode"] = ref_out_leg
            # the like-for-like figure as a first-class key: the same frame with EVERY tensor the reference's eval forward produces materialised
            out["reference_outputs_value"] = ref_out_leg.get("frames_per_s")
        if stream_leg is not None:
            out["frame_stream"] = stream_leg
            out["frame_stream_median_value"] = stream_leg.get("median_frames_per_s")
            out["frame_stream_min_value"] = stream_leg.get("min_frames_per_s")
        if c:
            np_ = PLANE_PRODUCTS.get(args.precision)
            # the matrix-pipe view of the same stack: every f32 product of the pair model is `np_` bf16 plane products on
            # v_mfma_f32_32x32x16_bf16 (exact f32: one product on v_mfma_f32_32x32x2_f32); "useful" counts the pairs of the rulebooks,
            # not the dense-over-pairs work the tiles execute on absent neighbours
            if np_:
                mfma = dict(unit="TFLOP/s", dtype="bf16", plane_products_per_f32_product=np_, achieved=np_ * main_leg["tflops"], peak=BF16_MFMA_PEAK_TFLOPS,
                            frac=np_ * main_leg["tflops"] / BF16_MFMA_PEAK_TFLOPS, f32_equivalent_tflops=main_leg["tflops"])
            else:
                mfma = dict(unit="TFLOP/s", dtype="f32", achieved=main_leg["tflops"], peak=F32_MFMA_PEAK_TFLOPS, frac=main_leg["tflops"] / F32_MFMA_PEAK_TFLOPS)
            out["roofline"] = {
                # what the counters say (profiles/round3_pmc_sq.md): the stack is bound by the matrix pipe + its LDS operand feed, HBM moves
                # ~0.2x of the pair model.  achieved / peak / frac stay SURVEY.md 8(d)'s pair-model figure (algorithmic gather bytes over
                # the measured duration against the HBM peak), the headline the scope table defines; `mfma` and `hbm_physical_GBps` are the
                # physical utilisations of the two units.
                "bound": "mfma", "kernel": "sparse-conv stack: %d layers/frame (%d on k_tile_conv - %d of them inside chained persistent launches, "
                                           "ls3d_tile_conv_chain - + %d on k_gather_gemm) = %d kernel launches, two HIP-event brackets per frame"
                                           % (c["launches"], c["tile_launches"], c.get("chained_layers", 0), c["launches"] - c["tile_launches"],
                                              c.get("kernel_launches", c["launches"])),
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                "model": "pair model, SURVEY.md 8(d): sum over the layers of P_l * (Cin + Cout) * 4 bytes",
                # per LAYER (37 per frame; the unit of the pair model) and per kernel launch (a chained launch runs several layers)
                "avg_launch_us": 1e3 * mean_ms / max(c["launches"], 1), "avg_kernel_launch_us": 1e3 * mean_ms / max(c.get("kernel_launches", c["launches"]), 1),
                "layers_per_frame": c["launches"], "kernel_launches_per_frame": c.get("kernel_launches", c["launches"]),
                "algo_bytes_per_frame": c["algo_bytes"],
                "algo_bytes_per_launch": c["algo_bytes"] / max(c["launches"], 1), "tflops_useful": main_leg.get("tflops"),
                "sparse_conv_ms_per_frame": stack, "mfma": mfma, "hbm_physical_GBps": None,
                # the traffic no kernel can avoid (each referenced input row, each output row and the weights once per launch) and the
                # HBM time it stands for
                "b_min": {"bytes_per_frame": c["b_min_bytes"], "ms_at_hbm_peak": 1e3 * c["b_min_bytes"] / (HBM_PEAK_GBS * 1e9),
                          "achieved_GBps": c["b_min_bytes"] / (mean_ms * 1e-3) / 1e9 if mean_ms else 0.0,
                          "frac": c["b_min_bytes"] / (mean_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if mean_ms else 0.0},
            }
            if args.model == "sdseg3d" and args.points == 120000 and os.path.exists(PMC_RECORD):
                # counters cannot be collected from inside the timed run: separate rocprofv3 --pmc passes of this same command on the
                # same build (tools/collect_profiles.sh; the record names the commit and the kernels it saw)
                j = json.load(open(PMC_RECORD))
                r = j.get(args.precision)
                if r:
                    # the counters are per KERNEL launch (a chained launch runs several layers); per frame = x kernel launches, and per LAYER
                    # (the unit of `achieved`'s algorithmic bytes) = the frame's bytes / layers
                    frame_bytes = r["traffic_bytes_per_launch"] * c.get("kernel_launches", c["launches"])
                    out["roofline"]["traffic"] = frame_bytes / max(c["launches"], 1)
                    out["roofline"]["traffic_per_kernel_launch"] = r["traffic_bytes_per_launch"]
                    out["roofline"]["traffic_source"] = os.path.relpath(PMC_RECORD, ROOT) + ": (2*FETCH_SIZE+WRITE_SIZE)*1024 per sparse-conv kernel launch, corrected per MI355X_MICROARCH.md, x kernel launches per frame / layers per frame; kernels " + ", ".join(r.get("kernels", []))
                    out["roofline"]["hbm_physical_GBps"] = frame_bytes / (mean_ms * 1e-3) / 1e9 if mean_ms else None
                    if r.get("mfma_busy") is not None:
                        out["roofline"]["mfma"]["mfma_busy"] = r["mfma_busy"]
                        out["roofline"]["mfma"]["mfma_busy_source"] = os.path.relpath(PMC_RECORD, ROOT).replace(".json", "_sq.md") + ": SQ_VALU_MFMA_BUSY_CYCLES / SIMD cycles, time-weighted over the stack's kernels"
        if stages is not None:
            out["stages_ms"] = stages
        if stage_roof is not None:
            out["stage_rooflines"] = stage_roof
        for prec, leg in legs.items():
            key = {"f32": "exact_f32_mode", "bf16x3": "fast_mode", "bf16x8": "f32_grade_8_product_mode", "bf16x6": "f32_grade_6_product_mode"}[prec]
            out[key] = dict(precision=DTYPES[prec], value=leg["frames_per_s"], ms_per_step=leg["ms_per_step"], latency=leg["latency"],
                            sparse_conv_ms_per_frame=leg.get("conv_stack_ms"), roofline_frac=leg.get("roofline_frac"),
                            max_rel_logit_diff_vs_value_mode=leg["max_rel_logit_diff_vs_value_mode"],
                            argmax_agreement_vs_value_mode=leg["argmax_agreement_vs_value_mode"])
            if prec == "f32" and leg.get("tflops"):
                out[key]["mfma"] = dict(achieved_tflops=leg["tflops"], peak_tflops=F32_MFMA_PEAK_TFLOPS, frac=leg["tflops"] / F32_MFMA_PEAK_TFLOPS)
        if batched is not None:
            out["batched"] = batched
        if throughput is not None:
            out["throughput_mode"] = throughput
        if train is not None:
            out["train_step"] = train
        if mseg is not None:
            out["mseg3d"] = mseg
        if cfg4 is not None:
            out["mseg3d_bf16_fp8"] = cfg4
        if args.model == "mseg3d":
            out["metric"] = "frames/sec, MSeg3D forward (LiDAR + 6-cam features), 120k-pt nuScenes-style frame"
            out["config"]["workload"] = out["config"]["workload"].replace(
                "nuScenes LiDAR-only SDSeg3D (TransVFE->UNetSCN3D->PointSegBatchlossHead)",
                "nuScenes MSeg3D (ImprovedMeanVFE->UNetSCN3D->PointSegMSeg3DHead GF+SF-Phase, image_features [1,6,48,160,240])")
        if single and not args.no_cpu_baseline and args.model == "sdseg3d" and not SIM:
            out["cpu_baseline"] = cpu_baseline(args.cpu_points or args.points, 100, value_logits_cpu)
            if "parity" in out["cpu_baseline"]:  # GPU logits of the TIMED frame (value's arithmetic) vs the CPU oracle's
                out["parity_vs_cpu"] = out["cpu_baseline"].pop("parity")
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        # RCCL writes its version banner to the C stdout buffer, which would otherwise be flushed AFTER this line at exit: flush it
        # first so that the JSON record is the last line of the output
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
