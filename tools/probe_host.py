#!/usr/bin/env python
"""Where does a frame's time go on the host?  (measurement helper)  eager capacity-mode forward: host seconds per call without any
synchronisation vs GPU time per frame; FrameGraph: host time of copy + replay, GPU time between the events around the replay."""
import os, sys, time, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from lidarseg3d_amd import ops, synth, graph, detectors
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
ops.set_precision("bf16x6")
model, _ = bench.build_model(dev)
f = synth.lidar_frame(120000, seed=100, **synth.NUSC)
pts = torch.from_numpy(np.concatenate([np.zeros((f.shape[0], 1), np.float32), f], 1)).to(dev)
ex = dict(points=pts, batch_size=1)
out = {}
with torch.no_grad():
    for _ in range(5): model(dict(ex), return_loss=False)
    torch.cuda.synchronize()
    K = 30
    t0 = time.perf_counter(); host = []
    for _ in range(K):
        a = time.perf_counter(); model(dict(ex), return_loss=False); host.append(time.perf_counter() - a)
    th = time.perf_counter() - t0
    torch.cuda.synchronize(); tt = time.perf_counter() - t0
    out["eager"] = dict(host_ms_per_call_mean=1e3 * th / K, host_ms_per_call_median=1e3 * sorted(host)[K // 2], total_ms_per_frame=1e3 * tt / K)
    # the same with the end-of-frame count check removed (pure submission time)
    bb = model.backbone; orig = bb.geometry_check
    bb.geometry_check = lambda d: (d.pop("geometry_record", None), True)[1]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K): model(dict(ex), return_loss=False)
    th = time.perf_counter() - t0; torch.cuda.synchronize(); tt = time.perf_counter() - t0
    out["eager_no_check"] = dict(host_ms_per_call=1e3 * th / K, total_ms_per_frame=1e3 * tt / K)
    bb.geometry_check = orig
    fg = graph.FrameGraph(model, ex)
    for _ in range(3): fg(ex, clone=False)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
    torch.cuda.synchronize(); t0 = time.perf_counter(); hr = []
    for a, b in evs:
        t1 = time.perf_counter(); a.record(); fg.graph.replay(); b.record(); hr.append(time.perf_counter() - t1)
        torch.cuda.current_stream().synchronize()
    tt = time.perf_counter() - t0
    gpu = sorted(a.elapsed_time(b) for a, b in evs)
    out["graph"] = dict(replay_host_ms_median=1e3 * sorted(hr)[K // 2], gpu_ms_median=gpu[K // 2], gpu_ms_min=gpu[0], total_ms_per_frame=1e3 * tt / K)
    # back-to-back replays without a sync in between (GPU-side time per frame of the captured work)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K): fg.graph.replay()
    torch.cuda.synchronize(); out["graph_back_to_back_ms_per_frame"] = 1e3 * (time.perf_counter() - t0) / K
print(json.dumps(out, indent=1))
