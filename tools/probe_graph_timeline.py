"""When does every ops.* call of a CAPTURED frame start and end during a replay of its hipGraph, on which stream?  (rocprofv3 serialises a frame's
kernels onto one queue; the eager probe - tools/probe_frame_timeline.py - also measures the host.)  Every public function of lidarseg3d_amd.ops is
wrapped for the measurement: a one-lane kernel that writes the 100 MHz wall clock (ls3d_stamp) in front of the call and one behind it, on the stream
that is current at the call - graph nodes like the others.  ~2 us per stamp: the probed frame is a few per cent longer than the plain one (both
printed).  Times are relative to the frame's first stamp, median over the replays; `wait` = start minus the end of the previous call on that stream.

    python tools/probe_graph_timeline.py [--model sdseg3d|mseg3d] [--min-us 5]"""
import argparse
import os
import statistics
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench

SKIP = ("set_", "get_", "use_", "make_", "registered_host", "check", "precision", "planes", "stamp", "tile_chain_enabled", "tile_chain_pays", "tile_products",
        "empty_rows", "conv_out_shape")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="sdseg3d")
    ap.add_argument("--min-us", type=float, default=5.0)
    ap.add_argument("--replays", type=int, default=9)
    a = ap.parse_args()
    from lidarseg3d_amd import graph as lgraph, ops, synth
    dev = torch.device("cuda:0")
    ops.set_precision("bf16x6")
    model, _ = bench.build_model(dev, kind=a.model)
    f = synth.lidar_frame(120000, seed=100, **synth.NUSC)
    pts = torch.from_numpy(np.concatenate([np.zeros((len(f), 1), np.float32), f], 1)).to(dev)
    ex = dict(points=pts, batch_size=1)
    if a.model == "mseg3d":
        img, emb, cuv = synth.camera_inputs(120000, seed=100, ncam=6, c_img=48, h=160, w=240, batch=1)
        ex.update(points_cuv=torch.from_numpy(cuv).to(dev), image_features=torch.from_numpy(img).to(dev), camera_semantic_embeddings=torch.from_numpy(emb).to(dev))

    def timed(fg, n):
        ts = []
        for _ in range(n):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            fg(ex, clone=False)
            e.record()
            torch.cuda.synchronize()
            ts.append(s.elapsed_time(e))
        return statistics.median(ts)

    plain = lgraph.FrameGraph(model, ex)
    t_plain = timed(plain, 15)
    buf = torch.zeros(8192, dtype=torch.int64, device=dev)
    calls, slot, depth = [], [0], [0]

    def wrap(name, fn):
        def inner(*args, **kw):
            if depth[0]:
                return fn(*args, **kw)
            i = slot[0]
            slot[0] += 2
            st = torch.cuda.current_stream(dev)
            ops.stamp(buf, i)
            depth[0] += 1
            try:
                return fn(*args, **kw)
            finally:
                depth[0] -= 1
                ops.stamp(buf, i + 1)
                calls.append((i, name, st.cuda_stream))
        return inner

    stamp = ops.stamp
    for name, fn in list(vars(ops).items()):
        if isinstance(fn, types.FunctionType) and fn.__module__ == ops.__name__ and not name.startswith("_") and not name.startswith(SKIP):
            setattr(ops, name, wrap(name, fn))
    fwd = model.forward

    def forward(*args, **kw):  # every forward (warm-up, capture) numbers its calls from 0
        slot[0] = 0
        del calls[:]
        return fwd(*args, **kw)
    model.forward = forward
    probed = lgraph.FrameGraph(model, ex)
    seq = list(calls)
    t_probed = timed(probed, 5)
    runs = []
    for _ in range(a.replays):
        probed(ex, clone=False)
        torch.cuda.synchronize()
        runs.append(buf.cpu().numpy().astype(np.int64).copy())
    print("%s: plain graph %.3f ms, with %d stamps %.3f ms" % (a.model, t_plain, 2 * len(seq), t_probed))
    base = [min(r[i] for i, _, _ in seq) for r in runs]
    streams, last = {}, {}
    for i, name, sid in seq:
        t0 = statistics.median((r[i] - b) / 100.0 for r, b in zip(runs, base))
        t1 = statistics.median((r[i + 1] - b) / 100.0 for r, b in zip(runs, base))
        s = streams.setdefault(sid, len(streams))
        wait = t0 - last.get(s, 0.0)
        last[s] = t1
        if t1 - t0 >= a.min_us or wait >= a.min_us:
            print("s%d  %8.1f -> %8.1f us   dur %7.1f   wait %7.1f   %s" % (s, t0, t1, t1 - t0, wait, name))


if __name__ == "__main__":
    main()
