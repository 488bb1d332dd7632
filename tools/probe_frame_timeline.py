"""When does every ops.* call of an eager capacity-mode frame start and end, on which stream?  (rocprofv3 serialises the kernels of a frame onto one
queue, so its trace cannot show what the side streams - geometry, lateral blocks, neighbour search - overlap with.)  Every public function of
lidarseg3d_amd.ops is wrapped for the measurement: an event in front of the call and one behind it on the stream that is current at the call;
times are relative to the frame's start event, median over the timed frames.  `wait` = the call's start minus the end of the previous call on the
same stream (what the stream waited for: an event of another stream, or the host).

    python tools/probe_frame_timeline.py [--model sdseg3d|mseg3d] [--min-us 5]"""
import argparse
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench

SKIP = ("set_", "get_", "use_", "make_", "registered_host", "check", "precision", "planes")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="sdseg3d")
    ap.add_argument("--min-us", type=float, default=5.0)
    ap.add_argument("--frames", type=int, default=7)
    a = ap.parse_args()
    from lidarseg3d_amd import ops, synth
    dev = torch.device("cuda:0")
    ops.set_precision("bf16x6")
    model, _ = bench.build_model(dev, kind=a.model)
    f = synth.lidar_frame(120000, seed=100, **synth.NUSC)
    pts = torch.from_numpy(np.concatenate([np.zeros((len(f), 1), np.float32), f], 1)).to(dev)
    ex = dict(points=pts, batch_size=1)
    if a.model == "mseg3d":
        img, emb, cuv = synth.camera_inputs(120000, seed=100, ncam=6, c_img=48, h=160, w=240, batch=1)
        ex.update(points_cuv=torch.from_numpy(cuv).to(dev), image_features=torch.from_numpy(img).to(dev), camera_semantic_embeddings=torch.from_numpy(emb).to(dev))
    with torch.no_grad():
        for _ in range(3):
            model(dict(ex), return_loss=False)
    torch.cuda.synchronize()
    log = []
    depth = [0]

    def wrap(name, fn):
        def inner(*args, **kw):
            if depth[0]:  # a wrapped function called by a wrapped function: the outer pair covers it
                return fn(*args, **kw)
            st = torch.cuda.current_stream(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            depth[0] += 1
            try:
                return fn(*args, **kw)
            finally:
                depth[0] -= 1
                e1.record(st)
                log.append((name, st.cuda_stream, e0, e1))
        return inner

    import types
    for name, fn in list(vars(ops).items()):
        if isinstance(fn, types.FunctionType) and fn.__module__ == ops.__name__ and not name.startswith("_") and not name.startswith(SKIP):
            setattr(ops, name, wrap(name, fn))
    frames = []
    with torch.no_grad():
        for _ in range(a.frames):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            mark = len(log)
            s.record()
            model(dict(ex), return_loss=False)
            e.record()
            frames.append((s, e, mark, len(log)))
    torch.cuda.synchronize()
    per = frames[0][3] - frames[0][2]
    if any(fr[3] - fr[2] != per for fr in frames):
        print("frames differ in their call sequences:", [fr[3] - fr[2] for fr in frames])
    fr_ms = [s.elapsed_time(e) for s, e, _, _ in frames]
    print("%s eager frame with the probes: median %.3f ms (%d wrapped calls per frame)" % (a.model, statistics.median(fr_ms), per))
    streams = {}
    rows = []
    for i in range(per):
        name, sid = log[frames[0][2] + i][0], log[frames[0][2] + i][1]
        t0 = statistics.median(fr[0].elapsed_time(log[fr[2] + i][2]) for fr in frames) * 1e3
        t1 = statistics.median(fr[0].elapsed_time(log[fr[2] + i][3]) for fr in frames) * 1e3
        rows.append((t0, t1, streams.setdefault(sid, len(streams)), name))
    last = {}
    for t0, t1, s, name in rows:  # in host call order
        wait = t0 - last.get(s, 0.0)
        last[s] = t1
        if t1 - t0 >= a.min_us or wait >= a.min_us:
            print("s%d  %8.1f -> %8.1f us   dur %7.1f   wait %7.1f   %s" % (s, t0, t1, t1 - t0, wait, name))


if __name__ == "__main__":
    main()
