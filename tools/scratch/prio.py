"""does a high-priority main stream keep the side streams' geometry out of the convolutions' way?  eager and hipGraph frames, SDSeg3D / MSeg3D"""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from lidarseg3d_amd import graph as lgraph, ops, synth
dev = torch.device("cuda:0")
ops.set_precision("bf16x6")
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "?")
for kind in ("sdseg3d", "mseg3d"):
    model, _ = bench.build_model(dev, kind=kind)
    f = synth.lidar_frame(120000, seed=100, **synth.NUSC)
    pts = torch.from_numpy(np.concatenate([np.zeros((len(f), 1), np.float32), f], 1)).to(dev)
    ex = dict(points=pts, batch_size=1)
    if kind == "mseg3d":
        img, emb, cuv = synth.camera_inputs(120000, seed=100, ncam=6, c_img=48, h=160, w=240, batch=1)
        ex.update(points_cuv=torch.from_numpy(cuv).to(dev), image_features=torch.from_numpy(img).to(dev), camera_semantic_embeddings=torch.from_numpy(emb).to(dev))

    def run(fn, st, n=30):
        ts = []
        with torch.cuda.stream(st), torch.no_grad():
            for i in range(n + 5):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record(st); fn(); e.record(st); st.synchronize()
                if i >= 5:
                    ts.append(s.elapsed_time(e))
        return statistics.median(ts)
    for name, prio in (("normal", 0), ("high", -1), ("normal", 0), ("high", -1)):
        st = torch.cuda.Stream(dev, priority=prio)
        t_e = run(lambda: model(dict(ex), return_loss=False), st)
        with torch.cuda.stream(st):
            fg = lgraph.FrameGraph(model, ex, stream=st)
        t_g = run(lambda: fg(ex, clone=False), st)
        print("%-8s main stream priority %-6s (%d): eager %.3f ms, hipGraph %.3f ms" % (kind, name, st.priority, t_e, t_g), flush=True)
