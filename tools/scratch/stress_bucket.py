import sys, os, faulthandler, gc
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import lidarseg3d_amd as L
from lidarseg3d_amd import graph, models_cfg, ops, synth
dev = torch.device("cuda:0")
ops.set_precision("bf16x6")
VAR = sys.argv[2] if len(sys.argv) > 2 else ""
KEEP = []
def scenario(kind, base, rep):
    model = L.build_detector(getattr(models_cfg, kind)(), train_cfg=None, test_cfg={}).eval()
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.random_state_dict(shapes, 5).items()})
    model.to(dev)
    sizes = [int(round(base * (0.9 + 0.2 * float(np.random.Generator(np.random.PCG64(1000 + sd)).uniform())))) for sd in range(8)]
    exs = []
    for sd, n in enumerate(sizes):
        f = synth.lidar_frame(n, seed=sd, **synth.NUSC)
        ex = dict(points=torch.from_numpy(np.concatenate([np.zeros((n, 1), np.float32), f], 1)).to(dev), batch_size=1, metadata=[dict(token="s%d" % sd)])
        if kind == "mseg3d":
            img, emb, cuv = synth.camera_inputs(n, seed=sd, ncam=6, c_img=48, h=40, w=60, batch=1)
            ex.update(points_cuv=torch.from_numpy(cuv).to(dev), image_features=torch.from_numpy(img).to(dev), camera_semantic_embeddings=torch.from_numpy(emb).to(dev))
        exs.append(ex)
    want = []
    with torch.no_grad():
        for ex in exs:
            want.append(model(dict(ex), return_loss=False)[0]["pred_point_sem_labels"].clone() if "noeager" not in VAR else None)
    # a plain FrameGraph first (like the other graph tests of the suite), dropped before the buckets are captured
    if "noplain" not in VAR:
        fg = graph.FrameGraph(model, exs[0])
        for _ in range(3):
            fg(exs[0])
        del fg
    bfg = graph.BucketedFrameGraph(model, bucket_points=16384)
    if "sharedpool" in VAR:
        bfg.pool = torch.cuda.graph_pool_handle()
    for it in range(3):
        for ex, w in zip(exs, want):
            out = bfg(ex)
            assert w is None or torch.equal(out[0]["pred_point_sem_labels"], w)
    if "keep" in VAR:
        KEEP.append((bfg, model))
    if "keepgraphs" in VAR:  # keep only the CUDAGraph objects alive, drop everything else
        KEEP.extend(g.graph for g in bfg.graphs.values())
    cap, fb = bfg.captures, bfg.fallbacks
    if "empty" in VAR:
        torch.cuda.synchronize(); bfg.graphs.clear(); gc.collect(); torch.cuda.empty_cache()
    if "resetonly" in VAR:
        torch.cuda.synchronize()
        for g in bfg.graphs.values():
            g.graph.reset()
    if "devsync" in VAR:
        torch.cuda.synchronize(); bfg.graphs.clear(); gc.collect(); torch.cuda.synchronize()
    if "reverse" in VAR:
        torch.cuda.synchronize()
        for k in list(bfg.graphs)[::-1]:
            del bfg.graphs[k]
            gc.collect()
    print("ok", kind, base, rep, "captures", cap, "fallbacks", fb, "mem GB %.1f" % (torch.cuda.memory_reserved() / 1e9), flush=True)
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    for kind in ("sdseg3d", "mseg3d"):
        scenario(kind, 60000 if "onebucket" in VAR else 66000, rep)
    gc.collect()
print("DONE", flush=True)
