"""create / replay / destroy plain FrameGraphs in a loop (mode: drop | keep)"""
import sys, os, faulthandler, gc
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import lidarseg3d_amd as L
from lidarseg3d_amd import graph, models_cfg, ops, synth
dev = torch.device("cuda:0")
ops.set_precision("bf16x6")
mode = sys.argv[1] if len(sys.argv) > 1 else "drop"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
keep = []
def make(kind, n, seed):
    model = L.build_detector(getattr(models_cfg, kind)(), train_cfg=None, test_cfg={}).eval()
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.random_state_dict(shapes, 5).items()})
    model.to(dev)
    f = synth.lidar_frame(n, seed=seed, **synth.NUSC)
    ex = dict(points=torch.from_numpy(np.concatenate([np.zeros((n, 1), np.float32), f], 1)).to(dev), batch_size=1)
    if kind == "mseg3d":
        img, emb, cuv = synth.camera_inputs(n, seed=seed, ncam=6, c_img=48, h=40, w=60, batch=1)
        ex.update(points_cuv=torch.from_numpy(cuv).to(dev), image_features=torch.from_numpy(img).to(dev), camera_semantic_embeddings=torch.from_numpy(emb).to(dev))
    return model, ex
for rep in range(reps):
    for kind in ("sdseg3d", "mseg3d"):
        model, ex = make(kind, 60000 + 1000 * rep, rep)
        fg = graph.FrameGraph(model, ex)
        for _ in range(4):
            fg(ex)
        if mode == "keep":
            keep.append((fg, model))
        else:
            del fg, model
        print("ok", rep, kind, "mem GB %.1f" % (torch.cuda.memory_reserved() / 1e9), flush=True)
    if mode == "gc":
        gc.collect(); torch.cuda.empty_cache()
print("DONE", flush=True)
