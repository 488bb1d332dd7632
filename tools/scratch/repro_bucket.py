import sys, os, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import lidarseg3d_amd as L
from lidarseg3d_amd import graph, models_cfg, ops, synth
kind = sys.argv[1] if len(sys.argv) > 1 else "sdseg3d"
base = int(sys.argv[2]) if len(sys.argv) > 2 else 66000
clone = (sys.argv[3] != "noclone") if len(sys.argv) > 3 else True
dev = torch.device("cuda:0")
model = L.build_detector(getattr(models_cfg, kind)(), train_cfg=None, test_cfg={}).eval()
shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.random_state_dict(shapes, 5).items()})
model.to(dev)
ops.set_precision("bf16x6")
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
        want.append(model(dict(ex), return_loss=False)[0]["pred_point_sem_labels"].clone())
print("eager done", sizes, flush=True)
bfg = graph.BucketedFrameGraph(model, bucket_points=16384)
for it in range(2):
    for i, (ex, w) in enumerate(zip(exs, want)):
        print("pass", it, "frame", i, "n", sizes[i], "bucket", bfg.bucket(sizes[i]), "captures", bfg.captures, flush=True)
        out = bfg(ex, clone=clone)
        torch.cuda.synchronize()
        assert torch.equal(out[0]["pred_point_sem_labels"], w), "labels differ"
print("OK fallbacks", bfg.fallbacks, "captures", bfg.captures, flush=True)
