"""which Python lines of a Waymo MSeg3D training step call a given torch op (default aten::sum) on the GPU: torch.profiler with stacks (measurement helper)"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import lidarseg3d_amd as L
from lidarseg3d_amd import models_cfg, ops, synth


def main():
    want = sys.argv[1:] or ["aten::sum"]
    cfg, ncls, ncam, B, npts = synth.WAYMO, 23, 5, 2, 180000
    dev = "cuda:0"; torch.cuda.set_device(0); torch.manual_seed(0)
    ops.set_precision("bf16x6")
    model = L.build_detector(models_cfg.mseg3d(num_class=ncls, pc_range=cfg["pc_range"], voxel_size=cfg["voxel_size"]), train_cfg=None, test_cfg={}).to(dev).train()
    frames = [synth.lidar_frame(npts, seed=b, **cfg) for b in range(B)]
    pts = torch.from_numpy(np.concatenate([np.concatenate([np.full((f.shape[0], 1), b, np.float32), f], 1) for b, f in enumerate(frames)])).to(dev)
    v, c, n, nv = ops.voxelize_hard(pts, cfg["voxel_size"], cfg["pc_range"], 5, 300000 * B, batched=True)
    V = int(nv)
    ex = dict(points=pts, voxels=v[:V], coordinates=c[:V], num_points=n[:V], num_voxels=[0] * B, shape=[np.asarray(ops.make_grid(cfg["voxel_size"], cfg["pc_range"])[1])],
              voxel_sem_labels=torch.randint(0, ncls, (V,), device=dev), point_sem_labels=torch.randint(0, ncls, (pts.shape[0],), device=dev))
    img, emb, cuv = synth.camera_inputs(pts.shape[0], seed=0, ncam=ncam, c_img=48, h=160, w=240, num_class=ncls, batch=B)
    ex.update(image_features=torch.from_numpy(img).to(dev), camera_semantic_embeddings=torch.from_numpy(emb).to(dev), points_cuv=torch.from_numpy(cuv).to(dev))
    for _ in range(2):
        model.zero_grad(set_to_none=True)
        model(dict(ex), return_loss=True)["loss"][0].backward()
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        model.zero_grad(set_to_none=True)
        model(dict(ex), return_loss=True)["loss"][0].backward()
        torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0, 0.0])
    for e in prof.events():
        if e.name in want:
            st = [s for s in e.stack if "lidarseg3d_amd" in s][:3] or [s for s in e.stack][:4]
            key = (e.name, str(e.input_shapes)[:60], " <- ".join(s.split("/")[-1][:60] for s in st))
            agg[key][0] += 1
            agg[key][1] += e.device_time_total if hasattr(e, "device_time_total") else e.cuda_time_total
    for k, (cnt, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
        print("%3d x %8.1f us  %s %s | %s" % (cnt, us, k[0], k[1], k[2]))


if __name__ == "__main__":
    main()
