#!/usr/bin/env python
"""One SDSeg3D (or, --model mseg3d, MSeg3D LiDAR side + fusion head) training step (forward + loss + backward + SGD update) on a synthetic 120k-point frame: the reader and the head's
MLPs on torch autograd, voxelization / sparse convolutions (forward, dgrad, wgrad) / 3-NN search on the HIP kernels.  Prints one
JSON line; not the headline benchmark (bench.py is).  SURVEY.md 8f rank 1, single GPU (the gradient all-reduce is torch DDP)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=120000)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--model", choices=["sdseg3d", "mseg3d"], default="sdseg3d")
    ap.add_argument("--geometry", choices=["nusc", "waymo"], default="nusc",
                    help="waymo = BASELINE configs[3]: range [-75.2,-75.2,-2,75.2,75.2,4], voxel [0.1,0.1,0.15], 23 classes, 5 cameras")
    ap.add_argument("--precision", choices=["f32", "bf16x6", "bf16x8"], default="f32",
                    help="arithmetic of the SubM layers' forward and dgrad (ops.set_precision); the weight gradient is exact f32 in every mode")
    ap.add_argument("--frames", type=int, default=1, help="frames per GPU per step (configs[3]: 2)")
    ap.add_argument("--syncbn", action="store_true", help="BatchNorm1d -> CountSyncBatchNorm1d (statistics over all ranks, weighted by row counts)")
    ap.add_argument("--ddp", action="store_true", help="wrap the model in DistributedDataParallel (RCCL gradient all-reduce); "
                    "launch with python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 ... (N = 1 works too)")
    ap.add_argument("--profile-ops", type=int, default=0, help="print the N most expensive torch ops of one step with their input shapes and exit")
    args = ap.parse_args()
    import lidarseg3d_amd as L
    from lidarseg3d_amd import models_cfg, ops, synth
    cfg = synth.WAYMO if args.geometry == "waymo" else synth.NUSC
    ncls, ncam = (23, 5) if args.geometry == "waymo" else (17, 6)
    rank, local_rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    torch.cuda.set_device(local_rank)
    dev = "cuda:%d" % local_rank
    torch.manual_seed(0)
    ops.set_precision(args.precision)
    mcfg = getattr(models_cfg, args.model)(num_class=ncls, pc_range=cfg["pc_range"], voxel_size=cfg["voxel_size"])
    model = L.build_detector(mcfg, train_cfg=None, test_cfg={})
    if args.syncbn:
        from lidarseg3d_amd import syncbn
        model = syncbn.convert_sync_batchnorm(model)
    model = model.to(dev).train()
    net = model
    if args.ddp:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world)
        # conv_out (spconv_down2) feeds nothing the segmentation loss sees -> find_unused_parameters
        # one flat bucket: the whole gradient is ~40 MB (SDSeg3D) and xGMI rings are per-link bound, so one large all-reduce at the
        # end of backward beats several 25 MB ones; gradients live in the bucket (no extra copy)
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank], find_unused_parameters=True,
                                                        bucket_cap_mb=128, gradient_as_bucket_view=True)
    B = args.frames
    frames = [synth.lidar_frame(args.points, seed=rank * B + b, **cfg) for b in range(B)]
    pts = torch.from_numpy(np.concatenate([np.concatenate([np.full((f.shape[0], 1), b, np.float32), f], 1) for b, f in enumerate(frames)])).to(dev)
    v, c, n, nv = ops.voxelize_hard(pts, cfg["voxel_size"], cfg["pc_range"], 5, 300000 * B, batched=True)
    V = int(nv)
    ex = dict(points=pts, voxels=v[:V], coordinates=c[:V], num_points=n[:V], num_voxels=[0] * B,
              shape=[np.asarray(ops.make_grid(cfg["voxel_size"], cfg["pc_range"])[1])],
              voxel_sem_labels=torch.randint(0, ncls, (V,), device=dev), point_sem_labels=torch.randint(0, ncls, (pts.shape[0],), device=dev))
    if args.model == "mseg3d":  # camera CNN outputs at the shipped configs' shapes (6 / 5 cameras, 48 channels, 160x240 maps)
        img, emb, cuv = synth.camera_inputs(pts.shape[0], seed=rank, ncam=ncam, c_img=48, h=160, w=240, num_class=ncls, batch=B)
        ex.update(image_features=torch.from_numpy(img).to(dev), camera_semantic_embeddings=torch.from_numpy(emb).to(dev),
                  points_cuv=torch.from_numpy(cuv).to(dev))
    opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9)
    if args.profile_ops:  # which torch ops (with input shapes) the step spends its GPU time in: what is left on torch autograd
        from torch.profiler import profile, ProfilerActivity
        for _ in range(2):
            opt.zero_grad(set_to_none=True)
            net(dict(ex), return_loss=True)["loss"][0].backward()
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
            opt.zero_grad(set_to_none=True)
            net(dict(ex), return_loss=True)["loss"][0].backward()
            torch.cuda.synchronize()
        print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=args.profile_ops, max_name_column_width=60,
                                                                  max_shapes_column_width=90))
        return
    tf = tb = to = 0.0
    losses = []
    for it in range(args.warmup + args.steps):
        opt.zero_grad(set_to_none=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        loss = net(dict(ex), return_loss=True)["loss"][0]
        torch.cuda.synchronize(); t1 = time.perf_counter()
        loss.backward()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        opt.step()
        torch.cuda.synchronize(); t3 = time.perf_counter()
        losses.append(float(loss.detach()))
        if it >= args.warmup:
            tf += t1 - t0; tb += t2 - t1; to += t3 - t2
    k = args.steps
    if args.ddp:
        t = torch.tensor([tf, tb, to], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        tf, tb, to = (float(v) for v in t.tolist())
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"what": args.model + " training step (" + args.precision + "), %s geometry, %d frame(s) per GPU, %d GPU(s)%s%s"
                                  % (args.geometry, B, world, ", DDP" if args.ddp else "", ", count-weighted SyncBN" if args.syncbn else ""),
                          "points_per_frame": args.points, "frames_per_gpu": B, "active_voxels": V,
                          "forward_ms": 1e3 * tf / k, "backward_ms": 1e3 * tb / k, "optimizer_ms": 1e3 * to / k,
                          "step_ms": 1e3 * (tf + tb + to) / k, "frames_per_s": world * B * k / (tf + tb + to),
                          "loss_first": losses[0], "loss_last": losses[-1], "steps": k}))


if __name__ == "__main__":
    main()
