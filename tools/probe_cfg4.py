#!/usr/bin/env python
"""configs[4] accuracy probe (needs the MI355X): MSeg3D logits at |logit|max = 10 against the CPU oracle for the combinations of the convolution
arithmetic (bf16x6 / bf16) and the SF-Phase attention operands (f32 / bf16 / fp8), at 30k and 120k points."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import lidarseg3d_amd as L
from lidarseg3d_amd import models_cfg, ops, synth
from oracle import ref as orc

cfg = synth.NUSC
dev = torch.device("cuda:0")
cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
model = L.build_detector(models_cfg.mseg3d(), train_cfg=None, test_cfg={}).eval()
shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
sd = {k: torch.from_numpy(v) for k, v in synth.random_state_dict(shapes, 5).items()}
model.to(dev)
for n in [int(a) for a in sys.argv[1:]] or [30000, 120000]:
    frame = synth.lidar_frame(n, seed=12, **cfg)
    img, emb, cuv = synth.camera_inputs(n, seed=4, ncam=6, c_img=48, h=40, w=60, batch=1)
    fwd = lambda s_: orc.mseg3d_forward(s_, [frame], torch.from_numpy(cuv), torch.from_numpy(img), torch.from_numpy(emb), cfg["voxel_size"], cfg["pc_range"])["out_logits"]
    want = fwd(sd)
    f = 10.0 / float(want.abs().max())
    sd10 = dict(sd); sd10["point_head.out_cls_layers.weight"] = sd["point_head.out_cls_layers.weight"] * f; sd10["point_head.out_cls_layers.bias"] = sd["point_head.out_cls_layers.bias"] * f
    model.load_state_dict(sd10)
    want = fwd(sd10)
    pts = cu(np.concatenate([np.zeros((n, 1), np.float32), frame], 1))
    ex = dict(points=pts, batch_size=1, points_cuv=cu(cuv), image_features=cu(img), camera_semantic_embeddings=cu(emb))
    for prec in ("bf16x6", "bf16"):
        for att in ("f32", "bf16", "fp8"):
            ops.set_precision(prec); ops.set_sffm_attention(att)
            with torch.no_grad():
                model(dict(ex), return_loss=False)
            got = model.point_head.forward_ret_dict["out_logits"].cpu()
            d = (got - want).abs()
            print(json.dumps(dict(points=n, conv=prec, attention=att, max_abs=float(d.max()), rms=float(d.pow(2).mean().sqrt()),
                                  argmax=float((got.argmax(1) == want.argmax(1)).float().mean()))), flush=True)
ops.set_precision("bf16x6"); ops.set_sffm_attention("f32")
