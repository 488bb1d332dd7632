#!/usr/bin/env python
"""tall-skinny Linear weight gradient: torch (hipBLASLt) gy^T x vs ops.linear_wgrad (ls3d_spconv_wgrad on the identity table)"""
import os, sys, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from lidarseg3d_amd import ops
dev = "cuda:0"
out = []
for n, cin, cout in ((360000, 96, 96), (360000, 96, 192), (360000, 192, 96), (360000, 64, 64), (241000, 32, 64), (1200000, 64, 64), (1200000, 64, 128), (1200000, 64, 192), (360000, 128, 64), (360000, 64, 23)):
    x = torch.randn(n, cin, device=dev); gy = torch.randn(n, cout, device=dev) * 0.1
    want = (gy.double().t() @ x.double())
    rec = dict(n=n, cin=cin, cout=cout)
    for name, fn in (("torch", lambda: gy.t() @ x), ("ls3d_f32", lambda: ops.linear_wgrad(x, gy, 0)), ("ls3d_x6", lambda: ops.linear_wgrad(x, gy, 6))):
        for _ in range(3): r = fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): r = fn()
        b.record(); torch.cuda.synchronize()
        rec[name + "_ms"] = a.elapsed_time(b) / 10
        rec[name + "_relerr"] = float((r.double() - want).norm() / want.norm())
    out.append(rec); print(json.dumps(rec), flush=True)
