#!/usr/bin/env python
"""Where the class-embedding side of the SF-Phase decoder (k_sffm_memory, one workgroup per frame) spends its time: the tracing build records the shader
clock at every phase boundary (include/ls3d.h: ls3d_sffm_memory_trace).  Needs the MI355X.   python tools/trace_memory.py [--cls 17] [--layers 6]"""
import argparse
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lidarseg3d_amd import _lib, ops, point_heads  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cls", type=int, default=17)
    ap.add_argument("--layers", type=int, default=6)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = point_heads.SemanticFeatureFusionModule(64, 48, 32, d_model=96, nhead=4, num_decoder_layers=a.layers, dim_feedforward=192).to(dev).eval()
    L, E = 2 * a.cls, 96
    model = m.packed()["memory"]
    mem = torch.randn(L, E, device=dev)
    kv = torch.empty((2 * a.layers, 1, E, L), dtype=torch.float32, device=dev)
    n = 2 + 4 * a.layers
    trace = torch.zeros((n,), dtype=torch.int64, device=dev)
    lib = _lib.load()
    rows = []
    for _ in range(a.reps):
        _lib.check(lib.ls3d_sffm_memory_trace(ctypes.c_void_p(mem.data_ptr()), 1, L, model.embed, model.heads, model.num_layers, model.layers,
                                              ctypes.c_void_p(kv.data_ptr()), ctypes.c_void_p(trace.data_ptr()), ops._stream(mem)), "ls3d_sffm_memory_trace")
        torch.cuda.synchronize()
        rows.append(trace.cpu().tolist())
    t = torch.tensor(rows[a.reps // 2:], dtype=torch.float64)  # warm repetitions
    d = (t[:, 1:] - t[:, :-1]).median(0).values.tolist()
    names = []
    for l in range(a.layers):
        names += ["L%d qkv(+kv)" % l, "L%d attention" % l, "L%d out-proj" % l, "L%d norm1" % l]
    names.append("last kv")
    per = {}
    for nm, v in zip(names, d):
        per.setdefault(nm.split(" ", 1)[1] if nm.startswith("L") else nm, []).append(v)
    total = float((t[:, -1] - t[:, 0]).median())
    print(json.dumps(dict(cls=a.cls, layers=a.layers, total_cycles=total, median_cycles_per_phase={k: sum(v) / len(v) for k, v in per.items()},
                          share={k: sum(v) / total for k, v in per.items()}, all=[round(v) for v in d])))


if __name__ == "__main__":
    main()
