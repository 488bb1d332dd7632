"""The SF-Phase decoder alone on the GPU (k_sffm_decoder_rt / k_sffm_decoder): time per call for a 120k-point frame, and with parts of the
register-resident kernel left out (SfParams::ablate) - where its time goes.   python tools/bench_decoder.py [--points 120000] [--cls 17]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=120000)
    ap.add_argument("--cls", type=int, default=17)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "decoder.json"))
    a = ap.parse_args()
    from lidarseg3d_amd import ops, point_heads
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = point_heads.SemanticFeatureFusionModule(64, 48, 64, d_model=96, nhead=4, num_decoder_layers=6, dim_feedforward=192).eval().to(dev)
    n = a.points
    x = torch.randn(n, 64, device=dev)
    e1, e2 = torch.randn(1, 48, a.cls, 1, device=dev), torch.randn(1, 64, a.cls, 1, device=dev)
    bidx = torch.zeros(n, device=dev)
    pts = torch.cat([bidx[:, None], torch.randn(n, 3, device=dev)], 1).contiguous()
    times = []
    orig = ops.sffm_decoder

    def timed(*args, **kw):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        r = orig(*args, **kw)
        e.record()
        times.append((s, e))
        return r
    ops.sffm_decoder = timed
    rec = {}
    cases = [("f32 (LDS-tile kernel)", "f32", 0), ("bf16x6 (register-resident)", "bf16x6", 0), ("  no attention", "bf16x6", 1), ("  no FFN", "bf16x6", 2),
             ("  no LayerNorm", "bf16x6", 4), ("  GEMMs without MFMAs", "bf16x6", 8), ("  GEMMs without weight staging", "bf16x6", 16),
             ("  no attention, no FFN", "bf16x6", 3), ("  GEMMs: neither", "bf16x6", 24), ("  only GEMM staging + barriers", "bf16x6", 1 | 4 | 8),
             ("  only GEMM MFMAs", "bf16x6", 1 | 4 | 16)]
    print("(the indented rows leave parts of the kernel out and need a library built with -DRT_ABLATE_HOOKS=1 (csrc/sffm.hip); in the product build the\n"
          " hooks are compiled out and they time the full kernel)")
    for name, prec, ab in cases:
        ops.set_precision(prec)
        ops._SFFM_ABLATE = ab
        del times[:]
        with torch.no_grad():
            for _ in range(a.reps + 3):
                m(x, e1, e2, bidx, 1, points=pts)
        torch.cuda.synchronize()
        ms = sorted(s.elapsed_time(e) for s, e in times[3:])
        rec[name.strip()] = ms[len(ms) // 2]
        print("%-36s %.3f ms" % (name, ms[len(ms) // 2]))
    ops._SFFM_ABLATE = 0
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(dict(points=n, cls=a.cls, ms=rec), open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
