#!/usr/bin/env python
"""A/B runs of bench.py under different environment knobs, one compact line per leg (and gpurun_out/ab_<tag>.json).

    python tools/ab.py TAG [--args "--steps 10 ..."] name1:ENV=VAL,ENV2=VAL name2: ...
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1]
    rest = sys.argv[2:]
    extra = "--steps 10 --warmup 3 --no-extra-modes --no-cpu-baseline"
    if rest and rest[0] == "--args":
        extra, rest = rest[1], rest[2:]
    out = {}
    for leg in rest:
        name, _, envs = leg.partition(":")
        env = dict(os.environ)
        for kv in filter(None, envs.split(",")):
            k, _, v = kv.partition("=")
            env[k] = v
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra.split(), env=env, capture_output=True, text=True, cwd=ROOT)
        try:
            j = json.loads(r.stdout.strip().splitlines()[-1])
            rf = j.get("roofline") or {}
            out[name] = dict(env=envs, frames_per_s=j["value"], ms_per_step=j["ms_per_step"],
                             conv_stack_ms=(rf.get("sparse_conv_ms_per_frame") or {}).get("mean"), frac=rf.get("frac"), stages=j.get("stages_ms"))
            print("%-28s %7.2f frames/s  %6.3f ms/step  conv stack %s ms  stages %s" % (
                name, j["value"], j["ms_per_step"], "%.3f" % out[name]["conv_stack_ms"] if out[name]["conv_stack_ms"] else "-",
                {k: round(v, 3) for k, v in (j.get("stages_ms") or {}).items()}), flush=True)
        except Exception as e:
            out[name] = dict(env=envs, error=repr(e), rc=r.returncode, stderr=r.stderr[-2000:], stdout=r.stdout[-500:])
            print("%-28s FAILED rc=%s %r\n%s\n%s" % (name, r.returncode, e, r.stderr[-1500:], r.stdout[-300:]), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "ab_%s.json" % tag), "w"), indent=1)


if __name__ == "__main__":
    main()
