#!/usr/bin/env python
"""Where does hipcc wait for memory inside the loops of a kernel?  (No GPU needed: hipcc cross-compiles to gfx950 assembly.)

A software pipeline written in HIP source is only a pipeline if the compiler's `s_waitcnt vmcnt(N)` leave the prefetched loads in
flight.  Three things silently turn it into load -> wait -> use (round 3 found all three by reading listings, after two rounds of
reading counters): a select or branch on a loaded value right behind the load; loads under a per-lane condition (a path may skip
them, so hipcc cannot count on them and waits for an older hazard with a count that drains them); an index load issued BEHIND the
rows it must not wait for (vmcnt counts in order).  This tool compiles one csrc/*.hip to assembly and prints, per kernel and per
loop block, the instruction mix and every vmcnt wait with the number of loads the block has issued before it:

    python tools/asm_waits.py spconv_bwd.hip                      # all kernels of the file, loop blocks only
    python tools/asm_waits.py spconv.hip -k gather_gemm_bf16x3ILi4ELb1ELi3 --all-blocks

`vmcnt(0)` (or a count below the number of loads meant to stay in flight) inside a hot loop is the thing to look for;
tests/test_host_logic.py::test_weight_gradient_row_pipelines_are_not_drained pins the weight-gradient loops this way.
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lidarseg3d_amd import build as B  # noqa: E402


def classify(op):
    if "mfma" in op:
        return "mfma"
    if op.startswith(("global_load", "buffer_load", "flat_load")):
        return "load"
    if op.startswith(("global_store", "buffer_store", "flat_store", "global_atomic")):
        return "store"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    return "salu" if op.startswith("s_") else "valu"


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("source", help="file under lidarseg3d_amd/csrc/ (or a path)")
    ap.add_argument("-k", "--kernel", default="", help="substring of the mangled kernel name")
    ap.add_argument("--all-blocks", action="store_true", help="also blocks outside loops")
    a = ap.parse_args()
    src = a.source if os.path.exists(a.source) else os.path.join(B.CSRC, a.source)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.check_call([hipcc] + B.CFLAGS + ["-S", "--cuda-device-only", src, "-o", out], cwd=tmp, stderr=subprocess.DEVNULL)
        asm = open(out).read()
    for m in re.finditer(r"^(_Z\w+):\s*; @", asm, re.M):
        name = m.group(1)
        if a.kernel not in name:
            continue
        body = asm[m.end():asm.index("s_endpgm", m.end())]
        lines = []
        for blk in re.split(r"^(?=\.LBB\d+_\d+:)", body, flags=re.M):
            head = blk.split("\n", 1)[0]
            in_loop = "Loop" in head
            if not (in_loop or a.all_blocks):
                continue
            mix, waits, loads = Counter(), [], 0
            for ln in blk.splitlines()[1:]:
                t = ln.strip()
                if not t or t.startswith((";", ".")):
                    continue
                kind = classify(t.split()[0])
                mix[kind] += 1
                if kind == "load":
                    loads += 1
                w = re.search(r"vmcnt\((\d+)\)", t)
                if w:
                    waits.append("%s@%d" % (w.group(1), loads))
            if mix["load"] or mix["mfma"] or waits:
                label = head.split(":")[0] if head.startswith(".LBB") else "entry"
                depth = re.search(r"Depth=(\d+)", head)
                lines.append("  %-10s %s %-58s vmcnt(N)@loads-issued: %s" % (
                    label, ("d" + depth.group(1)) if depth else "  ",
                    " ".join("%s %d" % (k, mix[k]) for k in ("mfma", "load", "lds", "valu", "salu", "barrier", "branch") if mix[k]), " ".join(waits) or "-"))
        if lines:
            print(name[:110])
            print("\n".join(lines))
    return 0


if __name__ == "__main__":
    sys.exit(main())
