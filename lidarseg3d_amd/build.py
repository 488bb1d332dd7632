"""Build libls3d.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

    python -m lidarseg3d_amd.build            # build if stale
    python -m lidarseg3d_amd.build --force

hipcc cross-compiles without a GPU, so this also runs in the CPU-only development container
(__graft_entry__.build()).  The built .so is git-ignored but travels to the GPU box with the snapshot.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libls3d.so")
ARCH = "gfx950"
# -ffp-contract=off: every f32 multiply/add rounds where it is written (voxel centres, distances and the
# voxel-coordinate division must match the reference bit for bit); fused multiply-adds are explicit fmaf().
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=" + ARCH, "-ffp-contract=off",
         "-fno-gpu-rdc", "-Wno-unused-result", "-Wno-unused-value"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "ls3d.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not is_stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + sources() + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
