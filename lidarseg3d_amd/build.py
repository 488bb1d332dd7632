"""Build libls3d.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

    python -m lidarseg3d_amd.build            # build what is stale
    python -m lidarseg3d_amd.build --force

Every csrc/*.hip is compiled to its own object (in parallel, only when it or a header changed) and the objects are
linked into lidarseg3d_amd/libls3d.so.  hipcc cross-compiles without a GPU, so this also runs in the CPU-only
development container (__graft_entry__.build()).  The built .so is git-ignored but travels to the GPU box with the
snapshot; the objects live in lidarseg3d_amd/_obj/ (git- and gpurun-ignored).
"""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_obj")
LIB = os.path.join(HERE, "libls3d.so")
ARCH = "gfx950"
# -ffp-contract=off: every f32 multiply/add rounds where it is written (voxel centres, distances and the
# voxel-coordinate division must match the reference bit for bit); fused multiply-adds are explicit fmaf().
CFLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-ffp-contract=off", "-fno-gpu-rdc",
          "-Wno-unused-result", "-Wno-unused-value"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _headers():
    return glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "ls3d.h")]


def _obj(src):
    return os.path.join(OBJ, os.path.basename(src) + ".o")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def is_stale():
    return _stale(LIB, sources() + _headers())


def build(force=False, verbose=False):
    if not force and not is_stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    hdrs = _headers()
    todo = [s for s in sources() if force or _stale(_obj(s), [s] + hdrs)]

    def cc(src):
        cmd = [hipcc] + CFLAGS + ["-c", src, "-o", _obj(src)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(todo)))) as ex:
        list(ex.map(cc, todo))
    cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-fno-gpu-rdc"] + [_obj(s) for s in sources()] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
