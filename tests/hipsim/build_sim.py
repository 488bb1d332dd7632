"""Host build of the kernel sources against tests/hipsim (TEST INFRASTRUCTURE ONLY — see hip_runtime.h there).
Produces tests/hipsim/libls3d_sim.so with the same C ABI; only tests load it, explicitly."""
import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "lidarseg3d_amd", "csrc")
LIB = os.path.join(HERE, "libls3d_sim.so")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def _host_has_fma():
    try:
        return " fma " in open("/proc/cpuinfo").read().replace("\n", " ")
    except OSError:
        return False


_FMA = ["-mfma"] if _host_has_fma() else []


def build(force=False):
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    deps = srcs + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "*.cpp")) + \
        glob.glob(os.path.join(HERE, "hip", "*.h")) + [os.path.join(ROOT, "include", "ls3d.h")]
    if not force and os.path.exists(LIB) and all(os.path.getmtime(d) <= os.path.getmtime(LIB) for d in deps):
        return LIB
    objs = []
    for s in srcs + [os.path.join(HERE, "hipsim.cpp")]:
        o = os.path.join(HERE, os.path.basename(s) + ".o")
        # -mfma: fmaf() (the emulated MFMAs' 16 K multiply-adds per instruction, the kernels' explicit fmaf) inlines to vfmadd instead of a libm
        # call - the same correctly rounded result, ~3x faster suite; -ffp-contract=off still keeps a * b + c two roundings
        subprocess.check_call([CLANG, "-x", "c++", "-std=c++17", "-O2"] + _FMA + ["-fPIC", "-ffp-contract=off", "-Wno-psabi",
                               "-Wno-unused-value", "-I", HERE, "-c", s, "-o", o])
        objs.append(o)
    subprocess.check_call([CLANG, "-shared", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
