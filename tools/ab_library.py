"""run a bench / tool entry point on ANOTHER build of libls3d.so (A/B of compile-time variants in one gpurun call):
   python tools/ab_library.py <library.so> <script.py> [args ...]      e.g. tools/ab_library.py gpurun_in_ab/libls3d_scalar_sub.so bench.py --steps 20"""
import os, sys, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from lidarseg3d_amd import _lib
_lib.use_library_for_testing(os.path.abspath(sys.argv[1]))
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
