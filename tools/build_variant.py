"""Builds a variant of the library for A/B runs: python tools/build_variant.py NAME [-DFLAG ...] -> lungmask_amd/_ab/lib_NAME.so
(git-ignored, travels with the gpurun snapshot).  The product build is lungmask_amd/build.py; this only adds compile flags."""
import glob, os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "lungmask_amd", "csrc")
name, flags = sys.argv[1], sys.argv[2:]
out = os.path.join(ROOT, "lungmask_amd", "_ab")
obj = os.path.join(out, "obj_" + name)
os.makedirs(obj, exist_ok=True)
srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
def cc(s):
    o = os.path.join(obj, os.path.basename(s).replace(".hip", ".o"))
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", *flags, "-c", s, "-o", o], check=True)
    return o
with ThreadPoolExecutor(6) as ex:
    objs = list(ex.map(cc, srcs))
lib = os.path.join(out, f"lib_{name}.so")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs], check=True)
print(lib)
