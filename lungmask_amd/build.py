"""Builds liblungmask_hip.so (hipcc, gfx950) in-tree.  `python -m lungmask_amd.build`."""
from __future__ import annotations

import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "liblungmask_hip.so")


def _sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _headers():
    return sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(ROOT, "include", "lungmask_hip.h")]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _run_all(cmds, verbose):
    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("build step failed: %s\n%s" % (" ".join(cmd), r.stdout))
        return r.stdout

    with ThreadPoolExecutor(max_workers=6) as ex:
        return list(ex.map(run, cmds))


def build(force: bool = False, verbose: bool = True) -> str:
    """hipcc --offload-arch=gfx950 for every csrc/*.hip, linked into liblungmask_hip.so."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "_build")
    os.makedirs(objdir, exist_ok=True)
    hdrs = _headers()
    jobs, objs = [], []
    for src in _sources():
        obj = os.path.join(objdir, os.path.basename(src).replace(".hip", ".o"))
        objs.append(obj)
        if force or _newer(obj, [src] + hdrs):
            jobs.append([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-c", src, "-o", obj])
    _run_all(jobs, verbose)
    if jobs or not os.path.exists(LIB):
        _run_all([[hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs], verbose)
    return LIB


def build_emu(force: bool = False, verbose: bool = False) -> str:
    """TEST-ONLY: the same sources compiled by g++ against tests/emu/hip_emu.h
    (functional emulation of the kernels for the GPU-less CPU test-suite)."""
    emu = os.path.join(ROOT, "tests", "emu")
    outdir = os.path.join(emu, "_build")
    os.makedirs(outdir, exist_ok=True)
    lib = os.path.join(outdir, "liblungmask_emu.so")
    # Hardware fused multiply-add for the emulator's fmaf chains (the software fmaf of libm is ~2.5x slower over the whole CPU suite;
    # both are correctly rounded, so every bit-exactness test is unaffected -- contraction stays off).  Only when THIS machine has
    # the instructions, and a library built with another set of flags (it travels with gpurun snapshots) is rebuilt, not loaded.
    arch = []
    try:
        with open("/proc/cpuinfo") as f:
            cpu_flags = next((ln for ln in f if ln.startswith("flags")), "").split()
        if "fma" in cpu_flags and "avx2" in cpu_flags:
            arch = ["-mfma", "-mavx2"]
    except OSError:
        pass
    # LM_EMU_EXTRA_FLAGS: extra -D switches for a variant of the emulator (e.g. -DLM_H3_FOLD_SCALE=0 to run the suite on the other
    # form of the kernels); part of the stamp, so switching rebuilds
    extra_flags = os.environ.get("LM_EMU_EXTRA_FLAGS", "").split()
    arch = arch + extra_flags
    stamp = os.path.join(outdir, "flags.txt")
    want = " ".join(arch)
    have = open(stamp).read() if os.path.exists(stamp) else None
    if have != want:
        force = True
    extra = [os.path.join(emu, "hip_emu.h"), os.path.join(emu, "hip_emu_switch.cpp")]
    hdrs = _headers() + extra
    objs, cmds = [], []
    for s in _sources() + [extra[1]]:
        obj = os.path.join(outdir, os.path.basename(s).rsplit(".", 1)[0] + ".o")
        objs.append(obj)
        if force or _newer(obj, [s] + hdrs):
            cmds.append(["g++", "-x", "c++", "-std=c++17", "-O2"] + arch + ["-ffp-contract=off", "-fopenmp", "-fPIC", "-DLM_EMU_BUILD", "-I", emu, "-I", CSRC, "-c", s, "-o", obj])
    _run_all(cmds, verbose)
    if cmds or not os.path.exists(lib):
        _run_all([["g++", "-shared", "-fopenmp", "-o", lib] + objs], verbose)
        with open(stamp, "w") as f:
            f.write(want)
    return lib


if __name__ == "__main__":
    if "--emu" in sys.argv:
        print(build_emu(force="--force" in sys.argv, verbose=True))
    else:
        print(build(force="--force" in sys.argv))
