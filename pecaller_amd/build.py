"""Build the gfx950 shared library in-tree (pecaller_amd/libpemap_hip.so) with hipcc.

hipcc cross-compiles without a GPU.  -ffp-contract=off is part of the contract: the SW recurrence and the
likelihood sums must be plain IEEE fp64 add/sub/mul (no FMA) to reproduce the reference's gcc/x86-64 arithmetic.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpemap_hip.so")
SOURCES = ["pemap_capi.hip", "pecall_capi.hip"]
DEPS = ["pemap_kernels.hip.h", "pemap_seed.hip.h", "pemap_seed2.hip.h", "pemap_seed4.hip.h", "pemap_wave.hip.h", "pemap_sw.hip.h", "pemap_band.hip.h", "pemap_aux.hip.h", "pecall_kernels.hip.h", "pecall_site.hip.h", os.path.join("..", "..", "include", "pemap_hip.h")]
# -fno-honor-nans -mno-amdgpu-ieee: no NaN can arise on this path; without IEEE mode v_max_f64 needs no canonicalising copy
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-honor-nans", "-mno-amdgpu-ieee", "-fPIC", "-shared", "-Wall",
         "-Wno-unused-function", "-Wno-unused-value", "-Wno-uninitialized"]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    for f in SOURCES + DEPS:
        p = os.path.join(CSRC, f)
        if os.path.exists(p) and os.path.getmtime(p) > t:
            return True
    return False


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    cmd = [hipcc()] + FLAGS + ["-o", LIB] + srcs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(LIB)
