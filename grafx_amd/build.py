"""Build libgrafx_amd.so (all HIP kernels + the C ABI) in-tree for gfx950.

    python -m grafx_amd.build [--force]

hipcc cross-compiles without a GPU.  The library lands next to the sources
(grafx_amd/lib/libgrafx_amd.so) so it travels with the tree; it is git-ignored.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
# GRAFX_AMD_LIB: use (and build) another copy of the library instead of the default one -- how the A/B tools select a
# variant without ever overwriting the live library (tools/ab.sh, tools/build_variant.sh)
LIB = os.environ.get("GRAFX_AMD_LIB") or os.path.join(LIBDIR, "libgrafx_amd.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -pragma-unroll-threshold: the FFT tile passes are `#pragma unroll` loop nests around inline packed-FP32
# instructions; the default size cap stops unrolling them (and then every twiddle index is a run-time value).
# -amdgpu-schedule-relaxed-occupancy: the FFT-tile kernels sit at two waves per SIMD by design (LDS), so the scheduler
# may spend registers on latency instead of defending an occupancy it cannot reach (A/B on MI355X: 1-2 % on fftconv1).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize",
         "-mllvm", "-pragma-unroll-threshold=1048576", "-mllvm", "-amdgpu-schedule-relaxed-occupancy=true",
         ] + os.environ.get("GRAFX_HIPCC_FLAGS", "").split()


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _deps():
    return sources() + glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(f) > t for f in _deps())


def build(force=False, verbose=False):
    if not force and not is_stale():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = LIBDIR if "GRAFX_AMD_LIB" not in os.environ else os.path.join(LIBDIR, "obj_" + os.path.basename(LIB))
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        cmd = [HIPCC, *FLAGS, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"link failed:\n{res.stdout}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
