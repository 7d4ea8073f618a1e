"""Builds libggml-hip-cdna4.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this runs in the CPU-only build container; the resulting .so is
git-ignored but travels to the GPU box with the gpurun snapshot."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libggml-hip-cdna4.so")
SOURCES = ["cdna4_api.hip"]
DEPS = ["cdna4_common.cuh", "gemv.cuh", "convert.cuh", "gemm_mfma.cuh", "reduce.inc", "iq_grids_packed.inc",
        os.path.join("..", "..", "include", "ggml_hip_cdna4.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
         "-Wall", "-Wno-unused-function", "-I/opt/rocm/include",
         "-fno-slp-vectorize"]    # keep scalar v_fma_f32: v_pk_fma_f32 beside MFMAs is slower (MI355X guide, "price of one filler")


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    for f in SOURCES + DEPS:
        p = os.path.join(CSRC, f)
        if os.path.exists(p) and os.path.getmtime(p) > t:
            return True
    return False


def build_library(force=False, verbose=False):
    """Compile the library if sources are newer than the .so.  Returns the path."""
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        if os.path.exists(LIB):
            return LIB          # GPU box without a toolchain: use the prebuilt library
        raise RuntimeError("hipcc not found and no prebuilt %s" % LIB)
    cmd = [hipcc] + FLAGS + ["-o", LIB + ".tmp"] + [os.path.join(CSRC, s) for s in SOURCES] + ["-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
