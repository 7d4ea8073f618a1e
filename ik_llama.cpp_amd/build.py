"""Builds libggml-hip-cdna4.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this runs in the CPU-only build container; the resulting .so is git-ignored but travels
to the GPU box with the gpurun snapshot.  The library is split into translation units (one per weight type and kernel family) that
are compiled in parallel and cached by a content hash under build/ (a kernel edit rebuilds only the TUs that include it)."""
import hashlib
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJDIR = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libggml-hip-cdna4.so")
BASE_TYPES = [12, 13, 14, 20, 21, 22, 2, 8, 23, 6, 3, 7, 133, 139, 140, 144, 152, 10, 11, 137, 138, 16, 17, 18, 145, 156, 146, 141, 157, 39, 19, 29]    # Q4_K Q5_K Q6_K IQ4_NL IQ3_S IQ2_S Q4_0 Q8_0 IQ4_XS Q5_0 Q4_1 Q5_1 Q6_0 IQ4_K IQ5_K IQ4_KS IQ5_KS Q2_K Q3_K IQ2_K IQ3_K IQ2_XXS IQ2_XS IQ3_XXS IQ2_KS IQ3_KS IQ4_KSS IQ6_K IQ2_KL MXFP4 IQ1_S IQ1_M (enum ggml_type)
GEMV_ONLY_TYPES = [153, 154, 155, 158]   # IQ2_KT IQ3_KT IQ4_KT IQ1_KT (trellis): decode kernels; prompts go through the f16 GEMM instance (type 1)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
         "-Wall", "-Wno-unused-function", "-I/opt/rocm/include",
         "--offload-compress",    # compressed code objects: the library is ~100 translation units of template instantiations (96 MB plain); every gpurun call pushes it
         "-fno-slp-vectorize"]    # keep scalar v_fma_f32: v_pk_fma_f32 beside MFMAs is slower (MI355X guide, "price of one filler")
EXTRA_FLAGS = os.environ.get("CDNA4_BUILD_FLAGS", "").split()     # developer knob: e.g. -DGEMV_EXP_TIMELINE

COMMON = ["api_internal.h", "cdna4_common.cuh", os.path.join("..", "..", "include", "ggml_hip_cdna4.h")]
GEMV_DEPS = COMMON + ["gemv.cuh", "gemv_launch.cuh"]
GEMM_DEPS = COMMON + ["gemv.cuh", "gemm_mfma.cuh", "gemm_wlds.cuh", "gemm_pp.cuh"]


def translation_units():
    """[(object name, source, extra defines, dependency headers)]"""
    tus = [("cdna4_api", "cdna4_api.hip", [], GEMV_DEPS + GEMM_DEPS + ["reduce.inc", "iq_grids_packed.inc"]),
           ("convert", "convert.hip", [], COMMON + ["gemv.cuh", "convert.cuh"]),
           ("gemv_dual", "gemv_dual.hip", [], GEMV_DEPS), ("gemv_attn", "gemv_attn.hip", [], GEMV_DEPS + ["fa_decode.cuh"]), ("gemv_mfma", "gemv_mfma.hip", [], GEMV_DEPS), ("gemv_bitnet", "gemv_bitnet.hip", [], COMMON),
           ("retile_host", "retile_host.hip", ["-fno-vectorize", "-fno-slp-vectorize"], COMMON)]    # host-only bit shuffling; this clang's -O3 vectorizers widen its 4-byte accesses into 16-byte ones that leave the buffer (tests/test_retile_host.py guard bytes)
    tus.append(("gemm_ppf", "gemm_ppf.hip", [], GEMM_DEPS + ["gemm_ppf.cuh"]))
    if os.path.exists(os.path.join(CSRC, "ops.hip")):
        tus.append(("ops", "ops.hip", [], COMMON + ["fa_decode.cuh"]))
    if os.path.exists(os.path.join(CSRC, "flash_attn.hip")):
        tus.append(("flash_attn", "flash_attn.hip", [], COMMON))
    for t in BASE_TYPES:
        for up in (0, 1):
            tus.append(("gemv_%d_%s" % (t, "upgate" if up else "plain"), "gemv_inst.hip", ["-DINST_TYPE=%d" % t, "-DINST_UPGATE=%d" % up], GEMV_DEPS))
        tus.append(("gemm_%d" % t, "gemm_inst.hip", ["-DINST_TYPE=%d" % t], GEMM_DEPS))
    for t in GEMV_ONLY_TYPES:
        for up in (0, 1):
            tus.append(("gemv_%d_%s" % (t, "upgate" if up else "plain"), "gemv_inst.hip", ["-DINST_TYPE=%d" % t, "-DINST_UPGATE=%d" % up], GEMV_DEPS))
    tus.append(("gemm_1", "gemm_inst.hip", ["-DINST_TYPE=1"], GEMM_DEPS))
    return tus


def _digest(src, defines, deps, extra=()):
    h = hashlib.sha256()
    h.update(" ".join(FLAGS + EXTRA_FLAGS + list(extra) + defines).encode())
    for f in [src] + sorted(set(deps)):
        p = os.path.join(CSRC, f)
        if os.path.exists(p):
            h.update(f.encode()); h.update(open(p, "rb").read())
    return h.hexdigest()[:20]


def build_library(force=False, verbose=False, extra_flags=(), out=None, tag=None, only=None):
    """Compile stale translation units (in parallel) and link.  Returns the library path.
    extra_flags / out / tag: experiment variants of the SAME library (scripts/gemv_timeline.py, scripts/gemm_exp.py): objects are cached
    under build/<tag>/ and the result goes to `out` (select it at run time with CDNA4_LIB=<out>); only = the TU names the variant flags apply to (the others
    link as the base objects -- a variant of one weight type costs two compiles, not a hundred)."""
    LIB = out or globals()["LIB"]; OBJDIR = os.path.join(globals()["OBJDIR"], tag) if tag else globals()["OBJDIR"]; extra_flags = list(extra_flags)
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        if os.path.exists(LIB):
            return LIB          # GPU box without a toolchain: use the prebuilt library
        raise RuntimeError("hipcc not found and no prebuilt %s" % LIB)
    os.makedirs(OBJDIR, exist_ok=True)
    jobs, objs = [], []
    for name, src, defines, deps in translation_units():
        # only = TU names that get the variant flags (and a variant object); every other TU is the base object, compiled (or reused) under build/
        variant = only is None or name in only
        odir = OBJDIR if variant else globals()["OBJDIR"]; xf = extra_flags if variant else []
        obj = os.path.join(odir, name + ".o"); stamp = obj + ".sha"
        dg = _digest(src, defines, deps, xf)
        objs.append(obj)
        if force or not os.path.exists(obj) or not os.path.exists(stamp) or open(stamp).read() != dg:
            jobs.append((name, [hipcc] + FLAGS + EXTRA_FLAGS + xf + defines + ["-c", os.path.join(CSRC, src), "-o", obj], stamp, dg))
    if not jobs and os.path.exists(LIB) and all(os.path.getmtime(o) <= os.path.getmtime(LIB) for o in objs):
        return LIB

    def run(job):
        name, cmd, stamp, dg = job
        if verbose:
            print(" ".join(cmd), flush=True)
        if os.path.exists(stamp):
            os.remove(stamp)
        subprocess.check_call(cmd)
        open(stamp, "w").write(dg)
        return name
    nthreads = max(1, min(len(jobs), int(os.environ.get("CDNA4_BUILD_JOBS", os.cpu_count() or 4))))
    if jobs:
        with ThreadPoolExecutor(nthreads) as ex:
            list(ex.map(run, jobs))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB + ".tmp"] + objs + ["-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    import sys
    print(build_library(force="--force" in sys.argv, verbose=True))
