"""CPU-side checks of the drop-in boundary: the UNMODIFIED reference consumers (libggml built with -DGGML_USE_CUDA, libllama, llama-bench) link
against the shim, i.e. every ggml_backend_cuda_* symbol the reference binds (ggml/include/ggml-cuda.h:25-51, ggml-backend.cpp registry,
src/llama.cpp device enumeration) resolves.  No GPU, no compute.  Skipped where the binaries were not built (no /root/reference)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LL = os.path.join(ROOT, "oracle", "_ref", "llama")
BENCH = os.path.join(LL, "bin", "llama-bench")
SHIM = os.path.join(ROOT, "ik_llama.cpp_amd", "backend", "libggml-cuda-cdna4.so")

needs_build = pytest.mark.skipif(not os.path.exists(BENCH), reason="oracle/_ref/llama not built (make -C ik_llama.cpp_amd/backend -f Makefile.llama; needs /root/reference)")


def undefined(path):
    out = subprocess.run(["nm", "-D", "--undefined-only", path], stdout=subprocess.PIPE, check=True).stdout.decode()
    return {l.split()[-1].split("@")[0] for l in out.splitlines() if l.strip()}


def defined(path):
    out = subprocess.run(["nm", "-D", "--defined-only", path], stdout=subprocess.PIPE, check=True).stdout.decode()
    return {l.split()[-1].split("@")[0] for l in out.splitlines() if l.strip()}


@needs_build
def test_every_cuda_symbol_the_reference_binds_is_exported_by_the_shim():
    want = set()
    for lib in ("lib/libggml.so", "lib/libllama.so", "bin/llama-bench"):
        want |= {s for s in undefined(os.path.join(LL, lib)) if s.startswith("ggml_backend_cuda") or s.startswith("ggml_cuda")}
    assert len(want) >= 10, want                     # the boundary is not trivially empty
    missing = want - defined(SHIM)
    assert not missing, "unresolved by the shim: %s" % sorted(missing)


@needs_build
def test_llama_bench_resolves_all_libraries():
    r = subprocess.run(["ldd", BENCH], stdout=subprocess.PIPE, stderr=subprocess.STDOUT).stdout.decode()
    assert "libggml-cuda-cdna4.so" in r and "libggml-hip-cdna4.so" in r and "libllama.so" in r and "not found" not in r, r


@needs_build
def test_llama_bench_starts_without_a_gpu():
    """argument parsing + backend registration run; with no device the shim reports 0 devices instead of failing to load"""
    r = subprocess.run([BENCH, "--help"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    assert r.returncode == 0 and b"usage" in r.stdout.lower()


SERVER = os.path.join(LL, "bin", "llama-server")


@pytest.mark.skipif(not os.path.exists(SERVER), reason="llama-server not built (Makefile.llama)")
def test_llama_server_links_against_the_shim():
    """north_star names llama-server: the reference's server sources (+ examples/mtmd, vendored cpp-httplib), unmodified, resolve every backend symbol from the shim"""
    r = subprocess.run(["ldd", SERVER], stdout=subprocess.PIPE, stderr=subprocess.STDOUT).stdout.decode()
    assert "libggml-cuda-cdna4.so" in r and "libggml-hip-cdna4.so" in r and "libllama.so" in r and "not found" not in r, r
    missing = {s for s in undefined(SERVER) if s.startswith("ggml_backend_cuda") or s.startswith("ggml_cuda")} - defined(SHIM)
    assert not missing, missing
    h = subprocess.run([SERVER, "--help"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    assert h.returncode in (0, 1) and (b"--port" in h.stdout or b"--port" in h.stderr)          # (this fork's --help prints the usage and exits 1)
