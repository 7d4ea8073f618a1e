"""Stand-alone driver (run in a child process by tests/test_gpu_r4_host.py): the host re-tiled `_R4` types -- IQ2_K_R4 IQ3_K_R4 IQ4_K_R4 IQ5_K_R4
IQ4_KS_R4 IQ5_KS_R4 Q4_0_R8 Q5_0_R4 Q6_0_R4 Q8_0_R8 MXFP4_R8 Q2_K_R4 Q3_K_R4 IQ4_XS_R8 IQ2_XXS_R4 IQ2_XS_R4 IQ3_XXS_R4 IQ2_BN_R4 -- through the backend shim, with the REAL reference libggml as the host and its CPU backend (which has kernels for these
interleaved types, iqk_gemm_iqk_quants.cpp) as the comparison.  Per type: MUL_MAT decode (N = 1) and prompt (N = 40) on an interleaved tensor
uploaded in one piece, the interleaved bytes read back unchanged, and for one type a piecewise upload and a MUL_MAT_ID over interleaved experts.
Prints one line per case; exit code 0 = every case passed.

    python tests/r4_host_case.py [type-name ...]          e.g. iq4_k_r4 (default: all)
"""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
from common import NMSE_VS_CPU, activations, gaussian_weights_f32, nmse  # noqa: E402
from oracle import bindings as ob  # noqa: E402

F32, I32 = 0, 26
R4_HOST = {"iq2_k_r4": (ob.IQ2_K, 337), "iq3_k_r4": (ob.IQ3_K, 338), "iq4_k_r4": (ob.IQ4_K, 339), "iq5_k_r4": (ob.IQ5_K, 340), "iq4_ks_r4": (ob.IQ4_KS, 344), "iq5_ks_r4": (ob.IQ5_KS, 352),
           "q4_0_r8": (ob.Q4_0, 202), "q5_0_r4": (ob.Q5_0, 206), "q6_0_r4": (ob.Q6_0, 233), "q8_0_r8": (ob.Q8_0, 208), "mxfp4_r8": (ob.MXFP4, 353),
           "q2_k_r4": (ob.Q2_K, 210), "q3_k_r4": (ob.Q3_K, 211), "iq4_xs_r8": (ob.IQ4_XS, 223), "iq2_xxs_r4": (ob.IQ2_XXS, 216), "iq2_xs_r4": (ob.IQ2_XS, 217), "iq3_xxs_r4": (ob.IQ3_XXS, 218),
           "iq2_bn_r4": (ob.IQ2_BN, 335)}
# the reference's AVX-512 kernels of these base types saturate int16 pair sums; the device computes the exact sums
SATURATING = {ob.IQ4_K, ob.IQ5_K, ob.IQ4_KS, ob.IQ5_KS, ob.IQ4_XS}


def interleave(lib, r4, w, k):
    out = np.empty_like(w)
    assert lib.cdna4_retile_r4_host(r4, w.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), w.shape[0], k, 0, 1) == 0
    return out


def main(names):
    from ggml_host import GgmlHost
    h = GgmlHost(); g = h.g
    lib = C.CDLL(os.path.join(os.path.dirname(HERE), "ik_llama.cpp_amd", "libggml-hip-cdna4.so"))
    lib.cdna4_retile_r4_host.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int]
    gpu = h.shim.ggml_backend_cuda_init(0, None, None); cpu = g.ggml_backend_cpu_init(); g.ggml_backend_cpu_set_n_threads(cpu, 8)
    assert gpu
    failures = 0

    def report(case, ok, **kw):
        nonlocal failures
        failures += 0 if ok else 1
        print(json.dumps(dict(case=case, ok=bool(ok), **kw)), flush=True)

    for name in names:
        base, r4 = R4_HOST[name]
        m, k = 64, 1024
        wb = h.ref.quantize(base, gaussian_weights_f32(m, k, 60 + base)); w = interleave(lib, r4, wb, k)
        bar = 2e-2 if base in SATURATING else NMSE_VS_CPU
        for n in (1, 40):
            x = activations(n, k, 61 + n); back = {}

            def build(ctx):
                a = g.ggml_new_tensor_2d(ctx, r4, k, m); b = g.ggml_new_tensor_2d(ctx, F32, k, n); back["a"] = a
                return {"a": a, "b": b}, g.ggml_mul_mat(ctx, a, b)
            orig_free = g.ggml_backend_buffer_free

            def grab(buf):          # read the weight tensor back after the compute, before the buffer is freed
                out = np.empty_like(w); g.ggml_backend_tensor_get(back["a"], out.ctypes.data_as(C.c_void_p), 0, out.nbytes); back["bytes"] = out
                orig_free(buf)
            g.ggml_backend_buffer_free = grab
            try:
                got, sup = h.run(gpu, build, {"a": w, "b": x})
            finally:
                g.ggml_backend_buffer_free = orig_free
            def build_base(ctx):
                a = g.ggml_new_tensor_2d(ctx, base, k, m); b = g.ggml_new_tensor_2d(ctx, F32, k, n)
                return {"a": a, "b": b}, g.ggml_mul_mat(ctx, a, b)
            want, _ = h.run(cpu, build_base, {"a": wb, "b": x})        # the CPU backend on the BASE-type tensor: what "served as its base type" has to reproduce
            e = float(nmse(got, want)); ok = sup and e < bar and np.array_equal(back["bytes"], w); e_r = None
            if name != "q8_0_r8":       # ... and the CPU backend's own interleaved kernels on the file bytes (Q8_0_R8: AVX-512 builds expect bytes the loader has biased by 127, iqk_quantize.cpp:8439-8441)
                want_r, _ = h.run(cpu, build, {"a": w, "b": x}); e_r = float(nmse(got, want_r)); ok = ok and e_r < bar
            report("%s mul_mat n=%d" % (name, n), ok, supported=bool(sup), nmse=e, nmse_vs_cpu_interleaved=e_r, bar=bar, bytes_back=bool(np.array_equal(back["bytes"], w)))

    if "iq4_k_r4" in names:
        base, r4 = R4_HOST["iq4_k_r4"]; m, k = 64, 1024
        # piecewise upload (llama-model-loader.cpp:1204-1240): re-tiled at first use; a later overwrite goes back through the file layout
        ws = [interleave(lib, r4, h.ref.quantize(base, gaussian_weights_f32(m, k, 71 + i)), k) for i in range(2)]
        x = activations(1, k, 73)
        ctx = g.ggml_init(h.ref.InitParams(g.ggml_tensor_overhead() * 16 + g.ggml_graph_overhead() + (1 << 16), None, True))
        a = g.ggml_new_tensor_2d(ctx, r4, k, m); b = g.ggml_new_tensor_2d(ctx, F32, k, 1); o = g.ggml_mul_mat(ctx, a, b)
        gf = g.ggml_new_graph(ctx); g.ggml_build_forward_expand(gf, o)
        buf = g.ggml_backend_alloc_ctx_tensors(ctx, gpu)
        g.ggml_backend_tensor_set(b, x.ctypes.data_as(C.c_void_p), 0, x.nbytes)
        for i, pieces in enumerate((3, 2)):
            flat = np.ascontiguousarray(ws[i]).reshape(-1); step = (flat.size // pieces + 143) // 144 * 144
            for o0 in range(0, flat.size, step):
                part = np.ascontiguousarray(flat[o0:o0 + step]); g.ggml_backend_tensor_set(a, part.ctypes.data_as(C.c_void_p), o0, part.nbytes)
            assert g.ggml_backend_graph_compute(gpu, gf) == 0
            r = np.empty(m, np.float32); g.ggml_backend_tensor_get(o, r.ctypes.data_as(C.c_void_p), 0, r.nbytes)

            def build(c2):
                a2 = g.ggml_new_tensor_2d(c2, r4, k, m); b2 = g.ggml_new_tensor_2d(c2, F32, k, 1)
                return {"a": a2, "b": b2}, g.ggml_mul_mat(c2, a2, b2)
            want, _ = h.run(cpu, build, {"a": ws[i], "b": x})
            e = float(nmse(r, want)); report("iq4_k_r4 upload in %d pieces" % pieces, e < 2e-2, nmse=e)
        g.ggml_backend_buffer_free(buf); g.ggml_free(ctx)

        # MUL_MAT_ID over interleaved expert tensors (decode and a prompt batch)
        m, k, n_expert, n_used = 128, 512, 8, 2
        we = np.stack([interleave(lib, r4, h.ref.quantize(base, gaussian_weights_f32(m, k, 80 + e)), k) for e in range(n_expert)])
        for n_tok in (1, 40):
            x = activations(n_tok * n_used, k, 90 + n_tok).reshape(n_tok, n_used, k)
            ids = np.random.default_rng(5 + n_tok).integers(0, n_expert, size=(n_tok, n_used)).astype(np.int32)

            def build(ctx):
                a = g.ggml_new_tensor_3d(ctx, r4, k, m, n_expert); b = g.ggml_new_tensor_3d(ctx, F32, k, n_used, n_tok); i = g.ggml_new_tensor_2d(ctx, I32, n_used, n_tok)
                return {"a": a, "b": b, "i": i}, g.ggml_mul_mat_id(ctx, a, b, i)
            got, sup = h.run(gpu, build, {"a": we, "b": x, "i": ids}); want, _ = h.run(cpu, build, {"a": we, "b": x, "i": ids})
            e = float(nmse(got, want)); report("iq4_k_r4 mul_mat_id n_tok=%d" % n_tok, sup and e < 2e-2, supported=bool(sup), nmse=e)
    g.ggml_backend_free(gpu); g.ggml_backend_free(cpu)
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:] or list(R4_HOST)))
