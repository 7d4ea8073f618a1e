"""Minimal ctypes driver of the REAL reference ggml library (oracle/_ref) used as the HOST of our backend shim:
builds one-op graphs with the reference's own graph API and runs them on a ggml_backend_t (ours or the CPU backend).
This is the new-repo equivalent of tests/test-backend-ops.cpp's `ggml_backend_compare_graph_backend` flow."""
import ctypes as C
import os

import numpy as np

from oracle import bindings as ob

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "ik_llama.cpp_amd", "backend", "libggml-cuda-cdna4.so")


class GgmlHost:
    def __init__(self):
        self.ref = ob.Ref()                       # dlopen(RTLD_GLOBAL): the shim resolves ggml_* symbols from it
        g = self.g = self.ref.lib
        C.CDLL(os.path.join(ROOT, "ik_llama.cpp_amd", "libggml-hip-cdna4.so"), mode=C.RTLD_GLOBAL)
        s = self.shim = C.CDLL(SHIM, mode=C.RTLD_GLOBAL)
        P = C.c_void_p
        for name, res, args in [
            ("ggml_new_tensor_2d", P, [P, C.c_int, C.c_int64, C.c_int64]), ("ggml_new_tensor_3d", P, [P, C.c_int, C.c_int64, C.c_int64, C.c_int64]),
            ("ggml_mul_mat", P, [P, P, P]), ("ggml_mul_mat_id", P, [P, P, P, P]), ("ggml_fused_up_gate", P, [P, P, P, P, C.c_int]),
            ("ggml_moe_up_gate_ext", P, [P, P, P, P, P, P, P, C.c_int]),
            ("ggml_reduce", P, [P, C.POINTER(P), C.c_int, C.c_int]), ("ggml_new_tensor_1d", P, [P, C.c_int, C.c_int64]),
            ("ggml_new_graph", P, [P]), ("ggml_build_forward_expand", None, [P, P]), ("ggml_backend_alloc_ctx_tensors", P, [P, P]),
            ("ggml_backend_tensor_set", None, [P, P, C.c_size_t, C.c_size_t]), ("ggml_backend_tensor_get", None, [P, P, C.c_size_t, C.c_size_t]),
            ("ggml_backend_graph_compute", C.c_int, [P, P]), ("ggml_backend_cpu_init", P, []), ("ggml_backend_cpu_set_n_threads", None, [P, C.c_int]),
            ("ggml_backend_free", None, [P]), ("ggml_backend_buffer_free", None, [P]), ("ggml_free", None, [P]), ("ggml_nbytes", C.c_size_t, [P]),
            ("ggml_tensor_overhead", C.c_size_t, []), ("ggml_graph_overhead", C.c_size_t, []), ("ggml_backend_supports_op", C.c_bool, [P, P]),
            ("ggml_backend_name", C.c_char_p, [P]), ("ggml_backend_reg_get_count", C.c_size_t, []),
            # the non-mat-mul ops of a Llama / Mixtral graph (tests/test_gpu_ops.py)
            ("ggml_new_tensor_4d", P, [P, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int64]),
            ("ggml_add", P, [P, P, P]), ("ggml_mul", P, [P, P, P]), ("ggml_div", P, [P, P, P]), ("ggml_rms_norm", P, [P, P, C.c_float]),
            ("ggml_fused_rms_norm", P, [P, P, P, C.c_float]), ("ggml_cpy", P, [P, P, P]), ("ggml_cont", P, [P, P]),
            ("ggml_rope_ext", P, [P, P, P, P, C.c_int, C.c_int, C.c_int] + [C.c_float] * 6),
            ("ggml_view_3d", P, [P, P, C.c_int64, C.c_int64, C.c_int64, C.c_size_t, C.c_size_t, C.c_size_t]),
            ("ggml_view_2d", P, [P, P, C.c_int64, C.c_int64, C.c_size_t, C.c_size_t]),
            ("ggml_permute", P, [P, P, C.c_int, C.c_int, C.c_int, C.c_int]), ("ggml_get_rows", P, [P, P, P]),
            ("ggml_soft_max_ext", P, [P, P, P, C.c_float, C.c_float]), ("ggml_flash_attn_ext", P, [P, P, P, P, P, C.c_float, C.c_float, C.c_float]),
            ("ggml_argsort", P, [P, P, C.c_int]), ("ggml_sum_rows", P, [P, P]), ("ggml_mul_multi_add", P, [P, P, P]),
        ]:
            f = getattr(g, name); f.restype = res; f.argtypes = args
        s.ggml_backend_cuda_init.restype = P; s.ggml_backend_cuda_init.argtypes = [C.c_int, P, P]
        s.ggml_backend_is_cuda.restype = C.c_bool; s.ggml_backend_is_cuda.argtypes = [P]
        s.ggml_backend_cuda_get_device_count.restype = C.c_int

    def run(self, backend, build, inputs):
        """build(ctx) -> (dict name->tensor, out tensor); inputs: name -> numpy array.  Returns the output as float32 numpy."""
        g = self.g
        ctx = g.ggml_init(self.ref.InitParams(g.ggml_tensor_overhead() * 64 + g.ggml_graph_overhead() + (1 << 16), None, True))
        tensors, out = build(ctx)
        outs = out if isinstance(out, (list, tuple)) else [out]
        gf = g.ggml_new_graph(ctx)
        for o in outs:
            g.ggml_build_forward_expand(gf, o)
        buf = g.ggml_backend_alloc_ctx_tensors(ctx, backend)
        assert buf
        for name, arr in inputs.items():
            arr = np.ascontiguousarray(arr); assert g.ggml_nbytes(tensors[name]) == arr.nbytes, (name, g.ggml_nbytes(tensors[name]), arr.nbytes)
            g.ggml_backend_tensor_set(tensors[name], arr.ctypes.data_as(C.c_void_p), 0, arr.nbytes)
        supported = all(g.ggml_backend_supports_op(backend, o) for o in outs)
        st = g.ggml_backend_graph_compute(backend, gf)
        assert st == 0, st
        results = []
        for o in outs:
            res = np.empty(g.ggml_nbytes(o) // 4, np.float32)
            g.ggml_backend_tensor_get(o, res.ctypes.data_as(C.c_void_p), 0, res.nbytes)
            results.append(res)
        g.ggml_backend_buffer_free(buf); g.ggml_free(ctx)
        return (results if isinstance(out, (list, tuple)) else results[0]), supported
