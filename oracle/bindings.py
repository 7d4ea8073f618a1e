"""ctypes bindings for the oracle (liboracle.so) and, when usable on this host, the real reference library
(oracle/_ref/libggml_ref_{avx512,avx2}.so).  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py)."""
import ctypes as C
import os
import subprocess
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

# ggml_type ids (reference ggml/include/ggml.h:391-470)
Q4_K, Q5_K, Q6_K, Q8_K, IQ4_NL, IQ3_S, IQ2_S = 12, 13, 14, 15, 20, 21, 22
Q4_0, Q8_0, IQ4_XS = 2, 8, 23          # more types of SURVEY 8 f3 (legacy 32-blocks, IQ4_XS): no _R4 / _R8 forms here
Q5_0, IQ2_XXS, IQ2_XS, IQ3_XXS = 6, 16, 17, 18
Q4_1, Q5_1, Q6_0, Q2_K, Q3_K = 3, 7, 133, 10, 11
IQ2_K, IQ3_K, IQ4_K, IQ5_K, IQ4_KS, IQ5_KS = 137, 138, 139, 140, 144, 152
IQ2_KS, IQ3_KS = 145, 156            # f16 row scale
IQ4_KSS, IQ2_KL = 146, 157           # f32 / f16 row scale
IQ1_BN, IQ2_BN, Q8_K64 = 134, 135, 136      # BitNet (oracle only so far): f16 / f32 row scale, Q8_K64 activations
BITNET_TYPES = [IQ1_BN, IQ2_BN]
IQ2_KT, IQ3_KT, IQ4_KT, IQ1_KT = 153, 154, 155, 158      # trellis types: f32 row scale, Q8_2_X4 activations; decode units + de-quantization, prompts through the f16 route
KT_TYPES = [IQ1_KT, IQ2_KT, IQ3_KT, IQ4_KT]
MXFP4 = 39             # 17-byte 32-blocks, E8M0 scale
IQ1_S, IQ1_M = 19, 29   # 1.56 / 1.75 bpw ternary codebook types
IQ6_K = 141      # ik's non-linear types; the _KS ones carry an f32 row scale in front of the blocks
LEGACY_TYPES = [Q4_0, Q8_0, IQ4_XS, Q5_0, IQ2_XXS, IQ2_XS, IQ3_XXS, Q4_1, Q5_1, Q6_0, Q2_K, Q3_K, IQ2_K, IQ3_K, IQ4_K, IQ5_K, IQ4_KS, IQ5_KS, IQ2_KS, IQ3_KS, IQ4_KSS, IQ2_KL, IQ6_K, IQ1_S, IQ1_M, MXFP4]
ROW_META = {IQ1_KT: 4, IQ2_KT: 4, IQ3_KT: 4, IQ4_KT: 4, IQ1_BN: 2, IQ2_BN: 4, IQ4_KS: 4, IQ5_KS: 4, IQ2_KS: 2, IQ3_KS: 2, IQ4_KSS: 4, IQ2_KL: 2}        # type traits row_meta_size
Q8_2_X4, Q8_K32 = 99, 148
Q4_K_R4, Q5_K_R4, Q6_K_R4, IQ4_NL_R4, IQ3_S_R4, IQ2_S_R4 = 212, 213, 214, 220, 221, 222
BASE_TYPES = [Q4_K, Q5_K, Q6_K, IQ4_NL, IQ2_S, IQ3_S]
R4_TYPES = [Q4_K_R4, Q5_K_R4, Q6_K_R4, IQ4_NL_R4, IQ2_S_R4, IQ3_S_R4]
R4_OF = dict(zip(BASE_TYPES, R4_TYPES))
BASE_OF = {v: k for k, v in R4_OF.items()}
NAMES = {Q4_K: "q4_K", Q5_K: "q5_K", Q6_K: "q6_K", IQ4_NL: "iq4_nl", IQ2_S: "iq2_s", IQ3_S: "iq3_s",
         Q4_K_R4: "q4_k_r4", Q5_K_R4: "q5_k_r4", Q6_K_R4: "q6_k_r4", IQ4_NL_R4: "iq4_nl_r4",
         IQ2_S_R4: "iq2_s_r4", IQ3_S_R4: "iq3_s_r4", Q4_0: "q4_0", Q8_0: "q8_0", IQ4_XS: "iq4_xs",
         Q5_0: "q5_0", IQ2_XXS: "iq2_xxs", IQ2_XS: "iq2_xs", IQ3_XXS: "iq3_xxs",
         Q4_1: "q4_1", Q5_1: "q5_1", Q6_0: "q6_0", Q2_K: "q2_K", Q3_K: "q3_K",
         IQ2_K: "iq2_k", IQ3_K: "iq3_k", IQ4_K: "iq4_k", IQ5_K: "iq5_k", IQ4_KS: "iq4_ks", IQ5_KS: "iq5_ks", IQ2_KS: "iq2_ks", IQ3_KS: "iq3_ks", IQ4_KSS: "iq4_kss", IQ2_KL: "iq2_kl", IQ6_K: "iq6_k", IQ1_S: "iq1_s", IQ1_M: "iq1_m", MXFP4: "mxfp4", IQ1_KT: "iq1_kt", IQ2_KT: "iq2_kt", IQ3_KT: "iq3_kt", IQ4_KT: "iq4_kt"}
TYPE_SIZE = {Q4_K: 144, Q5_K: 176, Q6_K: 210, IQ4_NL: 18, IQ2_S: 82, IQ3_S: 110, Q4_0: 18, Q8_0: 34, IQ4_XS: 136, Q5_0: 22, IQ2_XXS: 66, IQ2_XS: 74, IQ3_XXS: 98, Q4_1: 20, Q5_1: 24, Q6_0: 26, Q2_K: 84, Q3_K: 110,
             IQ2_K: 76, IQ3_K: 110, IQ4_K: 144, IQ5_K: 176, IQ4_KS: 136, IQ5_KS: 168, IQ2_KS: 70, IQ3_KS: 102, IQ4_KSS: 128, IQ2_KL: 86, IQ6_K: 212, IQ1_S: 50, IQ1_M: 56, MXFP4: 17, IQ1_BN: 13, IQ2_BN: 16, IQ1_KT: 56, IQ2_KT: 68, IQ3_KT: 100, IQ4_KT: 128}
BLCK = {Q4_K: 256, Q5_K: 256, Q6_K: 256, IQ4_NL: 32, IQ2_S: 256, IQ3_S: 256, Q4_0: 32, Q8_0: 32, IQ4_XS: 256, Q5_0: 32, IQ2_XXS: 256, IQ2_XS: 256, IQ3_XXS: 256, Q4_1: 32, Q5_1: 32, Q6_0: 32, Q2_K: 256, Q3_K: 256,
        IQ2_K: 256, IQ3_K: 256, IQ4_K: 256, IQ5_K: 256, IQ4_KS: 256, IQ5_KS: 256, IQ2_KS: 256, IQ3_KS: 256, IQ4_KSS: 256, IQ2_KL: 256, IQ6_K: 256, IQ1_S: 256, IQ1_M: 256, MXFP4: 32, IQ1_BN: 64, IQ2_BN: 64, IQ1_KT: 256, IQ2_KT: 256, IQ3_KT: 256, IQ4_KT: 256}
for _b, _r in R4_OF.items():
    TYPE_SIZE[_r] = TYPE_SIZE[_b]; BLCK[_r] = BLCK[_b]


def row_size(t, k):
    return ROW_META.get(t, 0) + TYPE_SIZE[t] * (k // BLCK[t])


def vec_dot_type(t):
    if t in (Q4_K, Q5_K, Q6_K, IQ4_NL, IQ4_NL_R4, Q4_0, Q8_0, Q5_0, Q4_1, Q5_1, Q6_0, MXFP4, IQ1_KT, IQ2_KT, IQ3_KT, IQ4_KT):
        return Q8_2_X4
    if t in (Q4_K_R4, Q5_K_R4):
        return Q8_K32
    if t in (IQ1_BN, IQ2_BN):
        return Q8_K64
    return Q8_K


def act_row_size(vdt, k):
    return 32 + k if vdt == Q8_K64 else (k // 32) * 36 if vdt == Q8_2_X4 else (k // 256) * 296


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", HERE, "liboracle.so"])


class Oracle:
    """Plain-C restatement (oracle/iqk_oracle.c)."""

    def __init__(self):
        path = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            build_oracle()
        self.lib = L = C.CDLL(path)
        L.oracle_dequantize_row.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int64]
        L.oracle_dequantize_rows_r4.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int64]
        L.oracle_repack_r4.argtypes = [C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
        L.oracle_quantize_activations.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int64]
        L.oracle_dequantize_activations.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int64]
        L.oracle_mul_mat.argtypes = [C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_size_t,
                                     C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]
        L.oracle_mul_mat_f64.argtypes = [C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_size_t,
                                         C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64]
        L.oracle_fused_up_gate.argtypes = [C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                           C.c_size_t, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]
        L.oracle_fused_up_gate_ext.argtypes = [C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                               C.c_size_t, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int64]
        L.oracle_mul_mat_id.argtypes = [C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p,
                                        C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_void_p]

    # weights: uint8 array [M, row_size]; activations float32 [N, K]
    def dequantize(self, t, w, k):
        w = np.ascontiguousarray(w, dtype=np.uint8); m = w.shape[0]
        out = np.empty((m, k), np.float32)
        if t in BASE_OF:
            assert m % 4 == 0
            for r in range(0, m, 4):
                self.lib.oracle_dequantize_rows_r4(t, _p(w[r:]), _p(out[r:]), k)
        else:
            for r in range(m):
                self.lib.oracle_dequantize_row(t, _p(w[r:]), _p(out[r:]), k)
        return out

    def repack_r4(self, base_t, w, k):
        w = np.ascontiguousarray(w, dtype=np.uint8); out = np.empty_like(w)
        self.lib.oracle_repack_r4(base_t, w.shape[0], k, _p(w), _p(out))
        return out

    def quantize_activations(self, vdt, x):
        x = np.ascontiguousarray(x, dtype=np.float32); n, k = x.shape
        out = np.zeros((n, act_row_size(vdt, k)), np.uint8)
        for i in range(n):
            self.lib.oracle_quantize_activations(vdt, _p(x[i:]), _p(out[i:]), k)
        return out

    def dequantize_activations(self, vdt, q, k):
        q = np.ascontiguousarray(q, dtype=np.uint8); out = np.empty((q.shape[0], k), np.float32)
        for i in range(q.shape[0]):
            self.lib.oracle_dequantize_activations(vdt, _p(q[i:]), _p(out[i:]), k)
        return out

    def mul_mat(self, t, w, x):
        """CPU-path arithmetic (int8 activations). returns [N, M] f32."""
        w = np.ascontiguousarray(w, dtype=np.uint8); x = np.ascontiguousarray(x, dtype=np.float32)
        m, (n, k) = w.shape[0], x.shape
        out = np.empty((n, m), np.float32)
        self.lib.oracle_mul_mat(t, m, n, k, _p(w), w.shape[1], _p(x), k, _p(out), m)
        return out

    def mul_mat_f64(self, t, w, x):
        """fp64 accumulate of dequantised weights x given activations. returns (C, sum|terms|), each [N, M] f64."""
        w = np.ascontiguousarray(w, dtype=np.uint8); x = np.ascontiguousarray(x, dtype=np.float32)
        m, (n, k) = w.shape[0], x.shape
        out = np.empty((n, m), np.float64); ab = np.empty((n, m), np.float64)
        self.lib.oracle_mul_mat_f64(t, m, n, k, _p(w), w.shape[1], _p(x), k, _p(out), _p(ab), m)
        return out, ab

    def fused_up_gate(self, t, op, wu, wg, x, up_b=None, gate_b=None, limit=0.0):
        """act(gate.x + b_g) (x) clamp(up.x + b_u): biases f32 [M] or None, limit = op_params[1] (0 = off)."""
        wu = np.ascontiguousarray(wu, dtype=np.uint8); wg = np.ascontiguousarray(wg, dtype=np.uint8)
        x = np.ascontiguousarray(x, dtype=np.float32); m, (n, k) = wu.shape[0], x.shape
        out = np.empty((n, m), np.float32)
        ub = None if up_b is None else np.ascontiguousarray(up_b, dtype=np.float32)
        gb = None if gate_b is None else np.ascontiguousarray(gate_b, dtype=np.float32)
        self.lib.oracle_fused_up_gate_ext(t, op, m, n, k, _p(wu), _p(wg), wu.shape[1], _p(x), k,
                                          None if ub is None else _p(ub), None if gb is None else _p(gb), float(limit), _p(out), m)
        return out

    def mul_mat_id(self, t, ws, x, ids):
        """ws uint8 [E, M, rs]; x f32 [T, n_b, K]; ids i32 [T, n_used] -> [T, n_used, M]."""
        ws = np.ascontiguousarray(ws, dtype=np.uint8); x = np.ascontiguousarray(x, dtype=np.float32)
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        e, m, rs = ws.shape; tkn, nb, k = x.shape; nu = ids.shape[1]
        out = np.empty((tkn, nu, m), np.float32)
        self.lib.oracle_mul_mat_id(t, m, k, e, _p(ws), rs, _p(x), nb, _p(ids), nu, tkn, _p(out))
        return out


def _cpu_flags():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("flags"):
                return set(line.split(":", 1)[1].split())
    except OSError:
        pass
    return set()


def ref_path():
    """Best oracle/_ref variant this CPU can execute, or None."""
    flags = _cpu_flags()
    v4 = {"avx512f", "avx512bw", "avx512dq", "avx512vl", "avx512cd", "avx512_vnni"}
    v3 = {"avx2", "fma", "f16c", "bmi2"}
    cands = []
    if v4 <= flags:
        cands.append("avx512")
    if v3 <= flags:
        cands.append("avx2")
    for c in cands:
        p = os.path.join(HERE, "_ref", "libggml_ref_%s.so" % c)
        if os.path.exists(p):
            return p
    return None


class Ref:
    """The real reference CPU library (compiled from /root/reference by oracle/Makefile)."""

    def __init__(self, path=None):
        path = path or ref_path()
        if path is None:
            raise RuntimeError("no usable oracle/_ref/libggml_ref_*.so for this host")
        self.path = path
        self.variant = "avx512" if "avx512" in path else "avx2"
        self.lib = L = C.CDLL(path, mode=C.RTLD_GLOBAL)

        # ggml_init() builds the fp16->fp32 lookup table the reference (de)quantizers read
        # (ggml/src/ggml.c ggml_init; ggml-impl.h ggml_lookup_fp16_to_fp32); without it every scale reads 0.
        class _InitParams(C.Structure):
            _fields_ = [("mem_size", C.c_size_t), ("mem_buffer", C.c_void_p), ("no_alloc", C.c_bool)]
        L.ggml_init.restype = C.c_void_p; L.ggml_init.argtypes = [_InitParams]
        L.ggml_free.argtypes = [C.c_void_p]
        self.InitParams = _InitParams
        ctx = L.ggml_init(_InitParams(1 << 20, None, False)); L.ggml_free(ctx)
        L.ggml_quantize_chunk.restype = C.c_size_t
        L.ggml_quantize_chunk.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
                                          C.c_void_p, C.c_void_p]
        L.iqk_mul_mat.restype = C.c_bool
        L.iqk_mul_mat.argtypes = [C.c_long, C.c_long, C.c_long, C.c_int, C.c_void_p, C.c_long, C.c_int, C.c_void_p,
                                  C.c_long, C.c_void_p, C.c_long, C.c_int, C.c_int]
        for t in NAMES:
            f = getattr(L, "dequantize_row_" + NAMES[t]); f.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]; f.restype = None
        for n in ("quantize_row_q8_2_x4", "iqk_quantize_row_q8_K", "quantize_row_q8_K32", "quantize_row_q8_K64"):
            f = getattr(L, n); f.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]; f.restype = None
        for t in (Q4_K, Q5_K, Q6_K, IQ4_NL, IQ2_S, IQ3_S):
            L.ggml_quantize_init(t)

    def repack_tensor(self, t, w, k):
        """the reference's OWN offline repack (iqk_repack_tensor, iqk_quantize.cpp:8535: what `llama-quantize --repack` / -rtr run) applied to a
        ggml_tensor holding `w` (uint8 [M, row_size]); returns (new ggml type, repacked bytes)."""
        L = self.lib; m = w.shape[0]; w = np.ascontiguousarray(w)
        L.ggml_new_tensor_2d.restype = C.c_void_p; L.ggml_new_tensor_2d.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int64]
        L.ggml_get_data.restype = C.c_void_p; L.ggml_get_data.argtypes = [C.c_void_p]
        L.iqk_repack_tensor.restype = None; L.iqk_repack_tensor.argtypes = [C.c_void_p]
        ctx = L.ggml_init(self.InitParams(w.nbytes + (1 << 20), None, False))
        try:
            tensor = L.ggml_new_tensor_2d(ctx, t, k, m)
            data = L.ggml_get_data(tensor); C.memmove(data, _p(w), w.nbytes)
            L.iqk_repack_tensor(tensor)
            new_t = C.c_int.from_address(tensor).value          # ggml_tensor::type is the first member (ggml.h)
            out = np.empty_like(w); C.memmove(_p(out), data, w.nbytes)
        finally:
            L.ggml_free(ctx)
        return new_t, out

    def quantize(self, t, wf):
        """f32 [M,K] -> uint8 [M,row_size] with the reference quantizer (all-ones imatrix)."""
        wf = np.ascontiguousarray(wf, dtype=np.float32); m, k = wf.shape
        out = np.zeros((m, row_size(t, k)), np.uint8)
        im = np.ones(k, np.float32)
        n = self.lib.ggml_quantize_chunk(t, _p(wf), _p(out), 0, m, k, _p(im), None)
        assert n == out.size, (n, out.size)
        return out

    def dequantize(self, t, w, k):
        w = np.ascontiguousarray(w, dtype=np.uint8); m = w.shape[0]
        out = np.empty((m, k), np.float32)
        f = getattr(self.lib, "dequantize_row_" + NAMES[t])
        if t in BASE_OF:
            for r in range(0, m, 4):
                f(_p(w[r:]), _p(out[r:]), 4 * k)
        else:
            for r in range(m):
                f(_p(w[r:]), _p(out[r:]), k)
        return out

    def quantize_activations(self, vdt, x):
        x = np.ascontiguousarray(x, dtype=np.float32); n, k = x.shape
        out = np.zeros((n, act_row_size(vdt, k)), np.uint8)
        f = {Q8_2_X4: self.lib.quantize_row_q8_2_x4, Q8_K: self.lib.iqk_quantize_row_q8_K,
             Q8_K32: self.lib.quantize_row_q8_K32, Q8_K64: self.lib.quantize_row_q8_K64}[vdt]
        for i in range(n):
            f(_p(x[i:]), _p(out[i:]), k)
        return out

    def mul_mat(self, t, w, x, nth=1):
        """ggml_compute_forward_mul_mat's work for one 2-D weight: quantize src1 rows to vec_dot_type, then
        iqk_mul_mat split over `nth` threads exactly as ggml does (ith/nth)."""
        w = np.ascontiguousarray(w, dtype=np.uint8); x = np.ascontiguousarray(x, dtype=np.float32)
        m, (n, k) = w.shape[0], x.shape
        vdt = vec_dot_type(t)
        q = self.quantize_activations(vdt, x)
        out = np.zeros((n, m), np.float32)
        args = (m, n, k, t, _p(w), w.shape[1], vdt, _p(q), q.shape[1], _p(out), m)
        if nth == 1:
            ok = self.lib.iqk_mul_mat(*args, 0, 1)
            assert ok
        else:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(nth) as ex:
                oks = list(ex.map(lambda i: self.lib.iqk_mul_mat(*args, i, nth), range(nth)))
            assert all(oks)
        return out

    def fused_up_gate(self, t, op, wu, wg, x, up_b=None, gate_b=None, limit=0.0):
        """the reference's own fused kernel, iqk_moe_fused_up_gate (iqk_mul_mat.cpp:783-857), single thread, no row mapping."""
        wu = np.ascontiguousarray(wu, dtype=np.uint8); wg = np.ascontiguousarray(wg, dtype=np.uint8)
        x = np.ascontiguousarray(x, dtype=np.float32); m, (n, k) = wu.shape[0], x.shape
        vdt = vec_dot_type(t); q = self.quantize_activations(vdt, x)
        out = np.zeros((n, m), np.float32)
        ub = None if up_b is None else np.ascontiguousarray(up_b, dtype=np.float32)
        gb = None if gate_b is None else np.ascontiguousarray(gate_b, dtype=np.float32)
        f = self.lib.iqk_moe_fused_up_gate; f.restype = C.c_bool
        f.argtypes = [C.c_long, C.c_long, C.c_long, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_void_p, C.c_long,
                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_void_p, C.c_float, C.c_int, C.c_int]
        ok = f(m, n, k, n, op, t, _p(wu), _p(wg), wu.shape[1], vdt, _p(q), q.shape[1], None if ub is None else _p(ub), None if gb is None else _p(gb),
               _p(out), m * 4, 0, None, float(limit), 0, 1)
        assert ok
        return out

    def mul_mat_omp(self, oracle, t, w, q, vdt, n, k, out, nth):
        """timing helper (bench.py cpu_baseline): activations already quantized; the reference iqk_mul_mat is driven by an
        OpenMP team inside liboracle.so (oracle_ref_mul_mat_omp) the way ggml's thread pool drives it."""
        fn = C.cast(self.lib.iqk_mul_mat, C.c_void_p)
        f = oracle.lib.oracle_ref_mul_mat_omp
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_long, C.c_long, C.c_long, C.c_int, C.c_void_p, C.c_long, C.c_int, C.c_void_p, C.c_long,
                      C.c_void_p, C.c_long, C.c_int]
        ok = f(fn, w.shape[0], n, k, t, _p(w), w.shape[1], vdt, _p(q), q.shape[1], _p(out), w.shape[0], nth)
        assert ok
