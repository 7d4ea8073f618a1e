"""ctypes binding of include/ggml_hip_cdna4.h (the C-ABI drop-in boundary)."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))

# enum ggml_type values (reference ggml/include/ggml.h:391-470)
GGML_TYPE = dict(F32=0, F16=1, Q4_K=12, Q5_K=13, Q6_K=14, Q8_K=15, IQ4_NL=20, IQ3_S=21, IQ2_S=22, BF16=30,
                 Q8_2_X4=99, Q8_K32=148, Q4_K_R4=212, Q5_K_R4=213, Q6_K_R4=214, IQ4_NL_R4=220, IQ3_S_R4=221, IQ2_S_R4=222)
UNARY = dict(RELU=6, SILU=10, SWIGLU_OAI=14, GELU=15)         # enum ggml_unary_op values of THIS fork (ggml.h:721-743: GELU is 15, not mainline 8)
T = GGML_TYPE
BASE_TYPES = [T["Q4_K"], T["Q5_K"], T["Q6_K"], T["IQ4_NL"], T["IQ2_S"], T["IQ3_S"]]
R4_TYPES = [T["Q4_K_R4"], T["Q5_K_R4"], T["Q6_K_R4"], T["IQ4_NL_R4"], T["IQ2_S_R4"], T["IQ3_S_R4"]]
R4_OF = dict(zip(BASE_TYPES, R4_TYPES)); BASE_OF = {v: k for k, v in R4_OF.items()}
TYPE_SIZE = {12: 144, 13: 176, 14: 210, 20: 18, 22: 82, 21: 110, 2: 18, 8: 34, 23: 136, 6: 22, 16: 66, 17: 74, 18: 98, 3: 20, 7: 24, 133: 26, 10: 84, 11: 110, 137: 76, 138: 110, 139: 144, 140: 176, 144: 136, 152: 168, 145: 70, 156: 102, 146: 128, 157: 86, 141: 212, 19: 50, 29: 56, 39: 17, 15: 296, 148: 296, 99: 36, 134: 13, 135: 16, 153: 68, 154: 100, 155: 128, 158: 56}
BLCK_SIZE = {12: 256, 13: 256, 14: 256, 20: 32, 22: 256, 21: 256, 2: 32, 8: 32, 23: 256, 6: 32, 16: 256, 17: 256, 18: 256, 3: 32, 7: 32, 133: 32, 10: 256, 11: 256, 137: 256, 138: 256, 139: 256, 140: 256, 144: 256, 152: 256, 145: 256, 156: 256, 146: 256, 157: 256, 141: 256, 19: 256, 29: 256, 39: 32, 15: 256, 148: 256, 99: 32, 134: 64, 135: 64, 153: 256, 154: 256, 155: 256, 158: 256}
for _b, _r in R4_OF.items():
    TYPE_SIZE[_r] = TYPE_SIZE[_b]; BLCK_SIZE[_r] = BLCK_SIZE[_b]


ROW_META = {144: 4, 152: 4, 145: 2, 156: 2, 146: 4, 157: 2, 134: 2, 135: 4, 153: 4, 154: 4, 155: 4, 158: 4}      # IQ4_KS / IQ5_KS: f32, IQ2_KS / IQ3_KS: f16 row scale in front of the blocks (type traits row_meta_size)


Q8_K64 = 136          # BitNet activations: {float d[4]; float d * sum(q) [4]; int8 q[k]} per row


def row_size(t, k):
    if t == Q8_K64:
        return 32 + k
    return ROW_META.get(t, 0) + TYPE_SIZE[t] * (k // BLCK_SIZE[t])


def vec_dot_type(t):
    if t in (134, 135):
        return Q8_K64
    if t in (12, 13, 14, 20, 220, 2, 8, 6, 3, 7, 133, 39, 153, 154, 155, 158):
        return T["Q8_2_X4"]
    if t in (212, 213):
        return T["Q8_K32"]
    return T["Q8_K"]


def act_row_size(vdt, k):
    return row_size(vdt, k)


class Cdna4Error(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("cdna4 error %d: %s" % (code, msg)); self.code = code


def lib_path():
    # CDNA4_LIB lets a developer A/B an experimental build of the SAME library (e.g. another -DGEMV_DEPTH)
    return os.environ.get("CDNA4_LIB") or os.path.join(HERE, "libggml-hip-cdna4.so")


_lib = None

# name -> (restype, argtypes): every symbol include/ggml_hip_cdna4.h declares
_P, _I, _L, _Z, _I64 = C.c_void_p, C.c_int, C.c_long, C.c_size_t, C.c_int64
SIGNATURES = {
    "cdna4_get_device_count": (_I, []),
    "cdna4_get_device_description": (_I, [_I, C.c_char_p, _Z]),
    "cdna4_get_device_memory": (_I, [_I, C.POINTER(_Z), C.POINTER(_Z)]),
    "cdna4_init": (_P, [_I]),
    "cdna4_free": (None, [_P]),
    "cdna4_last_error": (C.c_char_p, []),
    "cdna4_version": (C.c_char_p, []),
    "cdna4_last_launch_info": (C.c_char_p, []),
    "cdna4_set_gemm_form": (_I, [_I]),
    "cdna4_handoff_selftest": (_I, [_P, _P]),
    "cdna4_handoff_mode": (_I, [_P]),
    "cdna4_reserve_workspace": (_I, [_P, _Z]),
    "cdna4_preload_type": (_I, [_I]),
    "cdna4_type_supported": (_I, [_I]),
    "cdna4_blck_size": (_I, [_I]),
    "cdna4_type_size": (_Z, [_I]),
    "cdna4_row_size": (_Z, [_I, _I64]),
    "cdna4_vec_dot_type": (_I, [_I]),
    "cdna4_dequantize_rows": (_I, [_P, _I, _P, _I64, _I64, _I64, _P, _I, _I64, _P]),
    "cdna4_quantize_rows": (_I, [_P, _I, _P, _I64, _I64, _I64, _P, _P]),
    "cdna4_mul_mat": (_I, [_P, _L, _L, _L, _I, _P, _L, _I, _P, _L, _P, _L, _P]),
    "cdna4_mul_mat_multi": (_I, [_P, _I, _P, _L, _L, _P, _P, _P, _I, _P, _L, _P, _P, _P]),
    "cdna4_mul_mat_4d": (_I, [_P] + [_L] * 13 + [_I, _P, _L, _I, _P, _L, _P, _L, _P]),
    "cdna4_fused_up_gate": (_I, [_P, _L, _L, _L, _I, _I, _P, _P, _L, _I, _P, _L, _P, _L, _P]),
    "cdna4_fused_up_gate_ext": (_I, [_P, _L, _L, _L, _I, _I, _P, _P, _L, _I, _P, _L, _P, _P, C.c_float, _P, _L, _P]),
    "cdna4_fused_up_gate_q8": (_I, [_P, _L, _L, _I, _I, _P, _P, _L, _P, _P, _P, C.c_float, _P, _P, _P]),
    "cdna4_mul_mat_id": (_I, [_P, _L, _L, _I, _I, _L, _I, _P, _L, _L, _P, _I, _L, _L, _P, _L, _P, _L, _L, _P]),
    "cdna4_moe_fused_up_gate": (_I, [_P, _L, _L, _I, _I, _L, _I, _I, _P, _P, _L, _L, _P, _I, _L, _L, _P, _L, _P, _L, _L, _P]),
    "cdna4_moe_fused_up_gate_ext": (_I, [_P, _L, _L, _I, _I, _L, _I, _I, _P, _P, _L, _L, _P, _I, _L, _L, _P, _L, _P, _L, _P, _L, C.c_float, _P, _L, _L, _P]),
    "cdna4_moe_ffn": (_I, [_P, _L, _L, _L, _I, _I, _L, _I, _I, _P, _P, _L, _L, _I, _P, _L, _L, _P, _I, _L, _L, _P, _L, _P, _L, _P, _L, C.c_float, _P, _L, _L, _P, _L, _L, _P]),
    "cdna4_set_prefill_mode": (_I, [_P, _I]),
    "cdna4_set_deterministic": (_I, [_P, _I]),
    "cdna4_op_rms_norm": (_I, [_P, _P, _P, C.c_float, _P, _P]),
    "cdna4_op_binary": (_I, [_P, _I, _P, _P, _P, _P]),
    "cdna4_op_rope": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _P]),
    "cdna4_workspace_epoch": (C.c_long, [_P]),
    "cdna4_op_cpy": (_I, [_P, _P, _P, _P]),
    "cdna4_op_cpy_indirect": (_I, [_P, _P, _P, _P, _P]),
    "cdna4_op_add_rms_norm": (_I, [_P, _P, _P, _P, _P, C.c_float, _P, _P]),
    "cdna4_mul_mat_multi_fused": (_I, [_P, _I, _P, C.c_long, C.c_long, _P, _P, _P, _I, _P, C.c_long, _P, _P, _P, _P]),
    "cdna4_attn_out_fused": (_I, [_P, _P, _P, _P, _P, _P, C.c_float, C.c_float, C.c_float, C.c_long, C.c_long, _I, _P, C.c_long, _P, _P, _P]),
    "cdna4_fused_up_gate_fused": (_I, [_P, C.c_long, C.c_long, C.c_long, _I, _I, _P, _P, C.c_long, _I, _P, C.c_long, _P, _P, C.c_float, _P, C.c_long, _P, _P]),
    "cdna4_op_moe_router": (_I, [_P] * 9 + [_I, _P]),
    "cdna4_op_rope_cache": (_I, [_P, _P, C.c_int64, _P, _I, _I] + [C.c_float] * 6 + [_P]),
    "cdna4_op_rope_cache_reset": (_I, [_P]),
    "cdna4_op_rope_store_kv": (_I, [_P] * 12 + [_I, _I, _I] + [C.c_float] * 6 + [_P]),
    "cdna4_op_moe_router_norm": (_I, [_P] * 4 + [C.c_float] + [_P] * 7 + [_I, _P]),
    "cdna4_op_norm_rope_store_kv": (_I, [_P] * 3 + [C.c_float, _P, _P, _P, C.c_float] + [_P] * 8 + [_I, _I, _I] + [C.c_float] * 6 + [_P]),
    "cdna4_op_get_rows": (_I, [_P, _P, _P, _P, _P]),
    "cdna4_op_soft_max": (_I, [_P, _P, _P, _P, C.c_float, C.c_float, _P]),
    "cdna4_op_flash_attn": (_I, [_P, _P, _P, _P, _P, _P, C.c_float, C.c_float, C.c_float, _P]),
    "cdna4_op_flash_attn_q8": (_I, [_P, _P, _P, _P, _P, _P, C.c_float, C.c_float, C.c_float, _P, _P]),
    "cdna4_op_argsort": (_I, [_P, _P, _P, _I, _P]),
    "cdna4_op_sum_rows": (_I, [_P, _P, _P, _P]),
    "cdna4_op_mul_multi_add": (_I, [_P, _P, _P, _P, _P]),
    "cdna4_op_mul_multi_add_res": (_I, [_P, _P, _P, _P, _P, _P]),
    "cdna4_window_create": (_P, [_P, _I, _I, _I64, _P]),
    "cdna4_window_attach": (_I, [_P, _I, _P]),
    "cdna4_window_all_reduce_sum": (_I, [_P, _P, _I64, _I, _I, _P]),
    "cdna4_window_all_reduce_sum_wire": (_I, [_P, _P, _I64, _I, _I, _I, _P]),
    "cdna4_window_free": (None, [_P]),
    "cdna4_op_mul_mat_dense": (_I, [_P, _P, _P, _P, _P]),
    "cdna4_repack_r4": (_I, [_P, _I, _P, _I64, _I64, _P, _P]),
    "cdna4_unrepack_r4": (_I, [_P, _I, _P, _I64, _I64, _P, _P]),
    "cdna4_invalidate_weight_cache": (_I, [_P, _P]),
    "cdna4_retile_r4_host": (_I, [_I, _P, _P, _I64, _I64, _I, _I]),
    "cdna4_retile_r4_host_base_type": (_I, [_I]),
    "cdna4_retile_r4_host_rows": (_I, [_I]),
    "cdna4_comm_unique_id": (_I, [_P]),
    "cdna4_comm_init": (_P, [_P, _P, _I, _I]),
    "cdna4_comm_free": (None, [_P]),
    "cdna4_all_reduce_sum": (_I, [_P, _P, _I64, _I, _P]),
    "cdna4_reduce_peers": (_I, [_P, _P, _I, C.c_uint, _I64, _I, _P]),
    "cdna4_reduce_peers_slice": (_I, [_P, _P, _I, C.c_uint, _I64, _I, _I, _I, _P]),
    "cdna4_time_mul_mat": (_I, [_P, _L, _L, _L, _I, _P, _I, _L, _P, _L, _P, _L, _I, _I, _P, C.POINTER(C.c_float)]),
}


def load_library(path=None):
    """dlopen the C-ABI library and bind every declared symbol.  Raises if the library or a symbol is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or lib_path()
    if not os.path.exists(p):
        raise FileNotFoundError("%s not found: build it with `python -m ik_llama.cpp_amd.build` / __graft_entry__.build(); "
                                "there is no CPU fallback" % p)
    # torch ships its own libamdhip64 / librccl; whichever HIP runtime is loaded FIRST serves the whole process.  Load torch's before ours
    # so that tensors and kernels share one runtime (loading this library first made a later `import torch` see no device).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(p)
    for name, (res, args) in SIGNATURES.items():
        try:
            f = getattr(lib, name)      # AttributeError if the symbol is not exported
        except AttributeError:
            if path is None:
                raise
            continue                    # an explicitly named build (bench.py --ab-lib: an OLDER revision) may lack the newer entry points
        f.restype = res; f.argtypes = args
    if path is None:
        _lib = lib
    return lib


class _Fusion(C.Structure):      # cdna4_fusion {norm_w, norm_eps, residual, qkv, add_b, add_dst} (include/ggml_hip_cdna4.h)
    _fields_ = [("norm_w", C.c_void_p), ("norm_eps", C.c_float), ("residual", C.c_void_p), ("qkv", C.c_void_p), ("add_b", C.c_void_p), ("add_dst", C.c_void_p)]


class CallPlan:
    """A recorded sequence of C-ABI calls with their fully converted arguments (Cdna4Backend.record()).  Replaying it skips the Python
    argument marshalling of the wrappers (~8.5 us per call -> ~2 us), which is what bounds an eagerly launched decode token (193 calls
    with tensor parallelism) once the kernels are a few microseconds long."""
    def __init__(self):
        self.calls = []

    def replay(self, check):
        for f, a in self.calls:
            rc = f(*a)
            if rc:
                check(rc)


class _Recorder:
    def __init__(self, lib, plan):
        self._lib, self._plan = lib, plan

    def __getattr__(self, name):
        f = getattr(self._lib, name)

        def rec(*a):
            self._plan.calls.append((f, a))
            return f(*a)
        return rec


class Cdna4Backend:
    """One instance per GPU (the reference creates one ggml_backend_t per device, ggml-cuda.cu:5392).

    Tensors are torch tensors on the backend's device: quantized weights are uint8 [rows, row_size] (the GGUF block
    bytes, unchanged), activations float32 [n, K]; results are float32 [n, rows] -- i.e. ggml's dst[ne1=n][ne0=rows]."""

    def __init__(self, device=0, lib_path=None):
        """lib_path: another build of the SAME library (bench.py --ab-lib: two builds timed in one process); default = the in-tree library"""
        import torch
        self.torch = torch
        self.lib = load_library(lib_path)
        if not torch.cuda.is_available() or self.lib.cdna4_get_device_count() <= device:
            raise RuntimeError("Cdna4Backend: no HIP device %d visible (this backend has no CPU fallback)" % device)
        self.device = torch.device("cuda", device)
        self.ctx = self.lib.cdna4_init(device)
        if not self.ctx:
            raise Cdna4Error(-3, self.lib.cdna4_last_error().decode())
        self.comm = None
        self.window = None

    def close(self):
        self.window_free()
        if self.comm:
            self.lib.cdna4_comm_free(self.comm); self.comm = None
        if self.ctx:
            self.lib.cdna4_free(self.ctx); self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- helpers
    def _check(self, rc):
        if rc != 0:
            raise Cdna4Error(rc, self.lib.cdna4_last_error().decode())

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def description(self):
        buf = C.create_string_buffer(256); self._check(self.lib.cdna4_get_device_description(self.device.index, buf, 256))
        return buf.value.decode()

    def set_gemm_form(self, form):
        """process-wide: 1 default, 0 per-wave de-quantizing prompt GEMM everywhere, 2 workgroup-shared weight tiles wherever they can run, 3 the ping-pong kernel wherever it can run (cdna4_set_gemm_form)"""
        self._check(self.lib.cdna4_set_gemm_form(form))

    def handoff_mode(self):
        """0 fence-free in-launch hand-offs validated by the start-up self-test, 1 fenced by request, 2 fenced after a failed self-test, -1 untested (cdna4_handoff_mode)"""
        return int(self.lib.cdna4_handoff_mode(self.ctx))

    def last_launch_info(self):
        """which prompt-GEMM instantiation / grid served this thread's last Ny > 8 mat-mul launch, as a dict (cdna4_last_launch_info); {} before the first one"""
        txt = self.lib.cdna4_last_launch_info().decode()
        out = {"kernel": txt.split(" ")[0]} if txt else {}
        for kv in txt.split(" ")[1:]:
            k, v = kv.split("=")
            out[k] = v if k == "grid" else int(v)
        return out

    def record(self):
        """context manager: every C-ABI call made through this backend inside the block is executed AND recorded; returns the CallPlan.
        Valid as long as the tensors involved stay alive and the same stream is current at replay."""
        import contextlib

        @contextlib.contextmanager
        def cm():
            plan = CallPlan(); real = self.lib; self.lib = _Recorder(real, plan)
            try:
                yield plan
            finally:
                self.lib = real
        return cm()

    def reserve_workspace(self, nbytes):
        self._check(self.lib.cdna4_reserve_workspace(self.ctx, nbytes))

    def set_prefill_mode(self, mode):
        self._check(self.lib.cdna4_set_prefill_mode(self.ctx, mode))

    def set_deterministic(self, on):
        """split-K prompt launches add their slices in a fixed order (bit-reproducible prompts; slower)"""
        self._check(self.lib.cdna4_set_deterministic(self.ctx, 1 if on else 0))

    # ---- ops
    def dequantize(self, t, w, k, dtype=None):
        """type_traits[t].to_float over every row: uint8 [rows, row_size] -> [rows, k] f32 (or f16)."""
        torch = self.torch; dtype = dtype or torch.float32
        assert w.dtype == torch.uint8 and w.is_cuda and w.stride(1) == 1
        out = torch.empty((w.shape[0], k), dtype=dtype, device=self.device)
        self._check(self.lib.cdna4_dequantize_rows(self.ctx, t, w.data_ptr(), w.stride(0), w.shape[0], k, out.data_ptr(),
                                                   T["F32"] if dtype == torch.float32 else T["F16"], out.stride(0), self._stream()))
        return out

    def quantize_activations(self, vdt, x):
        """from_float of the CPU path's vec_dot_type: f32 [n, K] -> uint8 [n, row_size(vdt, K)] (reference byte layout)."""
        torch = self.torch
        assert x.dtype == torch.float32 and x.is_cuda and x.stride(1) == 1
        n, k = x.shape
        out = torch.zeros((n, row_size(vdt, k)), dtype=torch.uint8, device=self.device)
        self._check(self.lib.cdna4_quantize_rows(self.ctx, vdt, x.data_ptr(), x.stride(0) * 4, n, k, out.data_ptr(), self._stream()))
        return out

    def mul_mat(self, t, w, x, out=None, x_type=0):
        """GGML_OP_MUL_MAT: w uint8 [M, row_size] (type t), x f32 [N, K] (or pre-quantized uint8 rows with x_type) -> f32 [N, M]."""
        torch = self.torch
        assert w.dtype == torch.uint8 and w.is_cuda and w.stride(1) == 1 and x.is_cuda and x.stride(1) == 1
        m = w.shape[0]; n = x.shape[0]
        if x_type == 0:
            assert x.dtype == torch.float32; k = x.shape[1]; sb = x.stride(0) * 4
        else:
            assert x.dtype == torch.uint8; k = x.shape[1] - 32 if x_type == Q8_K64 else x.shape[1] // TYPE_SIZE[x_type] * BLCK_SIZE[x_type]; sb = x.stride(0)
        if out is None:
            out = torch.empty((n, m), dtype=torch.float32, device=self.device)
        self._check(self.lib.cdna4_mul_mat(self.ctx, m, n, k, t, w.data_ptr(), w.stride(0), x_type, x.data_ptr(), sb,
                                           out.data_ptr(), out.stride(0), self._stream()))
        return out

    def mul_mat_multi(self, types, weights, x, outs=None):
        """several MUL_MATs sharing src1 (q,k,v): same-type matrices are served by one launch when x has one row."""
        torch = self.torch; n, k = x.shape; nm = len(weights)
        if outs is None:
            outs = [torch.empty((n, w.shape[0]), dtype=torch.float32, device=self.device) for w in weights]
        La = C.c_long * nm; Ia = C.c_int * nm; Pa = C.c_void_p * nm
        self._check(self.lib.cdna4_mul_mat_multi(self.ctx, nm, La(*[w.shape[0] for w in weights]), n, k, Ia(*types), Pa(*[w.data_ptr() for w in weights]),
                                                 La(*[w.stride(0) for w in weights]), 0, x.data_ptr(), x.stride(0) * 4,
                                                 Pa(*[o.data_ptr() for o in outs]), La(*[o.stride(0) for o in outs]), self._stream()))
        return outs

    def fused_up_gate(self, t, w_up, w_gate, x, op=UNARY["SILU"], out=None, up_b=None, gate_b=None, limit=0.0):
        """GGML_OP_FUSED_UP_GATE: act(gate.x + gate_b) * clamp(up.x + up_b) -> f32 [N, M]  (biases f32 [M] or None; limit = op_params[1])."""
        torch = self.torch
        m = w_up.shape[0]; n, k = x.shape
        assert w_up.shape == w_gate.shape and w_up.stride(0) == w_gate.stride(0)
        for b in (up_b, gate_b):
            assert b is None or (b.dtype == torch.float32 and b.is_cuda and b.is_contiguous() and b.numel() == m)
        if out is None:
            out = torch.empty((n, m), dtype=torch.float32, device=self.device)
        self._check(self.lib.cdna4_fused_up_gate_ext(self.ctx, m, n, k, op, t, w_up.data_ptr(), w_gate.data_ptr(), w_up.stride(0), 0,
                                                     x.data_ptr(), x.stride(0) * 4, up_b.data_ptr() if up_b is not None else None,
                                                     gate_b.data_ptr() if gate_b is not None else None, float(limit),
                                                     out.data_ptr(), out.stride(0), self._stream()))
        return out

    def fused_up_gate_norm(self, t, w_up, w_gate, x, norm_w, eps=1e-5, op=UNARY["SILU"], out=None):
        """FUSED_RMS_NORM(x) * norm_w -> FUSED_UP_GATE as ONE call (cdna4_fused_up_gate_fused with cdna4_fusion.norm_w): what the shim issues for the ffn_norm + up*gate nodes of a
        decoded token (the mat-vec kernel's norm-carrying instantiation, FX = 1) and of a prompt ubatch (norm inside the activation-image launch)."""
        torch = self.torch
        m = w_up.shape[0]; n, k = x.shape
        assert w_up.shape == w_gate.shape and norm_w.dtype == torch.float32 and norm_w.numel() == k and norm_w.is_cuda and x.is_contiguous()
        if out is None:
            out = torch.empty((n, m), dtype=torch.float32, device=self.device)

        cache = self.__dict__.setdefault("_fx_cache", {})      # (a recorded call plan keeps the descriptor's ADDRESS: one live object per (norm weights, eps))
        key = (norm_w.data_ptr(), float(eps))
        fx = cache.get(key)
        if fx is None:
            fx = cache[key] = _Fusion(norm_w.data_ptr(), float(eps), None, None, None, None)
        self._check(self.lib.cdna4_fused_up_gate_fused(self.ctx, m, n, k, op, t, w_up.data_ptr(), w_gate.data_ptr(), w_up.stride(0), 0, x.data_ptr(), x.stride(0) * 4,
                                                       None, None, 0.0, out.data_ptr(), out.stride(0), C.addressof(fx), self._stream()))
        return out

    def fused_up_gate_q8(self, t, w_up, w_gate, x, op=UNARY["SILU"], out=None, q8_out=None, up_b=None, gate_b=None, limit=0.0):
        """decode fused up*gate that also emits the result as block_q8_2_x4 rows (input of the following mat-mul): returns (f32 [1, M], uint8 [1, M/128*144])."""
        torch = self.torch
        m = w_up.shape[0]; n, k = x.shape
        assert n == 1 and x.is_contiguous() and w_up.shape == w_gate.shape and w_up.stride(0) == w_gate.stride(0)
        if out is None:
            out = torch.empty((1, m), dtype=torch.float32, device=self.device)
        if q8_out is None:
            q8_out = torch.empty((1, m // 128 * 144), dtype=torch.uint8, device=self.device)
        self._check(self.lib.cdna4_fused_up_gate_q8(self.ctx, m, k, op, t, w_up.data_ptr(), w_gate.data_ptr(), w_up.stride(0), x.data_ptr(),
                                                    up_b.data_ptr() if up_b is not None else None, gate_b.data_ptr() if gate_b is not None else None,
                                                    float(limit), out.data_ptr(), q8_out.data_ptr(), self._stream()))
        return out, q8_out

    def mul_mat_id(self, t, ws, x, ids, out=None):
        """GGML_OP_MUL_MAT_ID: ws uint8 [E, M, row_size]; x f32 [T, n_b, K] (n_b in {1, n_used}); ids i32 [T, n_used] -> f32 [T, n_used, M]."""
        torch = self.torch
        e, m, _ = ws.shape; tk, nb, k = x.shape; nu = ids.shape[1]
        assert ids.dtype == torch.int32 and ids.is_cuda and x.is_contiguous() and ws.is_contiguous() and ids.is_contiguous()
        if out is None:
            out = torch.empty((tk, nu, m), dtype=torch.float32, device=self.device)
        self._check(self.lib.cdna4_mul_mat_id(self.ctx, m, k, e, nu, tk, t, ws.data_ptr(), ws.stride(1), ws.stride(0),
                                              x.data_ptr(), nb, x.stride(1) * 4, x.stride(0) * 4, ids.data_ptr(), ids.stride(0) * 4,
                                              out.data_ptr(), out.stride(1), out.stride(0), self._stream()))
        return out

    def moe_fused_up_gate(self, t, ws_up, ws_gate, x, ids, op=UNARY["SILU"], out=None, up_b=None, gate_b=None, limit=0.0):
        """GGML_OP_MOE_FUSED_UP_GATE; up_b / gate_b: f32 [E, M] per-expert biases or None."""
        torch = self.torch
        e, m, _ = ws_up.shape; tk, nb, k = x.shape; nu = ids.shape[1]
        for b in (up_b, gate_b):
            assert b is None or (b.dtype == torch.float32 and b.is_cuda and b.shape == (e, m) and b.stride(1) == 1)
        if out is None:
            out = torch.empty((tk, nu, m), dtype=torch.float32, device=self.device)
        self._check(self.lib.cdna4_moe_fused_up_gate_ext(self.ctx, m, k, e, nu, tk, op, t, ws_up.data_ptr(), ws_gate.data_ptr(), ws_up.stride(1),
                                                         ws_up.stride(0), x.data_ptr(), nb, x.stride(1) * 4, x.stride(0) * 4, ids.data_ptr(),
                                                         ids.stride(0) * 4,
                                                         up_b.data_ptr() if up_b is not None else None, up_b.stride(0) * 4 if up_b is not None else 0,
                                                         gate_b.data_ptr() if gate_b is not None else None, gate_b.stride(0) * 4 if gate_b is not None else 0,
                                                         float(limit), out.data_ptr(), out.stride(1), out.stride(0), self._stream()))
        return out

    def repack_r4(self, base_t, w, k):
        """iqk_repack_tensor: base-type rows -> row-interleaved _R4 rows (same shape / stride)."""
        out = self.torch.empty_like(w)
        self._check(self.lib.cdna4_repack_r4(self.ctx, base_t, w.data_ptr(), w.shape[0], k, out.data_ptr(), self._stream()))
        return out

    def unrepack_r4(self, base_t, w, k):
        """inverse of repack_r4 (bit-exact)."""
        out = self.torch.empty_like(w)
        self._check(self.lib.cdna4_unrepack_r4(self.ctx, base_t, w.data_ptr(), w.shape[0], k, out.data_ptr(), self._stream()))
        return out

    def invalidate_weight_cache(self, w=None):
        self._check(self.lib.cdna4_invalidate_weight_cache(self.ctx, w.data_ptr() if w is not None else None))

    # ---- tensor parallel reduce (GGML_OP_REDUCE)
    def comm_unique_id(self):
        buf = C.create_string_buffer(128); self._check(self.lib.cdna4_comm_unique_id(buf)); return buf.raw

    def comm_init(self, unique_id, rank, world):
        self.comm = self.lib.cdna4_comm_init(self.ctx, unique_id, rank, world)
        if not self.comm:
            raise Cdna4Error(-3, self.lib.cdna4_last_error().decode())

    def reduce(self, buf, wire=None):
        """GGML_OP_REDUCE (ADD), in place: every rank ends with the sum of all ranks' `buf`.  wire (f32 buf only): 16-bit type the partials may travel in."""
        torch = self.torch
        nb = buf.numel() * (buf.element_size() if wire is None else 2)
        if getattr(self, "window", None) and nb <= self.window_bytes and nb % 16 == 0 and buf.data_ptr() % 16 == 0 and buf.is_contiguous():    # one-shot over the IPC windows
            return self.window_reduce(buf, wire=wire)
        if wire is not None:                   # RCCL reduces in the wire type
            w = buf.to(wire); self.reduce(w); buf.copy_(w); return buf
        dt = {torch.float32: T["F32"], torch.float16: T["F16"], torch.bfloat16: T["BF16"]}[buf.dtype]
        self._check(self.lib.cdna4_all_reduce_sum(self.comm, buf.data_ptr(), buf.numel(), dt, self._stream()))
        return buf

    # ---- one-shot all-reduce over IPC-mapped windows (one process per GPU, no collective library)
    def window_create(self, rank, world, max_bytes):
        """returns this rank's 64-byte window handle; ship it to the peers and window_attach theirs"""
        buf = C.create_string_buffer(64)
        self.window = self.lib.cdna4_window_create(self.ctx, rank, world, max_bytes, buf)
        if not self.window:
            raise Cdna4Error(-3, self.lib.cdna4_last_error().decode())
        self.window_bytes = max_bytes
        return buf.raw

    def window_attach(self, peer_rank, handle):
        self._check(self.lib.cdna4_window_attach(self.window, peer_rank, handle))

    def window_reduce(self, buf, check=False, wire=None):
        """in place: every rank ends with the sum of all ranks' `buf` (same call order on every rank); wire = torch.bfloat16 / float16: f32 partials
        travel in 16 bits (converted inside the launch)"""
        torch = self.torch
        types = {torch.float32: T["F32"], torch.float16: T["F16"], torch.bfloat16: T["BF16"]}
        self._check(self.lib.cdna4_window_all_reduce_sum_wire(self.window, buf.data_ptr(), buf.numel(), types[buf.dtype], types[wire or buf.dtype],
                                                              1 if check else 0, self._stream()))
        return buf

    def window_free(self):
        if getattr(self, "window", None):
            self.lib.cdna4_window_free(self.window); self.window = None

    def reduce_peers(self, bufs, partial_mask=None, n_slices=1, q8_0=False):
        """in-process GGML_OP_REDUCE: bufs = list of same-shape tensors (or None); every tensor ends up holding the sum of the partials.
        q8_0: the tensors are uint8 images of block_q8_0 rows (34 bytes per 32 elements) -- the reference's reduce_type q8_0."""
        torch = self.torch
        ref = next(b for b in bufs if b is not None)
        if q8_0:
            assert ref.dtype == torch.uint8 and ref.numel() % 34 == 0
            return self._reduce_peers_raw(bufs, partial_mask, n_slices, ref.numel() // 34 * 32, 8)                # GGML_TYPE_Q8_0
        dt = {torch.float32: T["F32"], torch.float16: T["F16"], torch.bfloat16: T["BF16"]}[ref.dtype]
        return self._reduce_peers_raw(bufs, partial_mask, n_slices, ref.numel(), dt)

    def _reduce_peers_raw(self, bufs, partial_mask, n_slices, count, dt):
        if partial_mask is None:
            partial_mask = sum(1 << j for j, b in enumerate(bufs) if b is not None)
        arr = (C.c_void_p * len(bufs))(*[b.data_ptr() if b is not None else None for b in bufs])
        if n_slices > 1:      # the sliced form: one launch per slice (a multi-GPU host launches slice d on device d's context / stream)
            for sl in range(n_slices):
                self._check(self.lib.cdna4_reduce_peers_slice(self.ctx, arr, len(bufs), partial_mask, count, dt, sl, n_slices, self._stream()))
            return
        self._check(self.lib.cdna4_reduce_peers(self.ctx, arr, len(bufs), partial_mask, count, dt, self._stream()))

    def time_mul_mat(self, t, weights, x, out, warmup=3, iters=20):
        """avg ms per launch, HIP events on the launch stream; `weights` = list of rotating weight buffers (cold-cache)."""
        arr = (C.c_void_p * len(weights))(*[w.data_ptr() for w in weights])
        ms = C.c_float(0)
        w0 = weights[0]; n, k = x.shape
        self._check(self.lib.cdna4_time_mul_mat(self.ctx, w0.shape[0], n, k, t, arr, len(weights), w0.stride(0), x.data_ptr(), x.stride(0) * 4,
                                                out.data_ptr(), out.stride(0), warmup, iters, self._stream(), C.byref(ms)))
        return ms.value
