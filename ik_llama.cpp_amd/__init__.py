"""ik_llama.cpp_amd -- MI355X-native (gfx950) quantized mat-mul backend: host-side mirror of the reference interface.

The product is the C-ABI shared library `libggml-hip-cdna4.so` (include/ggml_hip_cdna4.h).  This package is the thin
Python host binding over it (ctypes; torch is used only for device memory / streams), mirroring the reference's
operator interface for this path: `mul_mat` (GGML_OP_MUL_MAT), `mul_mat_id` (MUL_MAT_ID), `fused_up_gate`
(FUSED_UP_GATE), `moe_fused_up_gate` (MOE_FUSED_UP_GATE), `dequantize` (type_traits.to_float), `quantize_activations`
(type_traits.from_float of vec_dot_type) and `reduce` (GGML_OP_REDUCE).

There is NO CPU fallback: if the HIP library is missing or no GPU is visible, construction fails loudly.
"""
from .cdna4 import (Cdna4Backend, Cdna4Error, load_library, lib_path, GGML_TYPE, UNARY, row_size, vec_dot_type,  # noqa: F401
                    act_row_size, TYPE_SIZE, BLCK_SIZE, BASE_TYPES, R4_TYPES, R4_OF, BASE_OF)
