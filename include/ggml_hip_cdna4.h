/*
 * ggml_hip_cdna4.h -- C ABI of the MI355X (gfx950 / CDNA4) quantized mat-mul backend.
 *
 * This is the drop-in boundary for ik_llama.cpp's quantized mat-mul hot path.  Every entry point is
 * `extern "C"`, takes plain pointers / sizes (device pointers are raw HIP device addresses, streams are
 * raw hipStream_t passed as void*), and states which reference interface it replaces.  No torch, no ggml
 * types appear in the signatures, so the library can be bound from C, C++ (the ggml-backend shim in
 * ik_llama.cpp_amd/backend/), or ctypes (ik_llama.cpp_amd/cdna4.py).
 *
 * Type ids are the reference's `enum ggml_type` values (ggml/include/ggml.h:391-470), so a caller passes
 * `tensor->type` unchanged.
 *
 * Conventions (same as the reference's iqk C ABI, ggml/src/iqk/iqk_mul_mat.h:16-39):
 *   A  : quantized weights, Nx rows of ne00 elements, `strideA` BYTES between rows (ggml nb01)
 *   B  : activations, Ny rows ("columns" of the mat-mul) of ne00 elements, `strideB` BYTES between rows (nb11)
 *   C  : f32 result, C[iy * stride_C + ix], `stride_C` in ELEMENTS (nb1 / sizeof(float))
 * All functions return 0 on success, a negative CDNA4_E_* code otherwise (no exceptions, no abort: the
 * ggml shim turns errors into GGML_STATUS_FAILED / GGML_ABORT like ggml-cuda.cu:136-147 does).
 */
#ifndef GGML_HIP_CDNA4_H
#define GGML_HIP_CDNA4_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CDNA4_API __attribute__((visibility("default")))

/* ggml_type ids on the hot path (ggml/include/ggml.h:391-470) */
enum cdna4_type {
    CDNA4_TYPE_F32 = 0, CDNA4_TYPE_F16 = 1,
    CDNA4_TYPE_Q4_K = 12, CDNA4_TYPE_Q5_K = 13, CDNA4_TYPE_Q6_K = 14, CDNA4_TYPE_Q8_K = 15,
    CDNA4_TYPE_IQ4_NL = 20, CDNA4_TYPE_IQ3_S = 21, CDNA4_TYPE_IQ2_S = 22,
    CDNA4_TYPE_BF16 = 30,
    CDNA4_TYPE_Q8_2_X4 = 99, CDNA4_TYPE_Q8_K32 = 148,
    CDNA4_TYPE_Q4_K_R4 = 212, CDNA4_TYPE_Q5_K_R4 = 213, CDNA4_TYPE_Q6_K_R4 = 214,
    CDNA4_TYPE_IQ4_NL_R4 = 220, CDNA4_TYPE_IQ3_S_R4 = 221, CDNA4_TYPE_IQ2_S_R4 = 222,
    /* row-interleaved forms that are re-tiled to their base type on the HOST at upload (cdna4_retile_r4_host below) */
    CDNA4_TYPE_IQ2_K_R4 = 337, CDNA4_TYPE_IQ3_K_R4 = 338, CDNA4_TYPE_IQ4_K_R4 = 339, CDNA4_TYPE_IQ5_K_R4 = 340,
    CDNA4_TYPE_IQ4_KS_R4 = 344, CDNA4_TYPE_IQ5_KS_R4 = 352,
    CDNA4_TYPE_Q4_0_R8 = 202, CDNA4_TYPE_Q5_0_R4 = 206, CDNA4_TYPE_Q8_0_R8 = 208, CDNA4_TYPE_Q6_0_R4 = 233, CDNA4_TYPE_MXFP4_R8 = 353,
    CDNA4_TYPE_Q2_K_R4 = 210, CDNA4_TYPE_Q3_K_R4 = 211, CDNA4_TYPE_IQ4_XS_R8 = 223,
    CDNA4_TYPE_IQ2_XXS_R4 = 216, CDNA4_TYPE_IQ2_XS_R4 = 217, CDNA4_TYPE_IQ3_XXS_R4 = 218, CDNA4_TYPE_IQ2_BN_R4 = 335,
};
/* An _R4 tensor whose bytes the host already un-interleaved to the base tiling (cdna4_unrepack_r4 at upload: what the ggml shim does in
 * set_tensor): pass CDNA4_TYPE_PRETILED(type).  The mat-mul then runs the base-layout kernels on the bytes as they are
 * (no shadow copy) with the _R4 kernels' activation arithmetic (Q8_K32 / Q8_K, ggml.c:989,1019,1050,1837-1848). */
#define CDNA4_TYPE_PRETILED(r4_type) ((r4_type) + 1000)

/* enum ggml_unary_op values used by the fused up*gate epilogue (ggml/include/ggml.h GGML_UNARY_OP_*;
 * iqk_mul_mat.cpp:129-135) */
enum cdna4_unary { CDNA4_UNARY_RELU = 6, CDNA4_UNARY_SILU = 10, CDNA4_UNARY_SWIGLU_OAI = 14, CDNA4_UNARY_GELU = 15 };   /* ggml.h:721-743 of this fork */

enum cdna4_status {
    CDNA4_OK = 0,
    CDNA4_E_UNSUPPORTED = -1,   /* type / shape not handled: caller falls back (supports_op == false)     */
    CDNA4_E_INVALID     = -2,   /* bad argument                                                            */
    CDNA4_E_HIP         = -3,   /* HIP runtime error; see cdna4_last_error()                               */
    CDNA4_E_NOMEM       = -4,   /* workspace too small and growth not allowed (stream capture)             */
};

/* how the N > 8 (prompt) path treats activations */
enum cdna4_prefill_mode {
    CDNA4_PREFILL_MFMA_F16 = 0, /* dequant -> f16 tiles, v_mfma_f32_32x32x16_f16, f32 accumulate (default)   */
    CDNA4_PREFILL_INT8_DOT = 1, /* reuse the decode int8-dot kernels column-group by column-group (CPU-arith.) */
    CDNA4_PREFILL_MFMA_F16_EXACT = 2, /* as MFMA_F16, but every weight enters the tile as its L0 (to_float) value rounded ONCE to f16: the quantized
                                    * weights are de-quantized chunk-wise to f16 and run through the f16 instance of the same GEMM.  The default tiles of
                                    * Q4_K / Q6_K round the block scale d*sc (and dmin*m) to f16 BEFORE the product (one extra 2^-11 rounding, the same the
                                    * reference's own prompt repack applies, iqk_gemm_kquants.cpp:2241-2249); this mode is the run-time parity switch that
                                    * keeps the north-star bar (1e-3 of sum|w*x|) on inputs where one activation dominates a row.  ~2x slower.        */
};

typedef struct cdna4_context cdna4_context;

/* ---- device / context --------------------------------------------------------------------------------
 * replaces ggml_backend_cuda_get_device_count / _get_device_description / _get_device_memory / _init
 * (ggml/include/ggml-cuda.h:24-47; ggml-cuda.cu:5392-5420). */
CDNA4_API int            cdna4_get_device_count(void);
CDNA4_API int            cdna4_get_device_description(int device, char *buf, size_t buf_size);
CDNA4_API int            cdna4_get_device_memory(int device, size_t *free_bytes, size_t *total_bytes);
CDNA4_API cdna4_context *cdna4_init(int device);            /* NULL on bad device, like ggml_backend_cuda_init */
CDNA4_API void           cdna4_free(cdna4_context *ctx);
CDNA4_API const char    *cdna4_last_error(void);            /* thread-local message of the last failure        */
CDNA4_API const char    *cdna4_version(void);
/* Diagnostics: which kernel instantiation and grid the calling thread's last prompt-batch (Ny > 8) mat-mul launch used, e.g.
 * "gemm_mfma type=12 nt=4 upgate=1 kx=128 ks=1 mw=2 xw=0 part=0 grid=448x1x1 ksplit=1 g=4"; since round 6 also the decode mat-vec launches of the main families
 * ("gemv type=14 ncols=1 upgate=0 yiters=1 nr=2 lpr=64 fx=0 waves=4 grid=1024x1x1", "gemv_sliced ...").  Empty before the first such launch.  No reference
 * counterpart (the reference's tile choice is compile-time per ISA, iqk_mul_mat.cpp:537-571); the parity tests use it to prove WHICH geometry they compared. */
CDNA4_API const char    *cdna4_last_launch_info(void);
/* Start-up self-test of the fence-free in-launch hand-offs (split-K prompt GEMM slabs, split-KV decode attention partials: write-through stores + one agent-scope ticket, MI355X
 * guide Guideline 16) against their unsplit forms on the live device; cdna4_init runs it once.  cdna4_handoff_mode: 0 = fence-free forms validated, 1 = fenced forms by request
 * (CDNA4_SPLITK_FENCE=1), 2 = fenced forms after a failed self-test, -1 = not tested.  No reference counterpart (the reference's CPU path joins threads at a barrier,
 * ggml.c:17955-17964; its CUDA split-K / split-KV kernels use separate launches). */
CDNA4_API int            cdna4_handoff_selftest(cdna4_context *ctx, void *stream);
CDNA4_API int            cdna4_handoff_mode(const cdna4_context *ctx);
/* Diagnostics / A-B switch, process-wide: which prompt-GEMM form the dense launches of the six scope types take.  1 (default; CDNA4_GEMM_WLDS in the environment sets the
 * initial value): 256-token workgroup tiles whose weight tile is de-quantized once into LDS ("gemm_wlds") where that grid fills the GPU, the per-wave de-quantizing kernel
 * ("gemm_mfma") elsewhere; 0: "gemm_mfma" everywhere; 2: "gemm_wlds" wherever it can run, however few workgroups; 3: "gemm_pp" (round 6: the same tile with the two waves of a SIMD alternating between
 * matrix and load / de-quantize intervals) wherever it can run; 4: the large-batch route from 256 tokens on -- the weights expanded once per mat-mul into the f16 image the
 * fused kernels build in registers ("dequant_slab") and a type-independent f16 x f16 kernel over two LDS-DMA-fed images ("gemm_ppf"); form 1 takes it by itself from 2048
 * tokens on (Q4_K / Q5_K fused up*gate: 3072; CDNA4_PPF_MIN_N moves the threshold).  The reference's own large-batch route is convert + cuBLAS (ggml-cuda.cu:1723).
 * Same products, same accumulation order in every form: results do not change. */
CDNA4_API int            cdna4_set_gemm_form(int form);

/* Threading / streams: a context serves ONE stream at a time (one ggml backend = one context = one stream, like the CUDA backend's per-device
 * context): its workspace is shared by every call, so two host threads or two streams must not use the same context concurrently.  Different
 * contexts (also on the same device) are independent.
 * Determinism: decode results are bit-reproducible.  A prompt GEMM whose (rows x tokens) grid cannot fill the chip splits K over grid.z and
 * accumulates the partial sums with hardware f32 atomics, so its low-order bits can differ from run to run (CDNA4_GEMM_KSPLIT_MULT=0 turns the grid-level split off).
 *
 * Scratch for quantized activations / f16 activation tiles.  Grown on demand by the mat-mul entry points
 * unless the stream is capturing; call this up front (ggml's graph_plan / reserve step) to make the
 * compute path allocation-free.  Mirrors the CUDA backend's pool (ggml-cuda/common.cuh ggml_cuda_pool). */
CDNA4_API int cdna4_reserve_workspace(cdna4_context *ctx, size_t bytes);
/* Load the device code of the prompt-batch kernels for weights of `type` on the CURRENT device now rather than at their first launch (the runtime loads a translation unit's code
 * object lazily: a few ms inside the first prompt pass otherwise).  Optional; results never depend on it.  No reference counterpart (CUDA loads modules eagerly). */
CDNA4_API int cdna4_preload_type(int type);
/* number of times the workspace has been (re-)allocated: a HIP graph captured by the caller holds the workspace address of its capture time and must be
 * dropped when this changes */
CDNA4_API long cdna4_workspace_epoch(cdna4_context *ctx);

/* ---- type traits (a13: ggml.c:679-1960 type_traits[], ggml_row_size ggml.c:4808-4811) ------------------- */
/* Weight types served (enum ggml_type values of the reference, ggml.h:391-470): every base type of the reference CUDA backend's MUL_MAT list (ggml-cuda.cu:4855-4892) --
 * Q4_0 Q4_1 Q5_0 Q5_1 Q6_0 Q8_0 Q2_K Q3_K Q4_K Q5_K Q6_K IQ1_S IQ1_M IQ2_XXS IQ2_XS IQ2_S IQ3_XXS IQ3_S IQ4_NL IQ4_XS MXFP4 IQ2_K IQ3_K IQ4_K IQ5_K IQ6_K IQ2_KS IQ3_KS
 * IQ4_KS IQ5_KS IQ4_KSS IQ2_KL IQ1_BN IQ2_BN IQ1_KT IQ2_KT IQ3_KT IQ4_KT -- plus the row-interleaved Q4_K_R4 Q5_K_R4 Q6_K_R4 IQ4_NL_R4 IQ2_S_R4 IQ3_S_R4.
 * Eighteen further row-interleaved forms (IQ2_K_R4 ... IQ5_KS_R4, Q4_0_R8 ... IQ2_BN_R4) are not types of this table: the host converts them to their base type once, at upload
 * (cdna4_retile_r4_host below), and passes the base type id from then on.
 * Trellis types (IQ2_KT / IQ3_KT): mat-mul results carry the 1.05 / 1.01 row factor of the reference's mat-mul kernels, cdna4_dequantize_rows / cdna4_op_get_rows return its scalar
 * to_float. */
CDNA4_API int    cdna4_type_supported(int type);            /* 1 if MUL_MAT with this src0 type is handled     */
CDNA4_API int    cdna4_blck_size(int type);
CDNA4_API size_t cdna4_type_size(int type);
CDNA4_API size_t cdna4_row_size(int type, int64_t ne00);
CDNA4_API int    cdna4_vec_dot_type(int type);              /* activation quant type of the CPU path (a10)     */

/* ---- dequantize (L0 parity: bit-identical to type_traits.to_float / dequantize_row_*) -------------------
 * replaces dequantize_row_{q4_K,q5_K,q6_K,iq4_nl,iq2_s,iq3_s} (ggml-quants.c:2797,3015,3231,3913,3729,3793),
 * dequantize_row_*_r4 (iqk_quantize.cpp:6118,6229,6342,5255,7871,8063) and the CUDA dequantize_block_* kernels
 * (ggml-cuda/convert.cu:227-722).  dst_type is CDNA4_TYPE_F32 or CDNA4_TYPE_F16; dst stride in elements.
 * For _R4 types nrows % 4 == 0 and row r of the output is logical row r (de-interleaved). */
CDNA4_API int cdna4_dequantize_rows(cdna4_context *ctx, int type, const void *A, int64_t strideA,
                                    int64_t nrows, int64_t ne00, void *dst, int dst_type, int64_t dst_stride,
                                    void *stream);

/* ---- activation quantizers (a10) ------------------------------------------------------------------------
 * replaces quantize_row_q8_2_x4 (iqk_quantize.cpp:1175), iqk_quantize_row_q8_K (:3932), quantize_row_q8_K32 (:3936)
 * and the CUDA quantize_q8_1 kernels (ggml-cuda/quantize.cu).  Output byte layout == the reference's
 * block_q8_2_x4 / block_q8_K, bit for bit.  `dst` rows are cdna4_row_size(vec_dot_type, ne00) bytes apart. */
CDNA4_API int cdna4_quantize_rows(cdna4_context *ctx, int vec_dot_type, const float *B, int64_t strideB,
                                  int64_t nrows, int64_t ne00, void *dst, void *stream);

/* ---- MUL_MAT (a1-a9) --------------------------------------------------------------------------------------
 * replaces iqk_mul_mat (iqk_mul_mat.h:16-19) / ggml_compute_forward_mul_mat (ggml.c:17863-18096) and
 * ggml_cuda_mul_mat (ggml-cuda.cu:2645-2727).
 * typeB: CDNA4_TYPE_F32 (activations are quantized / converted on the device as part of the call) or, for
 * Ny <= 8 only, the matching cdna4_vec_dot_type(typeA) (already-quantized rows, the iqk_mul_mat contract).
 * Ny <= 8  : decode GEMV, int8 activations + exact int32 block sums (the CPU path's arithmetic).
 * Ny  > 8  : prefill, see cdna4_set_prefill_mode. */
CDNA4_API int cdna4_mul_mat(cdna4_context *ctx, long Nx, long Ny, long ne00,
                            int typeA, const void *A, long strideA,
                            int typeB, const void *B, long strideB,
                            float *C, long stride_C, void *stream);

/* Several weight matrices applied to the SAME activations (attention q/k/v): replaces the reference's fusion of consecutive
 * MUL_MATs that share src1 (ggml.c:17984-18000 CPU, ggml-cuda.cu:2570-2600 CUDA).  Matrices of equal type / row stride are
 * served by one decode launch (Ny == 1); anything else falls back to one cdna4_mul_mat per matrix.  n_mats <= 16. */
CDNA4_API int cdna4_mul_mat_multi(cdna4_context *ctx, int n_mats, const long *Nx, long Ny, long ne00,
                                  const int *typeA, const void *const *A, const long *strideA,
                                  int typeB, const void *B, long strideB,
                                  float *const *C, const long *stride_C, void *stream);

/* batched / broadcast form, replaces iqk_mul_mat_4d (iqk_mul_mat.h:21-26): strides nb02.. in bytes,
 * nb2/nb3 of the result in elements; ne12 % ne02 == 0 and ne13 % ne03 == 0 (ggml broadcast rule). */
CDNA4_API int cdna4_mul_mat_4d(cdna4_context *ctx, long Nx, long Ny, long ne00,
                               long ne02, long ne03, long ne12, long ne13,
                               long nb02, long nb03, long nb12, long nb13, long nb2, long nb3,
                               int typeA, const void *A, long strideA,
                               int typeB, const void *B, long strideB,
                               float *C, long stride_C, void *stream);

/* fused dst = unary(gate.x) * (up.x), replaces ggml_compute_forward_mul_mat_up_gate (ggml.c:18653-18722)
 * and ggml_cuda_up_gate_unary (ggml-cuda.cu:3542).  Aup/Agate have identical type and shape. */
CDNA4_API int cdna4_fused_up_gate(cdna4_context *ctx, long Nx, long Ny, long ne00, int unary_op,
                                  int typeA, const void *Aup, const void *Agate, long strideA,
                                  int typeB, const void *B, long strideB,
                                  float *C, long stride_C, void *stream);
/* the full epilogue of mul_mat_up_gate_NxM (iqk_mul_mat.cpp:136-236; arguments of iqk_moe_fused_up_gate iqk_mul_mat.h / .cpp:783-787):
 *   t = act(gate.x + gate_b) ; limit > 1e-6 => t = min(t, limit)
 *   u = up.x + up_b ; SWIGLU_OAI => u = 1 + clamp(u, -7, 7), else limit > 1e-6 => u = clamp(u, -limit, limit) ;  C = u * t
 * up_b / gate_b: device f32 [Nx] or NULL; limit = dst->op_params[1] (0 = off) */
CDNA4_API int cdna4_fused_up_gate_ext(cdna4_context *ctx, long Nx, long Ny, long ne00, int unary_op,
                                      int typeA, const void *Aup, const void *Agate, long strideA,
                                      int typeB, const void *B, long strideB,
                                      const float *up_b, const float *gate_b, float limit,
                                      float *C, long stride_C, void *stream);

/* Decode form (one activation row) that ALSO emits the result row quantized to block_q8_2_x4 (ggml-common.h:287-299) -- byte-identical to
 * quantize_row_q8_2_x4 (iqk_quantize.cpp:1072-1175) applied to the f32 result -- so that the mat-mul consuming it (ffn_down) can be called
 * with typeB = GGML_TYPE_Q8_2_X4 and skips its activation quantization.  The reference's CUDA path fuses the same re-quantisation between
 * the fused up*gate and the down mat-mul (ggml-cuda.cu:3062-3185).  Requires ne00 == 4096 (one 64-lane slice per row), Nx % 128 == 0, a
 * base (non-_R4) type; q8_out holds Nx / 128 blocks of 144 bytes.  CDNA4_E_UNSUPPORTED otherwise: call cdna4_fused_up_gate_ext instead. */
CDNA4_API int cdna4_fused_up_gate_q8(cdna4_context *ctx, long Nx, long ne00, int unary_op,
                                     int typeA, const void *Aup, const void *Agate, long strideA,
                                     const float *B, const float *up_b, const float *gate_b, float limit,
                                     float *C, void *q8_out, void *stream);

/* MUL_MAT_ID, replaces iqk_mul_mat_moe (iqk_mul_mat.h:28-31) / ggml_compute_forward_mul_mat_id (ggml.c:18100-18416)
 * and ggml_cuda_mul_mat_id (ggml-cuda.cu:2836-3033) WITHOUT the host-side row mapping / D2H sync:
 *   as  : [n_expert][Nx] rows, expert e starts at A + e*nb02
 *   B   : f32 [n_tokens][n_b][ne00] (n_b == n_used, or 1 = same activation for every slot), strides nb11 (slot), nb12 (token) bytes
 *   ids : device int32 [n_tokens][n_used] (row stride ids_nb1 bytes); id < 0 or >= n_expert => zero output row
 *   C   : f32 [n_tokens][n_used][Nx], strides nb1 (slot), nb2 (token) in elements */
CDNA4_API int cdna4_mul_mat_id(cdna4_context *ctx, long Nx, long ne00, int n_expert, int n_used, long n_tokens,
                               int typeA, const void *A, long strideA, long nb02,
                               const float *B, int n_b, long nb11, long nb12,
                               const int32_t *ids, long ids_nb1,
                               float *C, long nb1, long nb2, void *stream);

/* fused MoE up*gate, replaces iqk_moe_fused_up_gate (iqk_mul_mat.h:33-37) / ggml_cuda_moe_up_gate_unary
 * (ggml-cuda.cu:3035): C[t][s][:] = unary(gate_e.x) * (up_e.x), e = ids[t][s]. */
CDNA4_API int cdna4_moe_fused_up_gate(cdna4_context *ctx, long Nx, long ne00, int n_expert, int n_used, long n_tokens,
                                      int unary_op, int typeA, const void *Aup, const void *Agate, long strideA, long nb02,
                                      const float *B, int n_b, long nb11, long nb12,
                                      const int32_t *ids, long ids_nb1,
                                      float *C, long nb1, long nb2, void *stream);
/* ... with the per-expert biases of src[4] / src[5] (ggml.c:18429-18456,18577-18590): bias of expert e = (char *)up_b + e * up_b_nb1 */
CDNA4_API int cdna4_moe_fused_up_gate_ext(cdna4_context *ctx, long Nx, long ne00, int n_expert, int n_used, long n_tokens,
                                          int unary_op, int typeA, const void *Aup, const void *Agate, long strideA, long nb02,
                                          const float *B, int n_b, long nb11, long nb12,
                                          const int32_t *ids, long ids_nb1,
                                          const float *up_b, long up_b_nb1, const float *gate_b, long gate_b_nb1, float limit,
                                          float *C, long nb1, long nb2, void *stream);

/* The whole expert FFN block of a MoE layer as the CUDA backend runs it for decode-size batches -- MOE_FUSED_UP_GATE and the FOLLOWING
 * MUL_MAT_ID (down projection on the fused result, same ids) consumed together, two graph nodes in one call (ggml_cuda_moe_up_gate_unary,
 * ggml-cuda.cu:3062-3185):  C1[t][s][:] = act(gate_e.x + b_g) * (up_e.x + b_u)   (the first node's own output tensor, Nx_ff wide)
 *                           C2[t][s][:] = down_e . C1[t][s][:]                   (Nx_out wide), e = ids[t][s]. */
CDNA4_API int cdna4_moe_ffn(cdna4_context *ctx, long Nx_ff, long ne00, long Nx_out, int n_expert, int n_used, long n_tokens, int unary_op,
                            int type_up_gate, const void *Aup, const void *Agate, long stride_up_gate, long nb02_up_gate,
                            int type_down, const void *Adown, long stride_down, long nb02_down,
                            const float *B, int n_b, long nb11, long nb12, const int32_t *ids, long ids_nb1,
                            const float *up_b, long up_b_nb1, const float *gate_b, long gate_b_nb1, float limit,
                            float *C1, long c1_nb1, long c1_nb2, float *C2, long c2_nb1, long c2_nb2, void *stream);

CDNA4_API int cdna4_set_prefill_mode(cdna4_context *ctx, int mode);
/* Bit-reproducible prompts.  Prompt launches whose (rows x tokens) grid is smaller than the chip split K over several workgroups; by default the slices are accumulated
 * with f32 hardware atomics (the sum of the same terms in arrival order: the last bits of a prompt mat-mul can differ from run to run, as with the reference CUDA backend's
 * split-K / stream-K paths before their fix-up).  With `on` != 0 every slice stores its partial tile to the workspace and the workgroup that arrives last adds them in slice
 * order: identical bits every run, at the price of an agent-scope release / acquire pair per workgroup (+25-30 us on a 4096-row matrix at 512 tokens).  Decode launches
 * (N <= 8) and unsplit prompt launches are deterministic either way.  Environment: CDNA4_DETERMINISTIC=1. */
CDNA4_API int cdna4_set_deterministic(cdna4_context *ctx, int on);

/* ---- the non-mat-mul ops of a Llama / Mixtral graph: the steps on either side of the mat-mul path -------------------
 * Tensors are plain strided descriptors in ggml's convention (ne[] in elements, nb[] in BYTES, type = enum ggml_type; I32 = 26), so the
 * shim forwards `tensor->data / ne / nb` unchanged.  Each entry states the reference CPU function it restates and the CUDA file it replaces.
 * Shapes / types outside what an entry supports return CDNA4_E_UNSUPPORTED (supports_op == false). */
typedef struct cdna4_tensor { void *data; int type; int64_t ne[4]; int64_t nb[4]; } cdna4_tensor;
#define CDNA4_TYPE_I32 26

/* FUSED_RMS_NORM / RMS_NORM: y = x * rsqrt(mean(x^2) + eps) * w (w NULL: plain RMS_NORM); ggml.c:17420-17470, ggml-cuda/norm.cu */
/* ---- decode-token fusions across graph nodes (ggml-cuda fuses the same neighbours: ggml_cuda_op_fused_add_rms_norm, ggml-cuda.cu graph fusion) ----
 * norm_w / norm_eps: the f32 activation row is RMS-normed (x * rsqrt(mean(x^2) + eps) * norm_w) inside the mat-mul's prologue, before it is quantized --
 *                    FUSED_RMS_NORM + MUL_MAT(s) / FUSED_UP_GATE as one launch, the normed row is never written;
 * residual:          C = W x + residual (indexed like C) -- MUL_MAT + ADD as one launch.
 * One activation row (Ny == 1), f32, row length <= 8192; CDNA4_E_UNSUPPORTED otherwise (the caller issues the nodes separately). */
/* qkv (with norm_w, cdna4_mul_mat_multi_fused only): the epilogue of the q,k,v launch of one decoded token -- ROPE(q), ROPE(k), CPY(k -> K cache), CPY(v -> V cache) ride in
 * the mat-mul: rows of a kind-0 (Q) / kind-1 (K) matrix are rotated in NORM mode (pairs (2 i, 2 i + 1) of every head, the first n_dims of head_dim) with the context's rope
 * cache (cdna4_op_rope_cache of ONE token must be current); Q goes to C[i] as f32, K / V (kind 2: not rotated) as f16 to *kv_slot[i] (or kv_dst[i] when the slot is NULL). */
typedef struct cdna4_qkv_epilogue { int head_dim, n_dims; int kind[4]; void *kv_dst[4]; void *const *kv_slot[4]; } cdna4_qkv_epilogue;
/* PROMPT batches (Ny > 8, norm_w set, residual / qkv NULL; prefill mode MFMA_F16, base-type K-quant / IQ weights with a matrix-core tile, ne00 % 128 == 0, ne00 <= 16384, 16-byte aligned
 * rows): the RMS-normed rows go straight into the f16 activation image the GEMM streams -- FUSED_RMS_NORM + (f32 -> f16 image) as one launch, the normed f32 rows are never written.
 * add_b / add_dst (prompt batches only): the residual ADD in front of the norm rides along: x = B + add_b (rows at strideB), stored to add_dst (rows at strideB), then normed --
 * ADD + FUSED_RMS_NORM + image as one launch (ggml-cuda's ggml_cuda_op_fused_add_rms_norm + the quantize_q8_1 pass of the mat-mul behind it).  Same arithmetic, in the same order,
 * as cdna4_op_add_rms_norm / cdna4_op_rms_norm followed by the mat-mul. */
typedef struct cdna4_fusion { const float *norm_w; float norm_eps; const float *residual; const cdna4_qkv_epilogue *qkv; const float *add_b; float *add_dst; } cdna4_fusion;
CDNA4_API int cdna4_mul_mat_multi_fused(cdna4_context *ctx, int n_mats, const long *Nx, long Ny, long ne00, const int *typeA, const void *const *A, const long *strideA,
                                        int typeB, const void *B, long strideB, float *const *C, const long *stride_C, const cdna4_fusion *fx, void *stream);
CDNA4_API int cdna4_fused_up_gate_fused(cdna4_context *ctx, long Nx, long Ny, long ne00, int unary_op, int typeA, const void *A_up, const void *A_gate, long strideA,
                                        int typeB, const void *B, long strideB, const float *up_b, const float *gate_b, float limit, float *C, long stride_C,
                                        const cdna4_fusion *fx, void *stream);

/* One decoded token: FLASH_ATTN_EXT (q [128, 1, n_head], f16 K / V views of fewer keys than the split-KV threshold, optional f16 mask) + MUL_MAT (attn_output: Nx rows of
 * ne00 = 128 * n_head weights, type Q4_K / Q5_K / Q6_K / IQ4_NL) + ADD (residual, indexed like C) as ONE launch: the attention runs on the first n_head workgroups, the mat-vec
 * workgroups stream their weights meanwhile and take the attention row over agent-scope tickets.  `attn` = the FLASH_ATTN_EXT node's own result tensor ([128, n_head, 1], f32,
 * contiguous: it is written as in the unfused graph).  Bit-identical to cdna4_op_flash_attn + cdna4_mul_mat_multi_fused(residual).  CDNA4_E_UNSUPPORTED: issue the nodes separately.
 * Replaces ggml-cuda/fattn-vec-f16.cuh + mmvq.cu + binbcast.cu for llm_build_kqv's last three nodes (src/llama-build-context.cpp); CPU: ggml.c:22874-23160, ggml.c:17863. */
CDNA4_API int cdna4_attn_out_fused(cdna4_context *ctx, const cdna4_tensor *q, const cdna4_tensor *k, const cdna4_tensor *v, const cdna4_tensor *mask, const cdna4_tensor *attn,
                                   float scale, float max_bias, float softcap, long Nx, long ne00, int typeA, const void *A, long strideA, const float *residual, float *C, void *stream);

/* The attention of ONE decoded token on the per-head decode kernel (head size 128, f16 K / V of fewer keys than the split-KV threshold) that ALSO emits the result row quantized
 * to block_q8_2_x4 (n_head blocks of 144 bytes in `q8_out`; byte-identical to quantize_row_q8_2_x4, iqk_quantize.cpp:1072-1175, of the f32 row written to `dst`): the attn_output
 * mat-mul behind it is then called with typeB = GGML_TYPE_Q8_2_X4 (cdna4_mul_mat, or cdna4_mul_mat_multi_fused with a residual) and skips its activation quantization --
 * quantize src1 once, consume it everywhere, as ggml.c:17955-17964 / ggml-cuda.cu:2524-2600 do.  CDNA4_E_UNSUPPORTED for any other shape: call cdna4_op_flash_attn. */
CDNA4_API int cdna4_op_flash_attn_q8(cdna4_context *ctx, const cdna4_tensor *q, const cdna4_tensor *k, const cdna4_tensor *v, const cdna4_tensor *mask, const cdna4_tensor *dst,
                                     float scale, float max_bias, float softcap, void *q8_out, void *stream);
CDNA4_API int cdna4_op_rms_norm(cdna4_context *ctx, const cdna4_tensor *x, const cdna4_tensor *w, float eps, const cdna4_tensor *dst, void *stream);
/* ADD (op 0) / MUL (1) / DIV (2), src1 broadcast over src0 like ggml_can_repeat; ggml-cuda/binbcast.cu */
CDNA4_API int cdna4_op_binary(cdna4_context *ctx, int op, const cdna4_tensor *a, const cdna4_tensor *b, const cdna4_tensor *dst, void *stream);
/* ROPE, modes NORM (0) and NEOX (2), YaRN and frequency factors; pos = i32 [ne2]; ggml.c:20987-21230 (op_params 1,2,4-10), ggml-cuda/rope.cu */
CDNA4_API int cdna4_op_rope(cdna4_context *ctx, const cdna4_tensor *x, const int32_t *pos, const float *freq_factors, const cdna4_tensor *dst, int n_dims, int mode, int n_ctx_orig,
                            float freq_base, float freq_scale, float ext_factor, float attn_factor, float beta_fast, float beta_slow, void *stream);
/* CPY / DUP / CONT between f32 and f16 with arbitrary strides (KV-cache writes); ggml-cuda/cpy.cu */
/* (cos, sin) of every (token, rotated pair) computed once for the rope ops that follow on this context with the same positions / parameters (one graph: every layer
 * rotates with the same angles; ggml_rope_cache_init does the same on the CPU).  cdna4_op_rope_cache_reset invalidates it (call when a new graph starts). */
CDNA4_API int cdna4_op_rope_cache(cdna4_context *ctx, const int32_t *pos, int64_t n_tok, const float *freq_factors, int n_dims, int n_ctx_orig, float freq_base, float freq_scale,
                                  float ext_factor, float attn_factor, float beta_fast, float beta_slow, void *stream);
CDNA4_API int cdna4_op_rope_cache_reset(cdna4_context *ctx);
CDNA4_API int cdna4_op_cpy(cdna4_context *ctx, const cdna4_tensor *src, const cdna4_tensor *dst, void *stream);
/* the same with the destination base address read on the device from *dst_slot (dst->data is ignored when dst_slot != NULL): lets a captured HIP graph
 * be replayed while the KV-cache write position moves (ggml-cuda.cu:4480-4560 updates the copy kernels' parameters for the same purpose) */
CDNA4_API int cdna4_op_cpy_indirect(cdna4_context *ctx, const cdna4_tensor *src, const cdna4_tensor *dst, void *const *dst_slot, void *stream);
/* ADD + FUSED_RMS_NORM of one residual-stream row block in one pass (ggml-cuda.cu fuses the same pair: ggml_cuda_op_fused_add_rms_norm): sum = a + b, dst = rms_norm(sum) * w */
CDNA4_API int cdna4_op_add_rms_norm(cdna4_context *ctx, const cdna4_tensor *a, const cdna4_tensor *b, const cdna4_tensor *sum, const cdna4_tensor *w, float eps, const cdna4_tensor *dst, void *stream);
/* ROPE(q) + ROPE(k) + CPY(k -> f16 K-cache view) + CPY(v -> f16 V-cache view) of one layer in one launch (llm_build_kv_store; the reference fuses rope pairs:
 * ggml_cuda_op_fused_rope, ggml-cuda/rope.cu).  k_dst may be NULL when the rotated K is only consumed by the cache write.  k_slot / v_slot: as cdna4_op_cpy_indirect. */
CDNA4_API int cdna4_op_rope_store_kv(cdna4_context *ctx, const cdna4_tensor *q, const cdna4_tensor *q_dst, const cdna4_tensor *k, const cdna4_tensor *k_dst, const cdna4_tensor *k_cache, void *const *k_slot,
                                     const cdna4_tensor *v, const cdna4_tensor *v_cache, void *const *v_slot, const int32_t *pos, const float *freq_factors, int n_dims, int mode, int n_ctx_orig,
                                     float freq_base, float freq_scale, float ext_factor, float attn_factor, float beta_fast, float beta_slow, void *stream);
/* the same launch for attention with per-head norms (llm_build_mul_mat_qkv with q_norm / k_norm, llama-build-context.cpp:2481-2490; Qwen3, ...): FUSED_RMS_NORM(q) + ROPE(q) +
 * FUSED_RMS_NORM(k) + ROPE(k) + CPY(k -> K cache) + CPY(v -> V cache), bit-identical to the six launches.  q / k: UN-normed [head size, heads, tokens] f32 rows; q_norm / k_norm: one f32
 * row of head-size weights.  Whole-head rotation (n_dims == head size of 64, 128 or 256), the layouts of a llama graph and a current rope cache (cdna4_op_rope_cache) only:
 * CDNA4_E_UNSUPPORTED with nothing launched otherwise -- the caller then issues the nodes one by one. */
CDNA4_API int cdna4_op_norm_rope_store_kv(cdna4_context *ctx, const cdna4_tensor *q, const cdna4_tensor *q_norm, float eps_q, const cdna4_tensor *q_dst, const cdna4_tensor *k, const cdna4_tensor *k_norm, float eps_k,
                                          const cdna4_tensor *k_dst, const cdna4_tensor *k_cache, void *const *k_slot, const cdna4_tensor *v, const cdna4_tensor *v_cache, void *const *v_slot, const int32_t *pos,
                                          const float *freq_factors, int n_dims, int mode, int n_ctx_orig, float freq_base, float freq_scale, float ext_factor, float attn_factor, float beta_fast, float beta_slow,
                                          void *stream);
/* GET_ROWS: f32 / f16 / the six base quant types -> f32; ggml.c:19808, ggml-cuda/getrows.cu */
CDNA4_API int cdna4_op_get_rows(cdna4_context *ctx, const cdna4_tensor *src, const cdna4_tensor *ids, const cdna4_tensor *dst, void *stream);
/* SOFT_MAX(x * scale + slope * mask) over ne0; ggml.c:20300, ggml-cuda/softmax.cu */
CDNA4_API int cdna4_op_soft_max(cdna4_context *ctx, const cdna4_tensor *x, const cdna4_tensor *mask, const cdna4_tensor *dst, float scale, float max_bias, void *stream);
/* FLASH_ATTN_EXT: q f32 [D, n_tok, n_head], k / v f16 [D, n_kv, n_head_kv], mask f16 [n_kv, n_tok] -> dst f32 [D, n_head, n_tok];
 * ggml.c:22874-23160 (op_params: scale, max_bias, softcap), ggml-cuda/fattn*.cu */
CDNA4_API int cdna4_op_flash_attn(cdna4_context *ctx, const cdna4_tensor *q, const cdna4_tensor *k, const cdna4_tensor *v, const cdna4_tensor *mask, const cdna4_tensor *dst,
                                  float scale, float max_bias, float softcap, void *stream);
/* ARGSORT (MoE top-k): ids sorted by value, ties ordered like the reference's (value, index) pairs; iqk_cpu_ops.cpp:228-266, ggml-cuda/argsort.cu */
CDNA4_API int cdna4_op_argsort(cdna4_context *ctx, const cdna4_tensor *x, const cdna4_tensor *dst, int order_desc, void *stream);
CDNA4_API int cdna4_op_sum_rows(cdna4_context *ctx, const cdna4_tensor *x, const cdna4_tensor *dst, void *stream);
/* MUL_MULTI_ADD: dst[:, t] = sum_j a[:, j, t] * b[0, j, t] (weighted sum of the used experts); iqk_cpu_ops.cpp:430-500, ggml-cuda/multiadd.cu */
CDNA4_API int cdna4_op_mul_multi_add(cdna4_context *ctx, const cdna4_tensor *a, const cdna4_tensor *b, const cdna4_tensor *dst, void *stream);
/* the same + res[:, t]: MUL_MULTI_ADD followed by the residual ADD of the block as one launch */
CDNA4_API int cdna4_op_mul_multi_add_res(cdna4_context *ctx, const cdna4_tensor *a, const cdna4_tensor *b, const cdna4_tensor *res, const cdna4_tensor *dst, void *stream);
/* small dense MUL_MAT (f32 / f16 weights x f32 activations: the MoE router ffn_gate_inp) */
CDNA4_API int cdna4_op_mul_mat_dense(cdna4_context *ctx, const cdna4_tensor *w, const cdna4_tensor *x, const cdna4_tensor *dst, void *stream);
/* the MoE router of a batch in one launch: logits = w x, probs = softmax(logits), sorted = argsort descending, wsel = probs of the n_used best, wsum = their
 * sum, wnorm = wsel / wsum -- the MUL_MAT + SOFT_MAX + ARGSORT + GET_ROWS + SUM_ROWS + DIV chain of llm_build_moe_ffn (llama-build-context.cpp:1464-1556; CUDA:
 * ggml_cuda_op_topk_moe).  All six results are written.  w: [K, n_expert <= 64] f32 / f16, x: [K, n_tok] f32, sorted: i32 [n_expert, n_tok]. */
CDNA4_API int cdna4_op_moe_router(cdna4_context *ctx, const cdna4_tensor *w, const cdna4_tensor *x, const cdna4_tensor *logits, const cdna4_tensor *probs, const cdna4_tensor *sorted,
                                  const cdna4_tensor *wsel, const cdna4_tensor *wsum, const cdna4_tensor *wnorm, int n_used, void *stream);
/* the same launch with the FUSED_RMS_NORM in front of the router folded in (llm_build_moe_ffn reads ffn_norm's result for the router AND for the experts): x is the UN-normed
 * row, x_normed receives rms_norm(x) * norm_w (still needed by the experts), the logits are computed from it.  Bit-identical to cdna4_op_rms_norm + cdna4_op_moe_router.
 * f32 router weights of <= 8 experts, rows of 1024 ... 4096 values, 16-byte aligned rows: CDNA4_E_UNSUPPORTED otherwise (nothing launched). */
CDNA4_API int cdna4_op_moe_router_norm(cdna4_context *ctx, const cdna4_tensor *w, const cdna4_tensor *x, const cdna4_tensor *norm_w, float norm_eps, const cdna4_tensor *x_normed, const cdna4_tensor *logits,
                                       const cdna4_tensor *probs, const cdna4_tensor *sorted, const cdna4_tensor *wsel, const cdna4_tensor *wsum, const cdna4_tensor *wnorm, int n_used, void *stream);

/* ---- run-time repack to the row-interleaved layouts (a8) ------------------------------------------------
 * replaces iqk_repack_tensor (iqk_quantize.cpp:8535-8582): base type -> *_R4, on the device, out of place.
 * nrows % 4 == 0; row stride is unchanged. */
CDNA4_API int cdna4_repack_r4(cdna4_context *ctx, int base_type, const void *A, int64_t nrows, int64_t ne00,
                              void *dst, void *stream);
/* the inverse permutation (bit-exact): *_R4 rows -> base-type rows.  MI355X kernels run on the base tiling; mat-mul entry
 * points given an *_R4 typeA convert the tensor ONCE (keyed by its device pointer) and keep the base-layout shadow until
 * cdna4_invalidate_weight_cache(ctx, A) (A == NULL: everything) is called -- call it when the tensor's bytes change or its buffer
 * is freed (what ggml_backend_cuda_invalidate_graphs / buffer free mean for the CUDA backend).  The activation arithmetic stays
 * the _R4 kernels' (Q8_K32 / Q8_K), so results match the reference's _R4 CPU kernels, not the base-type ones. */
CDNA4_API int cdna4_unrepack_r4(cdna4_context *ctx, int base_type, const void *A, int64_t nrows, int64_t ne00,
                                void *dst, void *stream);
CDNA4_API int cdna4_invalidate_weight_cache(cdna4_context *ctx, const void *A);
/* Every other row-interleaved form whose base type is served -- IQ2_K_R4 IQ3_K_R4 IQ4_K_R4 IQ5_K_R4 IQ4_KS_R4 IQ5_KS_R4 (the ones the reference CUDA backend lists
 * for MUL_MAT, ggml-cuda.cu:4893-4898, and serves through de-quantization, ggml-cuda/convert.cu), Q4_0_R8 Q5_0_R4 Q6_0_R4 Q8_0_R8 MXFP4_R8 Q2_K_R4 Q3_K_R4 IQ4_XS_R8
 * IQ2_XXS_R4 IQ2_XS_R4 IQ3_XXS_R4 IQ2_BN_R4 (CPU-only in the reference) -- is re-tiled on the HOST, once, between the file bytes and the H2D copy: to_base != 0 turns
 * nrows interleaved rows into rows of the base type (which every mat-mul entry point then serves AS that base type: cdna4_retile_r4_host_base_type), to_base == 0
 * is the exact inverse = the reference's repack_* functions (iqk_quantize.cpp:5304-8088; byte-identical to iqk_repack_tensor, the function behind
 * `llama-quantize --repack` and -rtr; tests/test_retile_host.py).  Pure host code (no context, no device): src and dst are host buffers of
 * nrows * cdna4_row_size(base, ne00) bytes, out of place; nrows must be a multiple of the group size (cdna4_retile_r4_host_rows: 4 or 8), ne00 of the base type's
 * block size; n_threads <= 0: one per hardware thread.  Not covered: IQ1_S_R4 / IQ1_M_R4 (formats of their own: 32-weight blocks behind a row scale, no base-type
 * twin), Q8_K_R8 / Q8_KV_R8 / BF16_R16 (activation / KV-cache side types). */
CDNA4_API int cdna4_retile_r4_host(int r_type, const void *src, void *dst, int64_t nrows, int64_t ne00, int to_base, int n_threads);
CDNA4_API int cdna4_retile_r4_host_base_type(int r_type);       /* the base type id, -1 if r_type is not host re-tiled */
CDNA4_API int cdna4_retile_r4_host_rows(int r_type);            /* rows per interleaved group (4 or 8), 0 if r_type is not host re-tiled */

/* ---- GGML_OP_REDUCE (tensor-parallel sum of per-device partials) ------------------------------------------
 * replaces ggml_cuda_op_reduce (ggml-cuda/reduce.cu:125-598) and the NCCL bootstrap (ggml-cuda.cu:265-299).
 * One process per GPU: every rank creates a communicator from the same 128-byte unique id (rank 0 calls
 * cdna4_comm_unique_id and ships the bytes through whatever side channel the host has), then
 * cdna4_all_reduce_sum sums `count` elements in place over RCCL/xGMI; after it every rank holds the full sum
 * (the REDUCE node contract).  dtype is CDNA4_TYPE_F32 / F16 / BF16 (reduce.cu:131-134). */
#define CDNA4_UNIQUE_ID_BYTES 128
typedef struct cdna4_comm cdna4_comm;
CDNA4_API int  cdna4_comm_unique_id(void *id_out /* CDNA4_UNIQUE_ID_BYTES */);
CDNA4_API cdna4_comm *cdna4_comm_init(cdna4_context *ctx, const void *unique_id, int rank, int world_size);
CDNA4_API void cdna4_comm_free(cdna4_comm *comm);
CDNA4_API int  cdna4_all_reduce_sum(cdna4_comm *comm, void *buf, int64_t count, int dtype, void *stream);

/* In-process form of GGML_OP_REDUCE (ADD) for a host that drives several GPUs from ONE process (the reference's -sm graph design:
 * ggml.c:6166-6189 builds the node, reduce.cu:125-598 executes it on one backend that touches all devices' buffers).
 *   bufs[j], j < n (<= 16): device pointer of device j's tensor or NULL; bit j of partial_mask set = that buffer holds a partial sum
 *   (clear = copy-only target, op_params[4] of the node).  After the call EVERY non-NULL buffer holds the element-wise sum of the
 *   partials (f32 accumulate, ascending j).  dtype F32 / F16 / BF16; `count` elements; buffers 16-byte aligned.
 *   dtype Q8_0 (the reference's reduce_type q8_0, reduce.cu:20-43): the buffers hold block_q8_0 rows, `count` = elements (a multiple of 32); every 32-block is
 *   de-quantized, summed in f32 and re-quantized once (d = amax / 127 stored as f16, q = roundf(x / d)).
 * One launch on ctx's device reads / writes the peers' HBM directly (the caller enabled peer access and ordered the peers' streams
 * before / after `stream`, as the shim does with events).  The one-process-per-GPU design uses cdna4_all_reduce_sum instead. */
CDNA4_API int cdna4_reduce_peers(cdna4_context *ctx, void *const *bufs, int n, unsigned partial_mask, int64_t count, int dtype, void *stream);
/* The same reduce restricted to slice `slice` of `n_slices` equal parts of the vector (the last slice takes the ragged tail): for prompt-size messages the host launches
 * slice d on device d's context and stream, so that every GPU reduces its own 1/N of the vector and moves 2 (N-1)/N of the message over its own links instead of one GPU moving
 * 2 (N-1) x the message -- the reference's choice above its small-message threshold (reduce.cu:448-533, one kernel per device on its own stream).  All slices together
 * produce exactly the bits of one cdna4_reduce_peers call.  The caller orders the streams (every partial ready before any slice starts, every slice done before any
 * device continues). */
CDNA4_API int cdna4_reduce_peers_slice(cdna4_context *ctx, void *const *bufs, int n, unsigned partial_mask, int64_t count, int dtype, int slice, int n_slices, void *stream);

/* One-shot all-reduce for the one-process-per-GPU design without a collective library (reference: the P2P one-shot of ggml-cuda/reduce.cu:448-533, k_reduce_add_T -- every
 * GPU sums its peers' partials by loading their memory; there all devices live in one process, here the peers' memory is mapped through HIP IPC).
 * Every rank owns a WINDOW in device memory: two partial slots of `max_bytes` (parity of the call count) + an arrival flag.  Ranks exchange the CDNA4_IPC_HANDLE_BYTES handles
 * through the host's side channel and attach the peers' windows.  cdna4_window_all_reduce_sum (all ranks call it in the same order, in place on buf):
 *   (1) copies the rank's partial into its slot, (2) raises its flag to the call's epoch once every workgroup has stored its share, (3) waits -- BOUNDED -- until every
 *   peer's flag has reached the epoch, (4) sums all partials in rank order (every rank gets bit-identical results) into buf.
 * A peer that never arrives makes the kernel give up after ~1 s: the call then returns CDNA4_E_HIP after a stream synchronize only when `check` is non-zero (tests);
 * production callers run with check = 0 and stay asynchronous.  The epoch is a launch argument: the call cannot be captured into a HIP graph. */
#define CDNA4_IPC_HANDLE_BYTES 64
typedef struct cdna4_window cdna4_window;
CDNA4_API cdna4_window *cdna4_window_create(cdna4_context *ctx, int rank, int world_size, int64_t max_bytes, void *handle_out /* CDNA4_IPC_HANDLE_BYTES */);
CDNA4_API int  cdna4_window_attach(cdna4_window *win, int peer_rank, const void *handle);
CDNA4_API int  cdna4_window_all_reduce_sum(cdna4_window *win, void *buf, int64_t count, int dtype, int check, void *stream);
/* buf is f32, the partials travel as wire_dtype (CDNA4_TYPE_BF16 / _F16: the reference's reduce_type for prompt-size messages, src/llama.cpp:8147,8227-8242);
 * the conversion happens inside the launch, the sum of the rounded partials is accumulated and returned in f32.  wire_dtype == dtype: as above. */
CDNA4_API int  cdna4_window_all_reduce_sum_wire(cdna4_window *win, void *buf, int64_t count, int dtype, int wire_dtype, int check, void *stream);
CDNA4_API void cdna4_window_free(cdna4_window *win);

/* ---- measurement helper ----------------------------------------------------------------------------------
 * Times `iters` back-to-back launches of cdna4_mul_mat with HIP events on `stream` and returns the average
 * per-launch milliseconds (used by bench.py for roofline.achieved; the timed stream is the launch stream). */
CDNA4_API int cdna4_time_mul_mat(cdna4_context *ctx, long Nx, long Ny, long ne00, int typeA, const void *const *A_rot, int n_rot,
                                 long strideA, const float *B, long strideB, float *C, long stride_C,
                                 int warmup, int iters, void *stream, float *avg_ms);

#ifdef __cplusplus
}
#endif
#endif /* GGML_HIP_CDNA4_H */
