// api_internal.h -- what the translation units of libggml-hip-cdna4.so share: the context, error reporting, and the per-type
// launchers.  The library is built from several TUs (one per weight type and kernel family, ik_llama.cpp_amd/build.py) so that a
// kernel edit recompiles seconds, not minutes, of template instantiations; nothing here is exported.
#pragma once
#include "../../include/ggml_hip_cdna4.h"
#include "cdna4_common.cuh"

#include <cstdio>
#include <cstdlib>
#include <cstdarg>
#include <cstring>
#include <mutex>
#include <vector>

int cdna4_set_err(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));     // thread-local cdna4_last_error(); returns `code`
#define set_err cdna4_set_err
#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return cdna4_set_err(CDNA4_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

// One context per device; ONE stream at a time per context (the workspace and the event pair are not per-stream: the ggml
// scheduler drives a backend from one host thread on one stream, SURVEY 8b "Threading").
constexpr int CDNA4_KS_MAX_TILES = 16384;           // capacity of cdna4_context::ks_counters
struct cdna4_context {
    int device = 0;
    int num_cu = 256;
    size_t max_lds = 64 * 1024;
    const struct cdna4_fusion *fx = nullptr;            // set only for the duration of a cdna4_*_fused call (read where the decode launch arguments are filled)
    void *rope_table = nullptr;                         // per-graph (cos, sin) cache of the rope ops (ops.hip)
    void *fa_counters = nullptr; size_t fa_counters_bytes = 0;     // arrival counters of the split-KV decode attention (zero between launches)
    unsigned *ks_counters = nullptr;                   // arrival counters of the split-K prompt GEMM, one per output tile (zeroed once, re-armed by the kernel itself)
    struct { const void *pos = nullptr, *ff = nullptr; long n_tok = 0; int n_dims = 0; float theta_scale = 0, freq_scale = 0, ext_factor = 0, attn_factor = 0, corr0 = 0, corr1 = 0; } rope_key;
    long ws_epoch = 0;                                  // incremented whenever the workspace is re-allocated
    void *ws = nullptr; size_t ws_bytes = 0;          // scratch (f16 activations of the prefill path, MoE grouping tables, q8 images)
    uint16_t *grid = nullptr;                          // packed IQ2_S (1024) + IQ3_S (512) codebooks
    uint8_t *iq_tables = nullptr;                      // expanded codebooks + sign tables for the decode kernels
    int prefill_mode = CDNA4_PREFILL_MFMA_F16;
    bool deterministic = false;                         // cdna4_set_deterministic: split-K prompt launches add their slices in a fixed order (no atomics)
    // in-launch hand-offs (split-K prompt GEMM slabs, split-KV decode attention partials) travel as write-through (sc1) stores + one relaxed agent-scope ticket + sc1 loads, without
    // release / acquire fences (MI355X guide, Guideline 16 / "in-launch split-K reduction").  cdna4_handoff_selftest checks both against their unsplit forms on THIS device at
    // context creation; on a mismatch the context falls back to the fenced forms.  -1 not tested yet, 0 fence-free forms validated, 1 fenced by request (CDNA4_SPLITK_FENCE=1),
    // 2 fenced after a failed self-test.  selftest_unsplit: the self-test's reference runs (no K split over grid.z, per-head attention).
    int handoff = -1; bool selftest_unsplit = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // _R4 tensors handed to a mat-mul as they are (the shim converts at upload instead) are un-interleaved once and cached
    struct Shadow { const void *src; int type; long nrows, K, stride; void *base; };
    std::vector<Shadow> shadows; std::mutex shadow_mu;
};

// which instantiation served the calling thread's last prompt-GEMM launch (tests pin the geometry a shape takes: cdna4_last_launch_info)
void cdna4_note_launch(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
// decode launches (hot: ~130 per token when a graph is walked eagerly): the fields are only stored, cdna4_last_launch_info() formats them on demand
void cdna4_note_gemv(const char *kernel, int type, int ncols, int upgate, int yiters, int nr, int lpr, int fx, long wgs, unsigned grid_y, int waves);
int cdna4_gemm_form(void);      // cdna4_set_gemm_form: 1 default, 0 no workgroup-shared weight tiles, 2 shared weight tiles wherever the kernel can run

// opt a kernel in to > 64 KiB of dynamic LDS.  Function attributes are per DEVICE: tracked per (current device, function), thread-safe;
// a failed attempt is retried by the next call.
int cdna4_opt_in_lds(const void *func);

struct GemvArgs; struct GemmArgs; struct UpGateEpilogue;

// strided tensor descriptor handed to the non-mat-mul kernels by value (ggml convention: ne[] elements, nb[] bytes)
struct TD { char *data; long ne[4]; long nb[4]; };
static inline TD td_of(const cdna4_tensor *t) { TD d; d.data = (char *)t->data; for (int i = 0; i < 4; ++i) { d.ne[i] = t->ne[i]; d.nb[i] = t->nb[i]; } return d; }

// per-type launchers (each defined in its own TU: gemv_inst.hip / gemm_inst.hip compiled with -DINST_TYPE=<ggml_type>)
#define CDNA4_FOR_BASE_TYPES(X) X(12) X(13) X(14) X(20) X(21) X(22) X(2) X(8) X(23) X(6) X(3) X(7) X(133) X(139) X(140) X(144) X(152) X(10) X(11) X(137) X(138) X(16) X(17) X(18) X(145) X(156) X(146) X(141) X(157) X(39) X(19) X(29)
// decode-only types: their prompt batches are de-quantized to f16 (convert.hip) and run through the f16 instance of the MFMA GEMM (type 1)
#define CDNA4_FOR_GEMV_ONLY_TYPES(X) X(153) X(154) X(155) X(158)
#define CDNA4_DECL_GEMV(T) \
    int cdna4_gemv_launch_##T##_plain(const cdna4_context *ctx, int vdt, const GemvArgs &a, int ncols, unsigned grid_y, hipStream_t st); \
    int cdna4_gemv_launch_##T##_upgate(const cdna4_context *ctx, int vdt, const GemvArgs &a, int ncols, unsigned grid_y, hipStream_t st);
CDNA4_FOR_BASE_TYPES(CDNA4_DECL_GEMV)
CDNA4_FOR_GEMV_ONLY_TYPES(CDNA4_DECL_GEMV)
#undef CDNA4_DECL_GEMV
// mode: 0 dense / multi (tile shape chosen inside), 1 grouped (MUL_MAT_ID, nt given), upgate from a.A2
#define CDNA4_DECL_GEMM(T) int cdna4_gemm_launch_##T(int num_cu, const GemmArgs &a, int grouped_nt, hipStream_t st); int cdna4_gemm_preload_##T(void); \
                           int cdna4_dequant_slab_launch_##T(const GemmArgs &a, void *w16, int *pairing, hipStream_t st);
CDNA4_FOR_BASE_TYPES(CDNA4_DECL_GEMM)
CDNA4_DECL_GEMM(1)
#undef CDNA4_DECL_GEMM
int cdna4_gemm_ppf_launch(int num_cu, const GemmArgs &a, hipStream_t st);      // gemm_ppf.hip: f16 weight image x f16 activation image (large batches)
int cdna4_gemv_dual_launch(const cdna4_context *ctx, int type_a, const GemvArgs &a, const GemvArgs &b, hipStream_t st);   // -1: not applicable
// gemv_attn.hip: decode attention + attn_output mat-vec + residual in one launch (-1: not served); ops.hip: is this attention the per-head decode kernel's case?
bool cdna4_fa_is_plain_decode(const cdna4_context *ctx, const cdna4_tensor *q, const cdna4_tensor *k, const cdna4_tensor *v, const cdna4_tensor *mask, const cdna4_tensor *dst);
int cdna4_gemv_attn_launch(const cdna4_context *ctx, int type, const cdna4_tensor *q, const cdna4_tensor *k, const cdna4_tensor *v, const cdna4_tensor *mask, const cdna4_tensor *attn,
                           float scale, float max_bias, float softcap, const GemvArgs &a, unsigned *sync, hipStream_t st);

// BitNet (gemv_bitnet.hip)
int cdna4_launch_quantize_q8_k64(const void *B, long strideB, long nrows, long K, void *dst, long dst_row_bytes, hipStream_t st);
int cdna4_launch_gemv_bitnet(const cdna4_context *ctx, int type, const void *A, long strideA, long M, long K, const void *Xq, long xq_stride, int ncols, float *C, long stride_C, hipStream_t st);

// utility kernels (convert.hip)
int cdna4_launch_dequant(const cdna4_context *ctx, int type, const void *A, long strideA, long nrows, long K, void *dst, int dst_type, long dst_stride, hipStream_t st, bool matmul_value = false);
int cdna4_launch_quantize(int vdt, const void *B, long strideB, long nrows, long K, void *dst, long dst_row_bytes, hipStream_t st);
int cdna4_launch_repack(bool to_r4, int base, const void *src, void *dst, long nrows, long K, long stride, hipStream_t st);
int cdna4_launch_f32_to_f16_slab(const void *B, long strideB, long K, long nrows, void *dst, long xrows, float *xscale, hipStream_t st);
int cdna4_launch_norm_f16_slab(const void *B, const void *add_b, void *add_dst, long strideB, const float *w, float eps, long K, long nrows, void *dst, long xrows, float *xscale, hipStream_t st);   // ops.hip
int cdna4_launch_moe_sort(const int32_t *ids, long ids_nb1, int n_tokens, int n_used, int n_expert, int BN, int max_tiles, int *pairs_sorted, int *tiles,
                          float *C, long nb1, long nb2, int M, hipStream_t st);
int cdna4_launch_moe_gather_f16(const void *B, int n_b, long nb11, long nb12, int n_used, const int *pairs_sorted, long rows_pad, long pairs, long K, void *X, float *xscale, hipStream_t st);
int cdna4_launch_iq_tables_init(const uint16_t *packed, uint8_t *out);
int cdna4_launch_get_rows(const cdna4_context *ctx, const cdna4_tensor *src, const cdna4_tensor *ids, const cdna4_tensor *dst, hipStream_t st);
int cdna4_launch_reduce_peers(int num_cu, void *const *bufs, int n, unsigned partial_mask, long count, int dtype, int slice, int n_slices, hipStream_t st);

// gemv_mfma.hip: 2..16 pre-quantized activation columns on the int8 matrix cores (-1: type / shape not served)
int cdna4_gemv_mfma_launch(const cdna4_context *ctx, int type, const GemvArgs &a, int ncols, hipStream_t st);
// workspace growth outside stream capture (cdna4_api.hip); `epoch` counts re-allocations (captured graphs hold the old pointer)
int cdna4_ensure_ws(cdna4_context *ctx, size_t bytes, hipStream_t st);
// flash_attn.hip: prompt-batch attention on the matrix cores
size_t cdna4_flash_attn_mfma_workspace(const cdna4_tensor *k);
int cdna4_launch_flash_attn_mfma(const cdna4_tensor *q, const cdna4_tensor *k, const cdna4_tensor *v, const cdna4_tensor *mask, const cdna4_tensor *dst, void *vt,
                                 float scale, float max_bias, float softcap, hipStream_t st);
