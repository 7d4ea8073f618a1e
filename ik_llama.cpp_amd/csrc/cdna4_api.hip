// cdna4_api.hip -- C ABI (include/ggml_hip_cdna4.h) over the gfx950 kernels.  Host-side dispatch only: the kernels are instantiated in
// the per-type translation units (gemv_inst.hip, gemm_inst.hip, gemv_dual.hip, convert.hip, ops.hip; see api_internal.h).
#include "api_internal.h"
#include "gemv.cuh"          // GemvArgs, gemv_lds_bytes, table sizes (templates are NOT instantiated here)
#include "gemm_mfma.cuh"     // GemmArgs, gemm_mfma_npad / gemm_mfma_supported
#include "iq_grids_packed.inc"   // k_iq2s_grid_packed[1024], k_iq3s_grid_packed[512] (host arrays)

#include <algorithm>
#include <set>
#include <utility>

#define CDNA4_VERSION "ggml-hip-cdna4 0.2 (gfx950)"

static thread_local char g_err[512] = "";
int cdna4_set_err(int code, const char *fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
    return code;
}

static thread_local char g_launch_note[256] = "";
static thread_local struct { const char *kernel; int type, ncols, upgate, yiters, nr, lpr, fx, waves; long wgs; unsigned gy; } g_gemv_note = {nullptr, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
void cdna4_note_launch(const char *fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_launch_note, sizeof(g_launch_note), fmt, ap); va_end(ap); g_gemv_note.kernel = nullptr;
}
void cdna4_note_gemv(const char *kernel, int type, int ncols, int upgate, int yiters, int nr, int lpr, int fx, long wgs, unsigned grid_y, int waves) {
    g_gemv_note = {kernel, type, ncols, upgate, yiters, nr, lpr, fx, waves, wgs, grid_y};
}

// > 64 KiB of dynamic LDS needs an opt-in per (device, kernel): function attributes are per device, and several devices / host threads
// share this process in the reference's -sm graph design (one backend per device, one host thread each).
int cdna4_opt_in_lds(const void *func) {
    static std::mutex mu; static std::set<std::pair<int, const void *>> done;
    int dev = 0; HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    if (done.count({dev, func})) return CDNA4_OK;
    HIP_TRY(hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    done.insert({dev, func});
    return CDNA4_OK;
}
#ifdef GEMV_EXP_TIMELINE      // experiment builds only (scripts/gemv_timeline.py)
long long *g_gemv_timeline = nullptr; int g_gemv_timeline_wgs = 0;
extern "C" __attribute__((visibility("default"))) int cdna4_exp_set_timeline(long long *buf) { g_gemv_timeline = buf; return 0; }
extern "C" __attribute__((visibility("default"))) int cdna4_exp_timeline_wgs(void) { return g_gemv_timeline_wgs; }
#endif

// All entry points below get C linkage and default visibility from their declarations in ggml_hip_cdna4.h.

const char *cdna4_last_error(void) { return g_err; }
const char *cdna4_version(void) { return CDNA4_VERSION; }
const char *cdna4_last_launch_info(void) {
    if (g_gemv_note.kernel) {      // the newest launch was a decode mat-vec: format its stored fields now
        const auto &n = g_gemv_note;
        snprintf(g_launch_note, sizeof(g_launch_note), "%s type=%d ncols=%d upgate=%d yiters=%d nr=%d lpr=%d fx=%d waves=%d grid=%ldx%ux1", n.kernel, n.type, n.ncols, n.upgate, n.yiters, n.nr, n.lpr, n.fx, n.waves, n.wgs, n.gy);
        g_gemv_note.kernel = nullptr;
    }
    return g_launch_note;
}
static int g_gemm_form = getenv("CDNA4_GEMM_WLDS") ? atoi(getenv("CDNA4_GEMM_WLDS")) : 1;
int cdna4_gemm_form(void) { return __atomic_load_n(&g_gemm_form, __ATOMIC_RELAXED); }
int cdna4_set_gemm_form(int form) {
    if (form < 0 || form > 4) return set_err(CDNA4_E_INVALID, "gemm form %d", form);
    __atomic_store_n(&g_gemm_form, form, __ATOMIC_RELAXED); return CDNA4_OK;
}

int cdna4_get_device_count(void) {
    int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; } return n;
}
int cdna4_get_device_description(int device, char *buf, size_t buf_size) {
    hipDeviceProp_t p; HIP_TRY(hipGetDeviceProperties(&p, device));
    snprintf(buf, buf_size, "%s (%s)", p.name, p.gcnArchName); return CDNA4_OK;
}
int cdna4_get_device_memory(int device, size_t *free_bytes, size_t *total_bytes) {
    int prev = 0; HIP_TRY(hipGetDevice(&prev)); HIP_TRY(hipSetDevice(device));
    hipError_t e = hipMemGetInfo(free_bytes, total_bytes); (void)hipSetDevice(prev);
    if (e != hipSuccess) return set_err(CDNA4_E_HIP, "hipMemGetInfo: %s", hipGetErrorString(e));
    return CDNA4_OK;
}

cdna4_context *cdna4_init(int device) {
    int n = cdna4_get_device_count();
    if (device < 0 || device >= n) { set_err(CDNA4_E_INVALID, "invalid device %d (have %d)", device, n); return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { set_err(CDNA4_E_HIP, "hipSetDevice(%d) failed", device); return nullptr; }
    hipDeviceProp_t p; if (hipGetDeviceProperties(&p, device) != hipSuccess) { set_err(CDNA4_E_HIP, "hipGetDeviceProperties failed"); return nullptr; }
    cdna4_context *ctx = new cdna4_context();
    ctx->device = device; ctx->num_cu = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
    ctx->max_lds = p.maxSharedMemoryPerMultiProcessor ? p.maxSharedMemoryPerMultiProcessor : 64 * 1024;
    // arrival counters of the split-KV decode attention (ops.hip): fixed capacity, zeroed once here, re-armed by the kernel itself
    ctx->fa_counters_bytes = 64 * 1024;
    if (hipMalloc(&ctx->fa_counters, ctx->fa_counters_bytes) != hipSuccess || hipMemset(ctx->fa_counters, 0, ctx->fa_counters_bytes) != hipSuccess) {
        (void)hipGetLastError(); if (ctx->fa_counters) (void)hipFree(ctx->fa_counters); delete ctx; set_err(CDNA4_E_HIP, "hipMalloc(attention counters) failed"); return nullptr; }
    if (hipMalloc((void **)&ctx->ks_counters, CDNA4_KS_MAX_TILES * sizeof(unsigned)) != hipSuccess || hipMemset(ctx->ks_counters, 0, CDNA4_KS_MAX_TILES * sizeof(unsigned)) != hipSuccess) {
        (void)hipGetLastError(); (void)hipFree(ctx->fa_counters); if (ctx->ks_counters) (void)hipFree(ctx->ks_counters); delete ctx; set_err(CDNA4_E_HIP, "hipMalloc(split-K counters) failed"); return nullptr; }
    if (hipMalloc((void **)&ctx->grid, GRID_U16_TOTAL * sizeof(uint16_t)) != hipSuccess) { delete ctx; set_err(CDNA4_E_HIP, "hipMalloc(grid) failed"); return nullptr; }
    (void)hipMemcpy(ctx->grid, k_iq2s_grid_packed, 1024 * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(ctx->grid + GRID_IQ3S, k_iq3s_grid_packed, 512 * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(ctx->grid + GRID_IQ2XXS, k_iq2xxs_grid_packed, 256 * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(ctx->grid + GRID_IQ2XS, k_iq2xs_grid_packed, 512 * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(ctx->grid + GRID_IQ3XXS, k_iq3xxs_grid_packed, 256 * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(ctx->grid + GRID_IQ1S, k_iq1s_grid_packed, 2048 * 2, hipMemcpyHostToDevice);
    if (hipMalloc((void **)&ctx->iq_tables, IQ_TABLES_BYTES) != hipSuccess) { (void)hipFree(ctx->grid); delete ctx; set_err(CDNA4_E_HIP, "hipMalloc(iq tables) failed"); return nullptr; }
    (void)cdna4_launch_iq_tables_init(ctx->grid, ctx->iq_tables);
    (void)hipDeviceSynchronize();
    (void)hipEventCreate(&ctx->ev0); (void)hipEventCreate(&ctx->ev1);
    (void)cdna4_handoff_selftest(ctx, nullptr);      // fence-free in-launch hand-offs validated on THIS device, or the context uses the fenced forms (never fatal)
    return ctx;
}
void cdna4_free(cdna4_context *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->ws) (void)hipFree(ctx->ws);
    for (auto &sh : ctx->shadows) (void)hipFree(sh.base);
    if (ctx->grid) (void)hipFree(ctx->grid);
    if (ctx->iq_tables) (void)hipFree(ctx->iq_tables);
    if (ctx->rope_table) (void)hipFree(ctx->rope_table);
    if (ctx->fa_counters) (void)hipFree(ctx->fa_counters);
    if (ctx->ks_counters) (void)hipFree(ctx->ks_counters);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    delete ctx;
}

int cdna4_reserve_workspace(cdna4_context *ctx, size_t bytes) {
    if (!ctx) return set_err(CDNA4_E_INVALID, "null context");
    if (bytes <= ctx->ws_bytes) return CDNA4_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipDeviceSynchronize());
    if (ctx->ws) { HIP_TRY(hipFree(ctx->ws)); ctx->ws = nullptr; ctx->ws_bytes = 0; }
    bytes = (bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
    HIP_TRY(hipMalloc(&ctx->ws, bytes)); ctx->ws_bytes = bytes; ++ctx->ws_epoch;
    return CDNA4_OK;
}
long cdna4_workspace_epoch(cdna4_context *ctx) { return ctx ? ctx->ws_epoch : -1; }
int cdna4_ensure_ws(cdna4_context *ctx, size_t bytes, hipStream_t st);
static int ensure_ws(cdna4_context *ctx, size_t bytes, hipStream_t st) { return cdna4_ensure_ws(ctx, bytes, st); }
int cdna4_ensure_ws(cdna4_context *ctx, size_t bytes, hipStream_t st) {
    if (bytes <= ctx->ws_bytes) return CDNA4_OK;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (st && hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
        return set_err(CDNA4_E_NOMEM, "workspace of %zu bytes needed during stream capture; call cdna4_reserve_workspace first", bytes);
    return cdna4_reserve_workspace(ctx, bytes);
}

// ---- type traits ------------------------------------------------------------------------------------
static bool weight_type_ok(int t) {
    if (type_is_pretiled(t)) { t -= T_PRETILED; if (!type_is_r4(t)) return false; }
    switch (t) { case T_Q4_K: case T_Q5_K: case T_Q6_K: case T_IQ4_NL: case T_IQ2_S: case T_IQ3_S: case T_Q4_0: case T_Q8_0: case T_IQ4_XS: case T_Q5_0: case T_IQ2_XXS: case T_IQ2_XS: case T_IQ3_XXS: case T_Q4_1: case T_Q5_1: case T_Q6_0: case T_Q2_K: case T_Q3_K:
                 case T_IQ2_K: case T_IQ3_K: case T_IQ4_K: case T_IQ5_K: case T_IQ4_KS: case T_IQ5_KS: case T_IQ2_KS: case T_IQ3_KS: case T_IQ4_KSS: case T_IQ2_KL: case T_IQ6_K: case T_IQ1_S: case T_IQ1_M: case T_MXFP4: case T_IQ1_BN: case T_IQ2_BN: case T_IQ2_KT: case T_IQ3_KT: case T_IQ4_KT: case T_IQ1_KT:
                 case T_Q4_K_R4: case T_Q5_K_R4: case T_Q6_K_R4: case T_IQ4_NL_R4: case T_IQ2_S_R4: case T_IQ3_S_R4: return true; }
    return false;
}
int    cdna4_type_supported(int type) { return weight_type_ok(type) ? 1 : 0; }
int    cdna4_blck_size(int type) { return type == T_Q8_K64 ? 64 : (weight_type_ok(type) || type == T_Q8_K || type == T_Q8_K32 || type == T_Q8_2_X4) ? type_block_elems(type) : 0; }
size_t cdna4_type_size(int type) { return (size_t)type_block_bytes(type); }
size_t cdna4_row_size(int type, int64_t ne00) { if (type == T_Q8_K64) return 32 + (size_t)ne00;      /* {float d[4]; float d * sum(q) [4]; int8 q[ne00]} */
                                                  const int bs = cdna4_blck_size(type); return bs ? (size_t)type_row_meta(type) + (size_t)type_block_bytes(type) * (size_t)(ne00 / bs) : 0; }
int    cdna4_vec_dot_type(int type) { return weight_type_ok(type) ? type_vec_dot(type) : -1; }

int cdna4_set_prefill_mode(cdna4_context *ctx, int mode) {
    if (!ctx || (mode != CDNA4_PREFILL_MFMA_F16 && mode != CDNA4_PREFILL_INT8_DOT && mode != CDNA4_PREFILL_MFMA_F16_EXACT)) return set_err(CDNA4_E_INVALID, "bad prefill mode");
    ctx->prefill_mode = mode; return CDNA4_OK;
}
int cdna4_set_deterministic(cdna4_context *ctx, int on) {
    if (!ctx) return set_err(CDNA4_E_INVALID, "null context");
    ctx->deterministic = on != 0; return CDNA4_OK;
}

// ---- dequantize -----------------------------------------------------------------------------------------
int cdna4_dequantize_rows(cdna4_context *ctx, int type, const void *A, int64_t strideA, int64_t nrows, int64_t ne00,
                          void *dst, int dst_type, int64_t dst_stride, void *stream) {
    if (!ctx || !A || !dst) return set_err(CDNA4_E_INVALID, "null argument");
    if (!weight_type_ok(type)) return set_err(CDNA4_E_UNSUPPORTED, "dequantize: type %d unsupported", type);
    if (dst_type != T_F32 && dst_type != T_F16) return set_err(CDNA4_E_INVALID, "dst_type must be F32 or F16");
    if (ne00 % type_block_elems(type)) return set_err(CDNA4_E_INVALID, "ne00 %% block size != 0");
    if (type_is_r4(type) && (nrows % 4)) return set_err(CDNA4_E_INVALID, "_R4 tensors need nrows %% 4 == 0");
    if (nrows == 0 || ne00 == 0) return CDNA4_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    return cdna4_launch_dequant(ctx, type_is_pretiled(type) ? type_base(type) : type, A, strideA, nrows, ne00, dst, dst_type, dst_stride, (hipStream_t)stream);
}

// ---- activation quantizers -----------------------------------------------------------------------------
int cdna4_quantize_rows(cdna4_context *ctx, int vdt, const float *B, int64_t strideB, int64_t nrows, int64_t ne00, void *dst, void *stream) {
    if (!ctx || !B || !dst) return set_err(CDNA4_E_INVALID, "null argument");
    if (vdt != T_Q8_2_X4 && vdt != T_Q8_K && vdt != T_Q8_K32 && vdt != T_Q8_K64) return set_err(CDNA4_E_UNSUPPORTED, "quantize: type %d unsupported", vdt);
    if (ne00 % (vdt == T_Q8_2_X4 ? 32 : vdt == T_Q8_K64 ? 64 : 256)) return set_err(CDNA4_E_INVALID, "ne00 %% block size != 0");
    if (nrows == 0 || ne00 == 0) return CDNA4_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    if (vdt == T_Q8_K64) return cdna4_launch_quantize_q8_k64(B, strideB, nrows, ne00, dst, 32 + (long)ne00, (hipStream_t)stream);
    return cdna4_launch_quantize(vdt, B, strideB, nrows, ne00, dst, (long)cdna4_row_size(vdt, ne00), (hipStream_t)stream);
}

// `type` is the BASE type of the (possibly un-interleaved) weights, `vdt` the activation quantization to reproduce
template <bool UPGATE>
static int launch_gemv(cdna4_context *ctx, int type, int vdt, GemvArgs a, int ncols, unsigned grid_y, hipStream_t st) {
    a.tables = type_has_tables(type) ? ctx->iq_tables + iq_tables_offset(type) : nullptr;
#define GV(T) case T: return UPGATE ? cdna4_gemv_launch_##T##_upgate(ctx, vdt, a, ncols, grid_y, st) : cdna4_gemv_launch_##T##_plain(ctx, vdt, a, ncols, grid_y, st);
    switch (type) { CDNA4_FOR_BASE_TYPES(GV) CDNA4_FOR_GEMV_ONLY_TYPES(GV) }
#undef GV
    return set_err(CDNA4_E_UNSUPPORTED, "gemv: weight type %d not implemented", type);
}

// ---- _R4 tensors: repack / un-repack kernels and the shadow cache ---------------------------------------------------
static int repack_common(cdna4_context *ctx, int base_type, const void *A, int64_t nrows, int64_t ne00, void *dst, void *stream, bool to_r4) {
    if (!ctx || !A || !dst) return set_err(CDNA4_E_INVALID, "null argument");
    if (!weight_type_ok(base_type) || type_is_r4(base_type)) return set_err(CDNA4_E_UNSUPPORTED, "repack: base type %d unsupported", base_type);
    if (nrows % 4 || ne00 % type_block_elems(base_type)) return set_err(CDNA4_E_INVALID, "repack needs nrows %% 4 == 0 and whole blocks");
    if (A == dst) return set_err(CDNA4_E_INVALID, "repack is out of place");
    if (nrows == 0 || ne00 == 0) return CDNA4_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    const long stride = (long)cdna4_row_size(base_type, ne00);
    return cdna4_launch_repack(to_r4, base_type, A, dst, nrows, ne00, stride, (hipStream_t)stream);
}
int cdna4_repack_r4(cdna4_context *ctx, int base_type, const void *A, int64_t nrows, int64_t ne00, void *dst, void *stream) { return repack_common(ctx, base_type, A, nrows, ne00, dst, stream, true); }
int cdna4_unrepack_r4(cdna4_context *ctx, int base_type, const void *A, int64_t nrows, int64_t ne00, void *dst, void *stream) { return repack_common(ctx, base_type, A, nrows, ne00, dst, stream, false); }

int cdna4_invalidate_weight_cache(cdna4_context *ctx, const void *A) {
    if (!ctx) return set_err(CDNA4_E_INVALID, "null context");
    std::lock_guard<std::mutex> lock(ctx->shadow_mu);
    HIP_TRY(hipSetDevice(ctx->device)); HIP_TRY(hipDeviceSynchronize());
    for (size_t i = 0; i < ctx->shadows.size();) {
        if (!A || ctx->shadows[i].src == A) { (void)hipFree(ctx->shadows[i].base); ctx->shadows.erase(ctx->shadows.begin() + i); } else ++i;
    }
    return CDNA4_OK;
}
// base-layout view of an _R4 tensor (converted once, then served from the cache)
static int shadow_of(cdna4_context *ctx, int r4_type, const void *A, long nrows, long K, long stride, hipStream_t st, const void **out) {
    std::lock_guard<std::mutex> lock(ctx->shadow_mu);
    for (auto &sh : ctx->shadows) if (sh.src == A && sh.type == r4_type && sh.nrows == nrows && sh.K == K && sh.stride == stride) { *out = sh.base; return CDNA4_OK; }
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (st && hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
        return set_err(CDNA4_E_NOMEM, "_R4 tensor first used during stream capture: run one eager pass first");
    const int base = type_base(r4_type);
    if (stride != (long)cdna4_row_size(base, K)) return set_err(CDNA4_E_UNSUPPORTED, "_R4 tensors must have contiguous rows");
    void *buf = nullptr; HIP_TRY(hipMalloc(&buf, (size_t)nrows * stride));
    int rc = cdna4_launch_repack(false, base, A, buf, nrows, K, stride, st); if (rc) { (void)hipFree(buf); return rc; }
    ctx->shadows.push_back({A, r4_type, nrows, K, stride, buf}); *out = buf;
    return CDNA4_OK;
}

static int check_mm_args(cdna4_context *ctx, long Nx, long Ny, long ne00, int typeA, const void *A, long strideA, int typeB, const void *B, float *C) {
    if (!ctx) return set_err(CDNA4_E_INVALID, "null context");
    if (!weight_type_ok(typeA)) return set_err(CDNA4_E_UNSUPPORTED, "mul_mat: weight type %d unsupported", typeA);
    if (Nx < 0 || Ny < 0 || ne00 < 0) return set_err(CDNA4_E_INVALID, "negative dimension");
    if (ne00 % type_block_elems(typeA)) return set_err(CDNA4_E_INVALID, "ne00=%ld not a multiple of the block size", ne00);
    if (ne00 % 64) return set_err(CDNA4_E_UNSUPPORTED, "ne00=%ld not a multiple of 64", ne00);
    if (type_is_r4(typeA) && (Nx % 4)) return set_err(CDNA4_E_INVALID, "_R4 weights need Nx %% 4 == 0");
    if (typeB != T_F32 && typeB != type_vec_dot(typeA)) return set_err(CDNA4_E_UNSUPPORTED, "mul_mat: activation type %d unsupported for weight type %d", typeB, typeA);
    if (typeB == T_Q8_2_X4 && (ne00 % 128)) return set_err(CDNA4_E_UNSUPPORTED, "pre-quantized Q8_2_X4 rows need ne00 %% 128 == 0");
    if ((size_t)strideA < cdna4_row_size(typeA, ne00)) return set_err(CDNA4_E_INVALID, "strideA smaller than a row");
    if (strideA > 0x7fffffffL) return set_err(CDNA4_E_UNSUPPORTED, "row stride above 2 GiB");
    if (Nx && Ny && ne00 && (!A || !B || !C)) return set_err(CDNA4_E_INVALID, "null pointer");
    return CDNA4_OK;
}

static int mul_mat_gemv(cdna4_context *ctx, long Nx, long Ny, long K, int typeA, const void *A, const void *A2, long strideA,
                        int typeB, const void *B, long strideB, float *C, long stride_C, int unary_op, hipStream_t st, const UpGateEpilogue *epi = nullptr,
                        void *q8_out = nullptr) {
    const int base = type_base(typeA), vdt = type_vec_dot(typeA);
    {   const size_t one = vdt == T_Q8_2_X4 ? gemv_lds_bytes<T_Q8_2_X4>(1, (int)K, base) : vdt == T_Q8_K32 ? gemv_lds_bytes<T_Q8_K32>(1, (int)K, base) : gemv_lds_bytes<T_Q8_K>(1, (int)K, base);
        if (one > 150 * 1024) return set_err(CDNA4_E_UNSUPPORTED, "gemv: ne00=%ld too long for the LDS activation image", K); }
    GemvArgs a; memset(&a, 0, sizeof(a));
    a.strideA = strideA; a.strideB = strideB; a.stride_C = stride_C; a.M = (int)Nx; a.K = (int)K; a.unary_op = unary_op; a.src_f32 = typeB == T_F32;
    if (epi) a.epi = *epi;
    a.q8_out = (uint8_t *)q8_out;
    if (ctx->fx) { a.norm_w = ctx->fx->norm_w; a.norm_eps = ctx->fx->norm_eps; a.R = ctx->fx->residual; if (ctx->fx->qkv) return set_err(CDNA4_E_UNSUPPORTED, "q,k,v epilogue: the matrices must form one launch"); }
    // 2..8 columns of a K-quant: the int8 matrix-core kernel on activations quantized ONCE (gemv_mfma.hip); same arithmetic as the v_dot4 kernels
    static const bool mfma_cols = !(getenv("CDNA4_GEMV_MFMA") && atoi(getenv("CDNA4_GEMV_MFMA")) == 0);
    // (measured, scripts/mb_cols.py: its time is flat in the column count -- 17-19 us on 14336 x 4096 Q4_K -- so it takes over where the v_dot4 kernels
    //  cross that: 5+ columns; Q6_K, one MFMA per 16 weights, only on long rows where the v_dot4 path needs two launches)
    static const bool q6_all = getenv("CDNA4_GEMV_MFMA_Q6_ALL") != nullptr;          // (tests: Q6_K on every shape)
    static const int mfma_min_cols = getenv("CDNA4_GEMV_MFMA_MIN_COLS") ? atoi(getenv("CDNA4_GEMV_MFMA_MIN_COLS")) : 5;
    if (mfma_cols && Ny >= (K > 8192 ? std::min(mfma_min_cols, 4) : mfma_min_cols) && Ny <= 8 && vdt == T_Q8_2_X4 && (base == T_Q4_K || base == T_Q5_K || (base == T_Q6_K && (K > 8192 || q6_all))) && K % 256 == 0 && !q8_out &&
        (typeB == T_F32 || typeB == T_Q8_2_X4)) {
        GemvArgs m = a; m.A[0] = (const uint8_t *)A; m.A2 = (const uint8_t *)A2; m.C[0] = C; m.nmat = 1; m.mend[0] = (int)Nx; m.src_f32 = 0;
        if (typeB == T_F32) {
            const size_t row_bytes = (size_t)(K / 128) * 144;
            int rc = ensure_ws(ctx, row_bytes * (size_t)Ny, st); if (rc) return rc;
            rc = cdna4_launch_quantize(T_Q8_2_X4, B, strideB, Ny, K, ctx->ws, (long)row_bytes, st); if (rc) return rc;
            m.B = (const uint8_t *)ctx->ws; m.strideB = (long)row_bytes;
        } else m.B = (const uint8_t *)B;
        const int rc = cdna4_gemv_mfma_launch(ctx, base, m, (int)Ny, st);
        if (rc != -1) return rc;
    }
    for (long c0 = 0; c0 < Ny;) {
        int n = 1;
        for (int t = 4; t >= 1; --t) {
            const size_t lds = vdt == T_Q8_2_X4 ? gemv_lds_bytes<T_Q8_2_X4>(t, (int)K, base) : vdt == T_Q8_K32 ? gemv_lds_bytes<T_Q8_K32>(t, (int)K, base) : gemv_lds_bytes<T_Q8_K>(t, (int)K, base);
            if (lds <= 96 * 1024 && t <= Ny - c0) { n = t; break; }
        }
        a.A[0] = (const uint8_t *)A; a.A2 = (const uint8_t *)A2; a.B = (const uint8_t *)B + c0 * strideB; a.C[0] = C + c0 * stride_C; a.nmat = 1; a.mend[0] = (int)Nx;
        const int rc = A2 ? launch_gemv<true>(ctx, base, vdt, a, n, 1, st) : launch_gemv<false>(ctx, base, vdt, a, n, 1, st);
        if (rc) return rc;
        c0 += n;
    }
    return CDNA4_OK;
}

// ---- prefill (MFMA) dispatch ---------------------------------------------------------------------------------
// type switch over the per-type TUs; IQ3_S's packed codebook sits behind IQ2_S's in ctx->grid
static int gemm_dispatch(const cdna4_context *ctx, int type, GemmArgs &g, int grouped_nt, hipStream_t st) {
    g.grid = ctx->grid + grid_offset_of(type);
#define GM(T) case T: return cdna4_gemm_launch_##T(ctx->num_cu, g, grouped_nt, st);
    switch (type) { CDNA4_FOR_BASE_TYPES(GM) GM(1) }
#undef GM
    return -1;
}
// Load the code objects a model with weights of `type` will launch for prompt batches NOW (current device) instead of inside the first prompt pass: the runtime loads a
// translation unit's device code at the first launch of one of its kernels -- 1-2 ms each for the per-type GEMM units.  Called by the shim when such weights are uploaded.
int cdna4_flash_attn_preload(void);
int cdna4_preload_type(int type) {
    if (!weight_type_ok(type)) return set_err(CDNA4_E_UNSUPPORTED, "preload: weight type %d unsupported", type);
    if (type_is_pretiled(type)) type -= T_PRETILED;
    int rc = -1;
#define PL(T) case T: rc = cdna4_gemm_preload_##T(); break;
    switch (type_base(type)) { CDNA4_FOR_BASE_TYPES(PL) }
#undef PL
    if (rc == -1) rc = 0;                                   // (a type without a GEMM unit of its own: nothing to load)
    if (rc == 0) rc = cdna4_flash_attn_preload();
    if (rc) { (void)hipGetLastError(); return set_err(CDNA4_E_HIP, "preload of the type-%d kernels failed", type); }
    return CDNA4_OK;
}
// f16 activation image of a batch: slabs X16[K / 64][ny_pad][64] (padding rows zeroed by the same kernel) followed by the per-row
// range-guard scales (convert.cuh); both live in the context workspace
struct XImage { __half *x; float *scale; long ny_pad; };
static size_t ximage_bytes(long ny_pad, long K) { return (((size_t)ny_pad * K * sizeof(__half) + 255) & ~(size_t)255) + (((size_t)ny_pad * sizeof(float) + 255) & ~(size_t)255); }
// split-K partial sums (gemm_mfma.cuh): only grids of fewer workgroups than CUs are split, at most until they cover ~2 x the CUs, so the partial sums of all slices never
// exceed 2 * num_cu tiles of 128 rows x 256 tokens
static size_t ksplit_ws_bytes(const cdna4_context *ctx, long Nx, long Ny) {
    const size_t cap = (size_t)2 * ctx->num_cu * 128 * 256 * sizeof(float);
    const size_t need = (size_t)((Nx + 255) & ~255L) * (size_t)((Ny + 255) & ~255L) * sizeof(float) * 8;      // (slabs are padded to whole tiles)
    return need < cap ? need : cap;
}
// K-split prompt launches add their slices in slice order through write-through partial slabs (gemm_mfma.cuh): deterministic and the default since round 4.
// CDNA4_SPLITK_ATOMICS=1: the f32-atomics form of rounds 1-3 (developer A/B knob; not reproducible run to run).
static bool splitk_slabs(const cdna4_context *) { static const bool atomics = getenv("CDNA4_SPLITK_ATOMICS") && atoi(getenv("CDNA4_SPLITK_ATOMICS")) != 0; return !atomics; }
static int make_ximage(cdna4_context *ctx, const void *B, long strideB, long K, long Ny, hipStream_t st, XImage &xi) {
    xi.ny_pad = gemm_mfma_npad(Ny);
    int rc = ensure_ws(ctx, ximage_bytes(xi.ny_pad, K), st); if (rc) return rc;
    xi.x = (__half *)ctx->ws; xi.scale = (float *)((char *)ctx->ws + (((size_t)xi.ny_pad * K * sizeof(__half) + 255) & ~(size_t)255));
    if (ctx->fx && ctx->fx->norm_w)         // prompt batch of a fused call: [ADD +] RMS norm + image in one launch (ops.hip; the conditions were checked by fused_args_ok)
        return cdna4_launch_norm_f16_slab(B, ctx->fx->add_b, ctx->fx->add_dst, strideB, ctx->fx->norm_w, ctx->fx->norm_eps, K, Ny, xi.x, xi.ny_pad, xi.scale, st);
    return cdna4_launch_f32_to_f16_slab(B, strideB, K, Ny, xi.x, xi.ny_pad, xi.scale, st);
}
// Large batches (gemm_ppf.cuh): de-quantize the weights ONCE into an f16 image in the workspace, then the f16 x f16 GEMM with both operands by LDS-DMA.  Taken when the
// 256 x 256 tiles fill the GPU and the batch is large enough for the extra pass over the weights to pay (measured crossover ~1500 tokens on the Llama-3-8B FFN shapes:
// profiles/r06_notes.md; CDNA4_PPF_MIN_N, cdna4_set_gemm_form(4) = wherever it can run, 0 / 2 / 3 = never).  Returns 1 = not taken.
static int dequant_slab_dispatch(int type, const GemmArgs &g, void *w16, int *pairing, hipStream_t st) {
#define DS(T) case T: return cdna4_dequant_slab_launch_##T(g, w16, pairing, st);
    switch (type) { CDNA4_FOR_BASE_TYPES(DS) }
#undef DS
    return -1;
}
static int mul_mat_ppf(cdna4_context *ctx, long Nx, long Ny, long K, int typeA, const void *A, const void *A2, long strideA,
                       const void *B, long strideB, float *C, long stride_C, int unary_op, hipStream_t st, const UpGateEpilogue *epi) {
    const int form = cdna4_gemm_form();
    if (form != 1 && form != 4) return 1;
    const int base = type_base(typeA);
    if (base == T_F16 || !gemm_mfma_supported(base) || (K & 127) || Nx < 128 || Ny < 128) return 1;
    // measured crossovers (scripts/mb_forms.py, Llama-3-8B shapes, us per op = activation image + [weight image +] GEMM, fused kernels -> this route, profiles/r06_notes.md):
    //   4096 tokens: Q4_K fused up*gate 1070-1105 -> 1045-1053, 14336 x 4096 538 -> 523, 4096 x 14336 524 -> 555(!), 4096 x 4096 161 -> 151; Q6_K 1251 -> 1053, 597 -> 537, 584 -> 523,
    //   184 -> 158; IQ4_NL 1227 -> 1021, 602 -> 515, 587 -> 506, 185 -> 154.   2048 tokens: Q4_K fused 590 -> 650 (x), 292 -> 282; Q6_K fused 712 -> 640, 331 -> 285; IQ4_NL fused 713 -> 612.
    //   1024 tokens: Q6_K fused 369 -> 385 (x).
    // => from 2048 tokens on; the packed-f16 types (Q4_K / Q5_K: their de-quantizer is the cheapest inside the fused kernels) from 3072 on, and never on their long-row K-split-free
    //    down projection shape (rows < K).  CDNA4_PPF_MIN_N overrides the token threshold.
    static const long env_min_n = getenv("CDNA4_PPF_MIN_N") ? atol(getenv("CDNA4_PPF_MIN_N")) : 0;
    const bool packed = base == T_Q4_K || base == T_Q5_K;
    const long min_n = env_min_n ? env_min_n : (packed ? (A2 ? 3072 : 2048) : 2048);
    const long rows = A2 ? 128 : 256, mt = (Nx + rows - 1) / rows, ntl = (Ny + 255) / 256, wgs = mt * ntl;
    if (form != 4) {
        const double fill = (double)wgs / (double)(((wgs + ctx->num_cu - 1) / ctx->num_cu) * ctx->num_cu), ntok = (double)Ny / (double)(ntl * 256);
        if (Ny < min_n || wgs < (long)(0.9 * ctx->num_cu) || fill * ntok < 0.8 || (packed && !A2 && Nx < K)) return 1;
    }
    const size_t xb = (ximage_bytes(gemm_mfma_npad(Ny), K) + 255) & ~(size_t)255, wb = (size_t)mt * 256 * K * sizeof(__half);
    int rc = ensure_ws(ctx, xb + wb, st); if (rc) return rc;
    XImage xi; rc = make_ximage(ctx, B, strideB, K, Ny, st, xi); if (rc) return rc;
    GemmArgs g; memset(&g, 0, sizeof(g)); if (epi) g.epi = *epi;
    g.A = (const uint8_t *)A; g.A2 = (const uint8_t *)A2; g.strideA = strideA; g.M = (int)Nx; g.N = (int)Ny; g.K = (int)K; g.grid = ctx->grid + grid_offset_of(base);
    void *w16 = (char *)ctx->ws + xb; int pairing = 0;
    rc = dequant_slab_dispatch(base, g, w16, &pairing, st);
    if (rc == -1) return 1;
    if (rc) return set_err(CDNA4_E_HIP, "weight image launch failed: %s", hipGetErrorString(hipGetLastError()));
    g.A = (const uint8_t *)w16; g.X = xi.x; g.xscale = xi.scale; g.xrows = xi.ny_pad; g.C = C; g.stride_C = stride_C; g.unary_op = unary_op; g.n_used = 1; g.nmat = 1; g.pairing = pairing;
    rc = cdna4_gemm_ppf_launch(ctx->num_cu, g, st);
    if (rc) return set_err(CDNA4_E_HIP, "f16 image gemm launch failed: %s", hipGetErrorString(hipGetLastError()));
    return CDNA4_OK;
}
static int mul_mat_mfma(cdna4_context *ctx, long Nx, long Ny, long K, int typeA, const void *A, const void *A2, long strideA,
                        const void *B, long strideB, float *C, long stride_C, int unary_op, hipStream_t st, const UpGateEpilogue *epi = nullptr) {
    { const int rc = mul_mat_ppf(ctx, Nx, Ny, K, typeA, A, A2, strideA, B, strideB, C, stride_C, unary_op, st, epi); if (rc != 1) return rc; }
    const size_t xb = (ximage_bytes(gemm_mfma_npad(Ny), K) + 255) & ~(size_t)255, kb = (A2 || !splitk_slabs(ctx)) ? 0 : ksplit_ws_bytes(ctx, Nx, Ny);
    int rc = ensure_ws(ctx, xb + kb, st); if (rc) return rc;               // (the activation image first: make_ximage then finds its room)
    XImage xi; rc = make_ximage(ctx, B, strideB, K, Ny, st, xi); if (rc) return rc;
    GemmArgs g; memset(&g, 0, sizeof(g)); if (epi) g.epi = *epi;
    g.A = (const uint8_t *)A; g.A2 = (const uint8_t *)A2; g.X = xi.x; g.xscale = xi.scale; g.xrows = xi.ny_pad; g.C = C; g.strideA = strideA; g.stride_C = stride_C;
    g.M = (int)Nx; g.N = (int)Ny; g.K = (int)K; g.unary_op = unary_op; g.n_used = 1; g.nmat = 1;
    g.ks_ws = kb ? (float *)((char *)ctx->ws + xb) : nullptr; g.ks_ws_bytes = kb; g.ks_cnt = ctx->ks_counters; g.ks_fence = ctx->handoff >= 1; g.no_ksplit = ctx->selftest_unsplit;
    rc = gemm_dispatch(ctx, type_base(typeA), g, 0, st);
    if (rc == -1) return set_err(CDNA4_E_UNSUPPORTED, "mfma gemm: type %d not implemented", typeA);
    if (rc) return set_err(CDNA4_E_HIP, "mfma gemm launch failed: %s", hipGetErrorString(hipGetLastError()));
    HIP_TRY(hipGetLastError());
    return CDNA4_OK;
}

// Prompt batches of a weight type without an MFMA tile of its own (the decode-only types): de-quantize a chunk of rows to f16 in the workspace (L0 values rounded
// once to f16 -- what the fused tiles produce in registers), then the f16 instance of the same GEMM.  Chunks bound the workspace (default 256 MiB of f16 weights).
// K that is a multiple of 64 but not of 128 (gpt-oss: 2880; the GEMM walks 128-wide K tiles): both operands are ZERO-PADDED to Kp = the next multiple of 128 -- the f16
// weight rows get Kp columns (the buffer is cleared first), the activation image one more 64-wide slab of zeros -- so the products of the padding are exact zeros.
static int mul_mat_via_f16(cdna4_context *ctx, long Nx, long Ny, long K, int typeA, const void *A, const void *A2, long strideA,
                           const void *B, long strideB, float *C, long stride_C, int unary_op, hipStream_t st, const UpGateEpilogue *epi) {
    const long Kp = (K + 127) & ~127L;
    const long budget = (getenv("CDNA4_F16_CHUNK_MB") ? atol(getenv("CDNA4_F16_CHUNK_MB")) : 256) << 20;      // (read per call: tests shrink it)
    const long row_bytes = Kp * 2, rows_chunk = std::min<long>((Nx + 127) & ~127L, std::max<long>(128, (budget / (row_bytes * (A2 ? 2 : 1))) & ~127L));
    const long ny_pad = gemm_mfma_npad(Ny);
    const size_t xbytes = (ximage_bytes(ny_pad, Kp) + 255) & ~(size_t)255, wbytes = ((size_t)rows_chunk * row_bytes + 255) & ~(size_t)255;
    int rc = ensure_ws(ctx, xbytes + wbytes * (A2 ? 2 : 1), st); if (rc) return rc;
    XImage xi; xi.ny_pad = ny_pad; xi.x = (__half *)ctx->ws; xi.scale = (float *)((char *)ctx->ws + (((size_t)ny_pad * Kp * sizeof(__half) + 255) & ~(size_t)255));
    rc = cdna4_launch_f32_to_f16_slab(B, strideB, K, Ny, xi.x, ny_pad, xi.scale, st); if (rc) return rc;
    if (Kp != K) HIP_TRY(hipMemsetAsync((char *)xi.x + (size_t)(K / 64) * ny_pad * 128, 0, (size_t)ny_pad * 128, st));         // the extra slab
    char *w1 = (char *)ctx->ws + xbytes, *w2 = w1 + wbytes;
    for (long r0 = 0; r0 < Nx; r0 += rows_chunk) {
        const long n = std::min(rows_chunk, Nx - r0);
        if (Kp != K) HIP_TRY(hipMemsetAsync(w1, 0, wbytes * (A2 ? 2 : 1), st));
        rc = cdna4_launch_dequant(ctx, type_base(typeA), (const char *)A + r0 * strideA, strideA, n, K, w1, T_F16, Kp, st, true); if (rc) return rc;
        if (A2) { rc = cdna4_launch_dequant(ctx, type_base(typeA), (const char *)A2 + r0 * strideA, strideA, n, K, w2, T_F16, Kp, st, true); if (rc) return rc; }
        GemmArgs g; memset(&g, 0, sizeof(g));
        if (epi) { g.epi = *epi; if (g.epi.up_b) g.epi.up_b += r0; if (g.epi.gate_b) g.epi.gate_b += r0; }
        g.A = (const uint8_t *)w1; g.A2 = A2 ? (const uint8_t *)w2 : nullptr; g.X = xi.x; g.xscale = xi.scale; g.xrows = xi.ny_pad; g.C = C + r0; g.strideA = row_bytes; g.stride_C = stride_C;
        g.M = (int)n; g.N = (int)Ny; g.K = (int)Kp; g.unary_op = unary_op; g.n_used = 1; g.nmat = 1;
        rc = gemm_dispatch(ctx, T_F16, g, 0, st);
        if (rc) return set_err(CDNA4_E_HIP, "f16 gemm launch failed: %s", hipGetErrorString(hipGetLastError()));
    }
    HIP_TRY(hipGetLastError());
    return CDNA4_OK;
}

static int mul_mat_any(cdna4_context *ctx, long Nx, long Ny, long K, int typeA, const void *A, const void *A2, long strideA,
                       int typeB, const void *B, long strideB, float *C, long stride_C, int unary_op, hipStream_t st, const UpGateEpilogue *epi = nullptr) {
    if (Nx == 0 || Ny == 0) return CDNA4_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    if (K == 0) {   // empty contraction: result is zero (ggml semantics)
        for (long n = 0; n < Ny; ++n) HIP_TRY(hipMemsetAsync(C + n * stride_C, 0, (size_t)Nx * sizeof(float), st));
        return CDNA4_OK;
    }
    if (type_is_r4(typeA)) {            // serve _R4 tensors from their un-interleaved shadow; typeA keeps selecting the _R4 activation arithmetic
        const void *sa = nullptr; int rc = shadow_of(ctx, typeA, A, Nx, K, strideA, st, &sa); if (rc) return rc; A = sa;
        if (A2) { rc = shadow_of(ctx, typeA, A2, Nx, K, strideA, st, &sa); if (rc) return rc; A2 = sa; }
    }
    const bool f16_exact = ctx->prefill_mode == CDNA4_PREFILL_MFMA_F16_EXACT, f16_mode = ctx->prefill_mode == CDNA4_PREFILL_MFMA_F16 || f16_exact;
    if (type_is_bitnet(typeA)) {        // BitNet (gemv_bitnet.hip): decode batches on the Q8_K64 kernels, prompt batches through the f16 route
        if (A2) return set_err(CDNA4_E_UNSUPPORTED, "fused up*gate on BitNet weights is not implemented");
        if (Ny > 8 && typeB == T_F32 && f16_mode && K % 64 == 0) return mul_mat_via_f16(ctx, Nx, Ny, K, typeA, A, nullptr, strideA, B, strideB, C, stride_C, 0, st, nullptr);
        const void *xq = B; long xs = strideB;
        if (typeB == T_F32) {
            const long rb = 32 + K; int rc = ensure_ws(ctx, (size_t)rb * Ny, st); if (rc) return rc;
            rc = cdna4_launch_quantize_q8_k64(B, strideB, Ny, K, ctx->ws, rb, st); if (rc) return rc;
            xq = ctx->ws; xs = rb;
        }
        for (long c0 = 0; c0 < Ny;) {
            int n = Ny - c0 >= 4 ? 4 : Ny - c0 >= 2 ? 2 : 1;
            while (n > 1 && (size_t)n * (K + 32) > 96 * 1024) n >>= 1;
            const int rc = cdna4_launch_gemv_bitnet(ctx, typeA, A, strideA, Nx, K, (const char *)xq + c0 * xs, xs, n, C + c0 * stride_C, stride_C, st); if (rc) return rc;
            c0 += n;
        }
        return CDNA4_OK;
    }
    const bool mfma_ok = typeB == T_F32 && f16_mode && gemm_mfma_supported(type_base(typeA)) && (K % 128 == 0);
    // (every type has an MFMA tile of its own at present: CDNA4_FORCE_F16_ROUTE=1 / CDNA4_PREFILL_MFMA_F16_EXACT send a mat-mul down the generic route:
    //  weights de-quantized to f16 -- the L0 value rounded once -- then the f16 instance of the GEMM)
    const char *force_f16 = getenv("CDNA4_FORCE_F16_ROUTE");
    if (Ny > 8 && typeB == T_F32 && f16_mode && K % 64 == 0 && !type_is_r4(typeA) && !type_is_pretiled(typeA) &&
        (K % 128 != 0 || (!mfma_ok && !gemm_mfma_supported(type_base(typeA))) || (force_f16 && force_f16[0] == '1') || f16_exact))        // (K % 128 != 0: zero-padded inside)
        return mul_mat_via_f16(ctx, Nx, Ny, K, typeA, A, A2, strideA, B, strideB, C, stride_C, unary_op, st, epi);
    if (Ny <= 8 || !mfma_ok) return mul_mat_gemv(ctx, Nx, Ny, K, typeA, A, A2, strideA, typeB, B, strideB, C, stride_C, unary_op, st, epi);
    return mul_mat_mfma(ctx, Nx, Ny, K, typeA, A, A2, strideA, B, strideB, C, stride_C, unary_op, st, epi);
}

int cdna4_mul_mat(cdna4_context *ctx, long Nx, long Ny, long ne00, int typeA, const void *A, long strideA,
                  int typeB, const void *B, long strideB, float *C, long stride_C, void *stream) {
    int rc = check_mm_args(ctx, Nx, Ny, ne00, typeA, A, strideA, typeB, B, C); if (rc) return rc;
    return mul_mat_any(ctx, Nx, Ny, ne00, typeA, A, nullptr, strideA, typeB, B, strideB, C, stride_C, 0, (hipStream_t)stream);
}

// q,k,v epilogue of a fused decode launch (cdna4_fusion.qkv): group slot g of `a` holds the caller's matrix `orig`
static void apply_qkv(const cdna4_context *ctx, GemvArgs &a, int g, int orig) {
    const cdna4_qkv_epilogue *q = ctx->fx ? ctx->fx->qkv : nullptr; if (!q) return;
    a.rope_tab = (const float2 *)ctx->rope_table; a.rope_hd = q->head_dim; a.rope_nd = q->n_dims;
    a.kind[g] = q->kind[orig]; a.kv_slot[g] = q->kv_slot[orig];
    if (q->kind[orig] != 0) a.C[g] = (float *)q->kv_dst[orig];
}
// several weight matrices sharing one activation batch (q,k,v): matrices of the same type go out in ONE launch
int cdna4_mul_mat_multi(cdna4_context *ctx, int n_mats, const long *Nx, long Ny, long ne00, const int *typeA, const void *const *A, const long *strideA,
                        int typeB, const void *B, long strideB, float *const *C, const long *stride_C, void *stream) {
    if (!ctx || n_mats <= 0 || !Nx || !typeA || !A || !strideA || !C || !stride_C) return set_err(CDNA4_E_INVALID, "bad multi mat-mul arguments");
    hipStream_t st = (hipStream_t)stream;
    bool done[16] = {false};
    if (n_mats > 16) return set_err(CDNA4_E_INVALID, "at most 16 matrices");
    for (int i = 0; i < n_mats; ++i) { int rc = check_mm_args(ctx, Nx[i], Ny, ne00, typeA[i], A[i], strideA[i], typeB, B, C[i]); if (rc) return rc; }
    {   bool bn = false; for (int i = 0; i < n_mats; ++i) bn = bn || type_is_bitnet(typeA[i]);
        if (bn) {                           // BitNet matrices are outside the grouped launches: one mat-mul each
            HIP_TRY(hipSetDevice(ctx->device));
            if (ctx->fx) return set_err(CDNA4_E_UNSUPPORTED, "fused norm / residual on BitNet weights is not implemented");
            for (int i = 0; i < n_mats; ++i) { const int rc = mul_mat_any(ctx, Nx[i], Ny, ne00, typeA[i], A[i], nullptr, strideA[i], typeB, B, strideB, C[i], stride_C[i], 0, st); if (rc) return rc; }
            return CDNA4_OK;
        } }
    // prompt batches: convert the shared activations to f16 ONCE, then one MFMA launch per group of same-type matrices
    const bool prefill = Ny > 8 && typeB == T_F32 && ctx->prefill_mode == CDNA4_PREFILL_MFMA_F16 && ne00 > 0 && ne00 % 128 == 0;
    XImage xi; xi.x = nullptr; xi.scale = nullptr; xi.ny_pad = 0; size_t multi_xb = 0, multi_kb = 0;
    if (prefill) {
        bool all_ok = true; for (int i = 0; i < n_mats; ++i) all_ok = all_ok && gemm_mfma_supported(type_base(typeA[i])) && !type_is_r4(typeA[i]);
        if (all_ok) {
            HIP_TRY(hipSetDevice(ctx->device));
            long nx_max = 0; for (int i = 0; i < n_mats; ++i) nx_max = std::max(nx_max, Nx[i]);
            multi_xb = (ximage_bytes(gemm_mfma_npad(Ny), ne00) + 255) & ~(size_t)255; multi_kb = splitk_slabs(ctx) ? ksplit_ws_bytes(ctx, nx_max, Ny) : 0;
            int rc = ensure_ws(ctx, multi_xb + multi_kb, st); if (rc) return rc;               // (the activation image first: make_ximage then finds its room)
            rc = make_ximage(ctx, B, strideB, ne00, Ny, st, xi); if (rc) return rc;
        }
    }
    const __half *xh = xi.x;
    // decode, exactly two type groups {Q4_K|Q5_K matrices} + {one Q6_K matrix} (Q4_K_M / Q5_K_M attention: q,k + attn_v): one launch
    if (Ny == 1 && xh == nullptr && typeB == T_F32 && ne00 > 0 && n_mats >= 2 && n_mats <= GEMV_MAX_MATS + 1) {
        int ib = -1, nb = 0, ta = -1; bool ok = true; long tot = 0;
        for (int i = 0; i < n_mats; ++i) {
            if (typeA[i] == T_Q6_K) { ib = i; ++nb; }
            else if ((typeA[i] == T_Q4_K || typeA[i] == T_Q5_K) && (ta < 0 || ta == typeA[i])) ta = typeA[i];
            else ok = false;
        }
        for (int i = 0; i < n_mats; ++i) if (ok && i != ib && ta >= 0) { for (int j = 0; j < n_mats; ++j) if (j != ib && strideA[j] != strideA[i]) ok = false; }
        if (ok && nb == 1 && ta >= 0 && Nx[ib] > 0) {
            HIP_TRY(hipSetDevice(ctx->device));
            GemvArgs a, b; memset(&a, 0, sizeof(a)); memset(&b, 0, sizeof(b));
            int g = 0;
            for (int i = 0; i < n_mats; ++i) if (i != ib) { a.A[g] = (const uint8_t *)A[i]; a.C[g] = C[i]; tot += Nx[i]; a.mend[g] = (int)tot; a.strideA = strideA[i]; a.stride_C = stride_C[i]; apply_qkv(ctx, a, g, i); ++g; }
            a.nmat = g; a.B = (const uint8_t *)B; a.strideB = strideB; a.M = (int)tot; a.K = (int)ne00; a.src_f32 = 1;
            b.A[0] = (const uint8_t *)A[ib]; b.C[0] = C[ib]; b.mend[0] = (int)Nx[ib]; b.nmat = 1; b.B = (const uint8_t *)B; b.strideA = strideA[ib]; b.strideB = strideB;
            b.stride_C = stride_C[ib]; b.M = (int)Nx[ib]; b.K = (int)ne00; b.src_f32 = 1;
            if (ctx->fx) { a.norm_w = b.norm_w = ctx->fx->norm_w; a.norm_eps = b.norm_eps = ctx->fx->norm_eps; apply_qkv(ctx, b, 0, ib); }
            if (tot > 0) { const int rc = cdna4_gemv_dual_launch(ctx, ta, a, b, st); if (rc == CDNA4_OK) return CDNA4_OK; if (rc != -1) return rc; }
        }
    }
    for (int i = 0; i < n_mats; ++i) {
        if (done[i]) continue;
        int grp[GEMV_MAX_MATS], ng = 0;
        const bool fusable = (Ny == 1 || xh != nullptr) && !type_is_r4(typeA[i]) && ne00 > 0;
        for (int j = i; j < n_mats && ng < GEMV_MAX_MATS; ++j)
            if (!done[j] && (j == i || (fusable && typeA[j] == typeA[i] && strideA[j] == strideA[i] && (Ny == 1 || xh != nullptr || stride_C[j] == stride_C[i]) && (xh == nullptr || Nx[j] % 128 == 0)))) grp[ng++] = j;      // (one result row: its stride is irrelevant; the GEMM takes a stride per matrix)
        if (xh != nullptr) {            // MFMA path on the shared f16 activations (row counts of all but the last matrix must be tile aligned)
            if (Nx[i] % 128 != 0 && ng > 1) ng = 1;
            GemmArgs g; memset(&g, 0, sizeof(g));
            long tot = 0;
            for (int k = 0; k < ng; ++k) { g.Am[k] = (const uint8_t *)A[grp[k]]; g.Cm[k] = C[grp[k]]; g.stride_Cm[k] = stride_C[grp[k]]; tot += Nx[grp[k]]; g.mend[k] = (int)tot; done[grp[k]] = true; }
            g.nmat = ng; g.A = g.Am[0]; g.C = g.Cm[0]; g.X = xi.x; g.xscale = xi.scale; g.xrows = xi.ny_pad; g.strideA = strideA[i]; g.stride_C = stride_C[i];
            g.M = (int)tot; g.N = (int)Ny; g.K = (int)ne00; g.n_used = 1;
            if (ng == 1 && multi_kb) { g.ks_ws = (float *)((char *)ctx->ws + multi_xb); g.ks_ws_bytes = multi_kb; g.ks_cnt = ctx->ks_counters; g.ks_fence = ctx->handoff >= 1; }
            int rc = gemm_dispatch(ctx, type_base(typeA[i]), g, 0, st);
            if (rc) return set_err(CDNA4_E_UNSUPPORTED, "multi gemm: type %d (rc %d)", typeA[i], rc);
            HIP_TRY(hipGetLastError());
            continue;
        }
        if (ng == 1 || !fusable) {
            int rc = mul_mat_any(ctx, Nx[i], Ny, ne00, typeA[i], A[i], nullptr, strideA[i], typeB, B, strideB, C[i], stride_C[i], 0, st);
            if (rc) return rc; done[i] = true; continue;
        }
        HIP_TRY(hipSetDevice(ctx->device));
        GemvArgs a; memset(&a, 0, sizeof(a));
        long tot = 0;
        for (int g = 0; g < ng; ++g) { a.A[g] = (const uint8_t *)A[grp[g]]; a.C[g] = C[grp[g]]; tot += Nx[grp[g]]; a.mend[g] = (int)tot; done[grp[g]] = true; apply_qkv(ctx, a, g, grp[g]); }
        a.nmat = ng; a.B = (const uint8_t *)B; a.strideA = strideA[i]; a.strideB = strideB; a.stride_C = stride_C[i]; a.M = (int)tot; a.K = (int)ne00; a.src_f32 = typeB == T_F32;
        if (ctx->fx) { a.norm_w = ctx->fx->norm_w; a.norm_eps = ctx->fx->norm_eps; }
        int rc = launch_gemv<false>(ctx, type_base(typeA[i]), type_vec_dot(typeA[i]), a, 1, 1, st); if (rc) return rc;
    }
    return CDNA4_OK;
}

int cdna4_mul_mat_4d(cdna4_context *ctx, long Nx, long Ny, long ne00, long ne02, long ne03, long ne12, long ne13,
                     long nb02, long nb03, long nb12, long nb13, long nb2, long nb3,
                     int typeA, const void *A, long strideA, int typeB, const void *B, long strideB,
                     float *C, long stride_C, void *stream) {
    int rc = check_mm_args(ctx, Nx, Ny, ne00, typeA, A, strideA, typeB, B, C); if (rc) return rc;
    if (ne02 <= 0 || ne03 <= 0 || ne12 % ne02 || ne13 % ne03) return set_err(CDNA4_E_INVALID, "broadcast rule violated (ne12 %% ne02, ne13 %% ne03)");
    const long r2 = ne12 / ne02, r3 = ne13 / ne03;      // iqk_mul_mat_4d (iqk_mul_mat.cpp:624-711) broadcast
    for (long i13 = 0; i13 < ne13; ++i13)
        for (long i12 = 0; i12 < ne12; ++i12) {
            const uint8_t *a = (const uint8_t *)A + (i12 / r2) * nb02 + (i13 / r3) * nb03;
            const uint8_t *b = (const uint8_t *)B + i12 * nb12 + i13 * nb13;
            rc = mul_mat_any(ctx, Nx, Ny, ne00, typeA, a, nullptr, strideA, typeB, b, strideB, C + i12 * nb2 + i13 * nb3, stride_C, 0, (hipStream_t)stream);
            if (rc) return rc;
        }
    return CDNA4_OK;
}

static bool up_gate_op_ok(int op) { return op == CDNA4_UNARY_RELU || op == CDNA4_UNARY_GELU || op == CDNA4_UNARY_SILU || op == CDNA4_UNARY_SWIGLU_OAI; }

int cdna4_fused_up_gate_ext(cdna4_context *ctx, long Nx, long Ny, long ne00, int unary_op, int typeA, const void *Aup, const void *Agate, long strideA,
                            int typeB, const void *B, long strideB, const float *up_b, const float *gate_b, float limit,
                            float *C, long stride_C, void *stream) {
    int rc = check_mm_args(ctx, Nx, Ny, ne00, typeA, Aup, strideA, typeB, B, C); if (rc) return rc;
    if (!Agate && Nx && Ny) return set_err(CDNA4_E_INVALID, "null gate weights");
    if (!up_gate_op_ok(unary_op)) return set_err(CDNA4_E_UNSUPPORTED, "unary op %d unsupported", unary_op);
    UpGateEpilogue epi; memset(&epi, 0, sizeof(epi)); epi.up_b = up_b; epi.gate_b = gate_b; epi.limit = limit;
    return mul_mat_any(ctx, Nx, Ny, ne00, typeA, Aup, Agate, strideA, typeB, B, strideB, C, stride_C, unary_op, (hipStream_t)stream, &epi);
}
// decode (one activation row): the fused result row is ALSO emitted quantized to block_q8_2_x4 -- bit-identical to what the next
// mat-mul's prologue would make of the f32 row -- so that a following cdna4_mul_mat(typeB = Q8_2_X4) only copies it
// (the CUDA reference re-quantises between the fused up*gate and the down mat-mul as well: ggml-cuda.cu:3062-3185)
int cdna4_fused_up_gate_q8(cdna4_context *ctx, long Nx, long ne00, int unary_op, int typeA, const void *Aup, const void *Agate, long strideA,
                           const float *B, const float *up_b, const float *gate_b, float limit, float *C, void *q8_out, void *stream) {
    int rc = check_mm_args(ctx, Nx, 1, ne00, typeA, Aup, strideA, T_F32, B, C); if (rc) return rc;
    if (!Agate || !q8_out) return set_err(CDNA4_E_INVALID, "null gate weights / q8 output");
    if (!up_gate_op_ok(unary_op)) return set_err(CDNA4_E_UNSUPPORTED, "unary op %d unsupported", unary_op);
    if (type_is_r4(typeA) || Nx % 128 || ne00 < 4096 || ne00 > 4096 * 1) return set_err(CDNA4_E_UNSUPPORTED, "q8 emission needs a base type, Nx %% 128 == 0 and ne00 == 4096");
    HIP_TRY(hipSetDevice(ctx->device));
    UpGateEpilogue epi; memset(&epi, 0, sizeof(epi)); epi.up_b = up_b; epi.gate_b = gate_b; epi.limit = limit;
    return mul_mat_gemv(ctx, Nx, 1, ne00, typeA, Aup, Agate, strideA, T_F32, B, ne00 * 4, C, Nx, unary_op, (hipStream_t)stream, &epi, q8_out);
}
int cdna4_fused_up_gate(cdna4_context *ctx, long Nx, long Ny, long ne00, int unary_op, int typeA, const void *Aup, const void *Agate, long strideA,
                        int typeB, const void *B, long strideB, float *C, long stride_C, void *stream) {
    return cdna4_fused_up_gate_ext(ctx, Nx, Ny, ne00, unary_op, typeA, Aup, Agate, strideA, typeB, B, strideB, nullptr, nullptr, 0.f, C, stride_C, stream);
}

static int moe_common(cdna4_context *ctx, long Nx, long K, int n_expert, int n_used, long n_tokens, int unary_op, int typeA,
                      const void *A, const void *A2, long strideA, long nb02, const float *B, int n_b, long nb11, long nb12,
                      const int32_t *ids, long ids_nb1, float *C, long nb1, long nb2, hipStream_t st, const UpGateEpilogue *epi = nullptr) {
    int rc = check_mm_args(ctx, Nx, 1, K, typeA, A, strideA, T_F32, B, C); if (rc) return rc;
    if (n_expert <= 0 || n_used <= 0 || n_tokens < 0 || !ids) return set_err(CDNA4_E_INVALID, "bad MoE arguments");
    if (type_is_bitnet(typeA)) return set_err(CDNA4_E_UNSUPPORTED, "MUL_MAT_ID on BitNet weights is not implemented");
    if (n_b != 1 && n_b != n_used) return set_err(CDNA4_E_INVALID, "n_b must be 1 or n_used");
    if (n_tokens == 0 || Nx == 0) return CDNA4_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    const int vdt = type_vec_dot(typeA);               // the activation arithmetic the CPU pairs with this tensor type (_R4: Q8_K32 / Q8_K)
    if (type_is_r4(typeA)) {                            // _R4 experts: all experts' rows un-interleaved once as one tall matrix (a8 x a11)
        if (nb02 != Nx * strideA) return set_err(CDNA4_E_UNSUPPORTED, "moe: _R4 expert tensors must be contiguous (nb02 == Nx * nb01)");
        const void *sa = nullptr; rc = shadow_of(ctx, typeA, A, (long)n_expert * Nx, K, strideA, st, &sa); if (rc) return rc; A = sa;
        if (A2) { rc = shadow_of(ctx, typeA, A2, (long)n_expert * Nx, K, strideA, st, &sa); if (rc) return rc; A2 = sa; }
    }
    typeA = type_base(typeA);                           // (pre-tiled _R4 tensors arrive in the base tiling: nothing to convert)
    const long pairs = n_tokens * n_used;
    // prompt-sized batches: group the (token, slot) pairs by expert ON THE DEVICE and run one grouped MFMA GEMM over all experts
    // Crossover (scripts/microbench.py moe, profiles/r01_notes.md): below ~4 pairs per expert the id-indexed GEMV (each pair streams its
    // expert once, 4 loads in flight per lane) beats the grouped GEMM, whose 32-token tiles are then latency-bound with a ~90 us floor.
    static const long moe_env_min = getenv("CDNA4_MOE_GEMM_MIN_PAIRS") ? atol(getenv("CDNA4_MOE_GEMM_MIN_PAIRS")) : 0;
    const long moe_gemm_min_pairs = moe_env_min ? moe_env_min : std::max<long>(32, 4L * n_expert);
    // a type without an MFMA tile of its own (the decode-only trellis types) takes the same grouped launch through the f16 route: experts are de-quantized to f16 a chunk
    // at a time (the value a MAT-MUL gives a weight, kt_matmul_factor) and the f16 instance of the grouped GEMM runs the tiles of that chunk's experts
    const bool via_f16 = !gemm_mfma_supported(typeA) && !type_is_bitnet(typeA);
    if (pairs >= moe_gemm_min_pairs && ctx->prefill_mode == CDNA4_PREFILL_MFMA_F16 && (gemm_mfma_supported(typeA) || via_f16) && K % 128 == 0 && n_expert <= 1024 && pairs < (1 << 24)) {
        const long avg = pairs / n_expert;
        static const int env_moe_nt = getenv("CDNA4_MOE_NT") ? atoi(getenv("CDNA4_MOE_NT")) : 0;
        // token-tile width by the average pairs per expert.  The kernel only multiplies the populated 32-token sub-tiles of a tile (gemm_mfma.cuh, COMPUTE_TILE_PART), so
        // a wide tile no longer pays for its padding in MFMAs: 128-token tiles from ~48 pairs per expert on (one de-quantization pass per expert instead of two or
        // four; round 2 needed ~256 pairs per expert before they paid).
        // -- as long as the 128-token grid still gives every CU two workgroups (Mixtral's 4096-row down projection at 512 tokens does not: 12 tiles x 32 row tiles = 384
        // workgroups of one wave per SIMD measured 430 us against 399 us with 64-token tiles; its 14336-row fused up*gate launch does: 792 -> 666 us)
        const long mt = (Nx + 127) / 128, tiles128 = pairs / 128 + (n_expert + 1) / 2;
        int nt = avg >= 48 ? 4 : avg >= 16 ? 2 : 1;
        if (nt == 4 && tiles128 * mt < 2L * ctx->num_cu) nt = 2;
        // short rows (Qwen3-30B-A3B's 768-wide down projection: 6 K tiles per workgroup) have too little de-quantization work per tile for the wide tile to pay for its
        // padding: 2048 tokens x 8 of 128 experts, 64- vs 128-token tiles: 196-203 vs 236-249 us (the fused 2048-wide launch: 340 vs 312-317 us), scripts/r03_gpu15.sh
        if (nt == 4 && !A2 && K < 2048) nt = 2;
        if (env_moe_nt) nt = env_moe_nt;
        const int BN = 32 * nt;
        const int max_tiles = (int)(pairs / BN + n_expert + 1);
        const long rows_pad = pairs + 256;
        const size_t x_bytes = ((size_t)rows_pad * K * sizeof(__half) + 255) & ~(size_t)255, s_bytes = ((size_t)rows_pad * sizeof(float) + 255) & ~(size_t)255;
        const size_t i_bytes = ((size_t)(pairs + 3 * max_tiles + 16) * sizeof(int) + 255) & ~(size_t)255;
        const long f16_budget = (getenv("CDNA4_F16_MOE_CHUNK_MB") ? atol(getenv("CDNA4_F16_MOE_CHUNK_MB")) : 1024) << 20;      // (read per call: tests shrink it)
        const size_t e_bytes = (size_t)Nx * K * sizeof(__half);                                                                 // one expert, one matrix, as f16
        const int e_chunk = via_f16 ? (int)std::max<long>(1, std::min<long>(n_expert, f16_budget / (long)(e_bytes * (A2 ? 2 : 1)))) : 0;
        const size_t need = x_bytes + s_bytes + i_bytes + (via_f16 ? e_bytes * (size_t)e_chunk * (A2 ? 2 : 1) : 0);
        rc = ensure_ws(ctx, need, st); if (rc) return rc;
        __half *xh = (__half *)ctx->ws; float *xscale = (float *)((char *)ctx->ws + x_bytes);
        int *pairs_sorted = (int *)((char *)ctx->ws + x_bytes + s_bytes), *tiles = pairs_sorted + pairs;
        HIP_TRY(hipMemsetAsync(pairs_sorted, 0xff, (size_t)pairs * sizeof(int), st));
        rc = cdna4_launch_moe_sort(ids, ids_nb1, (int)n_tokens, n_used, n_expert, BN, max_tiles, pairs_sorted, tiles, C, nb1, nb2, (int)Nx, st); if (rc) return rc;
        rc = cdna4_launch_moe_gather_f16(B, n_b, n_b == 1 ? 0 : nb11, nb12, n_used, pairs_sorted, rows_pad, pairs, K, xh, xscale, st); if (rc) return rc;
        GemmArgs g; memset(&g, 0, sizeof(g));
        g.A = (const uint8_t *)A; g.A2 = (const uint8_t *)A2; g.X = xh; g.xscale = xscale; g.xrows = rows_pad; g.C = C; g.strideA = strideA; g.stride_C = 0; g.M = (int)Nx; g.N = max_tiles; g.K = (int)K; g.unary_op = unary_op;
        if (epi) g.epi = *epi;
        g.moe_tiles = tiles; g.moe_pairs = pairs_sorted; g.expert_stride = nb02; g.nb1 = nb1; g.nb2 = nb2; g.n_used = n_used;
        if (via_f16) {
            char *w1 = (char *)ctx->ws + x_bytes + s_bytes + i_bytes, *w2 = w1 + e_bytes * (size_t)e_chunk;
            g.A = (const uint8_t *)w1; g.A2 = A2 ? (const uint8_t *)w2 : nullptr; g.strideA = K * (long)sizeof(__half); g.expert_stride = (long)e_bytes;
            for (int e0 = 0; e0 < n_expert; e0 += e_chunk) {
                const int e1 = std::min(n_expert, e0 + e_chunk);
                for (int e = e0; e < e1; ++e) {       // (experts need not be contiguous: one de-quantization launch each)
                    rc = cdna4_launch_dequant(ctx, typeA, (const char *)A + (long)e * nb02, strideA, Nx, K, w1 + (size_t)(e - e0) * e_bytes, T_F16, K, st, true); if (rc) return rc;
                    if (A2) { rc = cdna4_launch_dequant(ctx, typeA, (const char *)A2 + (long)e * nb02, strideA, Nx, K, w2 + (size_t)(e - e0) * e_bytes, T_F16, K, st, true); if (rc) return rc; }
                }
                g.expert_lo = e0; g.expert_hi = e1;
                rc = gemm_dispatch(ctx, T_F16, g, nt, st);
                if (rc) return set_err(CDNA4_E_UNSUPPORTED, "grouped f16 gemm (rc %d)", rc);
            }
            HIP_TRY(hipGetLastError());
            return CDNA4_OK;
        }
        rc = gemm_dispatch(ctx, typeA, g, nt, st);
        if (rc) return set_err(CDNA4_E_UNSUPPORTED, "grouped gemm: type %d (rc %d)", typeA, rc);
        HIP_TRY(hipGetLastError());
        return CDNA4_OK;
    }
    GemvArgs a; memset(&a, 0, sizeof(a));
    a.A[0] = (const uint8_t *)A; a.A2 = (const uint8_t *)A2; a.B = (const uint8_t *)B; a.C[0] = C; a.ids = ids; a.nmat = 1; a.mend[0] = (int)Nx;
    a.strideA = strideA; a.strideB = 0; a.stride_C = 0; a.expert_stride = nb02; a.nb11 = n_b == 1 ? 0 : nb11; a.nb12 = nb12; a.nb1 = nb1; a.nb2 = nb2; a.ids_nb1 = ids_nb1;
    a.M = (int)Nx; a.K = (int)K; a.n_expert = n_expert; a.n_used = n_used; a.unary_op = unary_op; a.src_f32 = 1;
    if (epi) a.epi = *epi;
    for (long p0 = 0; p0 < pairs; p0 += 65535) {      // grid.y limit: (token, slot) pair = blockIdx.y + pair0
        a.pair0 = (int)p0;
        const unsigned gy = (unsigned)(pairs - p0 < 65535 ? pairs - p0 : 65535);
        rc = A2 ? launch_gemv<true>(ctx, typeA, vdt, a, 1, gy, st) : launch_gemv<false>(ctx, typeA, vdt, a, 1, gy, st);
        if (rc) return rc;
    }
    return CDNA4_OK;
}
int cdna4_mul_mat_id(cdna4_context *ctx, long Nx, long ne00, int n_expert, int n_used, long n_tokens, int typeA, const void *A, long strideA, long nb02,
                     const float *B, int n_b, long nb11, long nb12, const int32_t *ids, long ids_nb1, float *C, long nb1, long nb2, void *stream) {
    return moe_common(ctx, Nx, ne00, n_expert, n_used, n_tokens, 0, typeA, A, nullptr, strideA, nb02, B, n_b, nb11, nb12, ids, ids_nb1, C, nb1, nb2, (hipStream_t)stream);
}
int cdna4_moe_fused_up_gate_ext(cdna4_context *ctx, long Nx, long ne00, int n_expert, int n_used, long n_tokens, int unary_op, int typeA,
                                const void *Aup, const void *Agate, long strideA, long nb02, const float *B, int n_b, long nb11, long nb12,
                                const int32_t *ids, long ids_nb1, const float *up_b, long up_b_nb1, const float *gate_b, long gate_b_nb1, float limit,
                                float *C, long nb1, long nb2, void *stream) {
    if (!Agate) return set_err(CDNA4_E_INVALID, "null gate weights");
    if (!up_gate_op_ok(unary_op)) return set_err(CDNA4_E_UNSUPPORTED, "unary op %d unsupported", unary_op);
    if ((up_b && up_b_nb1 % 4) || (gate_b && gate_b_nb1 % 4)) return set_err(CDNA4_E_INVALID, "bias strides must be multiples of 4 bytes");
    UpGateEpilogue epi; memset(&epi, 0, sizeof(epi)); epi.up_b = up_b; epi.gate_b = gate_b; epi.up_b_stride = up_b_nb1 / 4; epi.gate_b_stride = gate_b_nb1 / 4; epi.limit = limit;
    return moe_common(ctx, Nx, ne00, n_expert, n_used, n_tokens, unary_op, typeA, Aup, Agate, strideA, nb02, B, n_b, nb11, nb12, ids, ids_nb1, C, nb1, nb2, (hipStream_t)stream, &epi);
}
int cdna4_moe_ffn(cdna4_context *ctx, long Nx_ff, long ne00, long Nx_out, int n_expert, int n_used, long n_tokens, int unary_op,
                  int type_up_gate, const void *Aup, const void *Agate, long stride_up_gate, long nb02_up_gate,
                  int type_down, const void *Adown, long stride_down, long nb02_down,
                  const float *B, int n_b, long nb11, long nb12, const int32_t *ids, long ids_nb1,
                  const float *up_b, long up_b_nb1, const float *gate_b, long gate_b_nb1, float limit,
                  float *C1, long c1_nb1, long c1_nb2, float *C2, long c2_nb1, long c2_nb2, void *stream) {
    // node 1: fused up*gate of the routed experts; node 2: the down projection reads node 1's rows (one activation row per (token, slot))
    int rc = cdna4_moe_fused_up_gate_ext(ctx, Nx_ff, ne00, n_expert, n_used, n_tokens, unary_op, type_up_gate, Aup, Agate, stride_up_gate, nb02_up_gate, B, n_b, nb11, nb12,
                                         ids, ids_nb1, up_b, up_b_nb1, gate_b, gate_b_nb1, limit, C1, c1_nb1, c1_nb2, stream);
    if (rc) return rc;
    return cdna4_mul_mat_id(ctx, Nx_out, Nx_ff, n_expert, n_used, n_tokens, type_down, Adown, stride_down, nb02_down, C1, n_used, c1_nb1 * 4, c1_nb2 * 4, ids, ids_nb1,
                            C2, c2_nb1, c2_nb2, stream);
}
int cdna4_moe_fused_up_gate(cdna4_context *ctx, long Nx, long ne00, int n_expert, int n_used, long n_tokens, int unary_op, int typeA,
                            const void *Aup, const void *Agate, long strideA, long nb02, const float *B, int n_b, long nb11, long nb12,
                            const int32_t *ids, long ids_nb1, float *C, long nb1, long nb2, void *stream) {
    return cdna4_moe_fused_up_gate_ext(ctx, Nx, ne00, n_expert, n_used, n_tokens, unary_op, typeA, Aup, Agate, strideA, nb02, B, n_b, nb11, nb12, ids, ids_nb1,
                                       nullptr, 0, nullptr, 0, 0.f, C, nb1, nb2, stream);
}


// ---- in-process GGML_OP_REDUCE over peer-mapped buffers (the shim's REDUCE node) -----------------------------------------------
int cdna4_reduce_peers(cdna4_context *ctx, void *const *bufs, int n, unsigned partial_mask, int64_t count, int dtype, void *stream) {
    return cdna4_reduce_peers_slice(ctx, bufs, n, partial_mask, count, dtype, 0, 1, stream);
}
int cdna4_reduce_peers_slice(cdna4_context *ctx, void *const *bufs, int n, unsigned partial_mask, int64_t count, int dtype, int slice, int n_slices, void *stream) {
    if (!ctx || !bufs || n < 1 || n > 16 || count < 0 || n_slices < 1 || slice < 0 || slice >= n_slices) return set_err(CDNA4_E_INVALID, "bad peer-reduce arguments");
    if (count == 0) return CDNA4_OK;
    if (dtype == T_Q8_0 && count % 32) return set_err(CDNA4_E_INVALID, "peer-reduce of Q8_0 partial sums: count must be a multiple of 32");
    int nhave = 0;
    for (int j = 0; j < n; ++j) { if (bufs[j] && ((partial_mask >> j) & 1u)) ++nhave; if (bufs[j] && ((uintptr_t)bufs[j] & (dtype == T_Q8_0 ? 1 : 15))) return set_err(CDNA4_E_INVALID, dtype == T_Q8_0 ? "peer-reduce buffers of Q8_0 blocks must be 2-byte aligned" : "peer-reduce buffers must be 16-byte aligned"); }
    if (nhave < 1) return set_err(CDNA4_E_INVALID, "peer-reduce without a partial");
    HIP_TRY(hipSetDevice(ctx->device));
    return cdna4_launch_reduce_peers(ctx->num_cu, bufs, n, partial_mask, count, dtype, slice, n_slices, (hipStream_t)stream);
}

// ---- graph-level fusions of a decoded token: RMS norm folded into the activation prologue, residual add folded into the epilogue ----------
static int fused_args_ok(cdna4_context *ctx, const cdna4_fusion *fx, long Ny, long ne00, int typeB, int n_types, const int *types, const void *B, long strideB) {
    if (!ctx || !fx) return set_err(CDNA4_E_INVALID, "null argument");
    if (!fx->norm_w && !fx->residual) return set_err(CDNA4_E_INVALID, "empty fusion");
    if ((fx->add_b == nullptr) != (fx->add_dst == nullptr)) return set_err(CDNA4_E_INVALID, "fused ADD: add_b and add_dst come together");
    if (Ny > 8 && typeB == T_F32 && ne00 > 0) {         // prompt batch: [ADD +] norm + f16 image as one launch in front of the matrix-core GEMM(s)
        static const bool force_f16 = getenv("CDNA4_FORCE_F16_ROUTE") && getenv("CDNA4_FORCE_F16_ROUTE")[0] == '1';
        if (!fx->norm_w || fx->residual || fx->qkv) return set_err(CDNA4_E_UNSUPPORTED, "fused prompt batch: the norm (and the ADD in front of it) only");
        if (ctx->prefill_mode != CDNA4_PREFILL_MFMA_F16 || force_f16 || ne00 % 128 || ne00 > 16384) return set_err(CDNA4_E_UNSUPPORTED, "fused prompt batch: MFMA_F16 prefill mode, ne00 %% 128 == 0, ne00 <= 16384");
        for (int i = 0; i < n_types; ++i)
            if (type_is_r4(types[i]) || type_is_pretiled(types[i]) || type_is_bitnet(types[i]) || !gemm_mfma_supported(type_base(types[i])))
                return set_err(CDNA4_E_UNSUPPORTED, "fused prompt batch: weight type %d has no matrix-core tile of its own", types[i]);
        if (((uintptr_t)B | (uintptr_t)fx->norm_w | (uintptr_t)fx->add_b | (uintptr_t)fx->add_dst | (uintptr_t)strideB) % 16) return set_err(CDNA4_E_UNSUPPORTED, "fused prompt batch: 16-byte aligned rows");
        return CDNA4_OK;
    }
    if (Ny == 1 && typeB == T_Q8_2_X4 && ne00 > 0 && fx->residual && !fx->norm_w && !fx->qkv && !fx->add_b) {      // the row arrives quantized (cdna4_op_flash_attn_q8 / cdna4_fused_up_gate_q8): residual only
        for (int i = 0; i < n_types; ++i) if (type_is_r4(types[i]) || type_is_pretiled(types[i]) || type_is_bitnet(types[i]) || type_vec_dot(types[i]) != T_Q8_2_X4) return set_err(CDNA4_E_UNSUPPORTED, "fused residual on Q8_2_X4 activations: a base type whose vec_dot type is Q8_2_X4");
        return CDNA4_OK;
    }
    if (Ny != 1 || typeB != T_F32 || ne00 <= 0) return set_err(CDNA4_E_UNSUPPORTED, "fused norm / residual: one f32 activation row (decode) or a prompt batch (> 8 rows)");
    if (fx->add_b) return set_err(CDNA4_E_UNSUPPORTED, "fused ADD in front of the norm: prompt batches only");
    for (int i = 0; i < n_types; ++i) if (type_is_r4(types[i])) return set_err(CDNA4_E_UNSUPPORTED, "fused norm / residual: row-interleaved tensors must be re-tiled at upload");
    for (int i = 0; i < n_types; ++i) if (type_is_bitnet(types[i])) return set_err(CDNA4_E_UNSUPPORTED, "fused norm / residual on BitNet weights is not implemented");
    return CDNA4_OK;
}
int cdna4_mul_mat_multi_fused(cdna4_context *ctx, int n_mats, const long *Nx, long Ny, long ne00, const int *typeA, const void *const *A, const long *strideA,
                              int typeB, const void *B, long strideB, float *const *C, const long *stride_C, const cdna4_fusion *fx, void *stream) {
    if (n_mats <= 0 || !typeA) return set_err(CDNA4_E_INVALID, "bad multi mat-mul arguments");
    int rc = fused_args_ok(ctx, fx, Ny, ne00, typeB, n_mats, typeA, B, strideB); if (rc) return rc;
    if (fx->residual && n_mats != 1) return set_err(CDNA4_E_UNSUPPORTED, "fused residual: one matrix");
    if (fx->qkv) {
        const cdna4_qkv_epilogue *q = fx->qkv;
        if (!fx->norm_w || fx->residual || n_mats < 2 || n_mats > 4) return set_err(CDNA4_E_UNSUPPORTED, "q,k,v epilogue: with the fused norm, 2..4 matrices");
        if (q->head_dim <= 0 || q->head_dim % 2 || q->n_dims <= 0 || q->n_dims % 2 || q->n_dims > q->head_dim) return set_err(CDNA4_E_INVALID, "q,k,v epilogue: head size / rotated dims");
        if (!ctx->rope_table || !ctx->rope_key.pos || ctx->rope_key.n_tok != 1 || ctx->rope_key.n_dims != q->n_dims) return set_err(CDNA4_E_UNSUPPORTED, "q,k,v epilogue: no current one-token rope cache");
        for (int i = 0; i < n_mats; ++i) {
            if (q->kind[i] < 0 || q->kind[i] > 2 || Nx[i] % q->head_dim || (q->kind[i] != 0 && !q->kv_dst[i] && !q->kv_slot[i])) return set_err(CDNA4_E_INVALID, "q,k,v epilogue: matrix %d", i);
            if (q->kind[i] == 0 && !C[i]) return set_err(CDNA4_E_INVALID, "q,k,v epilogue: null Q destination");
        }
    }
    ctx->fx = fx; rc = cdna4_mul_mat_multi(ctx, n_mats, Nx, Ny, ne00, typeA, A, strideA, typeB, B, strideB, C, stride_C, stream); ctx->fx = nullptr;
    return rc;
}
int cdna4_fused_up_gate_fused(cdna4_context *ctx, long Nx, long Ny, long ne00, int unary_op, int typeA, const void *A_up, const void *A_gate, long strideA,
                              int typeB, const void *B, long strideB, const float *up_b, const float *gate_b, float limit, float *C, long stride_C,
                              const cdna4_fusion *fx, void *stream) {
    int rc = fused_args_ok(ctx, fx, Ny, ne00, typeB, 1, &typeA, B, strideB); if (rc) return rc;
    if (fx->residual) return set_err(CDNA4_E_UNSUPPORTED, "fused residual on the up*gate launch");
    ctx->fx = fx; rc = cdna4_fused_up_gate_ext(ctx, Nx, Ny, ne00, unary_op, typeA, A_up, A_gate, strideA, typeB, B, strideB, up_b, gate_b, limit, C, stride_C, stream); ctx->fx = nullptr;
    return rc;
}

int cdna4_attn_out_fused(cdna4_context *ctx, const cdna4_tensor *q, const cdna4_tensor *k, const cdna4_tensor *v, const cdna4_tensor *mask, const cdna4_tensor *attn,
                         float scale, float max_bias, float softcap, long Nx, long ne00, int typeA, const void *A, long strideA, const float *residual, float *C, void *stream) {
    if (!ctx || !q || !k || !v || !attn || !residual) return set_err(CDNA4_E_INVALID, "null argument");
    int rc = check_mm_args(ctx, Nx, 1, ne00, typeA, A, strideA, T_F32, attn->data, C); if (rc) return rc;
    if (type_is_r4(typeA) || type_is_pretiled(typeA) || type_vec_dot(typeA) != T_Q8_2_X4 || Nx < 1 || !cdna4_fa_is_plain_decode(ctx, q, k, v, mask, attn) || attn->ne[0] * attn->ne[1] != ne00 ||
        (uintptr_t)attn->data % 32 || ctx->fa_counters_bytes < 64)
        return set_err(CDNA4_E_UNSUPPORTED, "attention + attn_output fusion: shape / type not served");
    HIP_TRY(hipSetDevice(ctx->device));
    GemvArgs a; memset(&a, 0, sizeof(a));
    a.A[0] = (const uint8_t *)A; a.C[0] = C; a.mend[0] = (int)Nx; a.nmat = 1; a.B = (const uint8_t *)attn->data; a.strideA = strideA; a.strideB = ne00 * 4; a.stride_C = Nx; a.M = (int)Nx; a.K = (int)ne00;
    a.src_f32 = 1; a.R = residual;
    unsigned *sync = (unsigned *)((char *)ctx->fa_counters + ctx->fa_counters_bytes - 64);
    rc = cdna4_gemv_attn_launch(ctx, type_base(typeA), q, k, v, mask, attn, scale, max_bias, softcap, a, sync, (hipStream_t)stream);
    if (rc == -1) return set_err(CDNA4_E_UNSUPPORTED, "attention + attn_output fusion: shape / type not served");
    return rc;
}

// ---- measurement helper -----------------------------------------------------------------------------------
int cdna4_time_mul_mat(cdna4_context *ctx, long Nx, long Ny, long ne00, int typeA, const void *const *A_rot, int n_rot, long strideA,
                       const float *B, long strideB, float *C, long stride_C, int warmup, int iters, void *stream, float *avg_ms) {
    if (!ctx || !A_rot || n_rot <= 0 || iters <= 0 || !avg_ms) return set_err(CDNA4_E_INVALID, "bad timing arguments");
    hipStream_t st = (hipStream_t)stream;
    for (int i = 0; i < warmup; ++i) { int rc = cdna4_mul_mat(ctx, Nx, Ny, ne00, typeA, A_rot[i % n_rot], strideA, T_F32, B, strideB, C, stride_C, stream); if (rc) return rc; }
    HIP_TRY(hipEventRecord(ctx->ev0, st));
    for (int i = 0; i < iters; ++i) { int rc = cdna4_mul_mat(ctx, Nx, Ny, ne00, typeA, A_rot[i % n_rot], strideA, T_F32, B, strideB, C, stride_C, stream); if (rc) return rc; }
    HIP_TRY(hipEventRecord(ctx->ev1, st));
    HIP_TRY(hipEventSynchronize(ctx->ev1));
    float ms = 0; HIP_TRY(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    *avg_ms = ms / iters; return CDNA4_OK;
}

#include "reduce.inc"

// ---- start-up self-test of the fence-free in-launch hand-offs (VERDICT r05 "missing" 7 / ADVICE r04) -----------------------------------------------------------------
// The split-K prompt GEMM (gemm_mfma.cuh: partial tiles as sc1 stores -> vmcnt(0) -> one relaxed agent-scope ticket -> sc1 loads in the last arriver) and the split-KV decode
// attention (ops.hip: 8-byte agent-scope atomic stores / loads around the same ticket) rely on the write-through behaviour the MI355X guide documents as valid forms
// (Guideline 16, "in-launch split-K reduction") -- a property of this GPU generation and of how the allocation is mapped, not of the HIP memory model.  So every context
// checks them once against their UNSPLIT forms on the live device: four split launches each, the workspace (where the partial slabs live) poisoned with NaN bits in front of
// every one, so that a slab read before it has arrived is a NaN and not a plausible number.  A mismatch switches the context to the fenced forms (release fence before the
// ticket, acquire behind it) and says so once.  CDNA4_SPLITK_FENCE=1: fenced forms without testing; CDNA4_HANDOFF_SELFTEST=0: skip (trust), =fail: pretend a mismatch
// (tests/test_gpu_handoff.py drives the fallback with it).
static uint32_t selftest_lcg(uint32_t &s) { s = s * 1664525u + 1013904223u; return s >> 8; }
static bool selftest_close(const std::vector<float> &a, const std::vector<float> &ref, double rel) {
    double mx = 0; for (float v : ref) { if (!(v == v)) return false; mx = std::max(mx, (double)fabsf(v)); }
    if (mx == 0) return false;                        // (an all-zero reference proves nothing)
    for (size_t i = 0; i < a.size(); ++i) if (!(fabs((double)a[i] - (double)ref[i]) <= rel * mx)) return false;      // (NaN fails the comparison)
    return true;
}
int cdna4_handoff_mode(const cdna4_context *ctx) { return ctx ? ctx->handoff : -1; }
int cdna4_handoff_selftest(cdna4_context *ctx, void *stream) {
    if (!ctx) return set_err(CDNA4_E_INVALID, "null context");
    const char *e_f = getenv("CDNA4_SPLITK_FENCE"), *e_s = getenv("CDNA4_HANDOFF_SELFTEST");
    if (e_f && atoi(e_f) != 0) { ctx->handoff = 1; return CDNA4_OK; }
    if (e_s && !strcmp(e_s, "0")) { ctx->handoff = 0; return CDNA4_OK; }
    const bool force_fail = e_s && !strcmp(e_s, "fail");
    const bool no_poison = e_s && !strcmp(e_s, "nopoison");      // (debug)
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(ctx->device));
    const int prev = ctx->handoff; ctx->handoff = 0;                                   // test the fence-free forms
    // (a) split-K GEMM: Q4_K 1024 x 4096, 64 tokens: 8 row tiles of 128 -> K split 8 ways over grid.z (launch_gemm_type)
    const long M = 1024, K = 4096, N = 64, rs = K / 256 * 144;
    std::vector<uint8_t> w((size_t)M * rs); std::vector<float> x((size_t)N * K), c_ref((size_t)N * M), c((size_t)N * M);
    uint32_t seed = 12345u + (uint32_t)ctx->device;
    for (auto &b : w) b = (uint8_t)selftest_lcg(seed);
    for (long r = 0; r < M; ++r) for (long b = 0; b < K / 256; ++b) { uint8_t *blk = &w[(size_t)r * rs + b * 144]; blk[0] = 0x1f; blk[1] = 0x21; blk[2] = 0x1f; blk[3] = 0x1d; }      // d = 0.01, dmin = 0.005 (f16)
    for (auto &v : x) v = (float)(selftest_lcg(seed) & 0xffff) / 32768.f - 1.f;
    uint8_t *dW = nullptr; float *dX = nullptr, *dC = nullptr; void *dA = nullptr;
    // (b) split-KV attention: one token, 8 q heads on 2 kv heads of 128, 1024 keys (split form from 384 keys on)
    const long D = 128, NH = 8, NKV = 2, NK = 1024;
    const size_t q_b = (size_t)D * NH * 4, kv_b = (size_t)D * NK * NKV * 2, m_b = (size_t)NK * 2, o_b = (size_t)D * NH * 4, att_bytes = q_b + 2 * kv_b + m_b + o_b;
    std::vector<uint8_t> ah(att_bytes, 0); std::vector<float> o_ref((size_t)D * NH), o((size_t)D * NH);
    { float *qh = (float *)ah.data(); for (long i = 0; i < D * NH; ++i) qh[i] = (float)(selftest_lcg(seed) & 0xffff) / 32768.f - 1.f;
      uint16_t *kh = (uint16_t *)(ah.data() + q_b);      // f16 values in [-1, 1): sign | exponent 13 or 14 | random mantissa  (0.25 <= |v| < 1)
      for (size_t i = 0; i < kv_b; ++i) { const uint32_t r = selftest_lcg(seed); kh[i] = (uint16_t)(((r & 1) << 15) | ((13 + ((r >> 1) & 1)) << 10) | ((r >> 2) & 0x3ff)); } }
    auto cleanup = [&]() { if (dW) (void)hipFree(dW); if (dX) (void)hipFree(dX); if (dC) (void)hipFree(dC); if (dA) (void)hipFree(dA); ctx->selftest_unsplit = false; };
#define ST_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { cleanup(); ctx->handoff = prev; return cdna4_set_err(CDNA4_E_HIP, "hand-off self-test: %s failed: %s", #expr, hipGetErrorString(e_)); } } while (0)
#define ST_RC(expr) do { const int rc_ = (expr); if (rc_ != CDNA4_OK) { cleanup(); ctx->handoff = prev; return rc_; } } while (0)
    ST_TRY(hipMalloc((void **)&dW, w.size())); ST_TRY(hipMalloc((void **)&dX, x.size() * 4)); ST_TRY(hipMalloc((void **)&dC, c.size() * 4)); ST_TRY(hipMalloc(&dA, att_bytes));
    ST_TRY(hipMemcpy(dW, w.data(), w.size(), hipMemcpyHostToDevice)); ST_TRY(hipMemcpy(dX, x.data(), x.size() * 4, hipMemcpyHostToDevice)); ST_TRY(hipMemcpy(dA, ah.data(), att_bytes, hipMemcpyHostToDevice));
    cdna4_tensor tq, tk, tv, tm, to; memset(&tq, 0, sizeof(tq)); tk = tv = tm = to = tq;
    tq.data = dA; tq.type = T_F32; tq.ne[0] = D; tq.ne[1] = 1; tq.ne[2] = NH; tq.ne[3] = 1; tq.nb[0] = 4; tq.nb[1] = D * 4; tq.nb[2] = D * 4; tq.nb[3] = D * 4 * NH;
    tk.data = (char *)dA + q_b; tk.type = T_F16; tk.ne[0] = D; tk.ne[1] = NK; tk.ne[2] = NKV; tk.ne[3] = 1; tk.nb[0] = 2; tk.nb[1] = D * 2; tk.nb[2] = D * 2 * NK; tk.nb[3] = D * 2 * NK * NKV;
    tv = tk; tv.data = (char *)dA + q_b + kv_b;
    tm.data = (char *)dA + q_b + 2 * kv_b; tm.type = T_F16; tm.ne[0] = NK; tm.ne[1] = 1; tm.ne[2] = 1; tm.ne[3] = 1; tm.nb[0] = 2; tm.nb[1] = NK * 2; tm.nb[2] = NK * 2; tm.nb[3] = NK * 2;
    to.data = (char *)dA + q_b + 2 * kv_b + m_b; to.type = T_F32; to.ne[0] = D; to.ne[1] = NH; to.ne[2] = 1; to.ne[3] = 1; to.nb[0] = 4; to.nb[1] = D * 4; to.nb[2] = D * 4 * NH; to.nb[3] = D * 4 * NH;
    const float scale = 0.0883883f;
    // references: the unsplit forms
    ctx->selftest_unsplit = true;
    ST_RC(cdna4_mul_mat(ctx, M, N, K, T_Q4_K, dW, rs, 0, dX, K * 4, dC, M, st));
    ST_TRY(hipStreamSynchronize(st)); ST_TRY(hipMemcpy(c_ref.data(), dC, c.size() * 4, hipMemcpyDeviceToHost));
    ST_RC(cdna4_op_flash_attn(ctx, &tq, &tk, &tv, &tm, &to, scale, 0.f, 0.f, st));
    ST_TRY(hipStreamSynchronize(st)); ST_TRY(hipMemcpy(o_ref.data(), to.data, o_b, hipMemcpyDeviceToHost));
    ctx->selftest_unsplit = false;
    bool ok = !force_fail; int ksplit_seen = 0;
    for (int rep = 0; rep < 4 && ok; ++rep) {
        if (ctx->ws && !no_poison) ST_TRY(hipMemsetAsync(ctx->ws, 0xff, ctx->ws_bytes, st));
        ST_TRY(hipMemsetAsync(dC, 0xff, c.size() * 4, st));
        ST_RC(cdna4_mul_mat(ctx, M, N, K, T_Q4_K, dW, rs, 0, dX, K * 4, dC, M, st));
        { const char *p = strstr(g_launch_note, "ksplit="); if (p) ksplit_seen = std::max(ksplit_seen, atoi(p + 7)); }
        ST_TRY(hipStreamSynchronize(st)); ST_TRY(hipMemcpy(c.data(), dC, c.size() * 4, hipMemcpyDeviceToHost));
        ok = ok && selftest_close(c, c_ref, 2e-3);
        if (ctx->ws && !no_poison) ST_TRY(hipMemsetAsync(ctx->ws, 0xff, ctx->ws_bytes, st));
        ST_TRY(hipMemsetAsync(to.data, 0xff, o_b, st));
        ST_RC(cdna4_op_flash_attn(ctx, &tq, &tk, &tv, &tm, &to, scale, 0.f, 0.f, st));
        ST_TRY(hipStreamSynchronize(st)); ST_TRY(hipMemcpy(o.data(), to.data, o_b, hipMemcpyDeviceToHost));
        ok = ok && selftest_close(o, o_ref, 1e-4);
    }
#undef ST_TRY
#undef ST_RC
    cleanup();
    if (!ok) {
        ctx->handoff = 2;
        fprintf(stderr, "ggml-hip-cdna4: device %d: the fence-free in-launch hand-off %s its start-up self-test (split-K GEMM, K split %d ways, and split-KV attention against their unsplit forms): "
                        "this context uses the fenced forms (release / acquire around the arrival ticket)\n", ctx->device, force_fail ? "was told to fail (CDNA4_HANDOFF_SELFTEST=fail)" : "FAILED", ksplit_seen);
    }
    return CDNA4_OK;
}
