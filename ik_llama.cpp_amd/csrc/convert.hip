// convert.hip -- utility kernels of the library (convert.cuh) and their launch wrappers: L0 dequantizers, activation quantizers,
// _R4 repack, f16 slab conversion, MoE grouping, in-process peer reduce.  None is a hot kernel.
#include "api_internal.h"
#include "gemv.cuh"
#include "convert.cuh"
#include <algorithm>

template <int TYPE>
static int launch_dequant_t(const cdna4_context *ctx, const void *A, long strideA, long nrows, long K, void *dst, int dst_type, long dst_stride, hipStream_t st, float rs) {
    const long total = nrows * K; const int bs = 256; const unsigned grid = (unsigned)((total + bs - 1) / bs);
    if constexpr (!type_is_r4(TYPE)) {
        const int esz = dst_type == T_F32 ? 4 : 2;
        if (K % 8 == 0 && ((uintptr_t)dst % 16) == 0 && (dst_stride * esz) % 16 == 0) {
            const unsigned g8 = (unsigned)((total / 8 + bs - 1) / bs);
            if (dst_type == T_F32) hipLaunchKernelGGL((dequantize8_kernel<TYPE, float>), dim3(g8), dim3(bs), 0, st, (const uint8_t *)A, strideA, nrows, K, (float *)dst, dst_stride, ctx->grid, rs);
            else                   hipLaunchKernelGGL((dequantize8_kernel<TYPE, __half>), dim3(g8), dim3(bs), 0, st, (const uint8_t *)A, strideA, nrows, K, (__half *)dst, dst_stride, ctx->grid, rs);
            HIP_TRY(hipGetLastError());
            return CDNA4_OK;
        }
    }
    if (dst_type == T_F32) hipLaunchKernelGGL((dequantize_kernel<TYPE, float>), dim3(grid), dim3(bs), 0, st, (const uint8_t *)A, strideA, nrows, K, (float *)dst, dst_stride, ctx->grid, rs);
    else                   hipLaunchKernelGGL((dequantize_kernel<TYPE, __half>), dim3(grid), dim3(bs), 0, st, (const uint8_t *)A, strideA, nrows, K, (__half *)dst, dst_stride, ctx->grid, rs);
    HIP_TRY(hipGetLastError());
    return CDNA4_OK;
}
int cdna4_launch_dequant(const cdna4_context *ctx, int type, const void *A, long strideA, long nrows, long K, void *dst, int dst_type, long dst_stride, hipStream_t st, bool matmul_value) {
    const float rs = matmul_value ? kt_matmul_factor(type) : 1.0f;       // (trellis types only: the factor the reference's mat-mul kernels put on the row scale)
#define DQ(T) case T: return launch_dequant_t<T>(ctx, A, strideA, nrows, K, dst, dst_type, dst_stride, st, rs);
    switch (type) { DQ(T_Q4_K) DQ(T_Q5_K) DQ(T_Q6_K) DQ(T_IQ4_NL) DQ(T_IQ2_S) DQ(T_IQ3_S) DQ(T_Q4_0) DQ(T_Q8_0) DQ(T_IQ4_XS) DQ(T_Q5_0) DQ(T_IQ2_XXS) DQ(T_IQ2_XS) DQ(T_IQ3_XXS) DQ(T_Q4_1) DQ(T_Q5_1) DQ(T_Q6_0) DQ(T_Q2_K) DQ(T_Q3_K) DQ(T_IQ2_K) DQ(T_IQ3_K) DQ(T_IQ4_K) DQ(T_IQ5_K) DQ(T_IQ4_KS) DQ(T_IQ5_KS) DQ(T_IQ2_KS) DQ(T_IQ3_KS) DQ(T_IQ4_KSS) DQ(T_IQ2_KL) DQ(T_IQ6_K) DQ(T_IQ1_S) DQ(T_IQ1_M) DQ(T_MXFP4) DQ(T_IQ1_BN) DQ(T_IQ2_BN) DQ(T_IQ2_KT) DQ(T_IQ3_KT) DQ(T_IQ4_KT) DQ(T_IQ1_KT)
                    DQ(T_Q4_K_R4) DQ(T_Q5_K_R4) DQ(T_Q6_K_R4) DQ(T_IQ4_NL_R4) DQ(T_IQ2_S_R4) DQ(T_IQ3_S_R4) }
#undef DQ
    return set_err(CDNA4_E_UNSUPPORTED, "dequantize: type %d", type);
}

int cdna4_launch_quantize(int vdt, const void *B, long strideB, long nrows, long K, void *dst, long dst_row_bytes, hipStream_t st) {
    const long k8 = K / 8; const unsigned gy = (unsigned)std::min<long>(nrows, 32768), gz = (unsigned)((nrows + gy - 1) / gy);
    const dim3 grid((unsigned)((k8 + 255) / 256), gy, gz);
    if (vdt == T_Q8_2_X4)    hipLaunchKernelGGL((quantize_rows_kernel<T_Q8_2_X4>), grid, dim3(256), 0, st, (const uint8_t *)B, strideB, K, nrows, (uint8_t *)dst, dst_row_bytes);
    else if (vdt == T_Q8_K)  hipLaunchKernelGGL((quantize_rows_kernel<T_Q8_K>),    grid, dim3(256), 0, st, (const uint8_t *)B, strideB, K, nrows, (uint8_t *)dst, dst_row_bytes);
    else                     hipLaunchKernelGGL((quantize_rows_kernel<T_Q8_K32>),  grid, dim3(256), 0, st, (const uint8_t *)B, strideB, K, nrows, (uint8_t *)dst, dst_row_bytes);
    HIP_TRY(hipGetLastError());
    return CDNA4_OK;
}

template <bool TO_R4>
static int launch_repack_t(int base, const void *src, void *dst, long nrows, long K, long stride, hipStream_t st) {
    const long nthreads = nrows * (K / type_block_elems(base)); const unsigned grid = (unsigned)((nthreads + 127) / 128);
#define RP(T) case T: hipLaunchKernelGGL((repack_r4_kernel<T, TO_R4>), dim3(grid), dim3(128), 0, st, (const uint8_t *)src, (uint8_t *)dst, nrows, K, stride); break;
    switch (base) { RP(T_Q4_K) RP(T_Q5_K) RP(T_Q6_K) RP(T_IQ4_NL) RP(T_IQ2_S) RP(T_IQ3_S) default: return set_err(CDNA4_E_UNSUPPORTED, "repack: type %d", base); }
#undef RP
    HIP_TRY(hipGetLastError());
    return CDNA4_OK;
}
int cdna4_launch_repack(bool to_r4, int base, const void *src, void *dst, long nrows, long K, long stride, hipStream_t st) {
    return to_r4 ? launch_repack_t<true>(base, src, dst, nrows, K, stride, st) : launch_repack_t<false>(base, src, dst, nrows, K, stride, st);
}

// dst = f16 slab image of `xrows` rows (rows >= nrows are zero), xscale[xrows] = per-row range-guard scale (convert.cuh)
int cdna4_launch_f32_to_f16_slab(const void *B, long strideB, long K, long nrows, void *dst, long xrows, float *xscale, hipStream_t st) {
    hipLaunchKernelGGL(rows_to_f16_slab_kernel, dim3((unsigned)xrows), dim3(256), 0, st, (const uint8_t *)B, strideB, 1, 0L, 0L, 1, (const int *)nullptr, 0L, K, nrows, (__half *)dst, xrows, xscale);
    HIP_TRY(hipGetLastError());
    return CDNA4_OK;
}
int cdna4_launch_moe_gather_f16(const void *B, int n_b, long nb11, long nb12, int n_used, const int *pairs_sorted, long rows_pad, long pairs, long K, void *X, float *xscale, hipStream_t st) {
    hipLaunchKernelGGL(rows_to_f16_slab_kernel, dim3((unsigned)rows_pad), dim3(256), 0, st, (const uint8_t *)B, 0L, n_b, nb11, nb12, n_used, pairs_sorted, pairs, K, 0L, (__half *)X, rows_pad, xscale);
    HIP_TRY(hipGetLastError());
    return CDNA4_OK;
}
int cdna4_launch_moe_sort(const int32_t *ids, long ids_nb1, int n_tokens, int n_used, int n_expert, int BN, int max_tiles, int *pairs_sorted, int *tiles,
                          float *C, long nb1, long nb2, int M, hipStream_t st) {
    hipLaunchKernelGGL(moe_sort_kernel, dim3(1), dim3(1024), (size_t)(3 * n_expert + 2) * sizeof(int) + 1024 * sizeof(unsigned long long), st, ids, ids_nb1, n_tokens, n_used, n_expert, BN, max_tiles,
                       pairs_sorted, tiles, C, nb1, nb2, M);
    HIP_TRY(hipGetLastError());
    return CDNA4_OK;
}
int cdna4_launch_iq_tables_init(const uint16_t *packed, uint8_t *out) {
    hipLaunchKernelGGL(iq_tables_init_kernel, dim3(1), dim3(256), 0, 0, packed, out);
    HIP_TRY(hipGetLastError());
    return CDNA4_OK;
}

int cdna4_launch_reduce_peers(int num_cu, void *const *bufs, int n, unsigned partial_mask, long count, int dtype, int slice, int n_slices, hipStream_t st) {
    ReducePeersArgs a; memset(&a, 0, sizeof(a)); a.n = n; a.partial_mask = partial_mask; a.count = count;
    for (int j = 0; j < n; ++j) a.buf[j] = bufs[j];
    if (dtype == T_Q8_0) {           // `count` elements = count / 32 blocks; slices in whole blocks
        const long nb = count / 32, slb = (nb + n_slices - 1) / n_slices;
        a.v_begin = std::min<long>(nb, (long)slice * slb); a.v_end = std::min<long>(nb, a.v_begin + slb); a.tail = 0;
        const long nmine = a.v_end - a.v_begin; const unsigned gq = (unsigned)std::max<long>(1, std::min<long>((nmine + 63) / 64, 8L * num_cu));
        hipLaunchKernelGGL(reduce_peers_q8_0_kernel, dim3(gq), dim3(64), 0, st, a);
        HIP_TRY(hipGetLastError());
        return CDNA4_OK;
    }
    const long nv_all = count / (dtype == T_F32 ? 4 : 8), sl = (nv_all + n_slices - 1) / n_slices;
    a.v_begin = std::min<long>(nv_all, (long)slice * sl); a.v_end = std::min<long>(nv_all, a.v_begin + sl); a.tail = slice == n_slices - 1;
    const long nvec = a.v_end - a.v_begin; const unsigned grid = (unsigned)std::max<long>(1, std::min<long>((nvec + 255) / 256, 4L * num_cu));
    switch (dtype) {
        case T_F32:  hipLaunchKernelGGL(reduce_peers_kernel<float>, dim3(grid), dim3(256), 0, st, a); break;
        case T_F16:  hipLaunchKernelGGL(reduce_peers_kernel<_Float16>, dim3(grid), dim3(256), 0, st, a); break;
        case T_BF16: hipLaunchKernelGGL(reduce_peers_kernel<__bf16>, dim3(grid), dim3(256), 0, st, a); break;
        default: return set_err(CDNA4_E_UNSUPPORTED, "peer-reduce dtype %d unsupported", dtype);
    }
    HIP_TRY(hipGetLastError());
    return CDNA4_OK;
}

int cdna4_launch_get_rows(const cdna4_context *ctx, const cdna4_tensor *src, const cdna4_tensor *ids, const cdna4_tensor *dst, hipStream_t st) {
    long total = 1; for (int i = 0; i < 4; ++i) total *= dst->ne[i];
    const unsigned grid = (unsigned)std::min<long>((total + 255) / 256, 16L * ctx->num_cu);
#define GR(T) case T: hipLaunchKernelGGL(get_rows_kernel<T>, dim3(grid), dim3(256), 0, st, td_of(src), td_of(ids), td_of(dst), ctx->grid, total); break;
    switch (src->type) { GR(T_F32) GR(T_F16) GR(T_Q4_K) GR(T_Q5_K) GR(T_Q6_K) GR(T_IQ4_NL) GR(T_IQ2_S) GR(T_IQ3_S) GR(T_Q4_0) GR(T_Q8_0) GR(T_IQ4_XS) GR(T_Q5_0) GR(T_IQ2_XXS) GR(T_IQ2_XS) GR(T_IQ3_XXS) GR(T_Q4_1) GR(T_Q5_1) GR(T_Q6_0) GR(T_Q2_K) GR(T_Q3_K) GR(T_IQ2_K) GR(T_IQ3_K) GR(T_IQ4_K) GR(T_IQ5_K) GR(T_IQ4_KS) GR(T_IQ5_KS) GR(T_IQ2_KS) GR(T_IQ3_KS) GR(T_IQ4_KSS) GR(T_IQ2_KL) GR(T_IQ6_K) GR(T_IQ1_S) GR(T_IQ1_M) GR(T_MXFP4) GR(T_IQ2_KT) GR(T_IQ3_KT) GR(T_IQ4_KT) GR(T_IQ1_KT)
        default: return set_err(CDNA4_E_UNSUPPORTED, "get_rows: source type %d", src->type); }
#undef GR
    HIP_TRY(hipGetLastError());
    return CDNA4_OK;
}
