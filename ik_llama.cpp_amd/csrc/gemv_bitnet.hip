// gemv_bitnet.hip -- BitNet weights (IQ1_BN, IQ2_BN: SURVEY 8 f3) on gfx950: the Q8_K64 activation quantizer and the decode mat-vec.
//
// What it computes: the reference's mul_mat_iq1bn_q8_K64 / mul_mat_iq2bn_q8_K64 (iqk_gemm_1bit.cpp:1247-1447) with quantize_row_q8_K64 activations
// (iqk_quantize.cpp:586-690), restated for a 64-wide wavefront and pinned to oracle/iqk_oracle.c (itself pinned bit-exactly to the reference kernels,
// tests/test_oracle_vs_ref.py):
//   * weights are ternary, stored as u = value + 1 in {0, 1, 2}: IQ2_BN 2 bits each (16 bytes per 64: byte r holds elements r, r + 16, r + 32, r + 48), IQ1_BN base-3
//     digits (13 bytes per 64: five per byte, (3 ((b m) & 255)) >> 8 with m in {81, 27, 9, 3, 1}; the 16th element of every 16 comes from byte 12); one f32 (IQ2_BN)
//     or f16 (IQ1_BN) scale per ROW in front of the row's blocks;
//   * activations: ONE scale per class c = (j mod 16) / 4 for the whole row, d_c = max_c / 127, q = rne(x / d_c); row image {float d[4]; float d_c sum(q_c) [4]; int8 q[K]};
//   * result = d_row * hsum4( fma(d_c, (float)sum_c(u q), -d_c sum(q_c)) ): the four class sums are EXACT int32 sums over the whole row, so the device result equals
//     the oracle's bit for bit (same integers, same four fma, same hsum order (f0 + f2) + (f1 + f3)).
// MI355X mapping: a wave owns a row, lane l owns the 64-weight blocks l, l + 64, ... (one 16-byte / 13-byte piece each: HBM-bound, read once); the quantized activation
// rows of up to 8 columns sit in LDS; class sums meet through DPP-free integer butterflies once per row.  BitNet models are small (2 bits per weight): this is a
// correctness-first unit outside gemv_body's template family -- plain MUL_MAT only (prompts go through the f16 route: de-quantize + f16 MFMA GEMM).
#include "api_internal.h"
#include "cdna4_common.cuh"
#include <algorithm>

// ---- Q8_K64 -------------------------------------------------------------------------------------------------
// one workgroup per activation row; thread t only ever touches 4-element groups g = t, t + 256, ... whose class is g mod 4 = t mod 4
__global__ void __launch_bounds__(256) quantize_q8_k64_kernel(const uint8_t *B, long strideB, long K, uint8_t *dst, long dst_row_bytes) {
    __shared__ float s_max[256]; __shared__ int s_sum[256]; __shared__ float s_d[4];
    const long row = blockIdx.x; const int t = threadIdx.x; const long ng = K >> 2;
    const float *x = reinterpret_cast<const float *>(B + row * strideB);
    float mx = 0.f;
    for (long g = t; g < ng; g += 256) { const float4 v = *reinterpret_cast<const float4 *>(x + 4 * g); mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)))); }
    s_max[t] = mx; __syncthreads();
    if (t < 4) { float m = 0.f; for (int i = t; i < 256; i += 4) m = fmaxf(m, s_max[i]); s_d[t] = m / 127; }
    __syncthreads();
    const float d = s_d[t & 3], id = d > 0 ? 1 / d : 0.f;
    uint8_t *out = dst + row * dst_row_bytes; int8_t *qs = reinterpret_cast<int8_t *>(out + 32);
    int sum = 0;
    for (long g = t; g < ng; g += 256) {
        const float4 v = *reinterpret_cast<const float4 *>(x + 4 * g);
        int q0 = (int)rintf(id * v.x), q1 = (int)rintf(id * v.y), q2 = (int)rintf(id * v.z), q3 = (int)rintf(id * v.w);
        q0 = max(-128, min(127, q0)); q1 = max(-128, min(127, q1)); q2 = max(-128, min(127, q2)); q3 = max(-128, min(127, q3));
        sum += q0 + q1 + q2 + q3;
        *reinterpret_cast<uint32_t *>(qs + 4 * g) = (uint32_t)(q0 & 255) | ((uint32_t)(q1 & 255) << 8) | ((uint32_t)(q2 & 255) << 16) | ((uint32_t)(q3 & 255) << 24);
    }
    s_sum[t] = sum; __syncthreads();
    if (t < 4) { int s = 0; for (int i = t; i < 256; i += 4) s += s_sum[i]; reinterpret_cast<float *>(out)[t] = s_d[t]; reinterpret_cast<float *>(out)[4 + t] = s_d[t] * (float)s; }
}
int cdna4_launch_quantize_q8_k64(const void *B, long strideB, long nrows, long K, void *dst, long dst_row_bytes, hipStream_t st) {
    hipLaunchKernelGGL(quantize_q8_k64_kernel, dim3((unsigned)nrows), dim3(256), 0, st, (const uint8_t *)B, strideB, K, (uint8_t *)dst, dst_row_bytes);
    HIP_TRY(hipGetLastError());
    return CDNA4_OK;
}

// ---- decode mat-vec ---------------------------------------------------------------------------------------------
// class sums of one 64-weight block against one column: s[c] += sum over the 16 elements of class c of u * q
template <int TYPE>
__device__ __forceinline__ void bn_block_dot(const uint8_t *b, const uint32_t *y /* 16 dwords: the block's 64 int8 */, int (&s)[4]) {
    if (TYPE == T_IQ2_BN) {                           // dword c of the block = bytes 4c .. 4c + 3 = elements 16 i + 4c + {0..3}, i = bit pair
        const u128_a2 w = *reinterpret_cast<const u128_a2 *>(b); const uint32_t dw[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int i = 0; i < 4; ++i) s[c] = dot4((dw[c] >> (2 * i)) & 0x03030303u, y[4 * i + c], s[c]);
        }
    } else {                                          // IQ1_BN: element 16 i + r: r < 15 -> byte 3 i + r / 5, multiplier {81, 27, 9, 3, 1}[r % 5]; r = 15 -> byte 12, multiplier of i
        uint32_t bytes[13];
#pragma unroll
        for (int i = 0; i < 13; ++i) bytes[i] = b[i];
        constexpr uint32_t km[5] = {81, 27, 9, 3, 1};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                uint32_t w = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * c + e;
                    const uint32_t v = r < 15 ? (bytes[3 * i + r / 5] * km[r % 5]) & 255u : (bytes[12] * km[i]) & 255u;
                    w |= ((3u * v) >> 8) << (8 * e);
                }
                s[c] = dot4(w, y[4 * i + c], s[c]);
            }
        }
    }
}

// grid.x workgroups of 4 waves stride over the rows; LDS: [NCOLS][K] int8 + [NCOLS][8] floats
template <int TYPE, int NCOLS>
__global__ void __launch_bounds__(256) gemv_bn_kernel(const uint8_t *A, long strideA, int M, int K, const uint8_t *Xq, long xq_stride, float *C, long stride_C) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    int8_t *yq = reinterpret_cast<int8_t *>(smem); float *yd = reinterpret_cast<float *>(smem + (size_t)NCOLS * K);
    for (int i = threadIdx.x; i < NCOLS * (K >> 4); i += 256) { const int c = i / (K >> 4), j = i - c * (K >> 4);
        reinterpret_cast<uint4 *>(yq + (size_t)c * K)[j] = *reinterpret_cast<const uint4 *>(Xq + c * xq_stride + 32 + 16 * (long)j); }
    for (int i = threadIdx.x; i < NCOLS * 8; i += 256) yd[i] = reinterpret_cast<const float *>(Xq + (i >> 3) * xq_stride)[i & 7];
    __syncthreads();
    constexpr int TS = type_block_bytes(TYPE), META = type_row_meta(TYPE);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, U = K >> 6;
    for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
        const uint8_t *rp = A + (long)row * strideA;
        int s[NCOLS][4];
#pragma unroll
        for (int c = 0; c < NCOLS; ++c) { s[c][0] = s[c][1] = s[c][2] = s[c][3] = 0; }
        for (int u = lane; u < U; u += 64) {
            const uint8_t *b = rp + META + (long)u * TS;
#pragma unroll
            for (int c = 0; c < NCOLS; ++c) {
                uint32_t y[16];
                const uint4 *yp = reinterpret_cast<const uint4 *>(yq + (size_t)c * K + 64 * u);
#pragma unroll
                for (int i = 0; i < 4; ++i) { const uint4 v = yp[i]; y[4 * i] = v.x; y[4 * i + 1] = v.y; y[4 * i + 2] = v.z; y[4 * i + 3] = v.w; }
                bn_block_dot<TYPE>(b, y, s[c]);
            }
        }
#pragma unroll
        for (int c = 0; c < NCOLS; ++c) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) s[c][k] += __shfl_xor(s[c][k], off, 64);
            }
        }
        if (lane == 0) {
            const float d_row = TYPE == T_IQ2_BN ? __uint_as_float(reinterpret_cast<const u32_a2 *>(rp)->v) : half_bits_to_float(ld16(rp));
#pragma unroll
            for (int c = 0; c < NCOLS; ++c) {
                float f[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) f[k] = fmaf(yd[8 * c + k], (float)s[c][k], -yd[8 * c + 4 + k]);
                C[(long)c * stride_C + row] = d_row * ((f[0] + f[2]) + (f[1] + f[3]));
            }
        }
    }
}

template <int TYPE>
static int launch_bn(const cdna4_context *ctx, const void *A, long strideA, long M, long K, const void *Xq, long xq_stride, int ncols, float *C, long stride_C, hipStream_t st) {
    const size_t lds = (size_t)ncols * K + (size_t)ncols * 32;
    const unsigned grid = (unsigned)std::max<long>(1, std::min<long>((M + 3) / 4, 4L * ctx->num_cu));
#define BN_LAUNCH(N_) do { if (lds > 64 * 1024) { const int rc = cdna4_opt_in_lds((const void *)gemv_bn_kernel<TYPE, N_>); if (rc) return rc; }                          \
        hipLaunchKernelGGL((gemv_bn_kernel<TYPE, N_>), dim3(grid), dim3(256), lds, st, (const uint8_t *)A, strideA, (int)M, (int)K, (const uint8_t *)Xq, xq_stride, C, stride_C); } while (0)
    switch (ncols) { case 1: BN_LAUNCH(1); break; case 2: BN_LAUNCH(2); break; case 4: BN_LAUNCH(4); break; default: return set_err(CDNA4_E_INVALID, "bitnet gemv: %d columns", ncols); }
#undef BN_LAUNCH
    HIP_TRY(hipGetLastError());
    return CDNA4_OK;
}
// ncols in {1, 2, 4}; Xq = Q8_K64 rows (32 + K bytes each)
int cdna4_launch_gemv_bitnet(const cdna4_context *ctx, int type, const void *A, long strideA, long M, long K, const void *Xq, long xq_stride, int ncols, float *C, long stride_C, hipStream_t st) {
    if (type == T_IQ2_BN) return launch_bn<T_IQ2_BN>(ctx, A, strideA, M, K, Xq, xq_stride, ncols, C, stride_C, st);
    if (type == T_IQ1_BN) return launch_bn<T_IQ1_BN>(ctx, A, strideA, M, K, Xq, xq_stride, ncols, C, stride_C, st);
    return set_err(CDNA4_E_UNSUPPORTED, "bitnet gemv: type %d", type);
}
