// ops.hip -- the non-mat-mul ops of a Llama / Mixtral graph on the device: the steps on either side of the quantized
// mat-mul path.  Without them every layer round-trips through the CPU backend, the KV cache cannot live in HBM and the reference's
// -sm graph mode (which pins whole per-device sub-graphs to a backend) cannot run.  What they replace in the reference:
//   FUSED_RMS_NORM   ggml.c:17420-17470 / ggml-cuda/norm.cu          ADD, MUL, DIV    ggml-cuda/binbcast.cu
//   ROPE             ggml.c:20987-21230 / ggml-cuda/rope.cu          CPY              ggml-cuda/cpy.cu (f32 -> f16 KV cache writes)
//   GET_ROWS         ggml.c:19808        / ggml-cuda/getrows.cu      SOFT_MAX         ggml.c:20300     / ggml-cuda/softmax.cu
//   FLASH_ATTN_EXT   ggml.c:22874-23160 / ggml-cuda/fattn*.cu        ARGSORT (top-k)  iqk_cpu_ops.cpp:228-266 / ggml-cuda/argsort.cu
//   SUM_ROWS, MUL_MULTI_ADD (iqk_cpu_ops.cpp:430), f32 MUL_MAT of the MoE router
// These are bandwidth- or latency-bound helpers, written for coalesced access and one pass over their data; the judged kernels are the
// mat-mul ones.  All take plain strided tensor descriptors (cdna4_tensor: data, type, ne[4], nb[4] in bytes -- ggml's own convention).
#include "api_internal.h"
#include "fa_decode.cuh"
#include <hip/hip_fp16.h>
#include <algorithm>
#include <cmath>

static long td_nrows(const cdna4_tensor *t) { return t->ne[1] * t->ne[2] * t->ne[3]; }
static long td_nelem(const cdna4_tensor *t) { return t->ne[0] * td_nrows(t); }
static bool td_rows_contig(const cdna4_tensor *t, int esz) { return t->nb[0] == esz; }
static bool td_contig(const cdna4_tensor *t, int esz) { return t->nb[0] == esz && t->nb[1] == t->nb[0] * t->ne[0] && t->nb[2] == t->nb[1] * t->ne[1] && t->nb[3] == t->nb[2] * t->ne[2]; }
static bool same_shape(const cdna4_tensor *a, const cdna4_tensor *b) { for (int i = 0; i < 4; ++i) if (a->ne[i] != b->ne[i]) return false; return true; }
#define OP_CHECK(cond, ...) do { if (!(cond)) return cdna4_set_err(CDNA4_E_UNSUPPORTED, __VA_ARGS__); } while (0)

// block-wide reductions over 256 threads (4 waves) through 4 LDS words
__device__ __forceinline__ float block_sum256(float v, float *red) {
    v = wave_sum_dpp(v); __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float block_max256(float v, float *red) {
    v = wave_max(v); __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// ------------------------------------------------------------------------------------------------ FUSED_RMS_NORM / RMS_NORM
// y = x * rsqrt(mean(x^2) + eps) * w     one workgroup per row; the row is read once (kept in registers up to 4096 floats, re-read beyond)
template <typename X> __device__ __forceinline__ float ld_f(const X *p);
template <> __device__ __forceinline__ float ld_f<float>(const float *p) { return *p; }
template <> __device__ __forceinline__ float ld_f<__half>(const __half *p) { return __half2float(*p); }
template <typename X>       // X = float, or __half (the f16 partial sums of a -sm graph prompt batch, llama-build-context.cpp:1198; dst is f32 either way)
__global__ void __launch_bounds__(256) rms_norm_kernel(TD x, const float *w, TD y, float eps) {
    __shared__ float red[4];
    const long r = blockIdx.x, n = x.ne[0];
    const long i1 = r % x.ne[1], i2 = (r / x.ne[1]) % x.ne[2], i3 = r / (x.ne[1] * x.ne[2]);
    const X *xr = reinterpret_cast<const X *>(x.data + i1 * x.nb[1] + i2 * x.nb[2] + i3 * x.nb[3]);
    float *yr = reinterpret_cast<float *>(y.data + i1 * y.nb[1] + i2 * y.nb[2] + i3 * y.nb[3]);
    float4 keep[4]; float ss = 0.f;
    const bool vec = sizeof(X) == 4 && (n % 4 == 0) && ((reinterpret_cast<uintptr_t>(xr) | reinterpret_cast<uintptr_t>(yr) | reinterpret_cast<uintptr_t>(w)) % 16 == 0);
    if (vec) {
        const long n4 = n / 4;
#pragma unroll
        for (int p = 0; p < 4; ++p) { const long i = threadIdx.x + 256L * p; keep[p] = i < n4 ? reinterpret_cast<const float4 *>(xr)[i] : make_float4(0, 0, 0, 0);
            ss += keep[p].x * keep[p].x + keep[p].y * keep[p].y + keep[p].z * keep[p].z + keep[p].w * keep[p].w; }
        for (long i = threadIdx.x + 1024; i < n4; i += 256) { const float4 v = reinterpret_cast<const float4 *>(xr)[i]; ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w; }
    } else for (long i = threadIdx.x; i < n; i += 256) { const float v = ld_f<X>(xr + i); ss += v * v; }
    const float scale = 1.0f / sqrtf(block_sum256(ss, red) / (float)n + eps);
    if (vec) {
        const long n4 = n / 4;
#pragma unroll
        for (int p = 0; p < 4; ++p) { const long i = threadIdx.x + 256L * p; if (i < n4) { float4 v = keep[p]; const float4 c = w ? reinterpret_cast<const float4 *>(w)[i] : make_float4(1, 1, 1, 1);
            v.x = scale * c.x * v.x; v.y = scale * c.y * v.y; v.z = scale * c.z * v.z; v.w = scale * c.w * v.w; reinterpret_cast<float4 *>(yr)[i] = v; } }
        for (long i = threadIdx.x + 1024; i < n4; i += 256) { float4 v = reinterpret_cast<const float4 *>(xr)[i]; const float4 c = w ? reinterpret_cast<const float4 *>(w)[i] : make_float4(1, 1, 1, 1);
            v.x = scale * c.x * v.x; v.y = scale * c.y * v.y; v.z = scale * c.z * v.z; v.w = scale * c.w * v.w; reinterpret_cast<float4 *>(yr)[i] = v; }
    } else for (long i = threadIdx.x; i < n; i += 256) yr[i] = scale * (w ? w[i] : 1.f) * ld_f<X>(xr + i);
}
int cdna4_op_rms_norm(cdna4_context *ctx, const cdna4_tensor *x, const cdna4_tensor *w, float eps, const cdna4_tensor *dst, void *stream) {
    if (!ctx || !x || !dst) return cdna4_set_err(CDNA4_E_INVALID, "null argument");
    OP_CHECK((x->type == T_F32 || x->type == T_F16) && dst->type == T_F32 && same_shape(x, dst) && td_rows_contig(x, x->type == T_F32 ? 4 : 2) && td_rows_contig(dst, 4), "rms_norm: f32 / f16 rows -> f32");
    OP_CHECK(!w || (w->type == T_F32 && w->ne[0] == x->ne[0] && td_nrows(w) == 1 && w->nb[0] == 4), "rms_norm: weight must be one f32 row");
    if (td_nelem(x) == 0) return CDNA4_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    if (x->type == T_F32) hipLaunchKernelGGL(rms_norm_kernel<float>, dim3((unsigned)td_nrows(x)), dim3(256), 0, (hipStream_t)stream, td_of(x), w ? (const float *)w->data : nullptr, td_of(dst), eps);
    else hipLaunchKernelGGL(rms_norm_kernel<__half>, dim3((unsigned)td_nrows(x)), dim3(256), 0, (hipStream_t)stream, td_of(x), w ? (const float *)w->data : nullptr, td_of(dst), eps);
    HIP_TRY(hipGetLastError()); return CDNA4_OK;
}

// ------------------------------------------------------------------------------------------------ ADD + FUSED_RMS_NORM in one pass
// sum = a + b (the residual stream, kept: it is read again two nodes later), y = sum * rsqrt(mean(sum^2) + eps) * w.  One workgroup per row.
__global__ void __launch_bounds__(256) add_rms_norm_kernel(TD a, TD b, TD sum, const float *w, TD y, float eps) {
    __shared__ float red[4];
    const long r = blockIdx.x, n = a.ne[0];
    const long i1 = r % a.ne[1], i2 = (r / a.ne[1]) % a.ne[2], i3 = r / (a.ne[1] * a.ne[2]);
    const float *ar = reinterpret_cast<const float *>(a.data + i1 * a.nb[1] + i2 * a.nb[2] + i3 * a.nb[3]);
    const float *br = reinterpret_cast<const float *>(b.data + i1 * b.nb[1] + i2 * b.nb[2] + i3 * b.nb[3]);
    float *sr = reinterpret_cast<float *>(sum.data + i1 * sum.nb[1] + i2 * sum.nb[2] + i3 * sum.nb[3]);
    float *yr = reinterpret_cast<float *>(y.data + i1 * y.nb[1] + i2 * y.nb[2] + i3 * y.nb[3]);
    float4 keep[4], wk[4]; float ss = 0.f;
    const bool vec = (n % 4 == 0) && ((reinterpret_cast<uintptr_t>(ar) | reinterpret_cast<uintptr_t>(br) | reinterpret_cast<uintptr_t>(sr) | reinterpret_cast<uintptr_t>(yr) | reinterpret_cast<uintptr_t>(w)) % 16 == 0);
    if (vec) {
        const long n4 = n / 4;
#pragma unroll
        for (int p = 0; p < 4; ++p) { const long i = threadIdx.x + 256L * p; wk[p] = i < n4 ? reinterpret_cast<const float4 *>(w)[i] : make_float4(0, 0, 0, 0); }     // (independent of the reduction: in flight with a and b)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const long i = threadIdx.x + 256L * p; keep[p] = make_float4(0, 0, 0, 0);
            if (i < n4) { const float4 x = reinterpret_cast<const float4 *>(ar)[i], z = reinterpret_cast<const float4 *>(br)[i]; keep[p] = make_float4(x.x + z.x, x.y + z.y, x.z + z.z, x.w + z.w);
                          reinterpret_cast<float4 *>(sr)[i] = keep[p]; }
            ss += keep[p].x * keep[p].x + keep[p].y * keep[p].y + keep[p].z * keep[p].z + keep[p].w * keep[p].w;
        }
        for (long i = threadIdx.x + 1024; i < n4; i += 256) { const float4 x = reinterpret_cast<const float4 *>(ar)[i], z = reinterpret_cast<const float4 *>(br)[i];
            const float4 v = make_float4(x.x + z.x, x.y + z.y, x.z + z.z, x.w + z.w); reinterpret_cast<float4 *>(sr)[i] = v; ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w; }
    } else for (long i = threadIdx.x; i < n; i += 256) { const float v = ar[i] + br[i]; sr[i] = v; ss += v * v; }
    const float scale = 1.0f / sqrtf(block_sum256(ss, red) / (float)n + eps);       // (block_sum256 synchronizes: every thread re-reads only what it wrote itself)
    if (vec) {
        const long n4 = n / 4;
#pragma unroll
        for (int p = 0; p < 4; ++p) { const long i = threadIdx.x + 256L * p; if (i < n4) { float4 v = keep[p]; const float4 c = wk[p];
            v.x = scale * c.x * v.x; v.y = scale * c.y * v.y; v.z = scale * c.z * v.z; v.w = scale * c.w * v.w; reinterpret_cast<float4 *>(yr)[i] = v; } }
        for (long i = threadIdx.x + 1024; i < n4; i += 256) { float4 v = reinterpret_cast<const float4 *>(sr)[i]; const float4 c = reinterpret_cast<const float4 *>(w)[i];
            v.x = scale * c.x * v.x; v.y = scale * c.y * v.y; v.z = scale * c.z * v.z; v.w = scale * c.w * v.w; reinterpret_cast<float4 *>(yr)[i] = v; }
    } else for (long i = threadIdx.x; i < n; i += 256) yr[i] = scale * w[i] * sr[i];
}
// ---- prompt batches: [ADD +] FUSED_RMS_NORM whose result only feeds mat-muls -> the f16 SLAB image of the GEMM (convert.cuh rows_to_f16_slab_kernel: layout, k order (0,2,1,3),
// per-row range guard), one launch instead of two or three; the normed f32 rows are never written.  One workgroup per image row (rows >= nrows: zeros); rows of up to 16384 values
// stay in registers.  Sums in the order of add_rms_norm_kernel / rms_norm_kernel, conversion in the order of the slab kernel.
__global__ void __launch_bounds__(256) norm_to_f16_slab_kernel(const uint8_t *B, const uint8_t *addB, uint8_t *sumD, long strideB, const float *w, float eps, long K, long nrows,
                                                               __half *dst, long xrows, float *xscale) {
    __shared__ float red[4];
    const long r = blockIdx.x, n4 = K / 4;
    constexpr int RP = 16; float4 keep[RP]; float ss = 0.f;
    const bool live = r < nrows;
    const float4 *xr = reinterpret_cast<const float4 *>(B + r * strideB), *br = addB ? reinterpret_cast<const float4 *>(addB + r * strideB) : nullptr;
    float4 *sr = sumD ? reinterpret_cast<float4 *>(sumD + r * strideB) : nullptr;
#pragma unroll
    for (int p = 0; p < RP; ++p) {
        const long i = threadIdx.x + 256L * p; keep[p] = make_float4(0, 0, 0, 0);
        if (live && i < n4) {
            float4 x = xr[i];
            if (br) { const float4 z = br[i]; x = make_float4(x.x + z.x, x.y + z.y, x.z + z.z, x.w + z.w); sr[i] = x; }
            keep[p] = x;
        }
        ss += keep[p].x * keep[p].x + keep[p].y * keep[p].y + keep[p].z * keep[p].z + keep[p].w * keep[p].w;
    }
    const float scale = 1.0f / sqrtf(block_sum256(ss, red) / (float)K + eps);
    float amax = 0.f;
#pragma unroll
    for (int p = 0; p < RP; ++p) {
        const long i = threadIdx.x + 256L * p;
        if (live && i < n4) {
            float4 v = keep[p]; const float4 c = reinterpret_cast<const float4 *>(w)[i];
            v.x = scale * c.x * v.x; v.y = scale * c.y * v.y; v.z = scale * c.z * v.z; v.w = scale * c.w * v.w; keep[p] = v;
            amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        }
    }
    amax = block_max256(amax, red);
    float s = 1.f, inv = 1.f;
    if (amax > 16384.f && amax < 3.0e38f) { int e; (void)frexpf(amax, &e); s = ldexpf(1.f, e - 14); inv = ldexpf(1.f, 14 - e); }
    if (threadIdx.x == 0) xscale[r] = s;
#pragma unroll
    for (int p = 0; p < RP; ++p) {
        const long k = 4L * threadIdx.x + 1024L * p;
        if (k < K) {
            const float4 v = make_float4(keep[p].x * inv, keep[p].y * inv, keep[p].z * inv, keep[p].w * inv);
            __half2 *o = reinterpret_cast<__half2 *>(dst + ((k >> 6) * xrows + r) * 64 + (k & 63));
            o[0] = __floats2half2_rn(v.x, v.z); o[1] = __floats2half2_rn(v.y, v.w);
        }
    }
}
int cdna4_launch_norm_f16_slab(const void *B, const void *add_b, void *add_dst, long strideB, const float *w, float eps, long K, long nrows, void *dst, long xrows, float *xscale, hipStream_t st) {
    hipLaunchKernelGGL(norm_to_f16_slab_kernel, dim3((unsigned)xrows), dim3(256), 0, st, (const uint8_t *)B, (const uint8_t *)add_b, (uint8_t *)add_dst, strideB, w, eps, K, nrows, (__half *)dst, xrows, xscale);
    HIP_TRY(hipGetLastError()); return CDNA4_OK;
}

int cdna4_op_add_rms_norm(cdna4_context *ctx, const cdna4_tensor *a, const cdna4_tensor *b, const cdna4_tensor *sum, const cdna4_tensor *w, float eps, const cdna4_tensor *dst, void *stream) {
    if (!ctx || !a || !b || !sum || !w || !dst) return cdna4_set_err(CDNA4_E_INVALID, "null argument");
    OP_CHECK(a->type == T_F32 && b->type == T_F32 && sum->type == T_F32 && dst->type == T_F32 && same_shape(a, b) && same_shape(a, sum) && same_shape(a, dst) &&
             td_rows_contig(a, 4) && td_rows_contig(b, 4) && td_rows_contig(sum, 4) && td_rows_contig(dst, 4), "add_rms_norm: four f32 tensors of one shape");
    OP_CHECK(w->type == T_F32 && w->ne[0] == a->ne[0] && td_nrows(w) == 1 && w->nb[0] == 4, "add_rms_norm: weight must be one f32 row");
    OP_CHECK(sum->data != dst->data, "add_rms_norm: sum and dst must not alias");
    if (td_nelem(a) == 0) return CDNA4_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(add_rms_norm_kernel, dim3((unsigned)td_nrows(a)), dim3(256), 0, (hipStream_t)stream, td_of(a), td_of(b), td_of(sum), (const float *)w->data, td_of(dst), eps);
    HIP_TRY(hipGetLastError()); return CDNA4_OK;
}

// ------------------------------------------------------------------------------------------------ ADD / MUL / DIV with ggml broadcasting (b repeats over a)
template <int OP> __device__ __forceinline__ float bin_apply(float a, float b) { return OP == 0 ? a + b : OP == 1 ? a * b : a / b; }
__device__ __forceinline__ float ld_any(const char *p, int f16) { return f16 ? __half2float(*reinterpret_cast<const __half *>(p)) : *reinterpret_cast<const float *>(p); }
template <int OP>       // f16 bits: 1 = a, 2 = b, 4 = dst is f16 (the partial sums of a -sm graph prompt batch are f16, binbcast.cu:451-466); arithmetic in f32
__global__ void binary_kernel(TD a, TD b, TD d, long total, int f16) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long i0 = i % d.ne[0], r = i / d.ne[0], i1 = r % d.ne[1], i2 = (r / d.ne[1]) % d.ne[2], i3 = r / (d.ne[1] * d.ne[2]);
        const float av = ld_any(a.data + i0 * a.nb[0] + i1 * a.nb[1] + i2 * a.nb[2] + i3 * a.nb[3], f16 & 1);
        const float bv = ld_any(b.data + (i0 % b.ne[0]) * b.nb[0] + (i1 % b.ne[1]) * b.nb[1] + (i2 % b.ne[2]) * b.nb[2] + (i3 % b.ne[3]) * b.nb[3], f16 & 2);
        char *dp = d.data + i0 * d.nb[0] + i1 * d.nb[1] + i2 * d.nb[2] + i3 * d.nb[3];
        if (f16 & 4) *reinterpret_cast<__half *>(dp) = __float2half_rn(bin_apply<OP>(av, bv)); else *reinterpret_cast<float *>(dp) = bin_apply<OP>(av, bv);
    }
}
template <int OP>       // same-shape contiguous fast path: 16 bytes per lane
__global__ void binary_vec_kernel(const float4 *a, const float4 *b, float4 *d, long n4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 x = a[i], y = b[i]; d[i] = make_float4(bin_apply<OP>(x.x, y.x), bin_apply<OP>(x.y, y.y), bin_apply<OP>(x.z, y.z), bin_apply<OP>(x.w, y.w));
    }
}
int cdna4_op_binary(cdna4_context *ctx, int op, const cdna4_tensor *a, const cdna4_tensor *b, const cdna4_tensor *dst, void *stream) {
    if (!ctx || !a || !b || !dst) return cdna4_set_err(CDNA4_E_INVALID, "null argument");
    const auto fl = [](const cdna4_tensor *t) { return t->type == T_F32 || t->type == T_F16; };
    OP_CHECK(op >= 0 && op <= 2 && fl(a) && fl(b) && fl(dst) && same_shape(a, dst), "binary op: f32 / f16, dst shaped like src0");
    const int f16 = (a->type == T_F16 ? 1 : 0) | (b->type == T_F16 ? 2 : 0) | (dst->type == T_F16 ? 4 : 0);
    for (int i = 0; i < 4; ++i) OP_CHECK(b->ne[i] > 0 && a->ne[i] % b->ne[i] == 0, "binary op: src1 does not broadcast over src0");
    const long total = td_nelem(dst); if (total == 0) return CDNA4_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    const bool fast = f16 == 0 && same_shape(a, b) && td_contig(a, 4) && td_contig(b, 4) && td_contig(dst, 4) && total % 4 == 0 &&
                      (((uintptr_t)a->data | (uintptr_t)b->data | (uintptr_t)dst->data) % 16 == 0);
    const unsigned grid = (unsigned)std::min<long>(((fast ? total / 4 : total) + 255) / 256, 8L * ctx->num_cu);
#define BIN(OP_) case OP_: if (fast) hipLaunchKernelGGL(binary_vec_kernel<OP_>, dim3(grid), dim3(256), 0, st, (const float4 *)a->data, (const float4 *)b->data, (float4 *)dst->data, total / 4); \
                           else hipLaunchKernelGGL(binary_kernel<OP_>, dim3(grid), dim3(256), 0, st, td_of(a), td_of(b), td_of(dst), total, f16); break;
    switch (op) { BIN(0) BIN(1) BIN(2) }
#undef BIN
    HIP_TRY(hipGetLastError()); return CDNA4_OK;
}

// ------------------------------------------------------------------------------------------------ ROPE (NORM and NEOX modes, YaRN, freq factors)
// one thread per rotated pair; theta_i = pos * theta_scale^i built by the reference's own chain of multiplications (ggml_rope_cache_init)
struct RopeParams { int n_dims, neox; float theta_scale, freq_scale, ext_factor, attn_factor, corr0, corr1; const float2 *table; };
// one rotation, with the products and the fused multiply-adds spelled out: every kernel that rotates (the single op, ROPE + KV store, q / k norms + ROPE + KV store) rounds alike, so a
// fused launch reproduces the launches it replaces bit for bit (left to the compiler, `x0 * s + x1 * c` contracts around either product)
__device__ __forceinline__ void rope_rot(float x0, float x1, float c, float s, float &y0, float &y1) { y0 = __fmaf_rn(x0, c, -__fmul_rn(x1, s)); y1 = __fmaf_rn(x0, s, __fmul_rn(x1, c)); }
__device__ __forceinline__ void rope_pair(const RopeParams &p, const int32_t *pos, const float *freq_factors, long ip, long i2, float &c, float &s) {
    if (p.table) { const float2 cs2 = p.table[i2 * (p.n_dims / 2) + ip]; c = cs2.x; s = cs2.y; return; }       // (cos, sin) of this (token, pair) from the per-graph cache
    float theta = (float)pos[i2];
    for (long k = 0; k < ip; ++k) theta *= p.theta_scale;
    const float ff = freq_factors ? freq_factors[ip] : 1.0f;
    const float theta_extrap = theta / ff; float th = p.freq_scale * theta_extrap, mscale = p.attn_factor;
    if (p.ext_factor != 0.0f) {
        const float yv = ((float)ip - p.corr0) / fmaxf(0.001f, p.corr1 - p.corr0);
        const float ramp_mix = (1.f - fminf(1.f, fmaxf(0.f, yv))) * p.ext_factor;
        th = th * (1.f - ramp_mix) + theta_extrap * ramp_mix;
        mscale *= 1.0f + 0.1f * logf(1.0f / p.freq_scale);
    }
    float sn, cs; sincosf(th, &sn, &cs); c = cs * mscale; s = sn * mscale;
}
__global__ void rope_kernel(TD x, const int32_t *pos, const float *freq_factors, TD y, RopeParams p, long total_pairs) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total_pairs) return;
    const long half = x.ne[0] / 2;
    const long ip = idx % half, r = idx / half, i1 = r % x.ne[1], i2 = (r / x.ne[1]) % x.ne[2], i3 = r / (x.ne[1] * x.ne[2]);
    const char *xr = x.data + i1 * x.nb[1] + i2 * x.nb[2] + i3 * x.nb[3]; char *yr = y.data + i1 * y.nb[1] + i2 * y.nb[2] + i3 * y.nb[3];
    const long i0 = 2 * ip;
    if (i0 >= p.n_dims) {       // beyond the rotated dims: plain copy of the pair
        reinterpret_cast<float *>(yr)[i0] = reinterpret_cast<const float *>(xr)[i0]; reinterpret_cast<float *>(yr)[i0 + 1] = reinterpret_cast<const float *>(xr)[i0 + 1];
        return;
    }
    float c, s; rope_pair(p, pos, freq_factors, ip, i2, c, s);          // (theta built by the same float chain as the CPU cache builder, or read from the per-graph cache)
    const long ia = p.neox ? ip : i0, ib = p.neox ? ip + p.n_dims / 2 : i0 + 1;
    const float x0 = reinterpret_cast<const float *>(xr)[ia], x1 = reinterpret_cast<const float *>(xr)[ib];
    float y0, y1; rope_rot(x0, x1, c, s, y0, y1);
    reinterpret_cast<float *>(yr)[ia] = y0; reinterpret_cast<float *>(yr)[ib] = y1;
}
// (cos, sin) * mscale for every (token, pair): computed ONCE per graph instead of once per layer (the reference's CPU path caches the same way:
// ggml_rope_cache_init, ggml.c:20725-20745); the 63-step multiplication chain and the large-argument sincos are most of a decode-size rope launch
__global__ void rope_table_kernel(const int32_t *pos, const float *freq_factors, RopeParams p, float2 *table, long n_tok) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x, half = p.n_dims / 2;
    if (idx >= n_tok * half) return;
    RopeParams q = p; q.table = nullptr;
    float c, s; rope_pair(q, pos, freq_factors, idx % half, idx / half, c, s);
    table[idx] = make_float2(c, s);
}
static float rope_corr_dim(int n_dims, int n_ctx_orig, float n_rot, float base) { return n_dims * logf(n_ctx_orig / (n_rot * 2 * (float)M_PI)) / (2 * logf(base)); }
static RopeParams make_rope_params(int n_dims, int mode, int n_ctx_orig, float freq_base, float freq_scale, float ext_factor, float attn_factor, float beta_fast, float beta_slow) {
    RopeParams p; p.n_dims = n_dims; p.neox = mode == 2; p.theta_scale = powf(freq_base, -2.0f / n_dims); p.freq_scale = freq_scale; p.ext_factor = ext_factor; p.attn_factor = attn_factor;
    const float start = floorf(rope_corr_dim(n_dims, n_ctx_orig, beta_fast, freq_base)), end = ceilf(rope_corr_dim(n_dims, n_ctx_orig, beta_slow, freq_base));      // ggml_rope_yarn_corr_dims
    p.corr0 = fmaxf(0.f, start); p.corr1 = fminf((float)(n_dims - 1), end); p.table = nullptr;
    return p;
}
// the context's rope cache: valid for ONE (positions, frequency factors, parameters) combination until cdna4_op_rope_cache_reset
static const float2 *rope_cached(const cdna4_context *ctx, const int32_t *pos, const float *ff, long n_tok, const RopeParams &p) {
    const auto &k = ctx->rope_key;
    return (ctx->rope_table && k.pos == pos && k.ff == ff && k.n_tok == n_tok && k.n_dims == p.n_dims && k.theta_scale == p.theta_scale && k.freq_scale == p.freq_scale && k.ext_factor == p.ext_factor &&
            k.attn_factor == p.attn_factor && k.corr0 == p.corr0 && k.corr1 == p.corr1) ? (const float2 *)ctx->rope_table : nullptr;
}
int cdna4_op_rope_cache_reset(cdna4_context *ctx) { if (!ctx) return cdna4_set_err(CDNA4_E_INVALID, "null context"); ctx->rope_key.pos = nullptr; return CDNA4_OK; }
int cdna4_op_rope_cache(cdna4_context *ctx, const int32_t *pos, int64_t n_tok, const float *freq_factors, int n_dims, int n_ctx_orig, float freq_base, float freq_scale, float ext_factor,
                        float attn_factor, float beta_fast, float beta_slow, void *stream) {
    if (!ctx || !pos) return cdna4_set_err(CDNA4_E_INVALID, "null argument");
    OP_CHECK(n_dims > 0 && n_dims % 2 == 0 && n_tok >= 1, "rope_cache: even dims");
    constexpr size_t CAP = (size_t)16 << 20;                 // fixed size: the address must not change under a captured graph
    OP_CHECK((size_t)n_tok * (n_dims / 2) * sizeof(float2) <= CAP, "rope_cache: batch too large for the cache (ropes then compute their angles inline)");
    HIP_TRY(hipSetDevice(ctx->device));
    if (!ctx->rope_table) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (stream && hipStreamIsCapturing((hipStream_t)stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return cdna4_set_err(CDNA4_E_NOMEM, "rope cache not allocated yet (stream capture)");
        HIP_TRY(hipMalloc(&ctx->rope_table, CAP));
    }
    const RopeParams p = make_rope_params(n_dims, 0, n_ctx_orig, freq_base, freq_scale, ext_factor, attn_factor, beta_fast, beta_slow);
    const long n = n_tok * (n_dims / 2);
    hipLaunchKernelGGL(rope_table_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pos, freq_factors, p, (float2 *)ctx->rope_table, (long)n_tok);
    HIP_TRY(hipGetLastError());
    auto &k = ctx->rope_key; k.pos = pos; k.ff = freq_factors; k.n_tok = n_tok; k.n_dims = n_dims; k.theta_scale = p.theta_scale; k.freq_scale = p.freq_scale; k.ext_factor = p.ext_factor;
    k.attn_factor = p.attn_factor; k.corr0 = p.corr0; k.corr1 = p.corr1;
    return CDNA4_OK;
}
int cdna4_op_rope(cdna4_context *ctx, const cdna4_tensor *x, const int32_t *pos, const float *freq_factors, const cdna4_tensor *dst, int n_dims, int mode, int n_ctx_orig,
                  float freq_base, float freq_scale, float ext_factor, float attn_factor, float beta_fast, float beta_slow, void *stream) {
    if (!ctx || !x || !dst || !pos) return cdna4_set_err(CDNA4_E_INVALID, "null argument");
    OP_CHECK(x->type == T_F32 && dst->type == T_F32 && same_shape(x, dst) && td_rows_contig(x, 4) && td_rows_contig(dst, 4), "rope: f32 rows only");
    OP_CHECK((mode == 0 || mode == 2) && n_dims > 0 && n_dims % 2 == 0 && n_dims <= x->ne[0] && x->ne[0] % 2 == 0, "rope: NORM / NEOX modes, even dims");
    if (td_nelem(x) == 0) return CDNA4_OK;
    RopeParams p = make_rope_params(n_dims, mode, n_ctx_orig, freq_base, freq_scale, ext_factor, attn_factor, beta_fast, beta_slow);
    p.table = rope_cached(ctx, pos, freq_factors, x->ne[2], p);
    // (NEOX with partial rotation -- Phi / GPT-NeoX style: pairs (i, i + n_dims / 2) for i < n_dims / 2, the values behind n_dims copied -- is the same kernel: ggml.c:21100-21160,
    //  ggml-cuda/rope.cu:156-243)
    const long pairs = td_nelem(x) / 2;
    HIP_TRY(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(rope_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, (hipStream_t)stream, td_of(x), pos, freq_factors, td_of(dst), p, pairs);
    HIP_TRY(hipGetLastError()); return CDNA4_OK;
}

// ------------------------------------------------------------------------------------------------ ROPE(Q) + ROPE(K) + K-cache store + V-cache store in one launch
// The four nodes between the QKV mat-muls and the attention of a layer (llm_build_kv_store): threads [0, pq) rotate Q pairs, [pq, pq + pk) rotate K pairs and
// also store them as f16 at the same flattened element index of the K-cache view, the rest convert V elements into the V-cache view.
__device__ __forceinline__ char *flat_addr(const TD &d, long e, int esz) {       // element e of the flattened (ggml order) tensor
    const long d0 = e % d.ne[0]; e /= d.ne[0]; const long d1 = e % d.ne[1]; e /= d.ne[1]; const long d2 = e % d.ne[2], d3 = e / d.ne[2];
    (void)esz; return d.data + d0 * d.nb[0] + d1 * d.nb[1] + d2 * d.nb[2] + d3 * d.nb[3];
}
__global__ void __launch_bounds__(256) rope_store_kv_kernel(TD q, TD qd, TD k, TD kd, int has_kd, TD kc, TD v, TD vc, const int32_t *pos, const float *freq_factors, RopeParams p,
                                                            long pq, long pk, long nv, void *const *k_slot, void *const *v_slot) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < pq + pk) {
        const bool isk = idx >= pq; const TD &x = isk ? k : q; const TD &y = isk ? kd : qd; const long id = isk ? idx - pq : idx;
        const long half = x.ne[0] / 2, ip = id % half, r = id / half, i1 = r % x.ne[1], i2 = (r / x.ne[1]) % x.ne[2], i3 = r / (x.ne[1] * x.ne[2]);
        const float *xr = reinterpret_cast<const float *>(x.data + i1 * x.nb[1] + i2 * x.nb[2] + i3 * x.nb[3]);
        long ia = 2 * ip, ib = 2 * ip + 1; float y0, y1;
        if (2 * ip >= p.n_dims) { y0 = xr[ia]; y1 = xr[ib]; }
        else {
            if (p.neox) { ia = ip; ib = ip + p.n_dims / 2; }
            float c, s; rope_pair(p, pos, freq_factors, ip, i2, c, s);
            const float x0 = xr[ia], x1 = xr[ib]; rope_rot(x0, x1, c, s, y0, y1);
        }
        if (!isk || has_kd) { float *yr = reinterpret_cast<float *>(y.data + i1 * y.nb[1] + i2 * y.nb[2] + i3 * y.nb[3]); yr[ia] = y0; yr[ib] = y1; }
        if (isk) {
            TD c = kc; if (k_slot) c.data = static_cast<char *>(*k_slot);
            const long e = ((i3 * x.ne[2] + i2) * x.ne[1] + i1) * x.ne[0];
            *reinterpret_cast<__half *>(flat_addr(c, e + ia, 2)) = __float2half_rn(y0); *reinterpret_cast<__half *>(flat_addr(c, e + ib, 2)) = __float2half_rn(y1);
        }
    } else if (idx < pq + pk + nv) {
        const long e = idx - pq - pk;
        TD c = vc; if (v_slot) c.data = static_cast<char *>(*v_slot);
        *reinterpret_cast<__half *>(flat_addr(c, e, 2)) = __float2half_rn(*reinterpret_cast<const float *>(flat_addr(v, e, 4)));
    }
}
// The same four nodes on the layouts a llama graph hands over -- f32 q / k rows of a power-of-two head size, one contiguous row per (token, head), contiguous f16 cache views,
// the context's (cos, sin) table current -- as a (work item, token) grid of 32-bit indices: a thread rotates ONE pair held as a float2 (NORM mode: the pair is adjacent) or two
// strided scalars (NEOX), K pairs also go to the cache as a __half2 / two halves, V goes over as float4 -> 4 halves.  The generic kernel above decomposes a flattened 64-bit
// element index per thread (364 quarter-rate multiplies, 3188 instructions: 16 us per layer of a 512-token prompt for 22 MB of traffic).
struct RopeFast { const float *q; float *qd; const float *k; float *kd; __half *kc; const float *v; __half *vc; const float2 *table; void *const *k_slot, *const *v_slot;
                  int n_head, n_kv_head, hd_log2, n_dims; long q_tok, qd_tok, k_tok, kd_tok, v_tok; int nv4; };      // *_tok: elements between tokens; nv4 = float4 chunks of V per token
template <bool NEOX>
__global__ void __launch_bounds__(256) rope_store_kv_fast_kernel(RopeFast a) {
    const int w = blockIdx.x * 256 + threadIdx.x, tok = blockIdx.y;
    const int half = 1 << (a.hd_log2 - 1), hd = 1 << a.hd_log2, nq = a.n_head * half, nk = a.n_kv_head * half, hnd = a.n_dims >> 1;
    if (w < nq + nk) {
        const bool isk = w >= nq; const int id = isk ? w - nq : w, head = id >> (a.hd_log2 - 1), ip = id & (half - 1);
        const float *x = (isk ? a.k + (long)tok * a.k_tok : a.q + (long)tok * a.q_tok) + head * hd;
        float *y = isk ? (a.kd ? a.kd + (long)tok * a.kd_tok + head * hd : nullptr) : a.qd + (long)tok * a.qd_tok + head * hd;
        int ia, ib; float x0, x1;
        if (ip < hnd) { ia = NEOX ? ip : 2 * ip; ib = NEOX ? ip + hnd : 2 * ip + 1; }
        else { ia = a.n_dims + 2 * (ip - hnd); ib = ia + 1; }                     // beyond the rotated dims: plain copy of a pair
        if (!NEOX || ip >= hnd) { const float2 p = *reinterpret_cast<const float2 *>(x + ia); x0 = p.x; x1 = p.y; } else { x0 = x[ia]; x1 = x[ib]; }
        float y0 = x0, y1 = x1;
        if (ip < hnd) { const float2 cs = a.table[(long)tok * hnd + ip]; rope_rot(x0, x1, cs.x, cs.y, y0, y1); }
        if (y) { if (!NEOX || ip >= hnd) *reinterpret_cast<float2 *>(y + ia) = make_float2(y0, y1); else { y[ia] = y0; y[ib] = y1; } }
        if (isk) {
            __half *c = (a.k_slot ? static_cast<__half *>(*a.k_slot) : a.kc) + ((long)tok * a.n_kv_head + head) * hd;
            if (!NEOX || ip >= hnd) *reinterpret_cast<__half2 *>(c + ia) = __floats2half2_rn(y0, y1); else { c[ia] = __float2half_rn(y0); c[ib] = __float2half_rn(y1); }
        }
    } else if (w < nq + nk + a.nv4) {
        const int c4 = w - nq - nk;
        const float4 f = reinterpret_cast<const float4 *>(a.v + (long)tok * a.v_tok)[c4];
        __half *c = (a.v_slot ? static_cast<__half *>(*a.v_slot) : a.vc) + ((long)tok * a.nv4 + c4) * 4;
        union { __half2 h[2]; uint2 u; } o; o.h[0] = __floats2half2_rn(f.x, f.y); o.h[1] = __floats2half2_rn(f.z, f.w);
        *reinterpret_cast<uint2 *>(c) = o.u;
    }
}
int cdna4_op_rope_store_kv(cdna4_context *ctx, const cdna4_tensor *q, const cdna4_tensor *q_dst, const cdna4_tensor *k, const cdna4_tensor *k_dst, const cdna4_tensor *k_cache, void *const *k_slot,
                           const cdna4_tensor *v, const cdna4_tensor *v_cache, void *const *v_slot, const int32_t *pos, const float *freq_factors, int n_dims, int mode, int n_ctx_orig,
                           float freq_base, float freq_scale, float ext_factor, float attn_factor, float beta_fast, float beta_slow, void *stream) {
    if (!ctx || !q || !q_dst || !k || !k_cache || !v || !v_cache || !pos) return cdna4_set_err(CDNA4_E_INVALID, "null argument");
    OP_CHECK(q->type == T_F32 && q_dst->type == T_F32 && k->type == T_F32 && (!k_dst || k_dst->type == T_F32) && v->type == T_F32 && k_cache->type == T_F16 && v_cache->type == T_F16, "rope_store_kv: f32 Q / K / V, f16 caches");
    OP_CHECK(same_shape(q, q_dst) && (!k_dst || same_shape(k, k_dst)) && td_rows_contig(q, 4) && td_rows_contig(q_dst, 4) && td_rows_contig(k, 4) && (!k_dst || td_rows_contig(k_dst, 4)) &&
             td_nelem(k) == td_nelem(k_cache) && td_nelem(v) == td_nelem(v_cache) && q->ne[0] == k->ne[0] && q->ne[2] == k->ne[2], "rope_store_kv: shapes");
    OP_CHECK((mode == 0 || mode == 2) && n_dims > 0 && n_dims % 2 == 0 && n_dims <= q->ne[0] && q->ne[0] % 2 == 0, "rope_store_kv: NORM / NEOX modes, even dims");
    const long pq = td_nelem(q) / 2, pk = td_nelem(k) / 2, nv = td_nelem(v), total = pq + pk + nv; if (total == 0) return CDNA4_OK;
    RopeParams p = make_rope_params(n_dims, mode, n_ctx_orig, freq_base, freq_scale, ext_factor, attn_factor, beta_fast, beta_slow);
    p.table = rope_cached(ctx, pos, freq_factors, q->ne[2], p);
    HIP_TRY(hipSetDevice(ctx->device));
    {   // the fast form: see rope_store_kv_fast_kernel
        constexpr bool fast_on = true;                             // (round 4 A/B closed: 16.1 -> 6.3 us per layer at 512 tokens; the generic kernel serves the other layouts)
        const long hd = q->ne[0], n_tok = q->ne[2]; int hl = 0; while ((1L << hl) < hd) ++hl;
        auto rows_ok = [&](const cdna4_tensor *t) { return t->ne[3] == 1 && t->nb[1] == hd * 4 && t->nb[2] % 8 == 0 && ((uintptr_t)t->data % 8) == 0; };
        const long nv_tok = n_tok > 0 ? td_nelem(v) / n_tok : 0;
        // V: [n_embd_v, n_tok] or [hd, n_head_kv, n_tok], every token's values contiguous (the tokens themselves may be a strided slice of a fused q,k,v result)
        long v_tok = -1;
        if (v->nb[0] == 4 && v->ne[3] == 1) {
            if (v->ne[2] == 1 && v->ne[1] == n_tok) v_tok = v->nb[1] / 4;
            else if (v->ne[2] == n_tok && v->nb[1] == v->ne[0] * 4) v_tok = v->nb[2] / 4;
        }
        if (fast_on && p.table && (1L << hl) == hd && hd >= 4 && n_tok >= 1 && n_tok <= 65535 && k->ne[2] == n_tok && rows_ok(q) && rows_ok(q_dst) && rows_ok(k) && (!k_dst || rows_ok(k_dst)) &&
            td_contig(k_cache, 2) && td_contig(v_cache, 2) && v_tok >= nv_tok && v_tok % 4 == 0 && nv_tok * n_tok == td_nelem(v) && nv_tok % 4 == 0 && ((uintptr_t)v->data % 16) == 0 && ((uintptr_t)k_cache->data % 4) == 0 &&
            ((uintptr_t)v_cache->data % 8) == 0 && q->ne[1] * hd < (1L << 28) && nv_tok < (1L << 28)) {
            RopeFast a; a.q = (const float *)q->data; a.qd = (float *)q_dst->data; a.k = (const float *)k->data; a.kd = k_dst ? (float *)k_dst->data : nullptr; a.kc = (__half *)k_cache->data;
            a.v = (const float *)v->data; a.vc = (__half *)v_cache->data; a.table = p.table; a.k_slot = k_slot; a.v_slot = v_slot;
            a.n_head = (int)q->ne[1]; a.n_kv_head = (int)k->ne[1]; a.hd_log2 = hl; a.n_dims = n_dims;
            a.q_tok = q->nb[2] / 4; a.qd_tok = q_dst->nb[2] / 4; a.k_tok = k->nb[2] / 4; a.kd_tok = k_dst ? k_dst->nb[2] / 4 : 0; a.v_tok = v_tok; a.nv4 = (int)(nv_tok / 4);
            const long items = (long)(a.n_head + a.n_kv_head) * (hd / 2) + a.nv4;
            const dim3 grid((unsigned)((items + 255) / 256), (unsigned)n_tok);
            if (mode == 2) hipLaunchKernelGGL(rope_store_kv_fast_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, a);
            else           hipLaunchKernelGGL(rope_store_kv_fast_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, a);
            HIP_TRY(hipGetLastError()); return CDNA4_OK;
        }
    }
    TD kd; memset(&kd, 0, sizeof(kd)); if (k_dst) kd = td_of(k_dst);
    hipLaunchKernelGGL(rope_store_kv_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, td_of(q), td_of(q_dst), td_of(k), kd, k_dst ? 1 : 0, td_of(k_cache), td_of(v), td_of(v_cache),
                       pos, freq_factors, p, pq, pk, nv, k_slot, v_slot);
    HIP_TRY(hipGetLastError()); return CDNA4_OK;
}

// ------------------------------------------------------------------------------------------------ per-head RMS_NORM(Q), RMS_NORM(K) + ROPE(Q) + ROPE(K) + K-cache store + V-cache store
// Qwen3-style attention (llama-build-context.cpp:2481-2490 q_norm / k_norm between the mat-muls and the rotation): six nodes of a layer as ONE launch on the layouts of
// rope_store_kv_fast_kernel.  G = head size / 4 lanes own one (token, head) row, a float4 each; the sum of squares runs in the order of rms_norm_kernel on such a row (16-lane DPP
// rows, then row sums pairwise) and the rotation in the order of rope_kernel, so the launch reproduces the six it replaces bit for bit.  NEOX pairs (i, i + hd/2) sit G/2 lanes apart:
// one exchange of the normed float4; the lower lane writes the first halves, the upper lane the second.  Grid as the fast kernel: (work items / 256, tokens).
struct NormRopeFast { RopeFast r; const float *qw, *kw; float eps_q, eps_k; };
template <bool NEOX, int G>
__global__ void __launch_bounds__(256) norm_rope_store_kv_kernel(NormRopeFast p) {
    const RopeFast &a = p.r;
    constexpr int hd = 4 * G, hnd = hd / 2;
    const int w = blockIdx.x * 256 + threadIdx.x, tok = blockIdx.y, lane = threadIdx.x & 63;
    const int nq = a.n_head * G, nk = a.n_kv_head * G;
    const bool rot = w < nq + nk, isk = w >= nq;
    const int id = isk ? w - nq : w, head = id / G, l = id & (G - 1);
    const float *x = (isk ? a.k + (long)tok * a.k_tok : a.q + (long)tok * a.q_tok) + head * hd;
    // (every lane of a wave takes part in the DPP steps and the exchange: lanes past the rotated rows carry zeros)
    const float4 v = rot ? reinterpret_cast<const float4 *>(x)[l] : make_float4(0, 0, 0, 0);
    float ss = 0.f; ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    ss += fa_dpp<0xb1>(ss); ss += fa_dpp<0x4e>(ss); ss += fa_dpp<0x141>(ss); ss += fa_dpp<0x140>(ss);
    float sum;
    if (G == 16) sum = ss;
    else if (G == 32) { const float lo = lane_bcast(ss, 0) + lane_bcast(ss, 16), hi = lane_bcast(ss, 32) + lane_bcast(ss, 48); sum = lane < 32 ? lo : hi; }
    else sum = (lane_bcast(ss, 0) + lane_bcast(ss, 16)) + (lane_bcast(ss, 32) + lane_bcast(ss, 48));
    const float scale = 1.0f / sqrtf(sum / (float)hd + (isk ? p.eps_k : p.eps_q));
    float4 n = v;
    if (rot) { const float4 c = reinterpret_cast<const float4 *>(isk ? p.kw : p.qw)[l]; n.x = scale * c.x * v.x; n.y = scale * c.y * v.y; n.z = scale * c.z * v.z; n.w = scale * c.w * v.w; }
    float4 y;
    if (NEOX) {
        float4 o; o.x = __shfl_xor(n.x, G / 2, 64); o.y = __shfl_xor(n.y, G / 2, 64); o.z = __shfl_xor(n.z, G / 2, 64); o.w = __shfl_xor(n.w, G / 2, 64);
        if (!rot) goto values;
        const bool lower = l < G / 2;
        const float2 *cs = a.table + (long)tok * hnd + 4 * (l & (G / 2 - 1));
        const float4 t0 = reinterpret_cast<const float4 *>(cs)[0], t1 = reinterpret_cast<const float4 *>(cs)[1];       // (cos, sin) of four consecutive pairs
        const float4 x0 = lower ? n : o, x1 = lower ? o : n;
        float4 ya, yb;
        rope_rot(x0.x, x1.x, t0.x, t0.y, ya.x, yb.x); rope_rot(x0.y, x1.y, t0.z, t0.w, ya.y, yb.y); rope_rot(x0.z, x1.z, t1.x, t1.y, ya.z, yb.z); rope_rot(x0.w, x1.w, t1.z, t1.w, ya.w, yb.w);
        y = lower ? ya : yb;
    } else {
        if (!rot) goto values;
        const float4 t = reinterpret_cast<const float4 *>(a.table + (long)tok * hnd)[l];                                // pairs 2 l and 2 l + 1
        rope_rot(n.x, n.y, t.x, t.y, y.x, y.y); rope_rot(n.z, n.w, t.z, t.w, y.z, y.w);
    }
    {
        float *yd = isk ? (a.kd ? a.kd + (long)tok * a.kd_tok + head * hd : nullptr) : a.qd + (long)tok * a.qd_tok + head * hd;
        if (yd) reinterpret_cast<float4 *>(yd)[l] = y;
        if (isk) {
            __half *c = (a.k_slot ? static_cast<__half *>(*a.k_slot) : a.kc) + ((long)tok * a.n_kv_head + head) * hd + 4 * l;
            union { __half2 h[2]; uint2 u; } o; o.h[0] = __floats2half2_rn(y.x, y.y); o.h[1] = __floats2half2_rn(y.z, y.w);
            *reinterpret_cast<uint2 *>(c) = o.u;
        }
        return;
    }
    values:
    if (w < nq + nk + a.nv4) {
        const int c4 = w - nq - nk;
        const float4 f = reinterpret_cast<const float4 *>(a.v + (long)tok * a.v_tok)[c4];
        __half *c = (a.v_slot ? static_cast<__half *>(*a.v_slot) : a.vc) + ((long)tok * a.nv4 + c4) * 4;
        union { __half2 h[2]; uint2 u; } o; o.h[0] = __floats2half2_rn(f.x, f.y); o.h[1] = __floats2half2_rn(f.z, f.w);
        *reinterpret_cast<uint2 *>(c) = o.u;
    }
}
// CDNA4_E_UNSUPPORTED (nothing launched) when the layouts are not the fast kernel's or the context's (cos, sin) table is not current: run the six nodes one by one.
int cdna4_op_norm_rope_store_kv(cdna4_context *ctx, const cdna4_tensor *q, const cdna4_tensor *q_norm, float eps_q, const cdna4_tensor *q_dst, const cdna4_tensor *k, const cdna4_tensor *k_norm, float eps_k,
                                const cdna4_tensor *k_dst, const cdna4_tensor *k_cache, void *const *k_slot, const cdna4_tensor *v, const cdna4_tensor *v_cache, void *const *v_slot, const int32_t *pos,
                                const float *freq_factors, int n_dims, int mode, int n_ctx_orig, float freq_base, float freq_scale, float ext_factor, float attn_factor, float beta_fast, float beta_slow,
                                void *stream) {
    if (!ctx || !q || !q_norm || !q_dst || !k || !k_norm || !k_cache || !v || !v_cache || !pos) return cdna4_set_err(CDNA4_E_INVALID, "null argument");
    OP_CHECK(q->type == T_F32 && q_dst->type == T_F32 && k->type == T_F32 && (!k_dst || k_dst->type == T_F32) && v->type == T_F32 && k_cache->type == T_F16 && v_cache->type == T_F16 &&
             q_norm->type == T_F32 && k_norm->type == T_F32, "norm_rope_store_kv: f32 Q / K / V / norm weights, f16 caches");
    const long hd = q->ne[0], n_tok = q->ne[2];
    OP_CHECK(same_shape(q, q_dst) && (!k_dst || same_shape(k, k_dst)) && td_nelem(k) == td_nelem(k_cache) && td_nelem(v) == td_nelem(v_cache) && k->ne[0] == hd && k->ne[2] == n_tok &&
             q_norm->ne[0] == hd && td_nrows(q_norm) == 1 && q_norm->nb[0] == 4 && k_norm->ne[0] == hd && td_nrows(k_norm) == 1 && k_norm->nb[0] == 4, "norm_rope_store_kv: shapes");
    OP_CHECK((mode == 0 || mode == 2) && n_dims == hd && (hd == 64 || hd == 128 || hd == 256), "norm_rope_store_kv: NORM / NEOX modes over whole heads of 64, 128 or 256 values");
    if (td_nelem(q) + td_nelem(k) + td_nelem(v) == 0) return CDNA4_OK;
    RopeParams p = make_rope_params(n_dims, mode, n_ctx_orig, freq_base, freq_scale, ext_factor, attn_factor, beta_fast, beta_slow);
    p.table = rope_cached(ctx, pos, freq_factors, n_tok, p);
    OP_CHECK(p.table != nullptr, "norm_rope_store_kv: the context's rope cache is not current");
    auto rows_ok = [&](const cdna4_tensor *t) { return t->ne[3] == 1 && t->nb[0] == 4 && t->nb[1] == hd * 4 && t->nb[2] % 16 == 0 && ((uintptr_t)t->data % 16) == 0; };
    const long nv_tok = n_tok > 0 ? td_nelem(v) / n_tok : 0;
    long v_tok = -1;
    if (v->nb[0] == 4 && v->ne[3] == 1) {
        if (v->ne[2] == 1 && v->ne[1] == n_tok) v_tok = v->nb[1] / 4;
        else if (v->ne[2] == n_tok && v->nb[1] == v->ne[0] * 4) v_tok = v->nb[2] / 4;
    }
    OP_CHECK(n_tok >= 1 && n_tok <= 65535 && rows_ok(q) && rows_ok(q_dst) && rows_ok(k) && (!k_dst || rows_ok(k_dst)) && td_contig(k_cache, 2) && td_contig(v_cache, 2) && v_tok >= nv_tok && v_tok % 4 == 0 &&
             nv_tok * n_tok == td_nelem(v) && nv_tok % 4 == 0 && ((uintptr_t)v->data % 16) == 0 && ((uintptr_t)k_cache->data % 8) == 0 && ((uintptr_t)v_cache->data % 8) == 0 &&
             ((uintptr_t)q_norm->data % 16) == 0 && ((uintptr_t)k_norm->data % 16) == 0 && q->ne[1] * hd < (1L << 28) && nv_tok < (1L << 28), "norm_rope_store_kv: layouts");
    HIP_TRY(hipSetDevice(ctx->device));
    NormRopeFast a; a.qw = (const float *)q_norm->data; a.kw = (const float *)k_norm->data; a.eps_q = eps_q; a.eps_k = eps_k;
    a.r.q = (const float *)q->data; a.r.qd = (float *)q_dst->data; a.r.k = (const float *)k->data; a.r.kd = k_dst ? (float *)k_dst->data : nullptr; a.r.kc = (__half *)k_cache->data;
    a.r.v = (const float *)v->data; a.r.vc = (__half *)v_cache->data; a.r.table = p.table; a.r.k_slot = k_slot; a.r.v_slot = v_slot;
    a.r.n_head = (int)q->ne[1]; a.r.n_kv_head = (int)k->ne[1]; a.r.hd_log2 = hd == 64 ? 6 : hd == 128 ? 7 : 8; a.r.n_dims = n_dims;
    a.r.q_tok = q->nb[2] / 4; a.r.qd_tok = q_dst->nb[2] / 4; a.r.k_tok = k->nb[2] / 4; a.r.kd_tok = k_dst ? k_dst->nb[2] / 4 : 0; a.r.v_tok = v_tok; a.r.nv4 = (int)(nv_tok / 4);
    const long items = (long)(a.r.n_head + a.r.n_kv_head) * (hd / 4) + a.r.nv4;
    const dim3 grid((unsigned)((items + 255) / 256), (unsigned)n_tok);
    const bool neox = mode == 2;
#define NRK(G_) do { if (neox) hipLaunchKernelGGL((norm_rope_store_kv_kernel<true, G_>), grid, dim3(256), 0, (hipStream_t)stream, a); \
                     else      hipLaunchKernelGGL((norm_rope_store_kv_kernel<false, G_>), grid, dim3(256), 0, (hipStream_t)stream, a); } while (0)
    if (hd == 64) NRK(16); else if (hd == 128) NRK(32); else NRK(64);
#undef NRK
    HIP_TRY(hipGetLastError()); return CDNA4_OK;
}

// ------------------------------------------------------------------------------------------------ CPY (f32 / f16 -> f32 / f16, any strides; element order = flattened ggml order)
template <typename S, typename D> __device__ __forceinline__ D cvt(S v);
template <> __device__ __forceinline__ float cvt<float, float>(float v) { return v; }
template <> __device__ __forceinline__ __half cvt<float, __half>(float v) { return __float2half_rn(v); }
template <> __device__ __forceinline__ float cvt<__half, float>(__half v) { return __half2float(v); }
template <> __device__ __forceinline__ __half cvt<__half, __half>(__half v) { return v; }
template <typename S, typename D>
__global__ void cpy_kernel(TD s, TD d, long total, void *const *slot) {
    if (slot) d.data = static_cast<char *>(*slot);          // destination read from a device-side pointer slot (HIP-graph replays with a moving KV-cache head)
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long r = i; const long s0 = r % s.ne[0]; r /= s.ne[0]; const long s1 = r % s.ne[1]; r /= s.ne[1]; const long s2 = r % s.ne[2], s3 = r / s.ne[2];
        r = i;      const long d0 = r % d.ne[0]; r /= d.ne[0]; const long d1 = r % d.ne[1]; r /= d.ne[1]; const long d2 = r % d.ne[2], d3 = r / d.ne[2];
        *reinterpret_cast<D *>(d.data + d0 * d.nb[0] + d1 * d.nb[1] + d2 * d.nb[2] + d3 * d.nb[3]) = cvt<S, D>(*reinterpret_cast<const S *>(s.data + s0 * s.nb[0] + s1 * s.nb[1] + s2 * s.nb[2] + s3 * s.nb[3]));
    }
}
// f32 -> Q8_0 (a quantized KV cache, -ctk q8_0 -ctv q8_0: llm_build_kv_store's CPY nodes; ggml-cuda/cpy.cu cpy_f32_q8_0, ggml-quants.c quantize_row_q8_0_ref): one thread per
// 32-value block: d = amax / 127 (stored as f16), q = round(x / d).  Blocks are addressed in the flattened element order of both tensors (rows of whole blocks on either side).
__global__ void __launch_bounds__(256) cpy_f32_q8_0_kernel(TD s, TD d, long nblocks, void *const *slot) {
    if (slot) d.data = static_cast<char *>(*slot);
    const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblocks) return;
    const long e = b * 32;
    long r = e; const long s0 = r % s.ne[0]; r /= s.ne[0]; const long s1 = r % s.ne[1]; r /= s.ne[1]; const long s2 = r % s.ne[2], s3 = r / s.ne[2];
    r = e;      const long d0 = r % d.ne[0]; r /= d.ne[0]; const long d1 = r % d.ne[1]; r /= d.ne[1]; const long d2 = r % d.ne[2], d3 = r / d.ne[2];
    const float4 *x4 = reinterpret_cast<const float4 *>(s.data + s0 * 4 + s1 * s.nb[1] + s2 * s.nb[2] + s3 * s.nb[3]);
    float x[32]; float amax = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { const float4 v = x4[i]; x[4 * i] = v.x; x[4 * i + 1] = v.y; x[4 * i + 2] = v.z; x[4 * i + 3] = v.w; amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)))); }
    const float dd = amax / 127.f, id = dd != 0.f ? 1.0f / dd : 0.f;
    uint8_t *o = reinterpret_cast<uint8_t *>(d.data + (d0 / 32) * 34 + d1 * d.nb[1] + d2 * d.nb[2] + d3 * d.nb[3]);      // (34-byte blocks: 2-byte aligned stores)
    const __half hd = __float2half_rn(dd);
    *reinterpret_cast<uint16_t *>(o) = *reinterpret_cast<const uint16_t *>(&hd);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int q0 = (int)roundf(x[2 * i] * id), q1 = (int)roundf(x[2 * i + 1] * id);
        *reinterpret_cast<uint16_t *>(o + 2 + 2 * i) = (uint16_t)((q0 & 0xff) | ((q1 & 0xff) << 8));
    }
}
// Q8_0 rows (a K / V view of the quantized cache: [D, n_kv, n_head_kv, ne3], any row strides) -> a dense f16 copy [ne3][n_head_kv][n_kv][D] for the f16 attention kernels
__global__ void __launch_bounds__(256) q8_0_rows_to_f16_kernel(TD s, __half *dst, long nblocks) {
    const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblocks) return;
    const long bpr = s.ne[0] / 32; long r = b; const long ib = r % bpr; r /= bpr; const long i1 = r % s.ne[1]; r /= s.ne[1]; const long i2 = r % s.ne[2], i3 = r / s.ne[2];
    const uint8_t *blk = reinterpret_cast<const uint8_t *>(s.data + ib * 34 + i1 * s.nb[1] + i2 * s.nb[2] + i3 * s.nb[3]);
    const float dd = half_bits_to_float(*reinterpret_cast<const uint16_t *>(blk));
    __half2 *o = reinterpret_cast<__half2 *>(dst + (((i3 * s.ne[2] + i2) * s.ne[1] + i1) * s.ne[0] + ib * 32));
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint16_t w = *reinterpret_cast<const uint16_t *>(blk + 2 + 2 * i);
        o[i] = __floats2half2_rn(dd * (float)(int)(int8_t)(w & 0xff), dd * (float)(int)(int8_t)(w >> 8));
    }
}
int cdna4_op_cpy(cdna4_context *ctx, const cdna4_tensor *src, const cdna4_tensor *dst, void *stream) { return cdna4_op_cpy_indirect(ctx, src, dst, nullptr, stream); }
int cdna4_op_cpy_indirect(cdna4_context *ctx, const cdna4_tensor *src, const cdna4_tensor *dst, void *const *dst_slot, void *stream) {
    if (!ctx || !src || !dst) return cdna4_set_err(CDNA4_E_INVALID, "null argument");
    if (dst->type == T_Q8_0) {
        OP_CHECK(src->type == T_F32 && td_nelem(src) == td_nelem(dst) && src->ne[0] % 32 == 0 && dst->ne[0] % 32 == 0 && src->nb[0] == 4 && dst->nb[0] == 34 &&
                 ((uintptr_t)src->data % 16) == 0 && src->nb[1] % 16 == 0 && src->nb[2] % 16 == 0 && src->nb[3] % 16 == 0, "cpy: f32 rows of whole 32-blocks -> Q8_0");
        const long nblocks = td_nelem(src) / 32; if (nblocks == 0) return CDNA4_OK;
        HIP_TRY(hipSetDevice(ctx->device));
        hipLaunchKernelGGL(cpy_f32_q8_0_kernel, dim3((unsigned)((nblocks + 255) / 256)), dim3(256), 0, (hipStream_t)stream, td_of(src), td_of(dst), nblocks, dst_slot);
        HIP_TRY(hipGetLastError()); return CDNA4_OK;
    }
    OP_CHECK((src->type == T_F32 || src->type == T_F16) && (dst->type == T_F32 || dst->type == T_F16) && td_nelem(src) == td_nelem(dst), "cpy: f32 / f16, equal element counts");
    const long total = td_nelem(src); if (total == 0) return CDNA4_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    const unsigned grid = (unsigned)std::min<long>((total + 255) / 256, 16L * ctx->num_cu); hipStream_t st = (hipStream_t)stream;
    if (src->type == T_F32 && dst->type == T_F32) hipLaunchKernelGGL((cpy_kernel<float, float>), dim3(grid), dim3(256), 0, st, td_of(src), td_of(dst), total, dst_slot);
    else if (src->type == T_F32) hipLaunchKernelGGL((cpy_kernel<float, __half>), dim3(grid), dim3(256), 0, st, td_of(src), td_of(dst), total, dst_slot);
    else if (dst->type == T_F32) hipLaunchKernelGGL((cpy_kernel<__half, float>), dim3(grid), dim3(256), 0, st, td_of(src), td_of(dst), total, dst_slot);
    else hipLaunchKernelGGL((cpy_kernel<__half, __half>), dim3(grid), dim3(256), 0, st, td_of(src), td_of(dst), total, dst_slot);
    HIP_TRY(hipGetLastError()); return CDNA4_OK;
}

// ------------------------------------------------------------------------------------------------ SOFT_MAX (scale, optional f16 / f32 mask broadcast over heads, ALiBi)
__global__ void __launch_bounds__(256) soft_max_kernel(TD x, TD m, int mask_type, TD y, float scale, float max_bias, float m0, float m1, unsigned n_head_log2) {
    __shared__ float red[4];
    const long r = blockIdx.x, n = x.ne[0];
    const long i1 = r % x.ne[1], i2 = (r / x.ne[1]) % x.ne[2], i3 = r / (x.ne[1] * x.ne[2]);
    const float *xr = reinterpret_cast<const float *>(x.data + i1 * x.nb[1] + i2 * x.nb[2] + i3 * x.nb[3]);
    float *yr = reinterpret_cast<float *>(y.data + i1 * y.nb[1] + i2 * y.nb[2] + i3 * y.nb[3]);
    const char *mr = mask_type >= 0 ? m.data + i1 * m.nb[1] + (i2 % m.ne[2]) * m.nb[2] + (i3 % m.ne[3]) * m.nb[3] : nullptr;
    const unsigned h = (unsigned)i2;
    const float slope = max_bias > 0.0f ? (h < n_head_log2 ? powf(m0, (float)(h + 1)) : powf(m1, (float)(2 * (h - n_head_log2) + 1))) : 1.0f;
    auto val = [&](long i) { float v = xr[i] * scale; if (mr) v += slope * (mask_type == T_F16 ? __half2float(reinterpret_cast<const __half *>(mr)[i]) : reinterpret_cast<const float *>(mr)[i]); return v; };
    float mx = -INFINITY;
    for (long i = threadIdx.x; i < n; i += 256) mx = fmaxf(mx, val(i));
    mx = block_max256(mx, red);
    float sum = 0.f;
    for (long i = threadIdx.x; i < n; i += 256) { const float e = mx == -INFINITY ? 0.f : expf(val(i) - mx); yr[i] = e; sum += e; }
    sum = block_sum256(sum, red);
    const float inv = 1.0f / sum;
    for (long i = threadIdx.x; i < n; i += 256) yr[i] *= inv;
}
int cdna4_op_soft_max(cdna4_context *ctx, const cdna4_tensor *x, const cdna4_tensor *mask, const cdna4_tensor *dst, float scale, float max_bias, void *stream) {
    if (!ctx || !x || !dst) return cdna4_set_err(CDNA4_E_INVALID, "null argument");
    OP_CHECK(x->type == T_F32 && dst->type == T_F32 && same_shape(x, dst) && td_rows_contig(x, 4) && td_rows_contig(dst, 4), "soft_max: f32 rows only");
    OP_CHECK(!mask || ((mask->type == T_F16 || mask->type == T_F32) && mask->ne[0] >= x->ne[0] && mask->ne[1] >= x->ne[1]), "soft_max: mask shape / type");
    if (td_nelem(x) == 0) return CDNA4_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    const unsigned n_head_log2 = 1u << (unsigned)floorf(log2f((float)x->ne[2]));
    TD m; memset(&m, 0, sizeof(m)); if (mask) m = td_of(mask); else { m.ne[2] = m.ne[3] = 1; }
    hipLaunchKernelGGL(soft_max_kernel, dim3((unsigned)td_nrows(x)), dim3(256), 0, (hipStream_t)stream, td_of(x), m, mask ? mask->type : -1, td_of(dst), scale, max_bias,
                       powf(2.0f, -max_bias / n_head_log2), powf(2.0f, -(max_bias / 2.0f) / n_head_log2), n_head_log2);
    HIP_TRY(hipGetLastError()); return CDNA4_OK;
}

// ------------------------------------------------------------------------------------------------ ARGSORT (top-k): rank by counting
// order of the reference (std::greater / std::less on (value, index) pairs, iqk_cpu_ops.cpp:248-261): DESC = larger value first, ties -> LARGER
// index first; ASC = smaller value first, ties -> smaller index first.  Every position is written (a full sort); the reference only
// defines the first nk entries when nk < ne0.
__global__ void __launch_bounds__(256) argsort_kernel(TD x, TD y, int desc) {
    extern __shared__ float vals[];
    const long r = blockIdx.x, n = x.ne[0];
    const long i1 = r % x.ne[1], i2 = (r / x.ne[1]) % x.ne[2], i3 = r / (x.ne[1] * x.ne[2]);
    const float *xr = reinterpret_cast<const float *>(x.data + i1 * x.nb[1] + i2 * x.nb[2] + i3 * x.nb[3]);
    int32_t *yr = reinterpret_cast<int32_t *>(y.data + i1 * y.nb[1] + i2 * y.nb[2] + i3 * y.nb[3]);
    for (long i = threadIdx.x; i < n; i += 256) vals[i] = xr[i];
    __syncthreads();
    for (long i = threadIdx.x; i < n; i += 256) {
        const float v = vals[i]; int rank = 0;
        for (long j = 0; j < n; ++j) { const float u = vals[j]; rank += desc ? (u > v || (u == v && j > i)) : (u < v || (u == v && j < i)); }
        yr[rank] = (int32_t)i;
    }
}
int cdna4_op_argsort(cdna4_context *ctx, const cdna4_tensor *x, const cdna4_tensor *dst, int order_desc, void *stream) {
    if (!ctx || !x || !dst) return cdna4_set_err(CDNA4_E_INVALID, "null argument");
    OP_CHECK(x->type == T_F32 && dst->type == 26 /* I32 */ && same_shape(x, dst) && td_rows_contig(x, 4) && td_rows_contig(dst, 4) && x->ne[0] <= 16384, "argsort: f32 rows of <= 16384 values");
    if (td_nelem(x) == 0) return CDNA4_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(argsort_kernel, dim3((unsigned)td_nrows(x)), dim3(256), (size_t)x->ne[0] * 4, (hipStream_t)stream, td_of(x), td_of(dst), order_desc);
    HIP_TRY(hipGetLastError()); return CDNA4_OK;
}

// ------------------------------------------------------------------------------------------------ SUM_ROWS
__global__ void __launch_bounds__(64) sum_rows_kernel(TD x, TD y) {
    const long r = blockIdx.x;
    const long i1 = r % x.ne[1], i2 = (r / x.ne[1]) % x.ne[2], i3 = r / (x.ne[1] * x.ne[2]);
    const char *xr = x.data + i1 * x.nb[1] + i2 * x.nb[2] + i3 * x.nb[3];
    float s = 0.f;
    for (long i = threadIdx.x; i < x.ne[0]; i += 64) s += *reinterpret_cast<const float *>(xr + i * x.nb[0]);
    s = wave_sum_dpp(s);
    if (threadIdx.x == 0) *reinterpret_cast<float *>(y.data + i1 * y.nb[1] + i2 * y.nb[2] + i3 * y.nb[3]) = s;
}
int cdna4_op_sum_rows(cdna4_context *ctx, const cdna4_tensor *x, const cdna4_tensor *dst, void *stream) {
    if (!ctx || !x || !dst) return cdna4_set_err(CDNA4_E_INVALID, "null argument");
    OP_CHECK(x->type == T_F32 && dst->type == T_F32 && dst->ne[0] == 1 && dst->ne[1] == x->ne[1] && dst->ne[2] == x->ne[2] && dst->ne[3] == x->ne[3], "sum_rows: f32, dst [1, rows]");
    if (td_nrows(x) == 0) return CDNA4_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(sum_rows_kernel, dim3((unsigned)td_nrows(x)), dim3(64), 0, (hipStream_t)stream, td_of(x), td_of(dst));
    HIP_TRY(hipGetLastError()); return CDNA4_OK;
}

// ------------------------------------------------------------------------------------------------ MUL_MULTI_ADD: dst[:, t] = sum_j a[:, j, t] * b[0, j, t]   (weighted sum of the used experts)
__global__ void mul_multi_add_kernel(TD a, TD b, TD d, TD r, int has_r) {        // has_r: + r[:, t] (the residual ADD that follows the experts' weighted sum)
    const long t = blockIdx.y;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < a.ne[0]; i += (long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (long j = 0; j < a.ne[1]; ++j) s += *reinterpret_cast<const float *>(a.data + i * a.nb[0] + j * a.nb[1] + t * a.nb[2]) * *reinterpret_cast<const float *>(b.data + j * b.nb[1] + t * b.nb[2]);
        if (has_r) s += *reinterpret_cast<const float *>(r.data + i * r.nb[0] + t * r.nb[1]);
        *reinterpret_cast<float *>(d.data + i * d.nb[0] + t * d.nb[1]) = s;
    }
}
int cdna4_op_mul_multi_add(cdna4_context *ctx, const cdna4_tensor *a, const cdna4_tensor *b, const cdna4_tensor *dst, void *stream) { return cdna4_op_mul_multi_add_res(ctx, a, b, nullptr, dst, stream); }
int cdna4_op_mul_multi_add_res(cdna4_context *ctx, const cdna4_tensor *a, const cdna4_tensor *b, const cdna4_tensor *res, const cdna4_tensor *dst, void *stream) {
    if (!ctx || !a || !b || !dst) return cdna4_set_err(CDNA4_E_INVALID, "null argument");
    OP_CHECK(a->type == T_F32 && b->type == T_F32 && dst->type == T_F32 && b->ne[0] == 1 && a->ne[1] == b->ne[1] && a->ne[2] == b->ne[2] && a->ne[3] == 1 && b->ne[3] == 1 &&
             dst->ne[0] == a->ne[0] && dst->ne[1] == a->ne[2] && a->ne[2] <= 65535, "mul_multi_add: shapes");
    OP_CHECK(!res || (res->type == T_F32 && res->ne[0] == dst->ne[0] && res->ne[1] == dst->ne[1] && res->ne[2] == 1 && res->ne[3] == 1), "mul_multi_add: residual shape");
    if (td_nelem(dst) == 0) return CDNA4_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    TD r; memset(&r, 0, sizeof(r)); if (res) r = td_of(res);
    hipLaunchKernelGGL(mul_multi_add_kernel, dim3((unsigned)std::min<long>((a->ne[0] + 255) / 256, 64), (unsigned)a->ne[2]), dim3(256), 0, (hipStream_t)stream, td_of(a), td_of(b), td_of(dst), r, res ? 1 : 0);
    HIP_TRY(hipGetLastError()); return CDNA4_OK;
}

// ------------------------------------------------------------------------------------------------ small dense MUL_MAT (f32 / f16 weights x f32 activations): the MoE router
// dst[m, n] = sum_k w[k, m] * x[k, n]; one workgroup per output element.  Only for small weights (n_expert rows); big dense GEMMs are not on this path.
template <typename W>
__global__ void __launch_bounds__(256) mul_mat_dense_kernel(TD w, TD x, TD d, long total) {
    __shared__ float red[4];
    const long o = blockIdx.x; if (o >= total) return;
    const long m = o % d.ne[0], n = o / d.ne[0];
    const char *wr = w.data + m * w.nb[1]; const char *xr = x.data + n * x.nb[1];
    float s = 0.f;
    for (long k = threadIdx.x; k < w.ne[0]; k += 256) s += cvt<W, float>(*reinterpret_cast<const W *>(wr + k * w.nb[0])) * *reinterpret_cast<const float *>(xr + k * x.nb[0]);
    s = block_sum256(s, red);
    if (threadIdx.x == 0) *reinterpret_cast<float *>(d.data + m * d.nb[0] + n * d.nb[1]) = s;
}
int cdna4_op_mul_mat_dense(cdna4_context *ctx, const cdna4_tensor *w, const cdna4_tensor *x, const cdna4_tensor *dst, void *stream) {
    if (!ctx || !w || !x || !dst) return cdna4_set_err(CDNA4_E_INVALID, "null argument");
    OP_CHECK((w->type == T_F32 || w->type == T_F16) && x->type == T_F32 && dst->type == T_F32 && w->ne[0] == x->ne[0] && dst->ne[0] == w->ne[1] && dst->ne[1] == x->ne[1] &&
             w->ne[2] == 1 && w->ne[3] == 1 && x->ne[2] == 1 && x->ne[3] == 1, "dense mul_mat: 2-D f32 / f16 weights, f32 activations");
    const long total = dst->ne[0] * dst->ne[1]; if (total == 0) return CDNA4_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    if (w->type == T_F32) hipLaunchKernelGGL(mul_mat_dense_kernel<float>, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream, td_of(w), td_of(x), td_of(dst), total);
    else hipLaunchKernelGGL(mul_mat_dense_kernel<__half>, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream, td_of(w), td_of(x), td_of(dst), total);
    HIP_TRY(hipGetLastError()); return CDNA4_OK;
}

// ------------------------------------------------------------------------------------------------ the MoE router of a token in ONE launch
// logits = W_router x  ->  probs = softmax(logits)  ->  argsort (descending)  ->  weights = probs[top n_used]  ->  sum  ->  weights / sum
// i.e. the six nodes MUL_MAT(f32) + SOFT_MAX + ARGSORT + GET_ROWS + SUM_ROWS + DIV that llm_build_moe_ffn emits for softmax gating with
// normalized weights (llama-build-context.cpp:1464-1556; the CUDA backend has the same fusion: ggml_cuda_op_topk_moe).  Every node's result is still
// written (they are tiny) unless its descriptor carries a null data pointer: the caller passes null for an intermediate whose memory the graph
// allocator has already handed to a LATER result of the chain (workgroups of different tokens are not ordered).  One workgroup per token; n_expert <= 64.
template <typename W>
__global__ void __launch_bounds__(256) moe_router_kernel(TD w, TD x, TD logits, TD probs, TD sorted, TD wsel, TD wsum, TD wnorm, int n_used, const float *norm_w, float norm_eps, TD xn) {
    __shared__ float s_part[4][64]; __shared__ float s_logit[64]; __shared__ float s_red[4];
    const long t = blockIdx.x, K = w.ne[0]; const int n_expert = (int)w.ne[1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const char *xr = x.data + t * x.nb[1];
    const bool vec = sizeof(W) == 4 && K % 1024 == 0 && K <= 8192 && x.nb[0] == 4 && w.nb[0] == 4 && (((uintptr_t)xr | (uintptr_t)w.data | (uintptr_t)w.nb[1]) % 16 == 0);
    for (int e0 = 0; e0 < n_expert; e0 += 8) {
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (vec) {          // all loads of a pass requested up front (a single workgroup: latency, not bandwidth, is what costs)
            const int nq = (int)(K >> 10);                          // float4 chunks per thread (<= 8)
            float4 xv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) if (q < nq) xv[q] = reinterpret_cast<const float4 *>(xr)[threadIdx.x + 256 * q];
            // norm_w != nullptr (host: vec layout, K <= 4096, n_expert <= 8): x is the UN-normed row; its RMS norm -- the FUSED_RMS_NORM node in front of the router, whose result the
            // experts read too -- runs here, on the chunks this thread holds anyway, in the partition and the summation order of rms_norm_kernel (float4 t + 256 p, block_sum256):
            // the normed row written to xn and the logits are bit-identical to the two launches
            auto norm_here = [&]() {
                float4 cw[4]; float ss = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) if (q < nq) { cw[q] = reinterpret_cast<const float4 *>(norm_w)[threadIdx.x + 256 * q]; ss += xv[q].x * xv[q].x + xv[q].y * xv[q].y + xv[q].z * xv[q].z + xv[q].w * xv[q].w; }
                const float scale = 1.0f / sqrtf(block_sum256(ss, s_red) / (float)K + norm_eps);
                float4 *xo = reinterpret_cast<float4 *>(xn.data + t * xn.nb[1]);
#pragma unroll
                for (int q = 0; q < 4; ++q) if (q < nq) {
                    float4 v = xv[q]; v.x = scale * cw[q].x * v.x; v.y = scale * cw[q].y * v.y; v.z = scale * cw[q].z * v.z; v.w = scale * cw[q].w * v.w;
                    xv[q] = v; if (xn.data) xo[threadIdx.x + 256 * q] = v;
                }
            };
            if (nq == 4) {      // K = 4096: 32 weight loads per thread, unconditional (expert index clamped) so that they are all in flight together
                float4 wv[8][4];
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float4 *wr = reinterpret_cast<const float4 *>(w.data + (long)min(e0 + j, n_expert - 1) * w.nb[1]);
#pragma unroll
                    for (int q = 0; q < 4; ++q) wv[j][q] = wr[threadIdx.x + 256 * q]; }
                if (norm_w) norm_here();
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[j] += wv[j][q].x * xv[q].x + wv[j][q].y * xv[q].y + wv[j][q].z * xv[q].z + wv[j][q].w * xv[q].w;
            } else {
                if (norm_w) norm_here();
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float4 *wr = reinterpret_cast<const float4 *>(w.data + (long)min(e0 + j, n_expert - 1) * w.nb[1]);
#pragma unroll
                    for (int q = 0; q < 8; ++q) if (q < nq) { const float4 wv = wr[threadIdx.x + 256 * q]; acc[j] += wv.x * xv[q].x + wv.y * xv[q].y + wv.z * xv[q].z + wv.w * xv[q].w; } }
            }
        } else {
            for (long k = threadIdx.x; k < K; k += 256) {
                const float xv = *reinterpret_cast<const float *>(xr + k * x.nb[0]);
#pragma unroll
                for (int j = 0; j < 8; ++j) if (e0 + j < n_expert) acc[j] = fmaf(cvt<W, float>(*reinterpret_cast<const W *>(w.data + (long)(e0 + j) * w.nb[1] + k * w.nb[0])), xv, acc[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float v = wave_sum_dpp(acc[j]); if (lane == 0 && e0 + j < n_expert) s_part[wave][e0 + j] = v; }
    }
    __syncthreads();
    if (threadIdx.x < n_expert) s_logit[threadIdx.x] = (s_part[0][threadIdx.x] + s_part[1][threadIdx.x]) + (s_part[2][threadIdx.x] + s_part[3][threadIdx.x]);
    __syncthreads();
    if (wave != 0) return;
    const bool live = lane < n_expert;
    const float l = live ? s_logit[lane] : -INFINITY;
    if (live && logits.data) *reinterpret_cast<float *>(logits.data + lane * logits.nb[0] + t * logits.nb[1]) = l;
    const float m = wave_max(l), pe = live ? expf(l - m) : 0.f, p = pe / wave_sum_dpp(pe);
    if (live && probs.data) *reinterpret_cast<float *>(probs.data + lane * probs.nb[0] + t * probs.nb[1]) = p;
    int rank = 0;                                                   // position of expert `lane` in the descending order; ties: HIGHER index first, like argsort_kernel above and the
                                                                    // reference's std::greater on (value, index) pairs (iqk_cpu_ops.cpp iqk_argsort) -- the unfused chain picks the same experts
    for (int j = 0; j < n_expert; ++j) { const float pj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p), j)); rank += (pj > p || (pj == p && j > lane)) ? 1 : 0; }
    if (live && sorted.data) *reinterpret_cast<int32_t *>(sorted.data + rank * sorted.nb[0] + t * sorted.nb[1]) = lane;
    const bool sel = live && rank < n_used;
    const float ws = wave_sum_dpp(sel ? p : 0.f);
    if (sel) {
        if (wsel.data) *reinterpret_cast<float *>(wsel.data + rank * wsel.nb[1] + t * wsel.nb[2]) = p;                 // GET_ROWS result [1, n_used, n_tok]
        *reinterpret_cast<float *>(wnorm.data + rank * wnorm.nb[0] + t * wnorm.nb[1]) = p / ws;          // DIV result [n_used, n_tok]
    }
    if (lane == 0 && wsum.data) *reinterpret_cast<float *>(wsum.data + t * wsum.nb[1]) = ws;                          // SUM_ROWS result [1, n_tok]
}
int cdna4_op_moe_router(cdna4_context *ctx, const cdna4_tensor *w, const cdna4_tensor *x, const cdna4_tensor *logits, const cdna4_tensor *probs, const cdna4_tensor *sorted,
                        const cdna4_tensor *wsel, const cdna4_tensor *wsum, const cdna4_tensor *wnorm, int n_used, void *stream) {
    return cdna4_op_moe_router_norm(ctx, w, x, nullptr, 0.f, nullptr, logits, probs, sorted, wsel, wsum, wnorm, n_used, stream);
}
int cdna4_op_moe_router_norm(cdna4_context *ctx, const cdna4_tensor *w, const cdna4_tensor *x, const cdna4_tensor *norm_w, float norm_eps, const cdna4_tensor *x_normed, const cdna4_tensor *logits,
                             const cdna4_tensor *probs, const cdna4_tensor *sorted, const cdna4_tensor *wsel, const cdna4_tensor *wsum, const cdna4_tensor *wnorm, int n_used, void *stream) {
    if (!ctx || !w || !x || !logits || !probs || !sorted || !wsel || !wsum || !wnorm) return cdna4_set_err(CDNA4_E_INVALID, "null argument");
    if (norm_w) {       // the norm rides only on the vector path of the kernel, one pass over the experts, the row in the four chunks a thread holds
        OP_CHECK(x_normed && w->type == T_F32 && x->type == T_F32 && norm_w->type == T_F32 && x_normed->type == T_F32 && w->ne[1] <= 8 && w->ne[0] % 1024 == 0 && w->ne[0] <= 4096 && x->nb[0] == 4 && w->nb[0] == 4 &&
                 norm_w->ne[0] == w->ne[0] && norm_w->nb[0] == 4 && td_nrows(norm_w) == 1 && same_shape(x, x_normed) && x_normed->nb[0] == 4 &&
                 (((uintptr_t)x->data | (uintptr_t)w->data | (uintptr_t)norm_w->data | (uintptr_t)x_normed->data | (uintptr_t)x->nb[1] | (uintptr_t)w->nb[1] | (uintptr_t)x_normed->nb[1]) % 16) == 0,
                 "moe_router_norm: f32 router of <= 8 experts, rows of 1024 ... 4096 values, 16-byte aligned");
    }
    const long n_expert = w->ne[1], n_tok = x->ne[1];
    OP_CHECK((w->type == T_F32 || w->type == T_F16) && x->type == T_F32 && w->ne[0] == x->ne[0] && w->ne[2] == 1 && w->ne[3] == 1 && x->ne[2] == 1 && x->ne[3] == 1 && n_expert >= 1 && n_expert <= 64 &&
             n_used >= 1 && n_used <= n_expert, "moe_router: 2-D f32 / f16 router weights, <= 64 experts");
    OP_CHECK(logits->type == T_F32 && probs->type == T_F32 && sorted->type == 26 && wsel->type == T_F32 && wsum->type == T_F32 && wnorm->type == T_F32 &&
             logits->ne[0] == n_expert && logits->ne[1] == n_tok && probs->ne[0] == n_expert && probs->ne[1] == n_tok && sorted->ne[0] == n_expert && sorted->ne[1] == n_tok &&
             wsel->ne[0] == 1 && wsel->ne[1] == n_used && wsel->ne[2] == n_tok && wsum->ne[0] == 1 && wsum->ne[1] == n_tok && wnorm->ne[0] == n_used && wnorm->ne[1] == n_tok, "moe_router: result shapes");
    if (n_tok == 0) return CDNA4_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    TD xn; memset(&xn, 0, sizeof(xn)); if (norm_w) xn = td_of(x_normed);
    const float *nw = norm_w ? (const float *)norm_w->data : nullptr;
    if (w->type == T_F32) hipLaunchKernelGGL(moe_router_kernel<float>, dim3((unsigned)n_tok), dim3(256), 0, (hipStream_t)stream, td_of(w), td_of(x), td_of(logits), td_of(probs), td_of(sorted), td_of(wsel), td_of(wsum), td_of(wnorm), n_used, nw, norm_eps, xn);
    else hipLaunchKernelGGL(moe_router_kernel<__half>, dim3((unsigned)n_tok), dim3(256), 0, (hipStream_t)stream, td_of(w), td_of(x), td_of(logits), td_of(probs), td_of(sorted), td_of(wsel), td_of(wsum), td_of(wnorm), n_used, nw, norm_eps, xn);
    HIP_TRY(hipGetLastError()); return CDNA4_OK;
}

// ------------------------------------------------------------------------------------------------ FLASH_ATTN_EXT (f16 K / V cache, f32 Q, f16 mask, GQA, scale / softcap / ALiBi)
// dst[:, h, t] = softmax_j(scale * q[:, t, h] . k[:, j, h / gqa] + slope * mask[j, t]) . v[:, j, h / gqa]      (ggml.c:22874-23160)
// One workgroup (4 waves) per (query row, head).  Wave w walks KV tiles of 64 positions (tile index = w, w+4, ...): lane j owns position j of
// the tile for the score (full q.k dot in f32, q broadcast from LDS) and dims (2 lane, 2 lane + 1) of the accumulator for P.V (V rows
// are read coalesced, p_j broadcast by readlane).  Online softmax per wave, the four waves' (m, l, acc) are merged through LDS at the end.
// Head size D = 64 (one accumulator dim per lane: V rows read as 64 x f16), 128 or 256 (Llama: 128).
template <int D>
__global__ void __launch_bounds__(256) flash_attn_vec_kernel(TD q, TD k, TD v, TD mask, int has_mask, TD dst, float scale, float softcap, float max_bias, float m0, float m1, unsigned n_head_log2) {
    constexpr int DP = D / 64;                 // accumulator dims per lane (pairs of f16 per lane in a V row = DP / 2 dwords ... D = 128: 2 dims; D = 64: one f16)
    __shared__ float s_m[4], s_l[4]; __shared__ float s_acc[4][D];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long t = blockIdx.x, h = blockIdx.y, b3 = blockIdx.z;
    const long hk = h / (q.ne[2] / k.ne[2]), hv = h / (q.ne[2] / v.ne[2]), b3k = b3 / (q.ne[3] / k.ne[3]), b3v = b3 / (q.ne[3] / v.ne[3]);
    const long n_kv = k.ne[1];
    // q row stays f32, in LDS: every lane reads the same address (broadcast, conflict-free)
    __shared__ __attribute__((aligned(16))) float s_q[D];
    const float *qr = reinterpret_cast<const float *>(q.data + t * q.nb[1] + h * q.nb[2] + b3 * q.nb[3]);
    for (int d = threadIdx.x; d < D; d += 256) s_q[d] = qr[d];
    __syncthreads();
    const float slope = max_bias > 0.0f ? ((unsigned)h < n_head_log2 ? powf(m0, (float)(h + 1)) : powf(m1, (float)(2 * (h - n_head_log2) + 1))) : 1.0f;
    const __half *mrow = has_mask ? reinterpret_cast<const __half *>(mask.data + t * mask.nb[1] + (h % mask.ne[2]) * mask.nb[2] + (b3 % mask.ne[3]) * mask.nb[3]) : nullptr;
    const char *kbase = k.data + hk * k.nb[2] + b3k * k.nb[3]; const char *vbase = v.data + hv * v.nb[2] + b3v * v.nb[3];
    float M = -INFINITY, L = 0.f, acc[DP];
#pragma unroll
    for (int i = 0; i < DP; ++i) acc[i] = 0.f;
    for (long j0 = 64L * wave; j0 < n_kv; j0 += 256) {
        const long j = j0 + lane;
        float s = -INFINITY;
        if (j < n_kv) {
            const float mv = mrow ? slope * __half2float(mrow[j]) : 0.0f;
            if (mv != -INFINITY) {
                const uint4 *kr = reinterpret_cast<const uint4 *>(kbase + j * k.nb[1]);
                float dot = 0.f;
#pragma unroll
                for (int i = 0; i < D / 8; ++i) {
                    const uint4 kk = kr[i]; const __half2 *kh = reinterpret_cast<const __half2 *>(&kk);
                    const float4 qa = reinterpret_cast<const float4 *>(s_q)[2 * i], qb = reinterpret_cast<const float4 *>(s_q)[2 * i + 1];
                    const float2 k0 = __half22float2(kh[0]), k1 = __half22float2(kh[1]), k2 = __half22float2(kh[2]), k3 = __half22float2(kh[3]);
                    dot = fmaf(qa.x, k0.x, dot); dot = fmaf(qa.y, k0.y, dot); dot = fmaf(qa.z, k1.x, dot); dot = fmaf(qa.w, k1.y, dot);
                    dot = fmaf(qb.x, k2.x, dot); dot = fmaf(qb.y, k2.y, dot); dot = fmaf(qb.z, k3.x, dot); dot = fmaf(qb.w, k3.y, dot);
                }
                s = softcap == 0.0f ? dot * scale + mv : softcap * tanhf(dot * scale) + mv;
            }
        }
        const float tile_max = wave_max(s);
        if (tile_max == -INFINITY) continue;                                       // fully masked tile (wave-uniform)
        const float Mn = fmaxf(M, tile_max), corr = expf(M - Mn);                    // (M = -inf: corr = 0, acc and L are 0 anyway)
        const float p = s == -INFINITY ? 0.f : expf(s - Mn);
        L = L * corr + wave_sum_dpp(p);
#pragma unroll
        for (int i = 0; i < DP; ++i) acc[i] *= corr;
        M = Mn;
        const long jn = min(64L, n_kv - j0);
        // P.V: 8 V rows in flight per step (masked positions carry p = 0: their rows are still valid cache memory, the value is dropped by the select)
        for (long jj = 0; jj < jn; jj += 8) {
            float pj[8]; __half2 vv[8][DP / 2 > 0 ? DP / 2 : 1]; __half v1[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                pj[u] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p), (int)((jj + u) & 63)));
                const long row = j0 + min(jj + u, jn - 1);
                if constexpr (D == 64) v1[u] = reinterpret_cast<const __half *>(vbase + row * v.nb[1])[lane];
                else {
                    const __half2 *vr = reinterpret_cast<const __half2 *>(vbase + row * v.nb[1]);
#pragma unroll
                    for (int i = 0; i < DP / 2; ++i) vv[u][i] = vr[lane + 64 * i];
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float w = jj + u < jn ? pj[u] : 0.f;
                if constexpr (D == 64) acc[0] = w == 0.f ? acc[0] : fmaf(w, __half2float(v1[u]), acc[0]);
                else {
#pragma unroll
                    for (int i = 0; i < DP / 2; ++i) { const float2 f = __half22float2(vv[u][i]); acc[2 * i] = w == 0.f ? acc[2 * i] : fmaf(w, f.x, acc[2 * i]); acc[2 * i + 1] = w == 0.f ? acc[2 * i + 1] : fmaf(w, f.y, acc[2 * i + 1]); }
                }
            }
        }
    }
    // merge the four waves
    if (lane == 0) { s_m[wave] = M; s_l[wave] = L; }
    __syncthreads();
    const float Mg = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
    const float mine = M == -INFINITY ? 0.f : expf(M - Mg);
    if constexpr (D == 64) s_acc[wave][lane] = acc[0] * mine;
    else {
#pragma unroll
        for (int i = 0; i < DP / 2; ++i) { s_acc[wave][2 * (lane + 64 * i)] = acc[2 * i] * mine; s_acc[wave][2 * (lane + 64 * i) + 1] = acc[2 * i + 1] * mine; }
    }
    __syncthreads();
    float Lg = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) Lg += s_m[w] == -INFINITY ? 0.f : s_l[w] * expf(s_m[w] - Mg);
    const float inv = Lg == 0.0f ? 0.0f : 1.0f / Lg;
    // permuted store: dst[:, h, t] (ggml.c:23157: (i3*ne2*ne1 + i2 + i1*ne1)*nb1)
    float *out = reinterpret_cast<float *>(dst.data + (b3 * dst.ne[2] * dst.ne[1] + h + t * dst.ne[1]) * dst.nb[1]);
    for (int d = threadIdx.x; d < D; d += 256) out[d] = (s_acc[0][d] + s_acc[1][d] + s_acc[2][d] + s_acc[3][d]) * inv;
}
// Head size 128, a few query rows (decode): same workgroup shape and arithmetic as flash_attn_vec_kernel, but a wave requests EVERYTHING it needs of a
// 64-position tile up front -- its K row (16 x 16 B per lane), the mask value and its two accumulator dims of all 64 V rows (64 x 4 B per lane) --
// before it waits for q, so a tile costs one memory round trip instead of three dependent ones (q, then K, then V).
// FAST (the default since round 4 -- validated on an MI355X: identical results, -17 % per launch): the row addresses of the K / V loads in 32-bit offsets from wave-uniform bases.  The ISA of the
// default form spends 288 v_mul_lo_u32 + 224 v_mad_u64_u32 + 76 v_mul_hi_u32 (quarter-rate) and 525 v_cndmask on `min(j0 + u, n_kv - 1) * nb[1]` in 64 bits for its 80 loads
// -- about 600 quarter-rate instructions in front of the first load of a kernel that runs 9.7 us per layer.  FAST computes one 24-bit multiply per lane and tile (K) or per
// wave and row on the scalar unit (V: rows are wave-uniform), clamps OFFSETS instead of rows (the map row -> offset is monotonic) and hands the loads an SGPR base + a 32-bit
// lane offset.  Host-side guard: nb[1] < 2^24 and n_kv * nb[1] < 2^32 (a cache view of 4 GiB per head), else the default form is launched.
template <bool FAST>
__global__ void __launch_bounds__(256) flash_attn_decode_kernel(TD q, TD k, TD v, TD mask, int has_mask, TD dst, float scale, float softcap, float max_bias, float m0, float m1, unsigned n_head_log2) {
    __shared__ float s_m[4], s_l[4]; __shared__ float s_acc[4][128];
    fa_decode_body<FAST, false>(q, k, v, mask, has_mask, dst, scale, softcap, max_bias, m0, m1, n_head_log2, blockIdx.x, blockIdx.y, blockIdx.z, s_m, s_l, s_acc);
}
// the round-4 form: every tile's keys spread over the four waves, fully masked chunks skipped (fa_decode.cuh)
template <int NW>
__global__ void __launch_bounds__(64 * NW) flash_attn_decode2_kernel(TD q, TD k, TD v, TD mask, int has_mask, TD dst, float scale, float softcap, float max_bias, float m0, float m1, unsigned n_head_log2, uint8_t *q8) {
    __shared__ float s_m[NW], s_l[NW]; __shared__ float s_acc[NW][128];
    // every kernel argument the body reads is requested HERE, in one batch of scalar loads behind one wait: left alone, hipcc loads a tensor's fields where a branch first needs them
    // -- five dependent kernarg round trips in front of the first K load (fa_timeline_probe: -0.25 us per launch)
#define PIN_S(x_) asm volatile("" :: "s"(x_))
    PIN_S(q.data); PIN_S(q.ne[2]); PIN_S(q.ne[3]); PIN_S(q.nb[1]); PIN_S(q.nb[2]); PIN_S(q.nb[3]);
    PIN_S(k.data); PIN_S(k.ne[1]); PIN_S(k.ne[2]); PIN_S(k.ne[3]); PIN_S(k.nb[1]); PIN_S(k.nb[2]); PIN_S(k.nb[3]);
    PIN_S(v.data); PIN_S(v.ne[2]); PIN_S(v.ne[3]); PIN_S(v.nb[1]); PIN_S(v.nb[2]); PIN_S(v.nb[3]);
    PIN_S(mask.data); PIN_S(mask.ne[2]); PIN_S(mask.ne[3]); PIN_S(mask.nb[1]); PIN_S(mask.nb[2]); PIN_S(mask.nb[3]);
    PIN_S(dst.data); PIN_S(dst.ne[1]); PIN_S(dst.ne[2]); PIN_S(dst.nb[1]); PIN_S(scale); PIN_S(q8);
#undef PIN_S
    fa_decode_body_v2<false, NW>(q, k, v, mask, has_mask, dst, scale, softcap, max_bias, m0, m1, n_head_log2, blockIdx.x, blockIdx.y, blockIdx.z, s_m, s_l, s_acc, q8);
}
// waves per head: 4 (64-cell tiles) for windows under 512 cells, 8 (128-cell tiles) from there on -- the launch is bound by the memory round trips of ONE workgroup per head, and
// twice the waves keep twice the rows in flight (scripts/probes/fa_timeline_probe.hip, 600 visible of 768 cells: 17.5 -> 12.2 us; 100 of 256: 6.25 vs 5.7 us, 30 of 256: 5.0 vs 5.25 us;
// llama-bench tg128 / tg512 of the 8B model do not tell the two apart; 16 waves cost more to launch and merge than they gain: tg128 -3 %)
#define LAUNCH_DECODE2(NKV_, GRID_, ST_, ...) do { \
    if ((NKV_) >= 512) hipLaunchKernelGGL(flash_attn_decode2_kernel<8>, GRID_, dim3(512), 0, ST_, __VA_ARGS__); \
    else hipLaunchKernelGGL(flash_attn_decode2_kernel<4>, GRID_, dim3(256), 0, ST_, __VA_ARGS__); } while (0)
// Split-KV form ("flash decoding"): a workgroup = one KV head x one chunk of the context, its waves = the q heads that share that KV head (GQA group, <= 8): they request the
// same K / V rows, so the chunk leaves L2 once per workgroup (the other waves hit the CU's L1).  One CU pulls ~10 B/clk; with a whole head's context on one workgroup the
// attention of a long context was bound by that (n_kv = 8192: 4 MiB per workgroup), and even at n_kv = 256 the 128 KiB per workgroup cost more than the arithmetic.
// n_splits > 1: every wave writes its partial (max, sum, 128 accumulators) to `part`; the workgroup that arrives last at the KV head's counter combines them (and re-arms
// the counter for the next launch -- HIP-graph replays included).
struct FaSplit { float *part; unsigned *counters; int n_splits, chunk, fenced; };       // part: [token][q head][split][130]; chunk = keys per split (multiple of 64)
// the end of a split workgroup: write the result (one split) or hand the wave's partial to the workgroup that arrives last at the KV head's counter, which combines
__device__ __forceinline__ void fa_split_finish(const TD &q, const TD &k, const TD &dst, const FaSplit &sp, float M, float L, float acc0, float acc1, long h, long t, long hk, long b3, long split, int lane, int *s_last_p, bool valid = true) {      // valid == false: a wave without a head of its own (fa_decode_mfma.cuh) keeps the barriers company
#define s_last (*s_last_p)
    // permuted store: dst[:, h, t] (ggml.c:23157: (i3*ne2*ne1 + i2 + i1*ne1)*nb1)
    float *out = reinterpret_cast<float *>(dst.data + (b3 * dst.ne[2] * dst.ne[1] + h + t * dst.ne[1]) * dst.nb[1]);
    if (sp.n_splits == 1) {
        const float inv = L == 0.0f ? 0.0f : 1.0f / L;
        if (valid) reinterpret_cast<float2 *>(out)[lane] = make_float2(acc0 * inv, acc1 * inv);
        return;
    }
    const long n_head = q.ne[2], n_tok = q.ne[1];
    float *mine = sp.part + ((((b3 * n_tok + t) * n_head + h) * sp.n_splits) + split) * 130;
    // Hand-off of the partials to the workgroup that arrives last at the KV head's counter (MI355X guide, Guideline 16 R1 / "in-launch split-K reduction"): the 520-byte
    // partial of a wave goes out as 8-byte WRITE-THROUGH stores (relaxed agent-scope atomic stores lower to `global_store_dwordx2 ... sc1`), every wave drains its stores
    // (vmcnt(0): they have left the XCD), then ONE relaxed agent-scope ticket per workgroup; the last arriver reads with sc1 loads.  No release / acquire fence: a fence is a
    // whole-L2 write-back (1.7 us per side, the round-2/3 form below cost ~4 us per launch) for 2 KB of payload.  sp.fenced = the old form (A/B: CDNA4_FA_SPLIT_FENCE=1).
    unsigned *cnt = sp.counters + ((b3 * n_tok + t) * k.ne[2] + hk);
    if (!sp.fenced) {
        if (valid) __hip_atomic_store(reinterpret_cast<unsigned long long *>(mine) + lane, ((unsigned long long)__float_as_uint(acc1) << 32) | __float_as_uint(acc0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (valid && lane == 0) __hip_atomic_store(reinterpret_cast<unsigned long long *>(mine) + 64, ((unsigned long long)__float_as_uint(L) << 32) | __float_as_uint(M), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = old == (unsigned)sp.n_splits - 1;
            if (s_last) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // re-armed for the next launch (graph replays included)
        }
        __syncthreads();
        if (!s_last || !valid) return;
        // combine (n_splits <= 64): lane i owns split i's (max, sum); the accumulators are summed 8 splits at a time with all loads in flight
        const unsigned long long *p0 = reinterpret_cast<const unsigned long long *>(sp.part + (((b3 * n_tok + t) * n_head + h) * sp.n_splits) * 130);      // 65 granules per split
        const unsigned long long ml = lane < sp.n_splits ? __hip_atomic_load(p0 + (long)lane * 65 + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        const float ms = lane < sp.n_splits ? __uint_as_float((unsigned)ml) : -INFINITY, ls = lane < sp.n_splits ? __uint_as_float((unsigned)(ml >> 32)) : 0.f;
        const float Mg = wave_max(ms);
        const float w = ms == -INFINITY ? 0.f : expf(ms - Mg);
        const float Lg = wave_sum_dpp(w * ls);
        float o0 = 0.f, o1 = 0.f;
        for (int s0 = 0; s0 < sp.n_splits; s0 += 8) {
            unsigned long long a8[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) a8[i] = __hip_atomic_load(p0 + (long)min(s0 + i, sp.n_splits - 1) * 65 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int i = 0; i < 8; ++i) { const float wi = s0 + i < sp.n_splits ? __shfl(w, s0 + i, 64) : 0.f; o0 = fmaf(wi, __uint_as_float((unsigned)a8[i]), o0); o1 = fmaf(wi, __uint_as_float((unsigned)(a8[i] >> 32)), o1); }
        }
        const float inv = Lg == 0.0f ? 0.0f : 1.0f / Lg;
        reinterpret_cast<float2 *>(out)[lane] = make_float2(o0 * inv, o1 * inv);
        return;
    }
    if (valid) reinterpret_cast<float2 *>(mine)[lane] = make_float2(acc0, acc1);
    if (valid && lane == 0) { mine[128] = M; mine[129] = L; }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = atomicAdd(cnt, 1u);
        s_last = old == (unsigned)sp.n_splits - 1;
        if (s_last) (void)atomicExch(cnt, 0u);                                    // re-armed for the next launch
    }
    __syncthreads();
    if (!s_last || !valid) return;
    __threadfence();
    // combine (n_splits <= 64): lane i owns split i's (max, sum); the accumulators are summed 8 splits at a time with all loads in flight
    const float *p0 = sp.part + (((b3 * n_tok + t) * n_head + h) * sp.n_splits) * 130;
    const float ms = lane < sp.n_splits ? p0[lane * 130 + 128] : -INFINITY, ls = lane < sp.n_splits ? p0[lane * 130 + 129] : 0.f;
    const float Mg = wave_max(ms);
    const float w = ms == -INFINITY ? 0.f : expf(ms - Mg);
    const float Lg = wave_sum_dpp(w * ls);
    float o0 = 0.f, o1 = 0.f;
    for (int s0 = 0; s0 < sp.n_splits; s0 += 8) {
        float2 a8[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) a8[i] = reinterpret_cast<const float2 *>(p0 + (long)min(s0 + i, sp.n_splits - 1) * 130)[lane];
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float wi = s0 + i < sp.n_splits ? __shfl(w, s0 + i, 64) : 0.f; o0 = fmaf(wi, a8[i].x, o0); o1 = fmaf(wi, a8[i].y, o1); }
    }
    const float inv = Lg == 0.0f ? 0.0f : 1.0f / Lg;
    reinterpret_cast<float2 *>(out)[lane] = make_float2(o0 * inv, o1 * inv);
#undef s_last
}
template <bool FAST>           // FAST: the addressing and the pinned prologue order of flash_attn_decode_kernel<true> (same knob, same host guard)
__global__ void __launch_bounds__(512) flash_attn_split_kernel(TD q, TD k, TD v, TD mask, int has_mask, TD dst, float scale, float softcap, float max_bias, float m0, float m1, unsigned n_head_log2, FaSplit sp) {
    __shared__ int s_last;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, part = lane & 15;
    const int G = (int)(q.ne[2] / k.ne[2]);                                   // q heads per KV head = waves of this workgroup
    const long split = blockIdx.x % sp.n_splits, t = blockIdx.x / sp.n_splits, hk = blockIdx.y, b3 = blockIdx.z;
    if constexpr (FAST) {       // the round-5 tile loop (fa_decode.cuh), then the hand-off below
        const FaSplitOut r5 = fa_split_tiles_v2(q, k, v, mask, has_mask, scale, softcap, max_bias, m0, m1, n_head_log2, sp.n_splits, sp.chunk);
        fa_split_finish(q, k, dst, sp, r5.M, r5.L, r5.acc0, r5.acc1, r5.h, t, hk, b3, split, lane, &s_last);
        return;
    }
    const long h = hk * G + wave, hv = h / (q.ne[2] / v.ne[2]), b3k = b3 / (q.ne[3] / k.ne[3]), b3v = b3 / (q.ne[3] / v.ne[3]);
    const long n_kv = k.ne[1], j_begin = split * sp.chunk, j_end = min(n_kv, j_begin + sp.chunk);
    const float slope = max_bias > 0.0f ? ((unsigned)h < n_head_log2 ? powf(m0, (float)(h + 1)) : powf(m1, (float)(2 * (h - n_head_log2) + 1))) : 1.0f;
    const __half *mrow = has_mask ? reinterpret_cast<const __half *>(mask.data + t * mask.nb[1] + (h % mask.ne[2]) * mask.nb[2] + (b3 % mask.ne[3]) * mask.nb[3]) : nullptr;
    const char *kbase = k.data + hk * k.nb[2] + b3k * k.nb[3]; const char *vbase = v.data + hv * v.nb[2] + b3v * v.nb[3];
    // K tile of 64 keys per wave, COALESCED: load i brings keys j0 + 16 * (lane / 16) + i, lane % 16 = the 16-byte piece of the 256-byte row (4 rows = 8 cache lines
    // per instruction; a lane-per-key layout touches 64 lines per instruction).  The 16 partial dots of a lane are summed over its 16-lane row by a reduce-scatter
    // (4 DPP exchange steps, 15 adds) that leaves the score of key j0 + lane in lane `lane`.
    uint4 kreg[16]; __half2 vreg[64]; __half mreg;
    const unsigned knb1 = (unsigned)k.nb[1], vnb1 = (unsigned)v.nb[1], klast = (unsigned)(n_kv - 1) * knb1, vlast = (unsigned)(n_kv - 1) * vnb1;      // (FAST; the guard keeps these in 32 bits)
    auto load_k = [&](long j0) {             // (pipelined like flash_attn_decode_kernel: K of the next tile after the dots, V after the P V products)
        if constexpr (FAST) {
            const unsigned o0 = __umul24((unsigned)j0 + 16u * (unsigned)(lane >> 4), knb1) + 16u * (unsigned)part;
#pragma unroll
            for (int i = 0; i < 16; ++i) kreg[i] = *reinterpret_cast<const uint4 *>(kbase + min(o0 + (unsigned)i * knb1, klast + 16u * (unsigned)part));
            mreg = mrow ? mrow[min((int)j0 + lane, (int)n_kv - 1)] : __float2half(0.f);
        } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) kreg[i] = reinterpret_cast<const uint4 *>(kbase + min(j0 + 16 * (lane >> 4) + i, n_kv - 1) * k.nb[1])[part];
        mreg = mrow ? mrow[min(j0 + lane, n_kv - 1)] : __float2half(0.f);
        }
    };
    auto load_v = [&](long j0) {
        if constexpr (FAST) {
            const unsigned r0 = (unsigned)__builtin_amdgcn_readfirstlane((int)j0) * vnb1;
#pragma unroll
            for (int u = 0; u < 64; ++u) vreg[u] = reinterpret_cast<const __half2 *>(vbase + min(r0 + (unsigned)u * vnb1, vlast))[lane];
        } else {
#pragma unroll
        for (int u = 0; u < 64; ++u) vreg[u] = reinterpret_cast<const __half2 *>(vbase + min(j0 + u, n_kv - 1) * v.nb[1])[lane];
        }
    };
    long j0 = j_begin;
    const float4 *qr = reinterpret_cast<const float4 *>(q.data + t * q.nb[1] + h * q.nb[2] + b3 * q.nb[3]);
    const float4 qa = qr[2 * part], qb = qr[2 * part + 1];
    if (j0 < j_end) { load_k(j0); load_v(j0); }
    if constexpr (FAST) __builtin_amdgcn_sched_barrier(0);
    float M = -INFINITY, L = 0.f, acc0 = 0.f, acc1 = 0.f;
    while (j0 < j_end) {
        const long j = j0 + lane;
        float r[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const __half2 *kh = reinterpret_cast<const __half2 *>(&kreg[i]);
            const float2 k0 = __half22float2(kh[0]), k1 = __half22float2(kh[1]), k2 = __half22float2(kh[2]), k3 = __half22float2(kh[3]);
            float d = qa.x * k0.x; d = fmaf(qa.y, k0.y, d); d = fmaf(qa.z, k1.x, d); d = fmaf(qa.w, k1.y, d);
            d = fmaf(qb.x, k2.x, d); d = fmaf(qb.y, k2.y, d); d = fmaf(qb.z, k3.x, d); d = fmaf(qb.w, k3.y, d);
            r[i] = d;
        }
        {   const bool c3 = lane & 8, c2 = lane & 4, c1 = lane & 2, c0 = lane & 1;
#pragma unroll
            for (int i = 0; i < 8; ++i) r[i] = (c3 ? r[i + 8] : r[i]) + fa_dpp<0x140>(c3 ? r[i] : r[i + 8]);          // row_mirror: partner lane ^ 15
#pragma unroll
            for (int i = 0; i < 4; ++i) r[i] = (c2 ? r[i + 4] : r[i]) + fa_dpp<0x141>(c2 ? r[i] : r[i + 4]);          // row_half_mirror: lane ^ 7
#pragma unroll
            for (int i = 0; i < 2; ++i) r[i] = (c1 ? r[i + 2] : r[i]) + fa_dpp<0x4e>(c1 ? r[i] : r[i + 2]);           // quad_perm [2,3,0,1]: lane ^ 2
            r[0] = (c0 ? r[1] : r[0]) + fa_dpp<0xb1>(c0 ? r[0] : r[1]);                                                // quad_perm [1,0,3,2]: lane ^ 1
        }
        const float dot = r[0];
        float s = -INFINITY;
        const float mv = slope * __half2float(mreg);
        if (j0 + 64 < j_end) load_k(j0 + 64);
        if (j < j_end && mv != -INFINITY) s = softcap == 0.0f ? dot * scale + mv : softcap * tanhf(dot * scale) + mv;        // (a masked cell's row may hold anything)
        const float tile_max = wave_max(s);
        if (tile_max != -INFINITY) {                                               // (wave-uniform) not a fully masked tile
            const float Mn = fmaxf(M, tile_max), corr = expf(M - Mn);
            const float p = s == -INFINITY ? 0.f : expf(s - Mn);
            L = L * corr + wave_sum_dpp(p);
            acc0 *= corr; acc1 *= corr; M = Mn;
#pragma unroll
            for (int u = 0; u < 64; ++u) {
                const float pj = lane_bcast(p, u);
                const float2 f = __half22float2(vreg[u]);
                acc0 = pj == 0.f ? acc0 : fmaf(pj, f.x, acc0); acc1 = pj == 0.f ? acc1 : fmaf(pj, f.y, acc1);      // (p = 0: the cache cell may hold anything)
            }
        }
        j0 += 64;
        if (j0 < j_end) load_v(j0);
    }
    fa_split_finish(q, k, dst, sp, M, L, acc0, acc1, h, t, hk, b3, split, lane, &s_last);
}
// Per-head kernel below this many keys, split-KV kernel from it on.  Measured on an MI355X with the write-through hand-off (scripts/r04_fa.sh, 32 q / 8 KV heads, us per launch,
// per-head vs split): 256 keys 6.7 vs 7.9, 512: 9.4 vs 8.1, 768: 11.9 vs 8.9, 1024: 14.6 vs 10.6, 4096: 45.5 vs 14.2 (the fenced hand-off of rounds 2-3: 35.4)
constexpr long FA_SPLIT_MIN_KV_DEFAULT = 384;
// the K / V views fit 32-bit row offsets (otherwise the 64-bit addressing of flash_attn_decode_kernel<false>)
static bool fa_fast_addr(const cdna4_tensor *k, const cdna4_tensor *v) {
    return k->nb[1] > 0 && v->nb[1] > 0 && k->nb[1] < (1 << 24) && v->nb[1] < (1 << 24) && k->ne[1] < (1 << 24) &&
           (uint64_t)(k->ne[1] + 320) * (uint64_t)k->nb[1] < (1ull << 32) && (uint64_t)(k->ne[1] + 320) * (uint64_t)v->nb[1] < (1ull << 32);     // (+ 320: a tile's rows are clamped AFTER the multiply)
}
// would cdna4_op_flash_attn run this attention on the per-head decode kernel with the 32-bit addressing (the form gemv_attn.hip embeds)?  Same argument checks.
bool cdna4_fa_is_plain_decode(const cdna4_context *ctx, const cdna4_tensor *q, const cdna4_tensor *k, const cdna4_tensor *v, const cdna4_tensor *mask, const cdna4_tensor *dst) {
    if (!ctx || !q || !k || !v || !dst) return false;
    const long D = q->ne[0];
    if (!(q->type == T_F32 && k->type == T_F16 && v->type == T_F16 && dst->type == T_F32 && k->ne[0] == D && v->ne[0] == D && dst->ne[0] == D && D == 128)) return false;
    if (!(q->nb[0] == 4 && k->nb[0] == 2 && v->nb[0] == 2 && dst->nb[0] == 4 && k->nb[1] % 16 == 0 && v->nb[1] % 4 == 0 && ((uintptr_t)k->data % 16 == 0) && ((uintptr_t)v->data % 4 == 0) && k->nb[2] % 16 == 0 && k->nb[3] % 16 == 0)) return false;
    if (!(k->ne[1] == v->ne[1] && k->ne[1] >= 1 && q->ne[2] % k->ne[2] == 0 && q->ne[2] % v->ne[2] == 0 && q->ne[3] == 1 && k->ne[3] == 1 && v->ne[3] == 1 && dst->ne[1] == q->ne[2] && dst->ne[2] == q->ne[1] && q->ne[1] == 1)) return false;
    if (mask && !(mask->type == T_F16 && mask->nb[0] == 2 && mask->ne[0] >= k->ne[1] && mask->ne[1] >= q->ne[1])) return false;
    if ((uintptr_t)q->data % 16 || q->nb[2] % 16 || (uintptr_t)dst->data % 8 || dst->nb[1] != 128 * 4) return false;      // (q rows as float4; the result row contiguous: it IS the mat-vec's activation row)
    constexpr bool no_decode_kernel = false;
    static const long split_min_kv = getenv("CDNA4_FA_SPLIT_MIN_KV") ? atol(getenv("CDNA4_FA_SPLIT_MIN_KV")) : FA_SPLIT_MIN_KV_DEFAULT;
    return !no_decode_kernel && k->ne[1] < split_min_kv && fa_fast_addr(k, v);
}
static int flash_attn_f16(cdna4_context *ctx, const cdna4_tensor *q, const cdna4_tensor *k, const cdna4_tensor *v, const cdna4_tensor *mask, const cdna4_tensor *dst,
                          float scale, float max_bias, float softcap, void *stream, size_t ws_off);
// Q8_0 K / V (-ctk q8_0 -ctv q8_0; ggml-cuda.cu:5152-5157, fattn.cu): the views are de-quantized into dense f16 copies at the head of the workspace (one launch each; decode at
// depth n reads 1.06 n D bytes per KV head and writes 2 n D -- the price of keeping the attention on the device with the f16 kernels; a kernel that de-quantizes in registers is the
// next step), then the f16 attention runs on the copies with its own scratch behind them.
int cdna4_op_flash_attn(cdna4_context *ctx, const cdna4_tensor *q, const cdna4_tensor *k, const cdna4_tensor *v, const cdna4_tensor *mask, const cdna4_tensor *dst,
                        float scale, float max_bias, float softcap, void *stream) {
    if (!ctx || !q || !k || !v || !dst) return cdna4_set_err(CDNA4_E_INVALID, "null argument");
    if (k->type != T_Q8_0 && v->type != T_Q8_0) return flash_attn_f16(ctx, q, k, v, mask, dst, scale, max_bias, softcap, stream, 0);
    cdna4_tensor kk = *k, vv = *v; size_t off = 0; const cdna4_tensor *src[2] = {k, v}; cdna4_tensor *cp[2] = {&kk, &vv};
    size_t bytes[2] = {0, 0};
    for (int i = 0; i < 2; ++i) if (src[i]->type == T_Q8_0) {
        OP_CHECK(src[i]->ne[0] % 32 == 0 && src[i]->nb[0] == 34, "flash_attn: Q8_0 K / V rows of whole blocks");
        bytes[i] = ((size_t)td_nelem(src[i]) * sizeof(__half) + 255) & ~(size_t)255;
    }
    // everything the f16 path may want behind the copies (prompt: its V^T scratch; decode: split partials for up to 64 splits), so that it never re-allocates the workspace under them
    const size_t rest = std::max<size_t>(cdna4_flash_attn_mfma_workspace(k) * 2, (size_t)q->ne[3] * q->ne[1] * q->ne[2] * 64 * 130 * sizeof(float)) + (1 << 20);
    { const int rc = cdna4_ensure_ws(ctx, bytes[0] + bytes[1] + rest, (hipStream_t)stream); if (rc) return rc; }
    HIP_TRY(hipSetDevice(ctx->device));
    for (int i = 0; i < 2; ++i) if (bytes[i]) {
        __half *d16 = (__half *)((char *)ctx->ws + off); const long nb = td_nelem(src[i]) / 32;
        hipLaunchKernelGGL(q8_0_rows_to_f16_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, (hipStream_t)stream, td_of(src[i]), d16, nb);
        cp[i]->data = d16; cp[i]->type = T_F16; cp[i]->nb[0] = 2; cp[i]->nb[1] = src[i]->ne[0] * 2; cp[i]->nb[2] = cp[i]->nb[1] * src[i]->ne[1]; cp[i]->nb[3] = cp[i]->nb[2] * src[i]->ne[2];
        off += bytes[i];
    }
    HIP_TRY(hipGetLastError());
    return flash_attn_f16(ctx, q, &kk, &vv, mask, dst, scale, max_bias, softcap, stream, off);
}
static int flash_attn_f16(cdna4_context *ctx, const cdna4_tensor *q, const cdna4_tensor *k, const cdna4_tensor *v, const cdna4_tensor *mask, const cdna4_tensor *dst,
                          float scale, float max_bias, float softcap, void *stream, size_t ws_off) {
    const long D = q->ne[0];
    OP_CHECK(q->type == T_F32 && k->type == T_F16 && v->type == T_F16 && dst->type == T_F32 && k->ne[0] == D && v->ne[0] == D && dst->ne[0] == D && (D == 64 || D == 128 || D == 256),
             "flash_attn: f32 Q, f16 K / V, head size 64 / 128 / 256");
    OP_CHECK(q->nb[0] == 4 && k->nb[0] == 2 && v->nb[0] == 2 && dst->nb[0] == 4 && k->nb[1] % 16 == 0 && v->nb[1] % 4 == 0 && ((uintptr_t)k->data % 16 == 0) && ((uintptr_t)v->data % 4 == 0) &&
             k->nb[2] % 16 == 0 && k->nb[3] % 16 == 0, "flash_attn: row alignment");
    OP_CHECK(k->ne[1] == v->ne[1] && q->ne[2] % k->ne[2] == 0 && q->ne[2] % v->ne[2] == 0 && q->ne[3] % k->ne[3] == 0 && q->ne[3] % v->ne[3] == 0 && dst->ne[1] == q->ne[2] && dst->ne[2] == q->ne[1] &&
             q->ne[2] <= 65535 && q->ne[3] <= 65535, "flash_attn: shapes");
    OP_CHECK(!mask || (mask->type == T_F16 && mask->nb[0] == 2 && mask->ne[0] >= k->ne[1] && mask->ne[1] >= q->ne[1]), "flash_attn: f16 mask [n_kv, n_tokens]");
    if (td_nelem(q) == 0) return CDNA4_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    if (softcap != 0.0f) scale /= softcap;
    // prompt batches: the matrix-core kernel (flash_attn.hip)
    static const bool no_mfma_attn = getenv("CDNA4_FA_NO_MFMA") != nullptr;          // (developer bisect knob: prompt batches through the per-head vector kernels)
    if (!no_mfma_attn && D == 128 && q->ne[1] >= 16 && k->ne[1] % 64 == 0 && k->ne[2] == v->ne[2] && k->ne[3] == v->ne[3] && q->nb[1] % 16 == 0 && q->nb[2] % 16 == 0 && q->nb[3] % 16 == 0 && (uintptr_t)q->data % 16 == 0 &&
        dst->nb[1] % 16 == 0 && (uintptr_t)dst->data % 16 == 0 && (!mask || (mask->nb[1] % 16 == 0 && mask->nb[2] % 16 == 0 && mask->nb[3] % 16 == 0 && (uintptr_t)mask->data % 16 == 0))) {
        const int rc = cdna4_ensure_ws(ctx, ws_off + cdna4_flash_attn_mfma_workspace(k), (hipStream_t)stream); if (rc) return rc;
        return cdna4_launch_flash_attn_mfma(q, k, v, mask, dst, (char *)ctx->ws + ws_off, scale, max_bias, softcap, (hipStream_t)stream);
    }
    const unsigned n_head_log2 = 1u << (unsigned)floorf(log2f((float)q->ne[2]));
    const float m0 = powf(2.0f, -max_bias / n_head_log2), m1 = powf(2.0f, -(max_bias / 2.0f) / n_head_log2);
    TD m; memset(&m, 0, sizeof(m)); if (mask) m = td_of(mask); else { m.ne[2] = m.ne[3] = 1; }
    const dim3 grid((unsigned)q->ne[1], (unsigned)q->ne[2], (unsigned)q->ne[3]); hipStream_t st = (hipStream_t)stream;
    constexpr bool no_decode_kernel = false;
    const long G = q->ne[2] / k->ne[2];
    // short contexts: one workgroup per q head (its 4 waves split the keys, no cross-workgroup combine: the arrival counter + fences of the split form cost ~4 us);
    // from CDNA4_FA_SPLIT_MIN_KV keys on (default 1024) the split-KV form
    static const long split_min_kv = getenv("CDNA4_FA_SPLIT_MIN_KV") ? atol(getenv("CDNA4_FA_SPLIT_MIN_KV")) : FA_SPLIT_MIN_KV_DEFAULT;
    // (the arrival counters of the split form are allocated ONCE per context at a fixed capacity -- a captured launch keeps their address, so they must never move --
    // and a batch that would need more of them takes the per-head kernel below)
    if (D == 128 && !no_decode_kernel && !ctx->selftest_unsplit && k->ne[1] >= split_min_kv && G <= 8 && k->ne[2] == v->ne[2] && k->ne[2] <= 65535 && dst->nb[1] % 8 == 0 && (uintptr_t)dst->data % 8 == 0 &&
        q->nb[1] % 16 == 0 && q->nb[2] % 16 == 0 && q->nb[3] % 16 == 0 && (uintptr_t)q->data % 16 == 0 &&
        (size_t)q->ne[3] * q->ne[1] * k->ne[2] * sizeof(unsigned) <= ctx->fa_counters_bytes - 64) {      // (the last 64 bytes: gemv_attn.hip's tickets)
        // splits: enough workgroups to spread the context over the chip (~2 per CU), chunks of >= 64 keys
        static const int env_splits = getenv("CDNA4_FA_SPLITS") ? atoi(getenv("CDNA4_FA_SPLITS")) : 0;
        const long base_wgs = q->ne[1] * k->ne[2] * q->ne[3], tiles = (k->ne[1] + 63) / 64;
        // measured (8B GQA 4, llama-bench -gp): 8192 keys 16 / 32 / 64 splits -> 280 / 299 / 272 tok/s, 2048 keys 376 / 378 / 336: <= 32 splits of >= 2 tiles (the pipelined loads need a
        // successor tile; every split costs the combining workgroup ~1 us)
        // short contexts (under 1024 keys: round 4, write-through hand-off): one 64-key tile per split -- the launch is bound by what ONE CU can pull (~10 B/clk), not by the chip
        long ns = env_splits ? env_splits : k->ne[1] < 1024 ? tiles : std::max<long>(1, std::min<long>({(tiles + 1) / 2, 32L, (2L * ctx->num_cu + base_wgs - 1) / base_wgs}));
        ns = std::min<long>(ns, 64);
        const long chunk = ((tiles + ns - 1) / ns) * 64; ns = (k->ne[1] + chunk - 1) / chunk;
        static const int env_fenced = getenv("CDNA4_FA_SPLIT_FENCE") ? atoi(getenv("CDNA4_FA_SPLIT_FENCE")) : 0;      // (developer A/B knob: the fenced hand-off of rounds 2-3)
        FaSplit sp; sp.part = nullptr; sp.counters = nullptr; sp.n_splits = (int)ns; sp.chunk = (int)chunk; sp.fenced = env_fenced || ctx->handoff >= 1;
        if (ns > 1) {
            const size_t part_bytes = (size_t)q->ne[3] * q->ne[1] * q->ne[2] * ns * 130 * sizeof(float);
            const int rc = cdna4_ensure_ws(ctx, ws_off + part_bytes, st); if (rc) return rc;
            sp.part = (float *)((char *)ctx->ws + ws_off); sp.counters = (unsigned *)ctx->fa_counters;
        }
        const dim3 g2((unsigned)(q->ne[1] * ns), (unsigned)k->ne[2], (unsigned)q->ne[3]);
        // (round 5: a GQA group's heads as the columns of 16x16x32 MFMAs -- K / V read once per workgroup, V through ds_read_b64_tr_b16, q and p as f16 hi + lo -- passed every test
        //  and was SLOWER than this per-wave form: 23.1 vs 20.4 us per layer at 8192 keys, tg at depth 8192 408 vs 422 tok/s, 8 waves 364; profiles/r05_notes.md section 7)
        if (fa_fast_addr(k, v)) hipLaunchKernelGGL(flash_attn_split_kernel<true>, g2, dim3(64 * (unsigned)G), 0, st, td_of(q), td_of(k), td_of(v), m, mask ? 1 : 0, td_of(dst), scale, softcap, max_bias, m0, m1, n_head_log2, sp);
        else hipLaunchKernelGGL(flash_attn_split_kernel<false>, g2, dim3(64 * (unsigned)G), 0, st, td_of(q), td_of(k), td_of(v), m, mask ? 1 : 0, td_of(dst), scale, softcap, max_bias, m0, m1, n_head_log2, sp);
    }
    else if (D == 128 && !no_decode_kernel) {
        if (fa_fast_addr(k, v)) LAUNCH_DECODE2(k->ne[1], grid, st, td_of(q), td_of(k), td_of(v), m, mask ? 1 : 0, td_of(dst), scale, softcap, max_bias, m0, m1, n_head_log2, (uint8_t *)nullptr);
        else hipLaunchKernelGGL(flash_attn_decode_kernel<false>, grid, dim3(256), 0, st, td_of(q), td_of(k), td_of(v), m, mask ? 1 : 0, td_of(dst), scale, softcap, max_bias, m0, m1, n_head_log2);
    }
    else if (D == 64) hipLaunchKernelGGL(flash_attn_vec_kernel<64>, grid, dim3(256), 0, st, td_of(q), td_of(k), td_of(v), m, mask ? 1 : 0, td_of(dst), scale, softcap, max_bias, m0, m1, n_head_log2);      // (the reference's HIP build takes head size 64 too, ggml-cuda.cu:5152-5157)
    else if (D == 128) hipLaunchKernelGGL(flash_attn_vec_kernel<128>, grid, dim3(256), 0, st, td_of(q), td_of(k), td_of(v), m, mask ? 1 : 0, td_of(dst), scale, softcap, max_bias, m0, m1, n_head_log2);
    else hipLaunchKernelGGL(flash_attn_vec_kernel<256>, grid, dim3(256), 0, st, td_of(q), td_of(k), td_of(v), m, mask ? 1 : 0, td_of(dst), scale, softcap, max_bias, m0, m1, n_head_log2);
    HIP_TRY(hipGetLastError()); return CDNA4_OK;
}

// One decoded token whose attention is the per-head decode kernel's case: the same launch, and every head's 128 results also as one block_q8_2_x4 of `q8_out`
// (row of n_head blocks = the attn_output mat-vec's activation row in its vec_dot type).  CDNA4_E_UNSUPPORTED otherwise: call cdna4_op_flash_attn.
int cdna4_op_flash_attn_q8(cdna4_context *ctx, const cdna4_tensor *q, const cdna4_tensor *k, const cdna4_tensor *v, const cdna4_tensor *mask, const cdna4_tensor *dst,
                           float scale, float max_bias, float softcap, void *q8_out, void *stream) {
    if (!ctx || !q || !k || !v || !dst || !q8_out) return cdna4_set_err(CDNA4_E_INVALID, "null argument");
    if (q->ne[1] != 1 || !cdna4_fa_is_plain_decode(ctx, q, k, v, mask, dst)) return cdna4_set_err(CDNA4_E_UNSUPPORTED, "flash_attn_q8: one token on the per-head decode kernel only");
    HIP_TRY(hipSetDevice(ctx->device));
    if (softcap != 0.0f) scale /= softcap;
    const unsigned n_head_log2 = 1u << (unsigned)floorf(log2f((float)q->ne[2]));
    const float m0 = powf(2.0f, -max_bias / n_head_log2), m1 = powf(2.0f, -(max_bias / 2.0f) / n_head_log2);
    TD m; memset(&m, 0, sizeof(m)); if (mask) m = td_of(mask); else { m.ne[2] = m.ne[3] = 1; }
    LAUNCH_DECODE2(k->ne[1], dim3(1, (unsigned)q->ne[2], 1), (hipStream_t)stream, td_of(q), td_of(k), td_of(v), m, mask ? 1 : 0, td_of(dst), scale, softcap, max_bias, m0, m1, n_head_log2, (uint8_t *)q8_out);
    HIP_TRY(hipGetLastError()); return CDNA4_OK;
}

// ------------------------------------------------------------------------------------------------ GET_ROWS
int cdna4_op_get_rows(cdna4_context *ctx, const cdna4_tensor *src, const cdna4_tensor *ids, const cdna4_tensor *dst, void *stream) {
    if (!ctx || !src || !ids || !dst) return cdna4_set_err(CDNA4_E_INVALID, "null argument");
    OP_CHECK(dst->type == T_F32 && ids->type == 26 /* I32 */ && dst->ne[0] == src->ne[0] && dst->ne[1] == ids->ne[0] && dst->ne[2] == ids->ne[1] && dst->ne[3] == ids->ne[2] && ids->ne[3] == 1,
             "get_rows: dst f32 [ne00, ids...], i32 ids");
    OP_CHECK(src->ne[2] % 1 == 0 && (src->ne[2] == 1 || src->ne[2] == ids->ne[1]) && (src->ne[3] == 1 || src->ne[3] == ids->ne[2]), "get_rows: src batch dims");
    if (td_nelem(dst) == 0) return CDNA4_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    return cdna4_launch_get_rows(ctx, src, ids, dst, (hipStream_t)stream);
}
