// flash_attn.hip -- FLASH_ATTN_EXT for prompt batches on the matrix cores (gfx950, v_mfma_f32_32x32x16_f16).
//
// What it replaces: ggml_compute_forward_flash_attn_ext_f16 (ggml.c:22874-23160) / iqk_flash_attn_noalibi (iqk_flash_attn.cpp) on the CPU,
// ggml-cuda/fattn*.cu on CUDA.  Same math: dst[:, h, t] = softmax_j(scale q_t.k_j [softcap] + slope mask[j, t]) . v_j, online softmax.
//
// Formulation (everything "transposed", so that a query is a COLUMN of every MFMA result and its running max / sum / rescale factor is a
// per-lane scalar):
//   S^T[key, q]  = K[key, :] . Q[q, :]        A = K rows (LDS, 16 B per lane),  B = Q fragments (f16, registers, loaded once per wave)
//   O^T[d,  q] += V^T[d, key] P^T[key, q]     A = V^T rows (LDS),               B = P fragments = the S^T accumulators, converted in place
// The C layout of the 32x32 MFMA gives lane (q = l % 32, g = l / 32) the rows 8 (i / 4) + 4 g + i % 4 of a 32-row block (i = 0..15); the
// B operand of the next MFMA wants k-slots 8 g + j (j = 0..7) per 16-k step.  Loading K row pi(m) = m with bits 2 and 3 swapped into
// A-row m makes the two agree: accumulator i = 8 s + j of lane g IS key 16 s + 8 g + j of the block -- no shuffles, no LDS round trip
// for P, mask values are two 16-byte loads per lane and block.
// V is needed k-major (8 consecutive keys per lane for one d): a pre-pass transposes the V view once per launch into the context
// workspace ([head_kv][d][n_kv] f16, coalesced both ways through LDS); K tiles and V^T tiles are then staged through LDS with plain
// 16-byte loads, prefetched one tile ahead into registers.
// One workgroup = 4 waves x 32 queries of one head; KV tiles of 64 keys; a wave skips the MFMAs of a 32-key block whose mask is -inf for
// all its queries (the causal upper triangle).  Head size 128.
#include "api_internal.h"
#include <hip/hip_fp16.h>
#include <cmath>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

namespace {
constexpr int D = 128, BQ = 32, NW = 4, BK = 64;
constexpr int KROW = D * 2 + 16;            // LDS row pitch of the K tile (bytes): 272 -> 16 consecutive rows hit 16 distinct 16-byte bank slots
constexpr int VROW = BK * 2 + 16;           // LDS row pitch of the V^T tile: 144

__device__ __forceinline__ int pi_row(int m) { return (m & ~12) | ((m & 4) << 1) | ((m & 8) >> 1); }      // swap bits 2 and 3

__global__ void __launch_bounds__(256) transpose_v_kernel(TD v, __half *vt, long n_kv) {
    // block = (64 keys, one kv head, one batch): V[key][d] -> vt[d][key]
    __shared__ __half tile[BK][D + 2];
    const long k0 = 64L * blockIdx.x, hk = blockIdx.y, b3 = blockIdx.z;
    const char *src = v.data + hk * v.nb[2] + b3 * v.nb[3];
    for (int c = threadIdx.x; c < BK * (D / 2); c += 256) {
        const int key = c / (D / 2), d2 = c % (D / 2);
        const __half2 x = k0 + key < n_kv ? *reinterpret_cast<const __half2 *>(src + (k0 + key) * v.nb[1] + d2 * 4) : __floats2half2_rn(0.f, 0.f);
        tile[key][2 * d2] = x.x; tile[key][2 * d2 + 1] = x.y;
    }
    __syncthreads();
    __half *dst = vt + ((b3 * gridDim.y + hk) * D) * n_kv;
    for (int c = threadIdx.x; c < D * (BK / 2); c += 256) {
        const int d = c / (BK / 2), k2 = c % (BK / 2);
        if (k0 + 2 * k2 < n_kv) *reinterpret_cast<__half2 *>(dst + (long)d * n_kv + k0 + 2 * k2) = __halves2half2(tile[2 * k2][d], tile[2 * k2 + 1][d]);
    }
}

struct FaArgs {
    TD q, k, mask, dst; const __half *vt; long n_kv; int has_mask; float scale, softcap, max_bias, m0, m1; unsigned n_head_log2;
};

__global__ void __launch_bounds__(256) flash_attn_mfma_kernel(const FaArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char k_lds[BK * KROW];
    __shared__ __attribute__((aligned(16))) unsigned char v_lds[D * VROW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = lane & 31, g = lane >> 5;
    const long h = blockIdx.y, b3 = blockIdx.z, n_tok = a.q.ne[1], n_kv = a.n_kv;
    const long hk = h / (a.q.ne[2] / a.k.ne[2]), b3k = b3 / (a.q.ne[3] / a.k.ne[3]);
    const long q0 = (long)blockIdx.x * (BQ * NW) + wave * BQ, qi = q0 + m, qc = qi < n_tok ? qi : n_tok - 1;       // this lane's query (clamped for loads)
    const bool wave_active = q0 < n_tok;
    // ---- Q fragments: B operand of S^T, lane (q, g) holds d = 16 step + 8 g .. + 8 for step = 0..7
    half8 qf[D / 16];
    {
        const float *qr = reinterpret_cast<const float *>(a.q.data + qc * a.q.nb[1] + h * a.q.nb[2] + b3 * a.q.nb[3]);
#pragma unroll
        for (int s = 0; s < D / 16; ++s) {
            const float4 x = *reinterpret_cast<const float4 *>(qr + 16 * s + 8 * g), y = *reinterpret_cast<const float4 *>(qr + 16 * s + 8 * g + 4);
            qf[s][0] = (_Float16)x.x; qf[s][1] = (_Float16)x.y; qf[s][2] = (_Float16)x.z; qf[s][3] = (_Float16)x.w;
            qf[s][4] = (_Float16)y.x; qf[s][5] = (_Float16)y.y; qf[s][6] = (_Float16)y.z; qf[s][7] = (_Float16)y.w;
        }
    }
    const float slope = a.max_bias > 0.0f ? ((unsigned)h < a.n_head_log2 ? powf(a.m0, (float)(h + 1)) : powf(a.m1, (float)(2 * (h - a.n_head_log2) + 1))) : 1.0f;
    const char *mrow = a.has_mask ? a.mask.data + qc * a.mask.nb[1] + (h % a.mask.ne[2]) * a.mask.nb[2] + (b3 % a.mask.ne[3]) * a.mask.nb[3] : nullptr;
    const char *kbase = a.k.data + hk * a.k.nb[2] + b3k * a.k.nb[3];
    const __half *vtbase = a.vt + ((b3k * a.k.ne[2] + hk) * D) * n_kv;

    float16v o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = 0.f;
    float M = -INFINITY, L = 0.f;

    // tile prefetch registers: thread t owns chunks t, t + 256, ... (4 of each tile)
    uint4 kreg[4], vreg[4];
    auto load_tile = [&](long k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + 256 * i;
            kreg[i] = *reinterpret_cast<const uint4 *>(kbase + (k0 + (c >> 4)) * a.k.nb[1] + (c & 15) * 16);
            vreg[i] = *reinterpret_cast<const uint4 *>(vtbase + (long)(c >> 3) * n_kv + k0 + (c & 7) * 8);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + 256 * i;
            *reinterpret_cast<uint4 *>(k_lds + (c >> 4) * KROW + (c & 15) * 16) = kreg[i];
            *reinterpret_cast<uint4 *>(v_lds + (c >> 3) * VROW + (c & 7) * 16) = vreg[i];
        }
    };
    load_tile(0);
    for (long k0 = 0; k0 < n_kv; k0 += BK) {
        __syncthreads();                    // every wave is done with the previous tile
        store_tile();
        __syncthreads();
        if (k0 + BK < n_kv) load_tile(k0 + BK);
        if (!wave_active) continue;
#pragma unroll
        for (int kb = 0; kb < BK / 32; ++kb) {
            // mask of this lane's 16 keys of the block: keys 8 g + j and 16 + 8 g + j
            float mv[16];
            if (mrow) {
                const uint4 m0 = *reinterpret_cast<const uint4 *>(mrow + (k0 + 32 * kb + 8 * g) * 2), m1 = *reinterpret_cast<const uint4 *>(mrow + (k0 + 32 * kb + 16 + 8 * g) * 2);
                const __half2 *h0 = reinterpret_cast<const __half2 *>(&m0), *h1 = reinterpret_cast<const __half2 *>(&m1);
#pragma unroll
                for (int j = 0; j < 4; ++j) { const float2 x = __half22float2(h0[j]), y = __half22float2(h1[j]); mv[2 * j] = slope * x.x; mv[2 * j + 1] = slope * x.y; mv[8 + 2 * j] = slope * y.x; mv[8 + 2 * j + 1] = slope * y.y; }
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) mv[i] = 0.f;
            }
            bool any = false;
#pragma unroll
            for (int i = 0; i < 16; ++i) any |= mv[i] != -INFINITY;
            if (qi >= n_tok) any = false;
            if (!__any(any)) continue;                                              // the whole 32 x 32 block is masked for this wave
            // ---- S^T = K Q^T
            float16v sacc = 0.f;
            const unsigned char *krow = k_lds + (32 * kb + pi_row(m)) * KROW + 16 * g;
#pragma unroll
            for (int s = 0; s < D / 16; ++s) {
                const half8 kf = *reinterpret_cast<const half8 *>(krow + 32 * s);
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[s], sacc, 0, 0, 0);
            }
            // ---- online softmax (per-lane scalars: the lane's query)
            float sv[16], mloc = -INFINITY;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float x = a.softcap == 0.0f ? sacc[i] * a.scale : a.softcap * tanhf(sacc[i] * a.scale);
                sv[i] = mv[i] == -INFINITY ? -INFINITY : x + mv[i]; mloc = fmaxf(mloc, sv[i]);
            }
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
            const float Mn = fmaxf(M, mloc), mref = Mn == -INFINITY ? 0.f : Mn;
            const float corr = __expf(M - mref);                                    // M = -inf -> 0 (o and L are 0 then)
            float psum = 0.f; half8 pf[2];
#pragma unroll
            for (int i = 0; i < 16; ++i) { const float p = __expf(sv[i] - mref); psum += p; pf[i >> 3][i & 7] = (_Float16)p; }
            L = L * corr + psum; M = Mn;
            if (__any(corr != 1.0f)) {
#pragma unroll
                for (int db = 0; db < 4; ++db) o[db] *= corr;
            }
            // ---- O^T += V^T P^T
            const unsigned char *vrow = v_lds + m * VROW + (32 * kb + 8 * g) * 2;
#pragma unroll
            for (int db = 0; db < 4; ++db) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const half8 vf = *reinterpret_cast<const half8 *>(vrow + 32 * db * VROW + 32 * s);
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[s], o[db], 0, 0, 0);
                }
            }
        }
    }
    if (!wave_active || qi >= n_tok) return;
    L += __shfl_xor(L, 32, 64);             // (both halves are active for the shuffle: qi depends on m only)
    const float inv = L == 0.0f ? 0.0f : 1.0f / L;
    // permuted store: dst[:, h, t] (ggml.c:23157); lane (q, g) holds d = 32 db + 8 (i / 4) + 4 g + i % 4
    float *out = reinterpret_cast<float *>(a.dst.data + (b3 * a.dst.ne[2] * a.dst.ne[1] + h + qi * a.dst.ne[1]) * a.dst.nb[1]);
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            *reinterpret_cast<float4 *>(out + 32 * db + 8 * r + 4 * g) = make_float4(o[db][4 * r] * inv, o[db][4 * r + 1] * inv, o[db][4 * r + 2] * inv, o[db][4 * r + 3] * inv);
}
}  // namespace

size_t cdna4_flash_attn_mfma_workspace(const cdna4_tensor *k) { return (size_t)k->ne[3] * k->ne[2] * D * k->ne[1] * sizeof(__half); }

// preconditions (checked by the caller): head size 128, f32 Q rows / f16 K, V rows, n_kv % 64 == 0, 16-byte aligned K rows, V^T image in `vt`
int cdna4_launch_flash_attn_mfma(const cdna4_tensor *q, const cdna4_tensor *k, const cdna4_tensor *v, const cdna4_tensor *mask, const cdna4_tensor *dst, void *vt,
                                 float scale, float max_bias, float softcap, hipStream_t st) {
    const long n_kv = k->ne[1];
    hipLaunchKernelGGL(transpose_v_kernel, dim3((unsigned)(n_kv / BK), (unsigned)v->ne[2], (unsigned)v->ne[3]), dim3(256), 0, st, td_of(v), (__half *)vt, n_kv);
    FaArgs a; a.q = td_of(q); a.k = td_of(k); a.dst = td_of(dst); a.vt = (const __half *)vt; a.n_kv = n_kv; a.has_mask = mask ? 1 : 0;
    if (mask) a.mask = td_of(mask); else { memset(&a.mask, 0, sizeof(a.mask)); a.mask.ne[2] = a.mask.ne[3] = 1; }
    a.scale = scale; a.softcap = softcap; a.max_bias = max_bias;
    a.n_head_log2 = 1u << (unsigned)floorf(log2f((float)q->ne[2]));
    a.m0 = powf(2.0f, -max_bias / a.n_head_log2); a.m1 = powf(2.0f, -(max_bias / 2.0f) / a.n_head_log2);
    hipLaunchKernelGGL(flash_attn_mfma_kernel, dim3((unsigned)((q->ne[1] + BQ * NW - 1) / (BQ * NW)), (unsigned)q->ne[2], (unsigned)q->ne[3]), dim3(256), 0, st, a);
    return hipGetLastError() == hipSuccess ? CDNA4_OK : cdna4_set_err(CDNA4_E_HIP, "flash_attn_mfma launch failed");
}
