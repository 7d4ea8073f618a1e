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
    // block = (64 keys, one kv head, one batch): V[key][d] -> vt[d][key].  16-byte loads (all four of a thread in flight), 2-byte scatter into the LDS tile [d][64 keys + 8],
    // 16-byte reads of 8 keys of one d, 16-byte stores (a row of the tile = one 128-byte line of vt).  The launch is latency-bound: 1 MB for a 512-token prompt.
    constexpr int PITCH = BK + 8;                                    // halves: 144-byte rows (16-byte aligned)
    __shared__ __attribute__((aligned(16))) __half tile[D * PITCH];
    const long k0 = 64L * blockIdx.x, hk = blockIdx.y, b3 = blockIdx.z;
    const char *src = v.data + hk * v.nb[2] + b3 * v.nb[3];
    const bool vec = (v.nb[1] % 16 == 0) && ((uintptr_t)src % 16 == 0);
    uint4 x[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = threadIdx.x + 256 * i, key = c >> 4, ch = c & 15;
        x[i] = make_uint4(0, 0, 0, 0);
        if (k0 + key < n_kv) {
            const char *p = src + (k0 + key) * v.nb[1] + ch * 16;
            if (vec) x[i] = *reinterpret_cast<const uint4 *>(p);
            else { const unsigned *q = reinterpret_cast<const unsigned *>(p); x[i] = make_uint4(q[0], q[1], q[2], q[3]); }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = threadIdx.x + 256 * i, key = c >> 4, ch = c & 15;
        const __half *hv = reinterpret_cast<const __half *>(&x[i]);
#pragma unroll
        for (int j = 0; j < 8; ++j) tile[(8 * ch + j) * PITCH + key] = hv[j];
    }
    __syncthreads();
    __half *dst = vt + ((b3 * gridDim.y + hk) * D) * n_kv;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = threadIdx.x + 256 * i, d = c >> 3, k8 = c & 7;                     // 8 consecutive lanes = the 64 keys of one d
        if (k0 + 8 * k8 < n_kv) *reinterpret_cast<uint4 *>(dst + (long)d * n_kv + k0 + 8 * k8) = *reinterpret_cast<const uint4 *>(tile + d * PITCH + 8 * k8);
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

// ---- the same attention with the KEYS of one 32-query block spread over the four waves of a workgroup -----------------------------------------------------------------
// The kernel above gives a wave 32 queries and ALL keys: a 512-token prompt is 128 workgroups of one wave per SIMD, the last of which walks 16 blocks of 32 keys one after the
// other behind two barriers per tile -- 54 us for 2 GFLOP, all of it latency.  Here one workgroup owns ONE 32-query block of one head and wave w the 32-key blocks w, w + 4, ...:
// four times the workgroups (2 per CU), a quarter of the chain per wave, no LDS tile and no barrier inside the loop -- each wave loads its K rows and V^T rows straight into
// the MFMA operand registers (16-byte loads, the next block's K and mask requested before the current block's arithmetic).  A wave first looks at the masks of eight of its
// blocks at once (so a causal tail costs one load latency per eight blocks, not one per block) and only walks the visible ones.  The four partial (max, sum, O^T) triples are
// combined through LDS in wave order (wave r finishes the 32 head dims r owns): deterministic, the same sums launch after launch.
// Workgroup -> (kv head, q head of the group, query block): kv head = id % n_head_kv, so with 8 kv heads every XCD (workgroup id % 8) keeps ONE head's K / V in its L2.
struct FaSplitArgs {
    TD q, k, mask, dst; const __half *vt; long n_kv; int has_mask, n_qblk, gqa; float scale, softcap, max_bias, m0, m1; unsigned n_head_log2;
    TD v;           // VDIRECT: the V view itself (rows of 128 halves, 16-byte aligned); vt is not used
};

template <int W> __device__ __forceinline__ void fa_merge(float16v (&o)[4], float M, float L, float (*s_o)[3][16][64], float (*s_m)[64], float (*s_l)[64], int lane,
                                                          float *out, int g, bool store) {
    s_m[W][lane] = M; s_l[W][lane] = L;
#pragma unroll
    for (int db = 0; db < 4; ++db) {
        if (db == W) continue;
#pragma unroll
        for (int i = 0; i < 16; ++i) s_o[W][db < W ? db : db - 1][i][lane] = o[db][i];
    }
    __syncthreads();
    float Ms = s_m[0][lane];
#pragma unroll
    for (int w = 1; w < 4; ++w) Ms = fmaxf(Ms, s_m[w][lane]);
    float f[4], Ls = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) { const float mw = s_m[w][lane]; f[w] = mw == -INFINITY ? 0.f : __expf(mw - Ms); Ls += f[w] * s_l[w][lane]; }
    const float inv = Ls == 0.0f ? 0.0f : 1.0f / Ls;
    float16v r;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        float x = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) x += f[w] * (w == W ? o[W][i] : s_o[w][W < w ? W : W - 1][i][lane]);          // wave order: the same sum every launch
        r[i] = x * inv;
    }
    if (!store) return;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
        *reinterpret_cast<float4 *>(out + 32 * W + 8 * rr + 4 * g) = make_float4(r[4 * rr], r[4 * rr + 1], r[4 * rr + 2], r[4 * rr + 3]);
}

// VDIRECT (round 5): V^T operands without the transposing pre-pass.  A wave loads the 32 V rows of its key block as they lie in the cache (16 bytes per lane: key 4 it + l / 16,
// dims 8 (l % 16) .. + 7), writes them into its own 8.25 KiB LDS image and reads the MFMA A operands back with ds_read_b64_tr_b16, which hands lane l % 16 of a 16-lane group
// column l % 16 of a [4 keys][16 dims] block: two reads give the 8 consecutive keys 16 s + 8 g + 0..7 of dim 32 db + l % 32 -- exactly what the half8 load from the V^T image gave.
// Image: unit (db, s, r) = the four [4][16] blocks (g, dim half) one read instruction of the wave touches, 512 contiguous bytes (conflict-free like a linear ds_read_b64), units
// of one db skewed by 64 bytes so that the 16-byte row writes of a wave spread over the banks.  The image lives in the memory of s_o (used only by the final merge: one barrier
// in front of it).  Layouts: scripts/probes/mfma_layout_probe.hip.
typedef short fa_s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned fa_vimg_off(int db, int s_, int r, int q4) { return (unsigned)((((db * 2 + s_) * 2 + r) * 512) + db * 64 + q4 * 128); }
template <bool SOFTCAP, bool VDIRECT>
__global__ void __launch_bounds__(256, 2) flash_attn_mfma_split_kernel(const FaSplitArgs a) {
    __shared__ __attribute__((aligned(16))) float s_o[4][3][16][64];
    __shared__ float s_m[4][64], s_l[4][64];
    __shared__ half8 s_q[D / 16 * 2 * 32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = lane & 31, g = lane >> 5;
    const long n_tok = a.q.ne[1], n_kv = a.n_kv, n_head_kv = a.k.ne[2];
    long id = blockIdx.x;
    const long hk = id % n_head_kv; id /= n_head_kv;
    const long hg = id % a.gqa; id /= a.gqa;
    const long qb = a.n_qblk - 1 - id % a.n_qblk, b3 = id / a.n_qblk;                 // the longest chains (last query blocks of a causal prompt) first
    const long h = hk * a.gqa + hg, b3k = b3 / (a.q.ne[3] / a.k.ne[3]);
    const long qi = qb * BQ + m, qc = qi < n_tok ? qi : n_tok - 1;
    // the query block as f16 B fragments, shared by the four waves through LDS: slot (s, g, q) = the 8 head dims 16 s + 8 g .. of query q (a wave reads 1 KiB in a row);
    // wave w converts the slices s = 2 w, 2 w + 1
    {
        const float *qr = reinterpret_cast<const float *>(a.q.data + qc * a.q.nb[1] + h * a.q.nb[2] + b3 * a.q.nb[3]);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int s = 2 * wave + s2;
            const float4 x = *reinterpret_cast<const float4 *>(qr + 16 * s + 8 * g), y = *reinterpret_cast<const float4 *>(qr + 16 * s + 8 * g + 4);
            half8 t;
            t[0] = (_Float16)x.x; t[1] = (_Float16)x.y; t[2] = (_Float16)x.z; t[3] = (_Float16)x.w; t[4] = (_Float16)y.x; t[5] = (_Float16)y.y; t[6] = (_Float16)y.z; t[7] = (_Float16)y.w;
            s_q[(2 * s + g) * 32 + m] = t;
        }
    }
    const half8 *qfp = s_q + g * 32 + m;                                               // + 64 s
    const float slope = a.max_bias > 0.0f ? ((unsigned)h < a.n_head_log2 ? powf(a.m0, (float)(h + 1)) : powf(a.m1, (float)(2 * (h - a.n_head_log2) + 1))) : 1.0f;
    const char *mrow = a.has_mask ? a.mask.data + qc * a.mask.nb[1] + (h % a.mask.ne[2]) * a.mask.nb[2] + (b3 % a.mask.ne[3]) * a.mask.nb[3] + 16 * g : nullptr;
    __syncthreads();
    const char *krow = a.k.data + hk * a.k.nb[2] + b3k * a.k.nb[3] + pi_row(m) * a.k.nb[1] + 16 * g;        // + key0 * nb1 + 32 s
    const char *vrow = VDIRECT ? a.v.data + hk * a.v.nb[2] + b3k * a.v.nb[3] + (lane >> 4) * a.v.nb[1] + 16 * (lane & 15)          // + (32 b + 4 it) rows
                               : reinterpret_cast<const char *>(a.vt + ((b3k * n_head_kv + hk) * D + m) * n_kv) + 16 * g;  // + 32 db * n_kv * 2 + key0 * 2 + 32 s
    const long vdb = 64 * n_kv;
    unsigned char *vimg = reinterpret_cast<unsigned char *>(&s_o[0][0][0][0]) + wave * 8448;       // (VDIRECT; 4 x 8448 <= sizeof(s_o))
    const bool q_ok = qi < n_tok;

    float16v o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = 0.f;
    float M = -INFINITY, L = 0.f;

    const long nblk = n_kv / 32;
    constexpr int CH = 8;
    for (long c0 = wave; c0 < nblk; c0 += 4 * CH) {
        unsigned act = 0;
        if (mrow) {
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const long b = c0 + 4 * i;
                bool any = false;
                if (b < nblk) {
                    const uint4 x = *reinterpret_cast<const uint4 *>(mrow + b * 64), y = *reinterpret_cast<const uint4 *>(mrow + b * 64 + 32);
                    const unsigned ws[8] = {x.x, x.y, x.z, x.w, y.x, y.y, y.z, y.w};
#pragma unroll
                    for (int j = 0; j < 8; ++j) any |= ws[j] != 0xfc00fc00u;
                }
                if (__any(any && q_ok)) act |= 1u << i;
            }
        } else {
#pragma unroll
            for (int i = 0; i < CH; ++i) if (c0 + 4 * i < nblk) act |= 1u << i;
        }
        if (!act) continue;
        half8 kf[D / 16]; uint4 mk0 = {0, 0, 0, 0}, mk1 = {0, 0, 0, 0};
        auto load_k = [&](long b, half8 (&kk)[D / 16], uint4 &x, uint4 &y) {
            const char *kp = krow + 32 * b * a.k.nb[1];
#pragma unroll
            for (int s = 0; s < D / 16; ++s) kk[s] = *reinterpret_cast<const half8 *>(kp + 32 * s);
            if (mrow) { x = *reinterpret_cast<const uint4 *>(mrow + b * 64); y = *reinterpret_cast<const uint4 *>(mrow + b * 64 + 32); }
        };
        load_k(c0 + 4 * __builtin_ctz(act), kf, mk0, mk1);
        while (act) {
            const long b = c0 + 4 * __builtin_ctz(act); act &= act - 1;
            // the next visible block's K rows and mask first; this block's V^T rows are requested behind the S^T MFMAs (into the registers its K rows leave) and consumed
            // behind the soft-max
            half8 kn[D / 16]; uint4 mn0 = mk0, mn1 = mk1;
            if (act) load_k(c0 + 4 * __builtin_ctz(act), kn, mn0, mn1);
            float16v sacc = 0.f;
#pragma unroll
            for (int s = 0; s < D / 16; ++s) sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[s], qfp[64 * s], sacc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            half8 vf[8];
            if constexpr (VDIRECT) {
                const char *vp = vrow + 32 * b * a.v.nb[1];
#pragma unroll
                for (int it = 0; it < 8; ++it) vf[it] = *reinterpret_cast<const half8 *>(vp + 4 * it * a.v.nb[1]);       // row 4 it + l / 16 of the block, 16-byte piece l % 16
            } else {
                const char *vp = vrow + 64 * b;
#pragma unroll
                for (int db = 0; db < 4; ++db)
#pragma unroll
                    for (int s = 0; s < 2; ++s) vf[2 * db + s] = *reinterpret_cast<const half8 *>(vp + db * vdb + 32 * s);
            }
            __builtin_amdgcn_sched_barrier(0);
            float mv[16];
            if (mrow) {
                const __half2 *h0 = reinterpret_cast<const __half2 *>(&mk0), *h1 = reinterpret_cast<const __half2 *>(&mk1);
#pragma unroll
                for (int j = 0; j < 4; ++j) { const float2 x = __half22float2(h0[j]), y = __half22float2(h1[j]); mv[2 * j] = slope * x.x; mv[2 * j + 1] = slope * x.y; mv[8 + 2 * j] = slope * y.x; mv[8 + 2 * j + 1] = slope * y.y; }
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) mv[i] = 0.f;
            }
            float sv[16], mloc = -INFINITY;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float x = !SOFTCAP ? sacc[i] * a.scale : a.softcap * tanhf(sacc[i] * a.scale);
                sv[i] = (mv[i] == -INFINITY || !q_ok) ? -INFINITY : x + mv[i]; mloc = fmaxf(mloc, sv[i]);
            }
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
            const float Mn = fmaxf(M, mloc), mref = Mn == -INFINITY ? 0.f : Mn;
            const float corr = __expf(M - mref);
            float psum = 0.f; half8 pf[2];
#pragma unroll
            for (int i = 0; i < 16; ++i) { const float p = __expf(sv[i] - mref); psum += p; pf[i >> 3][i & 7] = (_Float16)p; }
            L = L * corr + psum; M = Mn;
            if (__any(corr != 1.0f)) {
#pragma unroll
                for (int db = 0; db < 4; ++db) o[db] *= corr;
            }
            if constexpr (VDIRECT) {
                // rows -> image: piece hl = l % 16 (dims 8 hl ..) of key 4 it + l / 16 lands in unit (db = hl / 4, s = it / 4, r = it % 2), block (g = (it / 2) % 2, dim half (hl / 2) % 2), row l / 16
                const int hl = lane & 15, j4 = lane >> 4;
#pragma unroll
                for (int it = 0; it < 8; ++it)
                    *reinterpret_cast<half8 *>(vimg + fa_vimg_off(hl >> 2, it >> 2, it & 1, ((it >> 1) & 1) * 2 + ((hl >> 1) & 1)) + j4 * 32 + (hl & 1) * 16) = vf[it];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // (this wave's own region: no barrier)
                const int q4 = lane >> 4;
#pragma unroll
                for (int db = 0; db < 4; ++db)
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const fa_s16x4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) fa_s16x4 *)(vimg + fa_vimg_off(db, s, 0, q4) + 8 * hl));
                        const fa_s16x4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) fa_s16x4 *)(vimg + fa_vimg_off(db, s, 1, q4) + 8 * hl));
                        uint4 vb; { const uint2 x0 = __builtin_bit_cast(uint2, r0), x1 = __builtin_bit_cast(uint2, r1); vb.x = x0.x; vb.y = x0.y; vb.z = x1.x; vb.w = x1.y; }
                        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, vb), pf[s], o[db], 0, 0, 0);
                    }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // (the reads are done before the next block's rows are written)
            } else {
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int s = 0; s < 2; ++s) o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[2 * db + s], pf[s], o[db], 0, 0, 0);
            }
            if (act) {
#pragma unroll
                for (int s = 0; s < D / 16; ++s) kf[s] = kn[s];
                mk0 = mn0; mk1 = mn1;
            }
        }
    }
    L += __shfl_xor(L, 32, 64);
    if constexpr (VDIRECT) __syncthreads();          // the V images of the other waves live in s_o: nobody merges before everybody has left the loop
    float *out = reinterpret_cast<float *>(a.dst.data + (b3 * a.dst.ne[2] * a.dst.ne[1] + h + (q_ok ? qi : 0) * a.dst.ne[1]) * a.dst.nb[1]);
    switch (wave) {                         // (compile-time register indices for the wave's own 32 head dims)
        case 0: fa_merge<0>(o, M, L, s_o, s_m, s_l, lane, out, g, q_ok); break;
        case 1: fa_merge<1>(o, M, L, s_o, s_m, s_l, lane, out, g, q_ok); break;
        case 2: fa_merge<2>(o, M, L, s_o, s_m, s_l, lane, out, g, q_ok); break;
        default: fa_merge<3>(o, M, L, s_o, s_m, s_l, lane, out, g, q_ok); break;
    }
}
}  // namespace

int cdna4_flash_attn_preload(void) { hipFuncAttributes at; return hipFuncGetAttributes(&at, (const void *)flash_attn_mfma_split_kernel<false, true>) == hipSuccess ? 0 : -2; }
size_t cdna4_flash_attn_mfma_workspace(const cdna4_tensor *k) { return (size_t)k->ne[3] * k->ne[2] * D * k->ne[1] * sizeof(__half); }

// preconditions (checked by the caller): head size 128, f32 Q rows / f16 K, V rows, n_kv % 64 == 0, 16-byte aligned K rows, V^T image in `vt`
int cdna4_launch_flash_attn_mfma(const cdna4_tensor *q, const cdna4_tensor *k, const cdna4_tensor *v, const cdna4_tensor *mask, const cdna4_tensor *dst, void *vt,
                                 float scale, float max_bias, float softcap, hipStream_t st) {
    const long n_kv = k->ne[1];
    const long n_qblk0 = (q->ne[1] + BQ - 1) / BQ;
    // the split kernel reads V in place when its rows are 16-byte aligned f16 rows (every llama KV cache view); the transposing pre-pass serves the other layouts and the fallback kernel
    const bool v_direct = n_qblk0 * q->ne[2] * q->ne[3] <= 0x7fffffffL && v->nb[0] == 2 && v->nb[1] % 16 == 0 && v->nb[2] % 16 == 0 && v->nb[3] % 16 == 0 && ((uintptr_t)v->data % 16) == 0 &&
                          v->ne[2] == k->ne[2] && v->ne[3] == k->ne[3];
    if (!v_direct) hipLaunchKernelGGL(transpose_v_kernel, dim3((unsigned)(n_kv / BK), (unsigned)v->ne[2], (unsigned)v->ne[3]), dim3(256), 0, st, td_of(v), (__half *)vt, n_kv);
    FaArgs a; a.q = td_of(q); a.k = td_of(k); a.dst = td_of(dst); a.vt = (const __half *)vt; a.n_kv = n_kv; a.has_mask = mask ? 1 : 0;
    if (mask) a.mask = td_of(mask); else { memset(&a.mask, 0, sizeof(a.mask)); a.mask.ne[2] = a.mask.ne[3] = 1; }
    a.scale = scale; a.softcap = softcap; a.max_bias = max_bias;
    a.n_head_log2 = 1u << (unsigned)floorf(log2f((float)q->ne[2]));
    a.m0 = powf(2.0f, -max_bias / a.n_head_log2); a.m1 = powf(2.0f, -(max_bias / 2.0f) / a.n_head_log2);
    const long n_qblk = (q->ne[1] + BQ - 1) / BQ, n_wg = n_qblk * q->ne[2] * q->ne[3];
    if (n_wg <= 0x7fffffffL) {          // (beyond 2^31 workgroups: the round-2 kernel below, one wave per 32 queries, its grid is three-dimensional)
        FaSplitArgs s; s.q = a.q; s.k = a.k; s.mask = a.mask; s.dst = a.dst; s.vt = a.vt; s.n_kv = n_kv; s.has_mask = a.has_mask; s.n_qblk = (int)n_qblk; s.gqa = (int)(q->ne[2] / k->ne[2]);
        s.scale = a.scale; s.softcap = a.softcap; s.max_bias = a.max_bias; s.m0 = a.m0; s.m1 = a.m1; s.n_head_log2 = a.n_head_log2;
        s.v = td_of(v);
        if (v_direct) { if (softcap != 0.0f) hipLaunchKernelGGL((flash_attn_mfma_split_kernel<true, true>), dim3((unsigned)n_wg), dim3(256), 0, st, s);
                        else hipLaunchKernelGGL((flash_attn_mfma_split_kernel<false, true>), dim3((unsigned)n_wg), dim3(256), 0, st, s); }
        else if (softcap != 0.0f) hipLaunchKernelGGL((flash_attn_mfma_split_kernel<true, false>), dim3((unsigned)n_wg), dim3(256), 0, st, s);
        else hipLaunchKernelGGL((flash_attn_mfma_split_kernel<false, false>), dim3((unsigned)n_wg), dim3(256), 0, st, s);
        return hipGetLastError() == hipSuccess ? CDNA4_OK : cdna4_set_err(CDNA4_E_HIP, "flash_attn_mfma (split) launch failed");
    }
    hipLaunchKernelGGL(flash_attn_mfma_kernel, dim3((unsigned)((q->ne[1] + BQ * NW - 1) / (BQ * NW)), (unsigned)q->ne[2], (unsigned)q->ne[3]), dim3(256), 0, st, a);
    return hipGetLastError() == hipSuccess ? CDNA4_OK : cdna4_set_err(CDNA4_E_HIP, "flash_attn_mfma launch failed");
}
