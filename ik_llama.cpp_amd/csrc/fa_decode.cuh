// fa_decode.cuh -- the per-head decode attention body (head size 128, one query row, f16 K / V), shared by flash_attn_decode_kernel (ops.hip) and the fused
// attention + attn_output launch (gemv_attn.hip).  Replaces ggml-cuda/fattn-vec-f16.cuh for one token (ggml.c:22874-23160 semantics: scale, optional soft-cap, ALiBi
// slope on the mask, masked cells contribute nothing).
#pragma once
#include "api_internal.h"
#include <hip/hip_fp16.h>

// wave-wide reductions on the DPP network (quad_perm lane^1, lane^2, row_half_mirror lane^7, row_mirror lane^15, then the four 16-lane rows by v_readlane): no LDS
// round trips (__shfl_xor is a ds_bpermute, ~100 clk of latency per step).  All 64 lanes end with the result.
template <int CTRL> __device__ __forceinline__ float fa_dpp(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false)); }
__device__ __forceinline__ float lane_bcast(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
template <int CTRL> __device__ __forceinline__ int fa_dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += fa_dpp<0xb1>(v); v += fa_dpp<0x4e>(v); v += fa_dpp<0x141>(v); v += fa_dpp<0x140>(v);
    return (lane_bcast(v, 0) + lane_bcast(v, 16)) + (lane_bcast(v, 32) + lane_bcast(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, fa_dpp<0xb1>(v)); v = fmaxf(v, fa_dpp<0x4e>(v)); v = fmaxf(v, fa_dpp<0x141>(v)); v = fmaxf(v, fa_dpp<0x140>(v));
    return fmaxf(fmaxf(lane_bcast(v, 0), lane_bcast(v, 16)), fmaxf(lane_bcast(v, 32), lane_bcast(v, 48)));
}

// One workgroup of 256 threads = one (token t, q head h, batch b3): wave w takes keys [64 w, 64 w + 64), then + 256, ...; the four waves merge through s_m / s_l / s_acc.
// PUBLISH: the 128 results leave as write-through stores (consumed by sibling workgroups of the same launch after an agent-scope ticket, gemv_attn.hip).
template <bool FAST, bool PUBLISH>
__device__ __forceinline__ void fa_decode_body(const TD &q, const TD &k, const TD &v, const TD &mask, int has_mask, const TD &dst, float scale, float softcap, float max_bias, float m0, float m1, unsigned n_head_log2,
                                               const long t, const long h, const long b3, float *s_m, float *s_l, float (*s_acc)[128]) {
    constexpr int D = 128;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, part = lane & 15;
    
    const long hk = h / (q.ne[2] / k.ne[2]), hv = h / (q.ne[2] / v.ne[2]), b3k = b3 / (q.ne[3] / k.ne[3]), b3v = b3 / (q.ne[3] / v.ne[3]);
    const long n_kv = k.ne[1];
    const float slope = max_bias > 0.0f ? ((unsigned)h < n_head_log2 ? powf(m0, (float)(h + 1)) : powf(m1, (float)(2 * (h - n_head_log2) + 1))) : 1.0f;
    const __half *mrow = has_mask ? reinterpret_cast<const __half *>(mask.data + t * mask.nb[1] + (h % mask.ne[2]) * mask.nb[2] + (b3 % mask.ne[3]) * mask.nb[3]) : nullptr;
    const char *kbase = k.data + hk * k.nb[2] + b3k * k.nb[3]; const char *vbase = v.data + hv * v.nb[2] + b3v * v.nb[3];
    // K tile of 64 keys per wave, COALESCED: load i brings keys j0 + 16 * (lane / 16) + i, lane % 16 = the 16-byte piece of the 256-byte row (4 rows = 8 cache lines
    // per instruction; a lane-per-key layout touches 64 lines per instruction).  The 16 partial dots of a lane are summed over its 16-lane row by a reduce-scatter
    // (4 DPP exchange steps, 15 adds) that leaves the score of key j0 + lane in lane `lane`.
    uint4 kreg[16]; __half2 vreg[64]; __half mreg;
    // unconditional loads (rows clamped into the view).  K and the mask of tile t + 1 are requested as soon as the dots of tile t have consumed kreg, V of tile t + 1 after the
    // P V products of tile t: the next tile's memory round trip runs under this tile's soft-max / P V arithmetic instead of after it
    const unsigned knb1 = (unsigned)k.nb[1], vnb1 = (unsigned)v.nb[1], klast = (unsigned)(n_kv - 1) * knb1, vlast = (unsigned)(n_kv - 1) * vnb1;      // (FAST; the guard keeps these in 32 bits)
    auto load_k = [&](long j0) {
        if constexpr (FAST) {
            const unsigned o0 = __umul24((unsigned)j0 + 16u * (unsigned)(lane >> 4), knb1) + 16u * (unsigned)part;       // this lane's piece of row j0 + 16 (lane / 16)
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
            const unsigned r0 = (unsigned)__builtin_amdgcn_readfirstlane((int)j0) * vnb1;                                 // the tile's first row: wave-uniform, on the scalar unit
#pragma unroll
            for (int u = 0; u < 64; ++u) vreg[u] = reinterpret_cast<const __half2 *>(vbase + min(r0 + (unsigned)u * vnb1, vlast))[lane];
        } else {
#pragma unroll
        for (int u = 0; u < 64; ++u) vreg[u] = reinterpret_cast<const __half2 *>(vbase + min(j0 + u, n_kv - 1) * v.nb[1])[lane];
        }
    };
    long j0 = 64L * wave;
    const float4 *qr = reinterpret_cast<const float4 *>(q.data + t * q.nb[1] + h * q.nb[2] + b3 * q.nb[3]);
    const float4 qa = qr[2 * part], qb = qr[2 * part + 1];
    load_k(j0); load_v(j0);
    // (ISA of the default form: the compiler sinks the 64 V loads below the dot products -- behind an s_waitcnt vmcnt(0) on the K tile -- so the V round trip starts only
    //  after the K round trip has ended: two dependent memory latencies in front of the first soft-max.  FAST pins the order: every load of the first tile is in flight before
    //  anything waits)
    if constexpr (FAST) __builtin_amdgcn_sched_barrier(0);
    float M = -INFINITY, L = 0.f, acc0 = 0.f, acc1 = 0.f;
    while (j0 < n_kv) {
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
        if (j0 + 256 < n_kv) load_k(j0 + 256);
        if (j < n_kv && mv != -INFINITY) s = softcap == 0.0f ? dot * scale + mv : softcap * tanhf(dot * scale) + mv;        // (a masked cell's row may hold anything)
        const float tile_max = wave_max(s);
        if (tile_max != -INFINITY) {                                               // (wave-uniform) not a fully masked tile
            const float Mn = fmaxf(M, tile_max), corr = expf(M - Mn);
            const float p = s == -INFINITY ? 0.f : expf(s - Mn);
            L = L * corr + wave_sum_dpp(p);
            acc0 *= corr; acc1 *= corr; M = Mn;
#pragma unroll
            for (int u = 0; u < 64; ++u) {
                const float pj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p), u));
                const float2 f = __half22float2(vreg[u]);
                acc0 = pj == 0.f ? acc0 : fmaf(pj, f.x, acc0); acc1 = pj == 0.f ? acc1 : fmaf(pj, f.y, acc1);      // (p = 0: the cache cell may hold anything)
            }
        }
        j0 += 256;
        if (j0 < n_kv) load_v(j0);
    }
    if (lane == 0) { s_m[wave] = M; s_l[wave] = L; }
    __syncthreads();
    const float Mg = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
    const float mine = M == -INFINITY ? 0.f : expf(M - Mg);
    s_acc[wave][2 * lane] = acc0 * mine; s_acc[wave][2 * lane + 1] = acc1 * mine;
    __syncthreads();
    float Lg = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) Lg += s_m[w] == -INFINITY ? 0.f : s_l[w] * expf(s_m[w] - Mg);
    const float inv = Lg == 0.0f ? 0.0f : 1.0f / Lg;
    float *out = reinterpret_cast<float *>(dst.data + (b3 * dst.ne[2] * dst.ne[1] + h + t * dst.ne[1]) * dst.nb[1]);
    if (threadIdx.x < D) {
        const float o = (s_acc[0][threadIdx.x] + s_acc[1][threadIdx.x] + s_acc[2][threadIdx.x] + s_acc[3][threadIdx.x]) * inv;
        if constexpr (PUBLISH) __hip_atomic_store(out + threadIdx.x, o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // write-through (sc1): read by other workgroups of the SAME launch
        else out[threadIdx.x] = o;
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------------------------------------
// Round 4: the same attention with the keys of EVERY 64-key tile spread over the four waves (wave w takes keys 16 w ... 16 w + 15 of each tile; lane = (key, 32-dim
// quarter of the head)), and chunks whose 16 mask cells are all -inf are neither loaded nor multiplied.
//   * Why: the launch is bound by what ONE CU can pull (measured slope: 2.6 us per 256 keys = ~49 GB/s per workgroup, scripts/r04_fa.sh), and llama.cpp pads the KV view to a
//     multiple of 256 cells while a decode step sees n_past + 1 of them: with wave w <-> keys [64 w, 64 w + 64) the workgroup fetched the whole padded window and the waves
//     behind the visible range idled; now the bytes fetched follow the VISIBLE keys (rounded to 16 per wave) and the four waves share them evenly.
//   * Round 5 (scripts/probes/fa_timeline_probe.hip: stamps of the phases of one launch): the mask cells of a wave's chunks of 16 tiles are requested TOGETHER, next to the
//     (speculative) K / V loads of the first of them, and give a bit set of live tiles: the loop walks live tiles only (a 256-cell window with 40 visible keys used to spend
//     three iterations of one mask round trip each on dead tiles), and the indices of a workgroup come from 32-bit divisions (the 64-bit forms were ~1500 scalar instructions
//     in front of the first load).  The waves merge behind ONE barrier (every reader rescales the four partial accumulators itself) and the q8 emission reduces over DPP.
// Arithmetic: q . k in f32 (32 dims per lane, 4-lane DPP sum), online soft-max per wave, P V in f32 with the probability broadcast by v_readlane; the four waves' (max, sum,
// accumulators) merge through LDS as before.  32-bit row offsets (host guard fa_fast_addr).
#ifndef FA_TL
#define FA_TL(i_)           // (scripts/probes/fa_timeline_probe.hip defines it: s_memtime stamps of one workgroup's phases)
#endif
template <bool PUBLISH, int NW = 4>
__device__ __forceinline__ void fa_decode_body_v2(const TD &q, const TD &k, const TD &v, const TD &mask, int has_mask, const TD &dst, float scale, float softcap, float max_bias, float m0, float m1, unsigned n_head_log2,
                                                  const long t, const long h, const long b3, float *s_m, float *s_l, float (*s_acc)[128], uint8_t *q8 = nullptr) {
    constexpr int D = 128;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, kq = lane >> 2, d4 = lane & 3;
    // (head and batch counts fit 32 bits: 32-bit unsigned divisions, and none at all for the usual batch of one)
    const unsigned uh = (unsigned)h, ub = (unsigned)b3, qh = (unsigned)q.ne[2], gk = qh / (unsigned)k.ne[2], gv = (unsigned)v.ne[2] == (unsigned)k.ne[2] ? gk : qh / (unsigned)v.ne[2];
    const unsigned hk = gk == 1 ? uh : uh / gk, hv = gv == gk ? hk : uh / gv;
    const unsigned b3k = ub == 0 ? 0 : ub / ((unsigned)q.ne[3] / (unsigned)k.ne[3]), b3v = ub == 0 ? 0 : ub / ((unsigned)q.ne[3] / (unsigned)v.ne[3]);
    const unsigned mh = (unsigned)mask.ne[2] == 1 ? 0 : uh % (unsigned)mask.ne[2], mb = ub == 0 ? 0 : ub % (unsigned)mask.ne[3];
    constexpr int TK = 16 * NW;                                                    // keys per tile: 16 per wave
    const int n_kv = (int)k.ne[1], n_tiles = (n_kv + TK - 1) / TK;
    const float slope = max_bias > 0.0f ? (uh < n_head_log2 ? powf(m0, (float)(h + 1)) : powf(m1, (float)(2 * (h - n_head_log2) + 1))) : 1.0f;
    const __half *mrow = has_mask ? reinterpret_cast<const __half *>(mask.data + t * mask.nb[1] + (long)mh * mask.nb[2] + (long)mb * mask.nb[3]) : nullptr;
    const char *kbase = k.data + (long)hk * k.nb[2] + (long)b3k * k.nb[3]; const char *vbase = v.data + (long)hv * v.nb[2] + (long)b3v * v.nb[3];
    const unsigned knb1 = (unsigned)k.nb[1], vnb1 = (unsigned)v.nb[1], klast = (unsigned)(n_kv - 1) * knb1, vlast = (unsigned)(n_kv - 1) * vnb1;
    // this lane's 32 q dims: pieces 4 i + d4 (8 dims each) of the row, i = 0..3 -- the four lanes of a key read 64 contiguous bytes of the K row per load instruction
    const float4 *qr = reinterpret_cast<const float4 *>(q.data + t * q.nb[1] + h * q.nb[2] + b3 * q.nb[3]);
    float4 qv[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) { qv[2 * i] = qr[2 * (4 * i + d4)]; qv[2 * i + 1] = qr[2 * (4 * i + d4) + 1]; }
    FA_TL(0);
    const int jw = 16 * wave;                                                     // first key of this wave's chunk inside a tile
    // (no branch around a mask load: hipcc ends every conditional load's block with s_waitcnt vmcnt(0), which serializes the look-ahead below into one memory round trip per tile
    //  and stalls the K prefetch of the loop behind its own mask cell; without a mask the load reads the head of the K view, n_kv halves are inside it, and the value is dropped)
    const __half *mld = mrow ? mrow : reinterpret_cast<const __half *>(kbase);
    const unsigned short mkeep = mrow ? 0xffffu : 0u;                             // (an AND, not a select: a select lets the compiler sink the load back under a branch)
    auto mask_of = [&](int tile) -> __half { return __ushort_as_half((unsigned short)(__half_as_ushort(mld[min(TK * tile + jw + kq, n_kv - 1)]) & mkeep)); };
    // one tile's K rows (this lane: 64 bytes of one key), V rows (16 keys x this lane's two dims) and mask cell; TWO such sets alternate: the loads of the next live tile are in
    // flight while the current one is multiplied (one set: the K / V of tile t + 1 were requested after the q . k / the P V of tile t -- a full memory round trip per tile)
    struct TileRegs { uint4 k[4]; __half2 v[16]; __half m; };
    auto load_tile = [&](TileRegs &r, int tile, bool with_mask) {
        const unsigned o = min(__umul24((unsigned)(TK * tile + jw + kq), knb1), klast) + 16u * (unsigned)d4;
#pragma unroll
        for (int i = 0; i < 4; ++i) r.k[i] = *reinterpret_cast<const uint4 *>(kbase + o + 64u * (unsigned)i);
        const unsigned r0 = (unsigned)__builtin_amdgcn_readfirstlane(TK * tile + jw) * vnb1;      // wave-uniform: scalar address arithmetic
#pragma unroll
        for (int u = 0; u < 16; ++u) r.v[u] = reinterpret_cast<const __half2 *>(vbase + min(r0 + (unsigned)u * vnb1, vlast))[lane];
        if (with_mask) r.m = mask_of(tile);
    };
    auto chunk_live = [&](__half mh_, int tile) -> bool {                        // wave-uniform: does any cell of this wave's chunk of `tile` count?
        const float mvv = __half2float(mh_);
        return __builtin_amdgcn_ballot_w64((TK * tile + jw + kq) < n_kv && mvv != -INFINITY) != 0;
    };
    float M = -INFINITY, L = 0.f, acc0 = 0.f, acc1 = 0.f;
    auto consume = [&](const TileRegs &r, int tile) {
        float d = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const __half2 *kh = reinterpret_cast<const __half2 *>(&r.k[i]);
            const float2 k0 = __half22float2(kh[0]), k1 = __half22float2(kh[1]), k2 = __half22float2(kh[2]), k3 = __half22float2(kh[3]);
            const float4 qa = qv[2 * i], qb = qv[2 * i + 1];
            d = fmaf(qa.x, k0.x, d); d = fmaf(qa.y, k0.y, d); d = fmaf(qa.z, k1.x, d); d = fmaf(qa.w, k1.y, d);
            d = fmaf(qb.x, k2.x, d); d = fmaf(qb.y, k2.y, d); d = fmaf(qb.z, k3.x, d); d = fmaf(qb.w, k3.y, d);
        }
        if (tile == 0) FA_TL(2);
        d += fa_dpp<0xb1>(d); d += fa_dpp<0x4e>(d);                                // the four quarters of a key: lanes ^ 1, ^ 2
        const int j = TK * tile + jw + kq;
        const float mvv = slope * __half2float(r.m);
        float sc = -INFINITY;
        if (j < n_kv && mvv != -INFINITY) sc = softcap == 0.0f ? d * scale + mvv : softcap * tanhf(d * scale) + mvv;      // (a masked cell's row may hold anything)
        const float tile_max = wave_max(sc);
        if (tile == 0) FA_TL(3);
        if (tile_max != -INFINITY) {                                               // (wave-uniform)
            const float Mn = fmaxf(M, tile_max), corr = expf(M - Mn);
            const float p = sc == -INFINITY ? 0.f : expf(sc - Mn);
            L = L * corr + wave_sum_dpp(d4 == 0 ? p : 0.f);                        // (one lane per key counts)
            acc0 *= corr; acc1 *= corr; M = Mn;
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const float pj = lane_bcast(p, 4 * u);
                const float2 f = __half22float2(r.v[u]);
                acc0 = pj == 0.f ? acc0 : fmaf(pj, f.x, acc0); acc1 = pj == 0.f ? acc1 : fmaf(pj, f.y, acc1);      // (p = 0: the cache cell may hold anything)
            }
        }
        if (tile == 0) FA_TL(4);
    };
    constexpr int CH = 8;                                                         // tiles per mask look-ahead: 512 cells with 4 waves, 1024 with 8 -- every window those launches serve by default (ops.hip LAUNCH_DECODE2)
    TileRegs ra, rb;
    for (int c0 = 0; c0 < n_tiles; c0 += CH) {
        __half mm[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) mm[i] = mask_of(min(c0 + i, n_tiles - 1));
        load_tile(ra, c0, false);                                                  // speculative: a decoded token's visible keys are a prefix of the window
        __builtin_amdgcn_sched_barrier(0);                                         // every load is in flight before anything waits
        if (c0 == 0) FA_TL(1);
        unsigned live = 0;
#pragma unroll
        for (int i = 0; i < CH; ++i) if (c0 + i < n_tiles && chunk_live(mm[i], c0 + i)) live |= 1u << i;
        live = (unsigned)__builtin_amdgcn_readfirstlane((int)live);
        if (!live) continue;
        auto next_live = [&]() -> int { if (!live) return -1; const int tn = c0 + __builtin_ctz(live); live &= live - 1; return tn; };
        int ta = next_live();
        ra.m = mm[0];
        if (ta != c0) load_tile(ra, ta, true);                                     // (the chunk's first tile is dead for this wave, a later one is not: a window that is no prefix)
        int tb = next_live();
        if (tb < 0) { consume(ra, ta); continue; }                                 // one live tile (a short context): nothing to prefetch
        // several live tiles: the prefetches are UNCONDITIONAL loads (past the last live tile they re-read the tile just multiplied: cache hits nobody waits for) -- a load under
        // `if (next >= 0)` merges with the register's old value at the join and hipcc closes the block with s_waitcnt vmcnt(0): the "prefetch" then completes before the
        // multiplication it was meant to overlap starts
        load_tile(rb, tb, true);
        for (;;) {                                                                 // ra holds live tile ta, rb holds live tile tb
            consume(ra, ta);
            const int tn = next_live(); load_tile(ra, tn >= 0 ? tn : ta, true);
            consume(rb, tb);
            if (tn < 0) break;
            const int tm = next_live(); load_tile(rb, tm >= 0 ? tm : tb, true);
            ta = tn;
            if (tm < 0) { consume(ra, ta); break; }
            tb = tm;
        }
    }
    FA_TL(5);
    // merge of the NW waves: (max, sum, un-scaled accumulators) through LDS, ONE barrier; a reader rescales the partial accumulators itself -- the products and the order of the
    // sums are those of the two-barrier form it replaces (each writer scaled its own accumulators, then a second barrier)
    if (lane == 0) { s_m[wave] = M; s_l[wave] = L; }
    s_acc[wave][2 * lane] = acc0; s_acc[wave][2 * lane + 1] = acc1;
    __syncthreads();
    FA_TL(6);
    if (threadIdx.x < D) {
        float Mg = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
#pragma unroll
        for (int w = 4; w < NW; w += 4) Mg = fmaxf(Mg, fmaxf(fmaxf(s_m[w], s_m[w + 1]), fmaxf(s_m[w + 2], s_m[w + 3])));
        float e[NW], Lg = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) { e[w] = s_m[w] == -INFINITY ? 0.f : expf(s_m[w] - Mg); Lg = __fadd_rn(Lg, s_m[w] == -INFINITY ? 0.f : __fmul_rn(s_l[w], e[w])); }
        const float inv = Lg == 0.0f ? 0.0f : 1.0f / Lg;
        FA_TL(7);
        float *out = reinterpret_cast<float *>(dst.data + (b3 * dst.ne[2] * dst.ne[1] + h + t * dst.ne[1]) * dst.nb[1]);
        const int c = threadIdx.x;
        float osum = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(s_acc[0][c], e[0]), __fmul_rn(s_acc[1][c], e[1])), __fmul_rn(s_acc[2][c], e[2])), __fmul_rn(s_acc[3][c], e[3]));
#pragma unroll
        for (int w = 4; w < NW; w += 4)
            osum = __fadd_rn(osum, __fadd_rn(__fadd_rn(__fmul_rn(s_acc[w][c], e[w]), __fmul_rn(s_acc[w + 1][c], e[w + 1])), __fadd_rn(__fmul_rn(s_acc[w + 2][c], e[w + 2]), __fmul_rn(s_acc[w + 3][c], e[w + 3]))));
        const float o = osum * inv;
        if constexpr (PUBLISH) __hip_atomic_store(out + threadIdx.x, o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else out[threadIdx.x] = o;
        // q8 != nullptr (one token): the head's 128 results ALSO leave as one block_q8_2_x4 (ggml-common.h:287-299; 4 x {bf16 d, int16 sum} + 128 int8), byte-identical to
        // quantize_row_q8_2_x4 (iqk_quantize.cpp:1072-1175) of the f32 row -- the attn_output mat-vec behind this launch then enters with typeB = Q8_2_X4 and skips the 16 KB
        // f32 read + quantization every one of its workgroups would repeat (the reference quantizes src1 once for all consumers too, ggml.c:17955-17964).  A head IS one
        // x4 super-block: no cross-workgroup step.  Same arithmetic as gemv.cuh's emit path: amax over the 32 lanes of a block, d = bf16(amax / 127), RNE, sum before saturation.
        FA_TL(8);
        if (q8) {
            // (32-lane reductions on the DPP network: 16-lane rows, then the two rows of a block by v_readlane -- the two waves that get here are complete)
            float amax = fabsf(o);
            amax = fmaxf(amax, fa_dpp<0xb1>(amax)); amax = fmaxf(amax, fa_dpp<0x4e>(amax)); amax = fmaxf(amax, fa_dpp<0x141>(amax)); amax = fmaxf(amax, fa_dpp<0x140>(amax));
            { const float lo = fmaxf(lane_bcast(amax, 0), lane_bcast(amax, 16)), hi = fmaxf(lane_bcast(amax, 32), lane_bcast(amax, 48)); amax = lane < 32 ? lo : hi; }
            const uint32_t db = float_to_bf16_bits(amax / 127.f);
            const float d = bf16_bits_to_float(db), id = d > 0 ? 1.f / d : 0.f;
            const int qv = (int)rintf(o * id);
            int isum = qv;
            isum += fa_dpp_i<0xb1>(isum); isum += fa_dpp_i<0x4e>(isum); isum += fa_dpp_i<0x141>(isum); isum += fa_dpp_i<0x140>(isum);
            { const int lo = __builtin_amdgcn_readlane(isum, 0) + __builtin_amdgcn_readlane(isum, 16), hi = __builtin_amdgcn_readlane(isum, 32) + __builtin_amdgcn_readlane(isum, 48); isum = lane < 32 ? lo : hi; }
            uint8_t *blk = q8 + (h + t * dst.ne[1]) * 144; const int ir = threadIdx.x >> 5;
            if ((threadIdx.x & 31) == 0) { *reinterpret_cast<uint16_t *>(blk + 2 * ir) = (uint16_t)db; *reinterpret_cast<int16_t *>(blk + 8 + 2 * ir) = (int16_t)isum; }
            blk[16 + threadIdx.x] = (uint8_t)((qv > 127 ? 127 : (qv < -128 ? -128 : qv)) & 255);
        }
        FA_TL(9);
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------------------------------------
// Split-KV decode attention ("flash decoding"), round-5 form of the per-wave tile loop (the hand-off to the combining workgroup is ops.hip's).  A workgroup = one KV head x one
// chunk of the window, wave w = q head hk G + w of the GQA group.  Tile = 64 keys per wave: K coalesced (load i brings keys j0 + 16 (lane / 16) + i, lane % 16 = the 16-byte
// piece of the 256-byte row), 16 partial dots per lane reduce-scattered over the 16-lane row, V as 64 x __half2 per lane.
// What changed against flash_attn_split_kernel<true> of rounds 2-4 (kept for views that need 64-bit offsets), all of it learned on the per-head kernel above:
//   * the wave index is made scalar (readfirstlane): h, the KV / mask row bases and their divisions run on the scalar unit in 32 bits (they were per-lane 64-bit divisions:
//     ~1900 vector instructions in front of the first load);
//   * the mask cell is loaded without a branch, and EVERY load is unconditional (past the chunk's end it re-reads the last tile): exact s_waitcnt vmcnt(N) instead of vmcnt(0)
//     at every join -- the K of tile i + 1 really is in flight under the soft-max and the P V of tile i;
//   * V moves as a ring of four quarters: the 16 rows of quarter c of tile i + 1 are requested as soon as quarter c of tile i has been multiplied (the whole V tile used to be
//     requested after the P V loop and waited for, un-overlapped, at the top of the next tile);
//   * a fully masked tile is multiplied like any other (probabilities 0) instead of branching around the P V loop -- the branch would put the quarter loads under a condition.
struct FaSplitOut { float M, L, acc0, acc1; long h; };
__device__ __forceinline__ FaSplitOut fa_split_tiles_v2(const TD &q, const TD &k, const TD &v, const TD &mask, int has_mask, float scale, float softcap, float max_bias, float m0, float m1, unsigned n_head_log2,
                                                        const int n_splits, const int chunk) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), part = lane & 15;
    const unsigned qh = (unsigned)q.ne[2], G = qh / (unsigned)k.ne[2];
    const unsigned split = blockIdx.x % (unsigned)n_splits, t = blockIdx.x / (unsigned)n_splits, hk = blockIdx.y, ub = blockIdx.z;
    const unsigned uh = hk * G + (unsigned)wave, hv = (unsigned)v.ne[2] == (unsigned)k.ne[2] ? hk : uh / (qh / (unsigned)v.ne[2]);
    const unsigned b3k = ub == 0 ? 0 : ub / ((unsigned)q.ne[3] / (unsigned)k.ne[3]), b3v = ub == 0 ? 0 : ub / ((unsigned)q.ne[3] / (unsigned)v.ne[3]);
    const unsigned mh = (unsigned)mask.ne[2] == 1 ? 0 : uh % (unsigned)mask.ne[2], mb = ub == 0 ? 0 : ub % (unsigned)mask.ne[3];
    const int n_kv = (int)k.ne[1], j_begin = (int)split * chunk, j_end = min(n_kv, j_begin + chunk);
    const float slope = max_bias > 0.0f ? (uh < n_head_log2 ? powf(m0, (float)(uh + 1)) : powf(m1, (float)(2 * (uh - n_head_log2) + 1))) : 1.0f;
    const char *kbase = k.data + (long)hk * k.nb[2] + (long)b3k * k.nb[3]; const char *vbase = v.data + (long)hv * v.nb[2] + (long)b3v * v.nb[3];
    const __half *mrow = has_mask ? reinterpret_cast<const __half *>(mask.data + (long)t * mask.nb[1] + (long)mh * mask.nb[2] + (long)mb * mask.nb[3]) : nullptr;
    const __half *mld = mrow ? mrow : reinterpret_cast<const __half *>(kbase);    // (branch-free mask loads: see fa_decode_body_v2)
    const unsigned short mkeep = mrow ? 0xffffu : 0u;
    const unsigned knb1 = (unsigned)k.nb[1], vnb1 = (unsigned)v.nb[1], klast = (unsigned)(n_kv - 1) * knb1, vlast = (unsigned)(n_kv - 1) * vnb1;
    const float4 *qr = reinterpret_cast<const float4 *>(q.data + (long)t * q.nb[1] + (long)uh * q.nb[2] + (long)ub * q.nb[3]);
    const float4 qa = qr[2 * part], qb = qr[2 * part + 1];
    uint4 kreg[16]; __half2 vreg[64]; __half mreg;
    auto load_k = [&](int j0) {
        const unsigned o0 = __umul24((unsigned)j0 + 16u * (unsigned)(lane >> 4), knb1) + 16u * (unsigned)part;
#pragma unroll
        for (int i = 0; i < 16; ++i) kreg[i] = *reinterpret_cast<const uint4 *>(kbase + min(o0 + (unsigned)i * knb1, klast + 16u * (unsigned)part));
        mreg = __ushort_as_half((unsigned short)(__half_as_ushort(mld[min(j0 + lane, n_kv - 1)]) & mkeep));
    };
    auto load_v_quarter = [&](int j0, int c) {
        const unsigned r0 = (unsigned)__builtin_amdgcn_readfirstlane(j0 + 16 * c) * vnb1;
#pragma unroll
        for (int u = 0; u < 16; ++u) vreg[16 * c + u] = reinterpret_cast<const __half2 *>(vbase + min(r0 + (unsigned)u * vnb1, vlast))[lane];
    };
    FaSplitOut o; o.M = -INFINITY; o.L = 0.f; o.acc0 = 0.f; o.acc1 = 0.f; o.h = uh;
    if (j_begin >= j_end) return o;
    load_k(j_begin);
#pragma unroll
    for (int c = 0; c < 4; ++c) load_v_quarter(j_begin, c);
    __builtin_amdgcn_sched_barrier(0);
    float M = -INFINITY, L = 0.f, acc0 = 0.f, acc1 = 0.f;
    for (int j0 = j_begin; j0 < j_end; j0 += 64) {
        const int jn = j0 + 64 < j_end ? j0 + 64 : j0;          // the tile the prefetches aim at (the last tile re-reads itself: hits nobody waits for)
        const int j = j0 + lane;
        float r[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const __half2 *kh = reinterpret_cast<const __half2 *>(&kreg[i]);
            const float2 k0 = __half22float2(kh[0]), k1 = __half22float2(kh[1]), k2 = __half22float2(kh[2]), k3 = __half22float2(kh[3]);
            float d = qa.x * k0.x; d = fmaf(qa.y, k0.y, d); d = fmaf(qa.z, k1.x, d); d = fmaf(qa.w, k1.y, d);
            d = fmaf(qb.x, k2.x, d); d = fmaf(qb.y, k2.y, d); d = fmaf(qb.z, k3.x, d); d = fmaf(qb.w, k3.y, d);
            r[i] = d;
        }
        const float mv = slope * __half2float(mreg);
        load_k(jn);                                               // (kreg and mreg are consumed)
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
        if (j < j_end && mv != -INFINITY) s = softcap == 0.0f ? dot * scale + mv : softcap * tanhf(dot * scale) + mv;        // (a masked cell's row may hold anything)
        const float tile_max = wave_max(s);
        const bool any = tile_max != -INFINITY;                    // (wave-uniform) a fully masked tile changes nothing: probabilities 0, correction 1
        const float Mn = any ? fmaxf(M, tile_max) : M, corr = any ? expf(M - Mn) : 1.0f;
        const float p = s == -INFINITY ? 0.f : expf(s - Mn);
        L = L * corr + wave_sum_dpp(p);
        acc0 *= corr; acc1 *= corr; M = Mn;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int u = 16 * c; u < 16 * c + 16; ++u) {
                const float pj = lane_bcast(p, u);
                const float2 f = __half22float2(vreg[u]);
                acc0 = pj == 0.f ? acc0 : fmaf(pj, f.x, acc0); acc1 = pj == 0.f ? acc1 : fmaf(pj, f.y, acc1);      // (p = 0: the cache cell may hold anything)
            }
            load_v_quarter(jn, c);
        }
    }
    o.M = M; o.L = L; o.acc0 = acc0; o.acc1 = acc1;
    return o;
}
