// gemv_mfma.hip -- decode with 2..8 activation columns (batched / speculative decode, MUL_MAT with ne11 = 2..8) on the int8 matrix cores.
//
// What it replaces: the reference's 1..8-column mat-vec kernels (ggml-cuda/mmvq-templates.cuh:68-150) and, on the CPU, the N <= 8 tiles of
// mul_mat_qX_K_q8_2_X4_T (iqk_gemm_kquants.cpp); SAME arithmetic as those and as gemv.cuh: activations quantized to block_q8_2_x4 (int8,
// bf16 scale per 32), exact int32 sums per 32-weight sub-block, f32 accumulate of d * sc * dy * isum - dmin * m * (dy * sum y).
//
// Why a second kernel: with v_dot4 every (weight unit, column) pair costs 16 dot instructions + the scale arithmetic, so 8 columns run ~3.6x
// the time of one (VALU-issue-bound, profiles/r01_notes.md).  v_mfma_i32_16x16x32_i8 contracts one 32-weight sub-block of 16 rows against
// 16 columns in ONE instruction:
//     D[col][row] = sum_k Y[col][k] W[row][k]        A = activations (lane (col = l % 16, kg = l / 16) holds k = 8 kg .. 8 kg + 7, ds_read_b64),
//                                                    B = weights     (lane (row = l % 16, kg) holds the same k of its row: 8 bytes of qs),
//     lane (row = l % 16, g = l / 16) receives the sums of ITS row for columns 4 g .. 4 g + 3: the row's scales are already in the lane.
// Q4_K / Q5_K: the 8 qs bytes 8 kg .. 8 kg + 7 of a 32-byte group give 8 weights of sub-block 2 j (low nibbles) and 8 of sub-block 2 j + 1
// (high nibbles): two MFMAs per 8-byte load, no cross-lane traffic.  Q6_K: one int8 scale per 16 weights -> one MFMA per 16-weight piece with
// the upper half of the k range zero (gfx950 has no 16x16x16 i8), lane (row, kg) holds l = 4 kg .. 4 kg + 3 of the piece.
// A wave owns 16 consecutive rows; when there are too few row groups to fill the chip, ks waves share a row group, split K by super-block and
// add their partial sums through LDS (fixed order).  The activations arrive pre-quantized (one quantize_rows launch per mat-mul instead of every
// workgroup re-quantizing 8 columns) and are staged in LDS in chunks of 4096 k with a column pitch of 4112 bytes (16 columns -> 16 distinct
// bank slots for the ds_read_b64 of a 32-lane group); block scales as [block][16 columns] so that a lane reads its 4 columns with one b128.
#include "gemv_launch.cuh"

typedef int int4v __attribute__((ext_vector_type(4)));

namespace {
__device__ __forceinline__ long u2l(uint32_t lo, uint32_t hi) { return (long)(((unsigned long)hi << 32) | lo); }

// one super-block of one row as this lane sees it
template <int TYPE> struct SbK;                   // Q4_K / Q5_K
template <> struct SbK<T_Q4_K> {
    static constexpr int BYTES = 144;
    uint4 h; uint2 q[4];
    __device__ __forceinline__ void load(const uint8_t *b, int kg) {
        h = ldw128(b);
#pragma unroll
        for (int j = 0; j < 4; ++j) q[j] = *reinterpret_cast<const uint2 *>(b + 16 + 32 * j + 8 * kg);
    }
    __device__ __forceinline__ void pieces(int j, long &lo, long &hi) const {
        lo = u2l(q[j].x & 0x0f0f0f0fu, q[j].y & 0x0f0f0f0fu); hi = u2l((q[j].x >> 4) & 0x0f0f0f0fu, (q[j].y >> 4) & 0x0f0f0f0fu);
    }
};
template <> struct SbK<T_Q5_K> {
    static constexpr int BYTES = 176;
    uint4 h; uint2 q[4], hb;
    __device__ __forceinline__ void load(const uint8_t *b, int kg) {
        h = ldw128(b); hb = *reinterpret_cast<const uint2 *>(b + 16 + 8 * kg);
#pragma unroll
        for (int j = 0; j < 4; ++j) q[j] = *reinterpret_cast<const uint2 *>(b + 48 + 32 * j + 8 * kg);
    }
    __device__ __forceinline__ void pieces(int j, long &lo, long &hi) const {
        lo = u2l((q[j].x & 0x0f0f0f0fu) | (((hb.x >> (2 * j)) & 0x01010101u) << 4), (q[j].y & 0x0f0f0f0fu) | (((hb.y >> (2 * j)) & 0x01010101u) << 4));
        hi = u2l(((q[j].x >> 4) & 0x0f0f0f0fu) | (((hb.x >> (2 * j + 1)) & 0x01010101u) << 4), ((q[j].y >> 4) & 0x0f0f0f0fu) | (((hb.y >> (2 * j + 1)) & 0x01010101u) << 4));
    }
};

typedef float float4v __attribute__((ext_vector_type(4)));
constexpr int KC_SB = 16, KC = 256 * KC_SB, KP = KC + 16, NBC = KC / 32;      // K chunk resident in LDS: 16 super-blocks, column pitch 4112 B, 128 blocks of 32

struct Stage { int8_t *yq; float *yd, *ys; };          // yq[col][KP]; yd / ys[block][16 columns] (a lane reads its 4 columns with one ds_read_b128)

// copy chunk `c` of the pre-quantized block_q8_2_x4 rows of `ncols` columns into LDS
__device__ __forceinline__ void stage_q8_2_x4(const GemvArgs &a, int ncols, int c, const Stage &s) {
    const int nb = a.K >> 5, nbc = min(NBC, nb - c * NBC);
    for (int i = threadIdx.x; i < ncols * nbc; i += blockDim.x) {
        const int col = i / nbc, bl = i - col * nbc, b = c * NBC + bl;
        const uint8_t *blk = a.B + (long)col * a.strideB + (long)(b >> 2) * 144; const int ir = b & 3;
        const float d = bf16_bits_to_float(ld16(blk + 2 * ir)); const int sm = (int)(short)ld16(blk + 8 + 2 * ir);
        s.yd[bl * 16 + col] = d; s.ys[bl * 16 + col] = d * (float)sm;
        const uint4 q0 = *reinterpret_cast<const uint4 *>(blk + 16 + 32 * ir), q1 = *reinterpret_cast<const uint4 *>(blk + 32 + 32 * ir);
        *reinterpret_cast<uint4 *>(s.yq + (long)col * KP + 32 * bl) = q0; *reinterpret_cast<uint4 *>(s.yq + (long)col * KP + 32 * bl + 16) = q1;
    }
}

__device__ __forceinline__ float4v cvt4(int4v v) { float4v f = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]}; return f; }

// ---- Q4_K / Q5_K: one super-block of one row as lane (row, kg) sees it + its contraction against the staged columns
template <int TYPE> struct WK : SbK<TYPE> {
    // bl0 = first 32-block of this super-block inside the staged chunk; ycol = this lane's A-operand base (column r, k-group kg)
    __device__ __forceinline__ void compute(int bl0, const int8_t *ycol, const float *ydl, const float *ysl, float4v &acc) const {
        // all 8 MFMAs of the super-block first (their results are needed ~8 passes later), then the scale arithmetic
        int4v s_lo[4], s_hi[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            long lo, hi; this->pieces(j, lo, hi);
            const int bl = bl0 + 2 * j;
            const long ya = *reinterpret_cast<const long *>(ycol + 32 * bl), yb = *reinterpret_cast<const long *>(ycol + 32 * bl + 32);
            const int4v z = {0, 0, 0, 0};
            s_lo[j] = __builtin_amdgcn_mfma_i32_16x16x32_i8(ya, lo, z, 0, 0, 0); s_hi[j] = __builtin_amdgcn_mfma_i32_16x16x32_i8(yb, hi, z, 0, 0, 0);
        }
        const float d = half_bits_to_float(this->h.x & 0xffff), dmin = half_bits_to_float(this->h.x >> 16);
        uint32_t sc03, sc47, mn03, mn47; k4_unpack_scales(this->h.y, this->h.z, this->h.w, sc03, sc47, mn03, mn47);
        float4v mins = {0.f, 0.f, 0.f, 0.f};
#ifdef GMF_EXP_NO_EPI
        acc += cvt4(s_lo[0] + s_hi[0] + s_lo[1] + s_hi[1] + s_lo[2] + s_hi[2] + s_lo[3] + s_hi[3]) * d; return;
#endif
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int bl = bl0 + 2 * j;
            const uint32_t scw = ((j & 2) ? sc47 : sc03) >> (16 * (j & 1)), mnw = ((j & 2) ? mn47 : mn03) >> (16 * (j & 1));
            const float d_lo = d * (float)(scw & 0xff), d_hi = d * (float)((scw >> 8) & 0xff), m_lo = dmin * (float)(mnw & 0xff), m_hi = dmin * (float)((mnw >> 8) & 0xff);
            const float4v dy_lo = *reinterpret_cast<const float4v *>(ydl + 16 * bl), dy_hi = *reinterpret_cast<const float4v *>(ydl + 16 * bl + 16);
            const float4v sy_lo = *reinterpret_cast<const float4v *>(ysl + 16 * bl), sy_hi = *reinterpret_cast<const float4v *>(ysl + 16 * bl + 16);
            // per column: fma(d_lo * dy_lo, isum_lo, .), fma(d_hi * dy_hi, isum_hi, .), fma(sy_lo, -m_lo, .), fma(sy_hi, -m_hi, .)   (gemv.cuh Unit<T_Q4_K>::dot);
            // the min terms are gathered in their own accumulator (independent dependency chain) and added once per super-block
            acc = __builtin_elementwise_fma(d_lo * dy_lo, cvt4(s_lo[j]), acc); acc = __builtin_elementwise_fma(d_hi * dy_hi, cvt4(s_hi[j]), acc);
            mins = __builtin_elementwise_fma(sy_lo, (float4v)(m_lo), mins);     mins = __builtin_elementwise_fma(sy_hi, (float4v)(m_hi), mins);
        }
        acc -= mins;
    }
    static constexpr int YK = 8;        // bytes of a lane's A-operand piece
};

// ---- Q6_K: 16-weight pieces with their own int8 scale -> one (half-filled) 16x16x32 MFMA each; lane (row, kg) holds l = 4 kg .. 4 kg + 3 of a piece
__device__ __forceinline__ uint32_t q6_sub32(uint32_t v) { return ((v | 0x80808080u) - 0x20202020u) ^ 0x80808080u; }      // per byte: q - 32, no cross-byte borrow
struct W6 {
    static constexpr int BYTES = 210, YK = 4;
    uint32_t la[2][2], lb[2][2], qh[2][2]; uint2 sc[2]; uint32_t dh;          // [half n][l0 / 16]
    __device__ __forceinline__ void load(const uint8_t *b, int kg) {
#pragma unroll
        for (int n = 0; n < 2; ++n) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int l = 16 * p + 4 * kg;
                const uint8_t *x = b + 64 * n + l, *y = b + 64 * n + 32 + l, *z = b + 128 + 32 * n + l;
                la[n][p] = ld32(x); lb[n][p] = ld32(y); qh[n][p] = ld32(z);     // (210-byte super-blocks: 2-byte aligned; single unaligned dword loads)
            }
            sc[n] = ld64(b + 192 + 8 * n);
        }
        dh = ld16(b + 208);
    }
    __device__ __forceinline__ void compute(int bl0, const int8_t *ycol, const float *ydl, const float *, float4v &acc) const {
        const float d = half_bits_to_float(dh);
#pragma unroll
        for (int n = 0; n < 2; ++n) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {            // l0 = 16 p; scales is = p + {0, 2, 4, 6} of this half
                const uint32_t s01 = sc[n].x >> (8 * p), s23 = sc[n].y >> (8 * p);
                const float ds[4] = {d * (float)(int)(int8_t)(s01 & 0xff), d * (float)(int)(int8_t)((s01 >> 16) & 0xff),
                                     d * (float)(int)(int8_t)(s23 & 0xff), d * (float)(int)(int8_t)((s23 >> 16) & 0xff)};
                const uint32_t A = la[n][p], B = lb[n][p], H = qh[n][p];
                const uint32_t q[4] = {q6_sub32((A & 0x0f0f0f0fu) | ((H & 0x03030303u) << 4)), q6_sub32((B & 0x0f0f0f0fu) | (((H >> 2) & 0x03030303u) << 4)),
                                       q6_sub32(((A >> 4) & 0x0f0f0f0fu) | (((H >> 4) & 0x03030303u) << 4)), q6_sub32(((B >> 4) & 0x0f0f0f0fu) | (((H >> 6) & 0x03030303u) << 4))};
                int4v sm[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {        // piece j: elements 128 n + 32 j + 16 p + [0, 16) -> activation block 4 n + j of the super-block
                    const uint32_t ya = *reinterpret_cast<const uint32_t *>(ycol + 32 * (bl0 + 4 * n + j) + 16 * p);
                    const int4v z = {0, 0, 0, 0};
                    sm[j] = __builtin_amdgcn_mfma_i32_16x16x32_i8(u2l(ya, 0u), u2l(q[j], 0u), z, 0, 0, 0);      // (k = 16 used: gfx950 has no 16x16x16 i8)
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4v dy = *reinterpret_cast<const float4v *>(ydl + 16 * (bl0 + 4 * n + j));
                    acc = __builtin_elementwise_fma(ds[j] * dy, cvt4(sm[j]), acc);                               // gemv.cuh Unit<T_Q6_K>::dot
                }
            }
        }
    }
};

// W = per-lane image of one super-block (WK<T_Q4_K>, WK<T_Q5_K>, W6).  blockDim = 64 * nw waves; `ks` waves share a 16-row group and split K by
// super-block (sb = kq, kq + ks, ...), nw / ks row groups per workgroup.  K is walked in LDS-resident chunks of 16 super-blocks.
template <class W, bool UPGATE>
__global__ void __launch_bounds__(512) gemv_mfma_kernel(const GemvArgs a, int ncols, int ks) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    Stage s; s.yq = reinterpret_cast<int8_t *>(smem); s.yd = reinterpret_cast<float *>(smem + (size_t)ncols * KP); s.ys = s.yd + NBC * 16;      // (KP is a multiple of 16)
    float *red = s.ys + NBC * 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6, r = lane & 15, kg = lane >> 4;
    const int nsb = a.K >> 8, rgw = nw / ks, rgi = wave / ks, kq = wave - rgi * ks;
    const long row0 = ((long)blockIdx.x * rgw + rgi) * 16, row = min(row0 + r, (long)a.M - 1);
    const bool active = row0 < a.M;
    const uint8_t *wrow = a.A[0] + row * a.strideA, *wrow2 = UPGATE ? a.A2 + row * a.strideA : nullptr;
    const int8_t *ycol = s.yq + (long)min(r, ncols - 1) * KP + W::YK * kg;      // A operand: this lane is (column r, k-group kg); columns past ncols re-read the last one (results unused)
    const float *ydl = s.yd + 4 * kg, *ysl = s.ys + 4 * kg;                     // D: this lane holds row r, columns 4 kg .. 4 kg + 3
    float4v acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
    // two super-blocks in flight ahead of the one being contracted (unconditional loads, clamped index: exact vmcnt)
    W cur, n1, n2, cur2, n12, n22;
    auto ld = [&](W &w, W &w2, int sb) { const int sc = min(sb, nsb - 1); w.load(wrow + (long)sc * W::BYTES, kg); if (UPGATE) w2.load(wrow2 + (long)sc * W::BYTES, kg); };
    ld(cur, cur2, kq); ld(n1, n12, kq + ks);
    int sb = kq;
    const int nchunk = (nsb + KC_SB - 1) / KC_SB;
    for (int c = 0; c < nchunk; ++c) {
        if (c > 0) __syncthreads();                    // everyone is done with the previous chunk
        for (int i = threadIdx.x; i < 2 * NBC * 16 / 4; i += blockDim.x) reinterpret_cast<float4v *>(s.yd)[i] = (float4v)(0.f);       // (columns past ncols read zeros)
        __syncthreads();
        stage_q8_2_x4(a, ncols, c, s);
        __syncthreads();
        const int sb_end = min(nsb, KC_SB * (c + 1));
        for (; sb < sb_end; sb += ks) {
#ifndef GMF_EXP_NO_LOAD          // (timing experiments, scripts/gmf_exp.py: results are wrong with these)
            ld(n2, n22, sb + 2 * ks);
#endif
            if (active) { cur.compute(8 * (sb - KC_SB * c), ycol, ydl, ysl, acc); if (UPGATE) cur2.compute(8 * (sb - KC_SB * c), ycol, ydl, ysl, acc2); }
            cur = n1; n1 = n2; if (UPGATE) { cur2 = n12; n12 = n22; }
        }
    }
    if (ks == 1) {
        if (active && row0 + r < a.M && 4 * kg < ncols) {
#pragma unroll
            for (int i = 0; i < 4; ++i) if (4 * kg + i < ncols) a.C[0][(long)(4 * kg + i) * a.stride_C + row0 + r] = UPGATE ? up_gate_combine(a.unary_op, acc[i], acc2[i], a.epi, row0 + r, 0) : acc[i];
        }
        return;
    }
    // partial sums of the ks waves of a row group -> C (wave-ordered sum: deterministic)
#pragma unroll
    for (int i = 0; i < 4; ++i) { red[((wave * 2 + 0) * 16 + r) * 16 + 4 * kg + i] = acc[i]; if (UPGATE) red[((wave * 2 + 1) * 16 + r) * 16 + 4 * kg + i] = acc2[i]; }
    __syncthreads();
    for (int o = threadIdx.x; o < rgw * 16 * ncols; o += blockDim.x) {
        const int g = o / (16 * ncols), oo = o - g * 16 * ncols, rr = oo & 15, cc = oo >> 4; const long orow = ((long)blockIdx.x * rgw + g) * 16 + rr;
        if (orow >= a.M) continue;
        float v = 0.f, v2 = 0.f;
        for (int w = g * ks; w < (g + 1) * ks; ++w) { v += red[((w * 2 + 0) * 16 + rr) * 16 + cc]; if (UPGATE) v2 += red[((w * 2 + 1) * 16 + rr) * 16 + cc]; }
        a.C[0][(long)cc * a.stride_C + orow] = UPGATE ? up_gate_combine(a.unary_op, v, v2, a.epi, orow, 0) : v;
    }
}

// ---- Q4_K / Q5_K with the weights staged through LDS: every 16-byte piece of a (16 rows x 1 super-block) tile is fetched exactly once by a
// global_load_lds_dwordx4 (144 / 176 contiguous bytes per row, LDS image = the tile row-major, no VGPR round trip); three tile buffers per
// wave, two tiles in flight ahead of the one being contracted, counted vmcnt -- no workgroup barrier in the weight path (the buffers are
// wave-private and a wave's LDS operations retire in order).  ds_read_b64 of lane (row, kg) at row * 144 (176) + 16 + 32 j + 8 kg: 16 rows x 2
// k-groups of a 32-lane access fall on 32 distinct bank pairs.
template <int TYPE, bool UPGATE>
__global__ void __launch_bounds__(512) gemv_mfma_lds_kernel(const GemvArgs a, int ncols, int ks) {
    typedef WK<TYPE> W;
    constexpr int BYTES = W::BYTES, PPR = BYTES / 16, NP = 16 * PPR, NI = 3, TILE = NI * 1024, NMAT = UPGATE ? 2 : 1;
    static_assert(NP <= NI * 64, "tile pieces");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    Stage s; s.yq = reinterpret_cast<int8_t *>(smem); s.yd = reinterpret_cast<float *>(smem + (size_t)ncols * KP); s.ys = s.yd + NBC * 16;
    float *red = s.ys + NBC * 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6, r = lane & 15, kg = lane >> 4;
    uint8_t *wbuf = reinterpret_cast<uint8_t *>(red + nw * 2 * 16 * 16) + (size_t)wave * (NMAT * 3 * TILE);
    const int nsb = a.K >> 8, rgw = nw / ks, rgi = wave / ks, kq = wave - rgi * ks;
    const long row0 = ((long)blockIdx.x * rgw + rgi) * 16;
    const bool active = row0 < a.M;
    // DMA source of this lane's three slots of a tile (slot p = lane + 64 i -> row p / PPR, piece p % PPR; slots past the tile re-read its last piece)
    long soff[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) { const int p = min(lane + 64 * i, NP - 1); soff[i] = min(row0 + p / PPR, (long)a.M - 1) * a.strideA + (p % PPR) * 16; }
    typedef __attribute__((address_space(3))) void lds_void_t;
    typedef const __attribute__((address_space(1))) void glb_void_t;
    const uint32_t wbuf_s = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)wbuf);
    auto dma = [&](int sb, int buf) {
        const long so = (long)min(sb, nsb - 1) * BYTES;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            __builtin_amdgcn_global_load_lds((glb_void_t *)(a.A[0] + soff[i] + so), (lds_void_t *)(uintptr_t)(wbuf_s + buf * TILE + i * 1024), 16, 0, 0);
            if (UPGATE) __builtin_amdgcn_global_load_lds((glb_void_t *)(a.A2 + soff[i] + so), (lds_void_t *)(uintptr_t)(wbuf_s + (3 + buf) * TILE + i * 1024), 16, 0, 0);
        }
    };
    const int8_t *ycol = s.yq + (long)min(r, ncols - 1) * KP + W::YK * kg;
    const float *ydl = s.yd + 4 * kg, *ysl = s.ys + 4 * kg;
    float4v acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
    dma(kq, 0); dma(kq + ks, 1);
    int sb = kq, it = 0;
    const int nchunk = (nsb + KC_SB - 1) / KC_SB;
    for (int c = 0; c < nchunk; ++c) {
        if (c > 0) __syncthreads();
        for (int i = threadIdx.x; i < 2 * NBC * 16 / 4; i += blockDim.x) reinterpret_cast<float4v *>(s.yd)[i] = (float4v)(0.f);
        __syncthreads();
        stage_q8_2_x4(a, ncols, c, s);
        __syncthreads();
        const int sb_end = min(nsb, KC_SB * (c + 1));
        for (; sb < sb_end; sb += ks, ++it) {
            const int b0 = it % 3, b2 = (it + 2) % 3;
            dma(sb + 2 * ks, b2);
            // tiles it, it + 1, it + 2 are outstanding in issue order: tile `it` has landed once at most 2 tiles' worth of loads remain
            __builtin_amdgcn_s_waitcnt(0x0f70 | ((2 * NI * NMAT) & 0xf) | (((2 * NI * NMAT) >> 4) << 14));      // vmcnt(2 * NI * NMAT); expcnt (bits 6:4) and lgkmcnt (11:8) at their maxima = not waited for
            __builtin_amdgcn_sched_barrier(0);
            if (active) {
                W cur; cur.load(wbuf + b0 * TILE + r * BYTES, kg); cur.compute(8 * (sb - KC_SB * c), ycol, ydl, ysl, acc);
                if (UPGATE) { W cur2; cur2.load(wbuf + (3 + b0) * TILE + r * BYTES, kg); cur2.compute(8 * (sb - KC_SB * c), ycol, ydl, ysl, acc2); }
            }
            __builtin_amdgcn_sched_barrier(0);         // the next DMA into this buffer (two iterations on) must stay behind these reads
        }
    }
    if (ks == 1) {
        if (active && row0 + r < a.M && 4 * kg < ncols) {
#pragma unroll
            for (int i = 0; i < 4; ++i) if (4 * kg + i < ncols) a.C[0][(long)(4 * kg + i) * a.stride_C + row0 + r] = UPGATE ? up_gate_combine(a.unary_op, acc[i], acc2[i], a.epi, row0 + r, 0) : acc[i];
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { red[((wave * 2 + 0) * 16 + r) * 16 + 4 * kg + i] = acc[i]; if (UPGATE) red[((wave * 2 + 1) * 16 + r) * 16 + 4 * kg + i] = acc2[i]; }
    __syncthreads();
    for (int o = threadIdx.x; o < rgw * 16 * ncols; o += blockDim.x) {
        const int g = o / (16 * ncols), oo = o - g * 16 * ncols, rr = oo & 15, cc = oo >> 4; const long orow = ((long)blockIdx.x * rgw + g) * 16 + rr;
        if (orow >= a.M) continue;
        float v = 0.f, v2 = 0.f;
        for (int w = g * ks; w < (g + 1) * ks; ++w) { v += red[((w * 2 + 0) * 16 + rr) * 16 + cc]; if (UPGATE) v2 += red[((w * 2 + 1) * 16 + rr) * 16 + cc]; }
        a.C[0][(long)cc * a.stride_C + orow] = UPGATE ? up_gate_combine(a.unary_op, v, v2, a.epi, orow, 0) : v;
    }
}

size_t mfma_lds_bytes(int ncols, int nw, size_t per_wave = 0) { return (size_t)nw * per_wave + (size_t)ncols * KP + (size_t)2 * NBC * 16 * sizeof(float) + (size_t)nw * 2 * 16 * 16 * sizeof(float); }

template <typename KERNEL>
int launch(const cdna4_context *ctx, KERNEL kernel, const GemvArgs &a, int ncols, hipStream_t st, size_t lds_per_wave = 0) {
    const int nsb = a.K >> 8; const long nrg = ((long)a.M + 15) / 16;
    // waves: ~8 per CU (2 per SIMD); ks waves split the K range of a row group when there are too few row groups
    static const int ks_env = getenv("CDNA4_GEMV_MFMA_KS") ? atoi(getenv("CDNA4_GEMV_MFMA_KS")) : 0;
    int ks = 1; while (ks < 8 && ks * 2 <= nsb && nrg * ks * 2 <= 8L * ctx->num_cu) ks *= 2;
    if (ks_env > 0) ks = std::min(ks_env, 8);
    const int nw = ks <= 4 ? 4 : ks;
    const size_t lds = mfma_lds_bytes(ncols, nw, lds_per_wave);
    if (lds > 160 * 1024 - 256) return -1;
    if (lds > 64 * 1024) { const int rc = cdna4_opt_in_lds((const void *)kernel); if (rc) return rc; }
    const long grid = (nrg + nw / ks - 1) / (nw / ks);
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(64 * nw), lds, st, a, ncols, ks);
    HIP_TRY(hipGetLastError());
    return CDNA4_OK;
}
}  // namespace

// 2..16 columns of pre-quantized (block_q8_2_x4) activations; returns -1 when the type / shape is not served by this kernel (the caller
// falls back to the v_dot4 kernels)
int cdna4_gemv_mfma_launch(const cdna4_context *ctx, int type, const GemvArgs &a, int ncols, hipStream_t st) {
    if (ncols < 2 || ncols > 16 || a.K % 256 != 0 || a.K <= 0 || a.nmat != 1 || a.ids || a.src_f32 || a.q8_out) return -1;
    const bool ug = a.A2 != nullptr;
    // default: per-lane 8-byte weight loads.  CDNA4_GEMV_MFMA_LDS=1 selects the LDS-DMA staged tiles (measured neutral at equal occupancy and slower once
    // its buffers cost the second workgroup per CU, profiles/r02_notes.md)
    static const bool direct = !(getenv("CDNA4_GEMV_MFMA_LDS") && atoi(getenv("CDNA4_GEMV_MFMA_LDS")) != 0);
    switch (type) {
        case T_Q4_K: if (direct) return ug ? launch(ctx, gemv_mfma_kernel<WK<T_Q4_K>, true>, a, ncols, st) : launch(ctx, gemv_mfma_kernel<WK<T_Q4_K>, false>, a, ncols, st);
                     return ug ? launch(ctx, gemv_mfma_lds_kernel<T_Q4_K, true>, a, ncols, st, 2 * 3 * 3072) : launch(ctx, gemv_mfma_lds_kernel<T_Q4_K, false>, a, ncols, st, 3 * 3072);
        case T_Q5_K: if (direct) return ug ? launch(ctx, gemv_mfma_kernel<WK<T_Q5_K>, true>, a, ncols, st) : launch(ctx, gemv_mfma_kernel<WK<T_Q5_K>, false>, a, ncols, st);
                     return ug ? launch(ctx, gemv_mfma_lds_kernel<T_Q5_K, true>, a, ncols, st, 2 * 3 * 3072) : launch(ctx, gemv_mfma_lds_kernel<T_Q5_K, false>, a, ncols, st, 3 * 3072);
        case T_Q6_K: return ug ? launch(ctx, gemv_mfma_kernel<W6, true>, a, ncols, st) : launch(ctx, gemv_mfma_kernel<W6, false>, a, ncols, st);
    }
    return -1;
}
