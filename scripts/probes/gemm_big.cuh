// gemm_big.cuh -- the prompt GEMM for LONG prompts (N a multiple of 256, >= 1024 tokens): one 4-wave workgroup per CU, ONE wave per SIMD with the whole register file.
//
// gemm_mfma_kernel (gemm_mfma.cuh) runs two waves per SIMD with 256 registers each: a wave owns 32 weight rows x 256 tokens (plain) or 32 rows x 128 tokens x {up, gate}
// (fused), i.e. 8 MFMAs per k-step, and every MFMA needs one A fragment (ds_read_b128) of its own in the plain form, every 4 (8) MFMAs one de-quantized B fragment.
// Here a wave owns 16 MFMAs per k-step in 256 accumulator registers:
//     plain : 64 weight rows (two 32-row blocks) x 256 tokens       fused : 32 rows x 256 tokens x {up, gate}
// so every A fragment feeds TWO MFMAs and every B fragment EIGHT: per MFMA half the LDS reads of the plain kernel, half the de-quantization work and half the
// LDS-DMA bytes of the fused one (the workgroup tile is 256 x 256 resp. 128 x 256 x 2 outputs on one 256-token activation tile).  With one wave per SIMD there is no
// partner wave to fill issue gaps, so the unrolled k-steps are interleaved by hand (sched_group_barrier): per MFMA (32 cycles of matrix pipe) ~2 VALU of the NEXT
// k-step's de-quantization and half a ds_read_b128 of its A fragments ride in the shadow of the pipe (MI355X guide: <= 5 single-issue instructions per MFMA gap).
// Same data path as gemm_mfma_kernel otherwise: per-type WTile (raw quant bytes -> registers, B fragments never touch LDS), activations f16 in the slab layout
// global -> LDS by LDS-DMA, XOR-swizzled, double buffered in 64-k tiles (one barrier per 64 MFMAs of a wave), XCD-aware tile order in L2-sized super-columns.
// Results are bit-identical to gemm_mfma_kernel's for the same K order (f32 accumulate of the same products in the same order: the MFMA chain per output is unchanged).
#pragma once

// V (scheduling variant; bit 0: sched_group_barrier interleave pattern, bit 1: software-pipelined B fragments -- the fragments of k-step s + 1 (and, at the end of a K
// tile, the scale preparation of the next one) are produced in program order BEFORE the MFMAs of k-step s, so that they have no dependence on them and can ride under them)
template <int TYPE, bool UPGATE, int V>
__global__ void __launch_bounds__(256, 1) gemm_big_kernel(const GemmArgs a) {
    constexpr bool SGB = (V & 1) != 0, PIPE = (V & 2) != 0;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int NT = 8, KX = 64, BN = 32 * NT, ROWB = KX * 2, PIECES = KX / 8, XT_BYTES = BN * ROWB, NXR = NT * KX / 64, NSUB = 128 / KX, SPS = 8 / NSUB;
    constexpr int NB = 2;                                    // B sources of a wave: plain = two 32-row blocks of the matrix, fused = the same 32 rows of up and of gate
    constexpr int MROWS = UPGATE ? 128 : 256;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, tg = threadIdx.x;
    const int MT = (a.M + MROWS - 1) / MROWS, T = gridDim.x;
    int tile;
    { const int b = blockIdx.x, xcd = b & 7, li = b >> 3, q = T >> 3, r = T & 7;
      tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + li; }
    const int G = a.m_major > 1 ? a.m_major : 1;
    const int sc = tile / (G * MT), rr = tile - sc * G * MT;
    const int m_tile = rr / G, n_tile = sc * G + (rr - m_tile * G);
    const int n0 = n_tile * BN;
    int mrow[NB]; bool m_ok[NB]; const uint8_t *wrow[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        mrow[j] = m_tile * MROWS + wave * (UPGATE ? 32 : 64) + (UPGATE ? 0 : 32 * j) + (lane & 31);
        m_ok[j] = mrow[j] < a.M; if (!m_ok[j]) mrow[j] = a.M - 1;
        wrow[j] = ((UPGATE && j) ? a.A2 : a.A) + (long)mrow[j] * a.strideA;
    }
    const int kt_begin = 0, kt_end = a.K >> 7;

    floatx16 acc[NB][NT];
#pragma unroll
    for (int j = 0; j < NB; ++j) { for (int t = 0; t < NT; ++t) { for (int r = 0; r < 16; ++r) acc[j][t][r] = 0.f; } }

    void *grid_lds = smem + 2 * XT_BYTES;
    if (TYPE == T_IQ2_S) expand_iq2s_grid(a.grid, grid_lds);
    if (TYPE == T_IQ3_S) expand_iq3s_grid(a.grid, grid_lds);

    // activation staging (see gemm_mfma_kernel): slot L = i * 256 + tid, row = L / PIECES, piece' = piece ^ ((row >> 1) & 7)
    const int xrow0 = tg / PIECES, xsw = (xrow0 >> 1) & 7, xpiece = (tg & (PIECES - 1)) ^ xsw;
    const long slab_bytes = a.xrows * 128, xtile_step = (KX / 64) * slab_bytes;
    const char *xthread = reinterpret_cast<const char *>(a.X) + (xpiece >> 3) * slab_bytes + (long)(n0 + xrow0) * 128 + (xpiece & 7) * 16;
    constexpr long xstep = (256 / PIECES) * 128;
    typedef __attribute__((address_space(3))) void lds_void_t;
    typedef const __attribute__((address_space(1))) void glb_void_t;
    const uint32_t xwave_s = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(smem + wave * 1024));
#define XB_ISSUE1(I_, XT_, BUF_) __builtin_amdgcn_global_load_lds((glb_void_t *)(xthread + (I_) * xstep + (long)(XT_) * xtile_step),              \
                                                                  (lds_void_t *)(uintptr_t)(xwave_s + (BUF_) * XT_BYTES + (I_) * 4096), 16, 0, 0)
    const int xt_last = NSUB * kt_end - 1;
    const uint8_t *xlane = smem + (lane & 31) * ROWB;
    const int hx = (WTile<TYPE>::HBIT * h) ^ ((lane >> 1) & 7);

    WTile<TYPE> w0[NB], w1[NB];
#pragma unroll
    for (int i_ = 0; i_ < NXR; ++i_) { XB_ISSUE1(i_, NSUB * kt_begin, 0); }
#pragma unroll
    for (int j = 0; j < NB; ++j) w0[j].load(wrow[j], kt_begin, h);
    half8 bfn[NB];                              // PIPE: the B fragments of the next k-step to run
    if (PIPE) {
        if (gemm_grid_lds_bytes(TYPE) > 0) __syncthreads();        // (the codebook in LDS is read by prepare / frag: every thread's share of the expansion must be there)
#pragma unroll
        for (int j = 0; j < NB; ++j) { w0[j].prepare(h, grid_lds); bfn[j] = w0[j].frag(0, h); }
    }
    int p = 0;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
#pragma unroll
        for (int hh = 0; hh < NSUB; ++hh) {
            __syncthreads();                   // (carries vmcnt(0)) tile in buffer p has landed for every wave; nobody reads buffer p^1 any more
            // (the LDS-DMA pieces are issued UNCONDITIONALLY -- past the last tile the last tile is fetched once more into the buffer nobody reads -- a branch per piece
            //  would cut the 64-MFMA sub-tile into scheduling regions of four MFMAs)
            const int xtn = min(NSUB * kt + hh + 1, xt_last);
            if (hh == 0) {
                const int ktn = min(kt + 1, kt_end - 1);
#pragma unroll
                for (int j = 0; j < NB; ++j) { w1[j].load(wrow[j], ktn, h); if (!PIPE) w0[j].prepare(h, grid_lds); }
            }
            const uint8_t *xb = xlane + p * XT_BYTES;
            half8 af[2][NT];
            { const int poff0 = (((WTile<TYPE>::kpiece(SPS * hh)) & (PIECES - 1)) ^ hx) << 4;
#pragma unroll
              for (int t = 0; t < NT; ++t) af[0][t] = *reinterpret_cast<const half8 *>(xb + t * (32 * ROWB) + poff0); }
#pragma unroll
            for (int s4 = 0; s4 < SPS; ++s4) {
                const int s = SPS * hh + s4;
                if (s4 < SPS - 1) {
                    const int poffn = (((WTile<TYPE>::kpiece(s + 1)) & (PIECES - 1)) ^ hx) << 4;
#pragma unroll
                    for (int t = 0; t < NT; ++t) af[(s4 + 1) & 1][t] = *reinterpret_cast<const half8 *>(xb + t * (32 * ROWB) + poffn);
                }
                half8 bf[NB];
                if (PIPE) {
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        bf[j] = bfn[j];
                        if (s < 7) bfn[j] = w0[j].frag(s + 1, h);
                        else { w1[j].prepare(h, grid_lds); bfn[j] = w1[j].frag(0, h); }        // (w1 was requested at the start of this K tile)
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < NB; ++j) bf[j] = w0[j].frag(s, h);
                }
#pragma unroll
                for (int t = 0; t < NT; ++t) {
#pragma unroll
                    for (int j = 0; j < NB; ++j) acc[j][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[s4 & 1][t], bf[j], acc[j][t], 0, 0, 0);
                    // the 8 LDS-DMA pieces of the next tile: one per 4 MFMAs, all within the first half of the tile's MFMAs (the barrier waits for the last one to land)
                    if (s4 < 2 && (t & 1) == 0) { XB_ISSUE1(4 * s4 + (t >> 1), xtn, p ^ 1); }
                }
            }
            if (SGB) {
                // interleave (one 64-MFMA sub-tile = one scheduling region): MFMA | 2 VALU | MFMA | 2 VALU + 1 LDS read ...  (masks: 0x8 MFMA, 0x2 VALU, 0x100 DS read)
#pragma unroll
                for (int i = 0; i < SPS * NT * NB / 2; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
            p ^= 1;
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) w0[j] = w1[j];
    }
#undef XB_ISSUE1
    // epilogue: C[token][row]; lanes 0..31 of a register hold 32 consecutive weight rows -> 128-byte stores
    // (unroll(full): 128 inlined copies of the fused epilogue are past the default unroll budget, and a rolled loop would index the accumulators dynamically -> scratch)
#pragma clang loop unroll(full)
    for (int t = 0; t < NT; ++t) {
#pragma clang loop unroll(full)
        for (int r = 0; r < 16; ++r) {
            const int tr = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
            const float xs = a.xscale ? a.xscale[n0 + tr] : 1.f;              // undo the f16 range-guard scale of this token (exact: a power of two)
            if (UPGATE) {
                if (m_ok[0]) a.C[(long)(n0 + tr) * a.stride_C + mrow[0]] = up_gate_combine(a.unary_op, acc[0][t][r] * xs, acc[1][t][r] * xs, a.epi, mrow[0], 0);
            } else {
#pragma unroll
                for (int j = 0; j < NB; ++j) if (m_ok[j]) a.C[(long)(n0 + tr) * a.stride_C + mrow[j]] = acc[j][t][r] * xs;
            }
        }
    }
}

// which types get the long-prompt kernel (two more instantiations per type: the six types of the scope table)
static constexpr bool gemm_big_type(int t) { return t == T_Q4_K || t == T_Q5_K || t == T_Q6_K || t == T_IQ4_NL || t == T_IQ2_S || t == T_IQ3_S; }

template <int TYPE, bool UPGATE, int V>
static int launch_gemm_big_v(const GemmArgs &a_in, hipStream_t st) {
    const size_t lds = (size_t)2 * 256 * 64 * 2 + gemm_grid_lds_bytes(TYPE);
    if (lds > 64 * 1024 && cdna4_opt_in_lds((const void *)gemm_big_kernel<TYPE, UPGATE, V>) != 0) return -2;
    GemmArgs a = a_in;
    const long ntl = a.N / 256, mrows = UPGATE ? 128 : 256;
    {   static const long budget = (getenv("CDNA4_GEMM_XBUDGET_MB") ? atol(getenv("CDNA4_GEMM_XBUDGET_MB")) : 4) << 20;
        const long tile_bytes = 256L * a.K * 2; long G = 1;
        for (long d = 1; d <= ntl; ++d) if (ntl % d == 0 && d * tile_bytes <= budget) G = d;
        a.m_major = (int)G; }
    hipLaunchKernelGGL((gemm_big_kernel<TYPE, UPGATE, V>), dim3((unsigned)(((a.M + mrows - 1) / mrows) * ntl)), dim3(256), lds, st, a);
    return 0;
}
#ifndef GEMM_BIG_DEFAULT_V
#define GEMM_BIG_DEFAULT_V 3
#endif
template <int TYPE, bool UPGATE>
static int launch_gemm_big(const GemmArgs &a, hipStream_t st) {
    if constexpr (TYPE == T_Q4_K) {          // developer A/B: the four scheduling variants exist for Q4_K only (CDNA4_GEMM_BIG_V = 0 .. 3)
        static const int v = getenv("CDNA4_GEMM_BIG_V") ? atoi(getenv("CDNA4_GEMM_BIG_V")) : GEMM_BIG_DEFAULT_V;
        switch (v) { case 0: return launch_gemm_big_v<TYPE, UPGATE, 0>(a, st); case 1: return launch_gemm_big_v<TYPE, UPGATE, 1>(a, st); case 2: return launch_gemm_big_v<TYPE, UPGATE, 2>(a, st); default: break; }
    }
    return launch_gemm_big_v<TYPE, UPGATE, GEMM_BIG_DEFAULT_V>(a, st);
}
