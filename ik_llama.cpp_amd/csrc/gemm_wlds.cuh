// gemm_wlds.cuh -- prompt GEMM for grids that fill the chip with 256-token tiles: the weight tile is de-quantized ONCE per workgroup into LDS.
//
// Why (profiles/r04_notes.md sections 8 / 12, VERDICT r04 "do this" 3): in gemm_mfma_kernel every wave de-quantizes the B fragments of its own 32 rows and reuses them for
// NT token tiles -- 4.9 VALU per MFMA on the fused up*gate launch, the time follows the weight type at equal MFMA work and stretches on boxes with lower compute clocks.
// Here a workgroup of 8 waves owns 256 tokens x 256 "virtual rows" (plain: 256 weight rows; fused up*gate: 128 rows of up + the same 128 rows of gate):
//   * thread (vrow = tid & 255, h = tid >> 8) loads the raw quant bytes of ITS row's 128-wide K tile (WTile<TYPE>::load, the same per-type tiles as gemm_mfma.cuh), turns them
//     into f16 with the same L0 arithmetic (WTile::frag) and stores 4 x 16 bytes per 64-wide stage into the weight image in LDS: 32 weights per thread and stage instead
//     of 64 per wave-lane and NT token tiles -- SQ_INSTS_VALU / SQ_INSTS_MFMA 8.5 -> 5.5 by the counters (profiles/r05_pmc_gemm_wlds.json; the design estimate was ~2: address
//     arithmetic, the fragment packing and the MFMAs themselves are in that count), and the time per stage follows the NON-MFMA instruction count (profiles/r06_notes.md section 3);
//   * a wave then computes 128 tokens x 64 virtual rows (4 x 2 accumulator tiles of v_mfma_f32_32x32x16_f16, 128 registers) with BOTH operands read from LDS
//     (6 ds_read_b128 per 8 MFMAs); plain: 2 x 32 rows, fused: 32 up rows + the same 32 gate rows (the epilogue combines them in registers);
//   * activations: the same f16 slab image as gemm_mfma_kernel, global -> LDS by global_load_lds_dwordx4, swizzle on the source address; the weight image uses the same
//     [row][8 pieces of 16 B] form, piece' = piece ^ ((row >> 1) & 7): ds_read_b128 and the ds_write_b128 of 8 consecutive rows are conflict-free;
//   * stages of 64 k, two LDS buffers per operand (4 x 32 KiB), ONE barrier per stage = per 32 MFMAs of a wave; the de-quantization + stores of stage t + 1 and the LDS-DMA
//     of its activations are spread over the four k-steps of stage t.
// Arithmetic: the products and the accumulation order of gemm_mfma_kernel (f16 weights from WTile::frag, f16 activations, the same 16 k-values per MFMA k-step, k-steps in
// the same order, f32 accumulate): bit-identical to an unsplit launch of the NT-tile kernel (tests/test_gpu_prefill.py compares the two).
#pragma once
#include "gemm_mfma.cuh"

__host__ __device__ constexpr bool gemm_wlds_type(int t) { return t == T_Q4_K || t == T_Q5_K || t == T_Q6_K || t == T_IQ4_NL || t == T_IQ2_S || t == T_IQ3_S; }
constexpr int WLDS_BT = 256, WLDS_STAGE = 32768;          // tokens per tile; bytes of one operand stage (256 rows x 64 k x 2 B)

template <int TYPE, bool UPGATE>
__global__ void __launch_bounds__(512, 2) gemm_wlds_kernel(const GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int ROWS = UPGATE ? 128 : 256;                 // weight rows per workgroup
    constexpr int HB = WTile<TYPE>::HBIT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    const int wt = wave >> 2, wr = wave & 3;                 // token half (128 tokens), virtual-row quarter (64 virtual rows); waves w and w + 4 share a SIMD: one of each half
    // ---- tile order: XCD-contiguous chunks of (super-column of G token tiles, weight tile outer inside it) -- as gemm_mfma_kernel
    const int MT = (a.M + ROWS - 1) / ROWS, T = gridDim.x;
    int tile;
    { const int b = blockIdx.x, xcd = b & 7, li = b >> 3, q = T >> 3, r = T & 7;
      tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + li; }
    const int G = a.m_major > 1 ? a.m_major : 1;
    const int sc = tile / (G * MT), rr = tile - sc * G * MT;
    const int m_tile = rr / G, n_tile = sc * G + (rr - m_tile * G);
    const int n0 = n_tile * WLDS_BT, n_valid = a.N - n0, m0 = m_tile * ROWS;

    uint8_t *abuf = smem, *bbuf = smem + 2 * WLDS_STAGE;
    void *grid_lds = smem + 4 * WLDS_STAGE;
    if (TYPE == T_IQ2_S) expand_iq2s_grid(a.grid, grid_lds);
    if (TYPE == T_IQ3_S) expand_iq3s_grid(a.grid, grid_lds);

    // ---- de-quantizer role: virtual row dv = tid & 255 (fused: rows 0..127 = up, 128..255 = gate), half dh = tid >> 8
    const int dv = tid & 255, dh = tid >> 8;
    int drow = m0 + (UPGATE ? (dv & 127) : dv); if (drow >= a.M) drow = a.M - 1;
    const uint8_t *wsrc = ((UPGATE && dv >= 128) ? a.A2 : a.A) + (long)drow * a.strideA;
    uint8_t *bdst = bbuf + dv * 128; const int dsw = ((dv >> 1) & 7) ^ ((dv & 1) << 2);      // weight image swizzle (see bsw below)
    // ---- activation staging: slot L = i * 512 + tid (16-byte units), row = L >> 3 = 64 i + (tid >> 3), physical piece tid & 7 holds logical piece (tid & 7) ^ swz(row)
    typedef __attribute__((address_space(3))) void lds_void_t;
    typedef const __attribute__((address_space(1))) void glb_void_t;
    const int xr0 = tid >> 3, xpiece = (tid & 7) ^ ((xr0 >> 1) & 7);
    const long slab_bytes = a.xrows * 128;
    const char *xsrc = reinterpret_cast<const char *>(a.X) + (long)(n0 + xr0) * 128 + xpiece * 16;
    const uint32_t xdst_s = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(abuf + wave * 1024));
    // The DMA is issued from inline asm (guide 5.7, glds16_asm): as a builtin it marks the wave's LGKM counter "out of order" for hipcc, and every ds_read wait behind it
    // becomes lgkmcnt(0) -- the fragment prefetch of the next k-step would be waited for at once.  Hidden from the compiler's counters it only ever makes hipcc's own vmcnt
    // waits longer (younger or older extra entries both raise the count); this kernel drains it itself (s_waitcnt vmcnt(0) in front of every barrier).
    auto x_issue = [&](int i, int st, int buf) {
        const char *g = xsrc + (long)st * slab_bytes + i * (64 * 128);
        const uint32_t l = __builtin_amdgcn_readfirstlane(xdst_s + buf * WLDS_STAGE + i * 8192); uint32_t keep;      // (provably wave-uniform for the "s" operand)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(l) : "memory");
    };
#define X_ISSUE(I_, ST_, BUF_) x_issue((I_), (ST_), (BUF_))
    // ---- consumer role: A rows 128 wt + 32 tt + (lane & 31); B rows (plain) 64 wr + 32 rt + (lane & 31), (fused) 32 wr + 128 rt + (lane & 31)
    // activation image (lane-linear LDS-DMA): piece' = piece ^ ((row >> 1) & 7).  Weight image (ds_write_b128: 8 consecutive rows per lane group, banks mod 32): the same with bit 2
    // flipped on odd rows, so that 8 consecutive rows store to 8 different 16-byte bank groups (with the plain form they hit 4: SQ_LDS_BANK_CONFLICT = 20 % of the LDS cycles)
    // while the 16 rows of a ds_read_b128 lane group still cover the 16 groups of the 64 read banks.
    const int lsw = ((lane & 31) >> 1) & 7, bsw = lsw ^ ((lane & 1) << 2);
    const uint8_t *ard = abuf + (128 * wt + (lane & 31)) * 128;
    const uint8_t *brd = bbuf + ((UPGATE ? 32 * wr : 64 * wr) + (lane & 31)) * 128;
    constexpr int BRT = (UPGATE ? 128 : 32) * 128;              // byte distance between the wave's two B row tiles

    floatx16 acc[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t) { for (int r = 0; r < 16; ++r) { acc[t][0][r] = 0.f; acc[t][1][r] = 0.f; } }

    const int KT = a.K >> 7, NS = 2 * KT;                    // 64-wide stages
    WTile<TYPE> w0, w1;
    // one de-quantized fragment of stage half hh (k-steps 4 hh .. 4 hh + 3 of the tile in w0) -> weight image `buf`
#define B_PUT(J_, HH_, BUF_) { const int s_ = 4 * (HH_) + (J_); const half8 f_ = w0.frag(s_, dh); \
        *reinterpret_cast<half8 *>(bdst + (BUF_) * WLDS_STAGE + ((((WTile<TYPE>::kpiece(s_) + HB * dh) & 7) ^ dsw) << 4)) = f_; }
    // prologue: stage 0
#pragma unroll
    for (int i = 0; i < 4; ++i) X_ISSUE(i, 0, 0);
    w0.load(wsrc, 0, dh);
    w1.load(wsrc, min(1, KT - 1), dh);
    __syncthreads();                                         // (codebook expansion visible before prepare() of the grid types reads it)
    w0.prepare(dh, grid_lds);
#pragma unroll
    for (int j = 0; j < 4; ++j) B_PUT(j, 0, 0);

    // Every stage issues the NEXT stage's activation DMA and weight fragments unconditionally (the last stage re-fetches itself into the idle buffers: nobody reads them):
    // no run-time guards inside the k-steps -- a guarded DMA piece cuts the tile body into scheduling regions (profiles/r04_notes.md section 5b).
    // k-step j of a stage contracts the logical pieces kpiece(j) ^ (HBIT h), h = 0 / 1 -- the pairing of gemm_mfma_kernel, so both kernels add the same products in the same order.
// experiment knobs (variant builds, scripts/wlds_exp.py): WLDS_EXP_PRIO = s_setprio(1) around every k-step's MFMA group; WLDS_EXP_TAILSB = a scheduling fence behind it (measured 1-6 % slower)
#ifdef WLDS_EXP_PRIO
#define WLDS_PRIO_UP __builtin_amdgcn_s_setprio(1);
#define WLDS_PRIO_DOWN __builtin_amdgcn_s_setprio(0);
#else
#define WLDS_PRIO_UP
#define WLDS_PRIO_DOWN
#endif
#ifdef WLDS_EXP_TAILSB
#define WLDS_TAIL_SB __builtin_amdgcn_sched_barrier(0);
#else
#define WLDS_TAIL_SB
#endif
#ifdef WLDS_EXP_PINGPONG
#define WLDS_MID_SB __builtin_amdgcn_sched_barrier(0);
#else
#define WLDS_MID_SB
#endif
#define PUT_STEP(J_, HH_, P_) { if ((J_) < 2) { X_ISSUE(2 * (J_), stn, (P_) ^ 1); X_ISSUE(2 * (J_) + 1, stn, (P_) ^ 1); } B_PUT((J_), 1 - (HH_), (P_) ^ 1) }
#define RD_FRAGS(J_, DST_, AP_, BP_) { const int pq_ = (WTile<TYPE>::kpiece(J_) & 7) ^ (HB * h), po_ = (pq_ ^ lsw) << 4, pb_ = (pq_ ^ bsw) << 4;               \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) af[DST_][t] = *reinterpret_cast<const half8 *>((AP_) + t * 4096 + po_);                   \
        bf[DST_][0] = *reinterpret_cast<const half8 *>((BP_) + pb_); bf[DST_][1] = *reinterpret_cast<const half8 *>((BP_) + BRT + pb_); }
#define STAGE(HH_, P_, ORD_)                                                                                                                       \
    {   /* hipcc does NOT wait for an unconditional LDS-DMA at __syncthreads() (ISA: the loop-top barrier carried lgkmcnt(0) only; wrong token rows in 25 % of a tile):        */ \
        /* every wave drains its own DMA (and its weight loads) explicitly, then the barrier publishes this stage: buffers P_ complete, buffers P_ ^ 1 free                  */ \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                                        \
        __syncthreads();                                                                                                                        \
        const int stn = min(2 * kt + (HH_) + 1, NS - 1);                                                                                        \
        const uint8_t *ap = ard + (P_) * WLDS_STAGE, *bp = brd + (P_) * WLDS_STAGE;                                                             \
        half8 af[2][4], bf[2][2];                                                                                                               \
        RD_FRAGS(0, 0, ap, bp)                                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);      /* the first k-step's fragments are on their way before anything else of the stage issues */  \
        if ((HH_) == 1) { w0 = w1; w0.prepare(dh, grid_lds); w1.load(wsrc, min(kt + 2, KT - 1), dh); }                                          \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                                         \
            /* fragments of k-step j + 1 are requested BEFORE the work of k-step j (pinned: left alone the scheduler sinks them to their first use and every k-step opens with */ \
            /* an exposed lgkmcnt(0) -- 9 per stage in the first version of this kernel); LDS returns in order, so the MFMAs of step j wait with a counted lgkmcnt             */ \
            if (j < 3) { RD_FRAGS(j + 1, (j + 1) & 1, ap, bp) }                                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                                                                  \
            if ((ORD_) == 0) { PUT_STEP(j, HH_, P_) WLDS_MID_SB }                                                                               \
            WLDS_PRIO_UP                                                                                                                        \
            _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                                                     \
                acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[j & 1][t], bf[j & 1][0], acc[t][0], 0, 0, 0);                             \
                acc[t][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[j & 1][t], bf[j & 1][1], acc[t][1], 0, 0, 0);                             \
            }                                                                                                                                   \
            WLDS_PRIO_DOWN                                                                                                                      \
            if ((ORD_) == 1) { WLDS_MID_SB PUT_STEP(j, HH_, P_) }                                                                               \
            WLDS_TAIL_SB                                                                                                                        \
        }                                                                                                                                       \
    }
#ifdef WLDS_EXP_PINGPONG
    // the two waves of a SIMD (token halves 0 / 1) run the k-step's two parts in opposite order: one de-quantizes while the other multiplies
    if (wt == 0) { for (int kt = 0; kt < KT; ++kt) { STAGE(0, 0, 0) STAGE(1, 1, 0) } }
    else         { for (int kt = 0; kt < KT; ++kt) { STAGE(0, 0, 1) STAGE(1, 1, 1) } }
#else
    for (int kt = 0; kt < KT; ++kt) { STAGE(0, 0, 0) STAGE(1, 1, 0) }
#endif
#undef STAGE
#undef RD_FRAGS
#undef PUT_STEP
#undef B_PUT
#undef X_ISSUE
    // ---- epilogue: C[token][row]; the per-token range-guard scales are staged in LDS once (see gemm_mfma_kernel)
    float *xs_lds = reinterpret_cast<float *>(smem);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the last stage's idle re-fetch lands in buffer 0, where the scales are staged)
    __syncthreads();
    for (int i = tid; i < WLDS_BT; i += 512) xs_lds[i] = (a.xscale && i < n_valid) ? a.xscale[n0 + i] : 1.f;
    __syncthreads();
#pragma unroll
    for (int rt = 0; rt < (UPGATE ? 1 : 2); ++rt) {
        const int row = m0 + (UPGATE ? 32 * wr : 64 * wr + 32 * rt) + (lane & 31);
        if (row < a.M) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int tr = 128 * wt + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (tr < n_valid) {
                        const float xs = xs_lds[tr];
                        float *dst = a.C + (long)(n0 + tr) * a.stride_C + row;
                        if (UPGATE) *dst = up_gate_combine(a.unary_op, acc[t][0][r] * xs, acc[t][1][r] * xs, a.epi, row, 0);
                        else *dst = acc[t][rt][r] * xs;
                    }
                }
            }
        }
    }
}

// 0 = launched, 1 = not this kernel's case (the caller goes on to gemm_mfma_kernel), -2 = HIP failure
template <int TYPE>
static int launch_gemm_wlds(int num_cu, const GemmArgs &a_in, hipStream_t st) {
    if constexpr (!gemm_wlds_type(TYPE)) { return 1; } else {
    const int env = cdna4_gemm_form();                       // cdna4_set_gemm_form / CDNA4_GEMM_WLDS: 0 = the per-wave de-quantizing kernel everywhere, 2 = this one wherever it can run
    if (!env || a_in.moe_tiles || a_in.nmat > 1 || (a_in.K & 127) || a_in.N < WLDS_BT / 2) return 1;
    // measured on MI355X (profiles/r05_notes.md): the fused up*gate launch gains 5 ... 17 % by type at 4096 tokens; the plain GEMMs (NT = 8: every B fragment already serves
    // 8 token tiles) lose 4 ... 7 % -- they keep the per-wave kernel unless form 2 asks for this one
    if (env != 2 && !a_in.A2) return 1;
    const int rows = a_in.A2 ? 128 : 256;
    const long mt = (a_in.M + rows - 1) / rows, ntl = (a_in.N + WLDS_BT - 1) / WLDS_BT, wgs = mt * ntl;
    const double fill = (double)wgs / (double)(((wgs + num_cu - 1) / num_cu) * num_cu);      // share of the last round of workgroups that is used
    const double ntok = (double)a_in.N / (double)(ntl * WLDS_BT);                             // share of the token tiles that is real tokens
    // measured (fused up*gate 14336 x 4096, us per launch, per-wave kernel -> this one): 4096 tokens (7 whole rounds of 256 workgroups) Q4_K 1115 -> 1037, Q6_K 1394 -> 1240,
    // IQ4_NL 1427 -> 1212; 2048 tokens (3.5 rounds) Q4_K 543 -> 575, Q6_K 723 -> 697; 512 tokens (224 workgroups) Q4_K 156 -> 165: the packed-f16 types (Q4_K / Q5_K, whose
    // per-wave de-quantization is cheapest) need whole rounds, the others win from 85 % fill on
    const double need = (TYPE == T_Q4_K || TYPE == T_Q5_K) ? 0.95 : 0.85;
    if (env != 2 && (wgs < (long)(1.7 * num_cu) || fill * ntok < need)) return 1;
    GemmArgs a = a_in;
    { const long budget = 4L << 20, tile_bytes = (long)WLDS_BT * a.K * 2; long G = 1;
      for (long d = 1; d <= ntl; ++d) if (ntl % d == 0 && d * tile_bytes <= budget) G = d;
      a.m_major = (int)G; }
    const size_t lds = 4 * WLDS_STAGE + gemm_grid_lds_bytes(TYPE);
    if (a.A2) {
        if (cdna4_opt_in_lds((const void *)gemm_wlds_kernel<TYPE, true>) != 0) return -2;
        hipLaunchKernelGGL((gemm_wlds_kernel<TYPE, true>), dim3((unsigned)wgs), dim3(512), lds, st, a);
    } else {
        if (cdna4_opt_in_lds((const void *)gemm_wlds_kernel<TYPE, false>) != 0) return -2;
        hipLaunchKernelGGL((gemm_wlds_kernel<TYPE, false>), dim3((unsigned)wgs), dim3(512), lds, st, a);
    }
    cdna4_note_launch("gemm_wlds type=%d nt=8 upgate=%d kx=64 ks=1 mw=2 xw=0 part=0 grid=%ldx1x1 ksplit=1 g=%d", TYPE, a.A2 ? 1 : 0, wgs, a.m_major);
    return 0;
    }
}
