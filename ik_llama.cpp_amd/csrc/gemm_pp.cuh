// gemm_pp.cuh -- prompt GEMM, "ping-pong" form: the two waves of every SIMD alternate between a matrix interval and a load / de-quantize interval.
//
// Why (profiles/r05_pmc_gemm_wlds.json): gemm_wlds_kernel keeps its eight waves in lock step -- one barrier per 64-wide stage, every wave reads
// fragments, de-quantizes and multiplies in the same order -- so the two waves of a SIMD want the matrix pipe at the same time and leave it idle at the same time (pipe busy
// 49 %), and every stage ends in `s_waitcnt vmcnt(0)` for loads issued inside that stage.  The guide's remedy for exactly this ceiling ("256^2 + 2 phases tops out at 800-900 TF,
// the 8-phase schedule reaches 1.3 PF") is a role split with loads that stay in flight across barriers.  This kernel is that idea built around the de-quantizer:
//   * same tile as gemm_wlds: 8 waves own 256 tokens x 256 virtual rows (fused up*gate: 128 rows x {up, gate}), 64-wide K stages, two LDS buffers per operand (4 x 32 KiB);
//     a wave multiplies 128 tokens (its GROUP's half: group = wave >> 2) x 64 virtual rows; waves w and w + 4 share a SIMD, one of each group;
//   * time is cut into intervals by workgroup barriers, four per stage.  In its MATRIX interval a wave issues 16 MFMAs (64 tokens x 64 rows x 64 k) and nothing else; in its
//     LOAD interval it requests the fragments of its next matrix interval (ds_read_b128, landed long before the interval ends), de-quantizes two of its four weight
//     fragments of the NEXT stage into the other weight buffer and (first load interval of a stage) issues its share of the next stage's activation DMA and the raw
//     weight loads of the tile after.  Group 1 runs one interval behind group 0, so on every SIMD one wave multiplies while the other loads: the matrix pipe sees a
//     continuous MFMA stream and the VALU / LDS work of the de-quantizer runs beside it instead of in front of it;
//   * no load is waited for in the interval it was issued in: the DMA of stage t + 1 is issued in the first load interval of stage t and drained at the end of the second
//     (two intervals = ~1000 cycles later); fragment reads are waited for at the end of their own load interval (they were its first instructions);
//   * hazards (I_i = interval i; group 0 runs LOAD1(t) COMP1(t) LOAD2(t) COMP2(t) in I_4t-1 .. I_4t+2, group 1 one later): activation buffer of stage t + 1 is last read
//     (as stage t - 1) in I_4t-2 and written from I_4t-1 on; complete (own vmcnt(0) + barrier) before I_4t+3 reads it.  Weight buffer of stage t + 1: last read in I_4t-4,
//     written I_4t-1 .. I_4t+2, every wave's ds_writes retired (lgkmcnt(0)) before the barrier that opens I_4t+3.
// Arithmetic: identical to gemm_wlds_kernel / gemm_mfma_kernel (WTile<TYPE>::frag weights, f16 activations, the same k-values per MFMA, k-steps of a stage in the same order
// into the same accumulator): bit-identical results (tests/test_gpu_prefill.py).
#pragma once
#include "gemm_wlds.cuh"

#ifndef GEMM_PP_PRIO
#define GEMM_PP_PRIO 1
#endif
#ifndef GEMM_PP_DQ_IN_COMP
#define GEMM_PP_DQ_IN_COMP 0
#endif
#ifndef GEMM_PP_DQ_VPM
#define GEMM_PP_DQ_VPM 4
#endif
// experiment builds (scripts/pp_exp.py): GEMM_PP_TIMELINE = s_memtime stamps of one workgroup's intervals (per wave: before the interval's closing lgkmcnt(0), after the barrier),
// kept in LDS and dumped at the end -- cdna4_exp_pp_timeline() copies them out; GEMM_PP_KO_* = knock-out builds (wrong results): no de-quantization arithmetic, no activation DMA,
// no fragment reads, no MFMAs
#ifdef GEMM_PP_TIMELINE
// per interval and wave one record of 8 stamps: [0] after the previous barrier, [1..3] section ends inside the interval (load intervals: fragment reads issued, de-quantization
// + weight-image stores issued, DMA / raw loads issued), [4] end of the interval's work (before the closing waits), [5] behind the PREVIOUS interval's closing waits (before its
// barrier).  s_memtime is asynchronous (LGKM counter): the stamps are only read behind the interval's own lgkmcnt(0), so taking one costs an issue slot, not a round trip.
__device__ unsigned long long g_pp_timeline[8 * 512];
extern "C" __attribute__((visibility("default"))) int cdna4_exp_pp_timeline(void *dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_pp_timeline), sizeof(g_pp_timeline)); }
#define PP_TL_BYTES (8 * 512 * 8)
#define PP_T(I_) asm volatile("s_memtime %0" : "=s"(tl_t##I_));
#define PP_TL_STORE { if (tl_on && tl_n < 64) { if (lane == 0) { tl_lds[8 * tl_n] = tl_t0; tl_lds[8 * tl_n + 1] = tl_t1; tl_lds[8 * tl_n + 2] = tl_t2; tl_lds[8 * tl_n + 3] = tl_t3; tl_lds[8 * tl_n + 4] = tl_t4; tl_lds[8 * tl_n + 5] = tl_t5; } ++tl_n; } \
                      tl_t1 = tl_t2 = tl_t3 = 0; }
#else
#define PP_TL_BYTES 0
#define PP_T(I_)
#define PP_TL_STORE
#endif

template <int TYPE, bool UPGATE>
__global__ void __launch_bounds__(512, 2) gemm_pp_kernel(const GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int ROWS = UPGATE ? 128 : 256;                 // weight rows per workgroup
    constexpr int HB = WTile<TYPE>::HBIT;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), h = lane >> 5;
    const int grp = wave >> 2, wr = wave & 3;                // token half (128 tokens) = role group; virtual-row quarter (64 virtual rows)
    // ---- tile order: as gemm_wlds_kernel
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

    // ---- de-quantizer role (as gemm_wlds_kernel): virtual row dv = tid & 255, half dh = tid >> 8 (= the wave's group)
    const int dv = tid & 255, dh = grp;
    int drow = m0 + (UPGATE ? (dv & 127) : dv); if (drow >= a.M) drow = a.M - 1;
    const uint8_t *wsrc = ((UPGATE && dv >= 128) ? a.A2 : a.A) + (long)drow * a.strideA;
    uint8_t *bdst = bbuf + dv * 128; const int dsw = ((dv >> 1) & 7) ^ ((dv & 1) << 2);
    // ---- activation staging (as gemm_wlds_kernel): slot L = i * 512 + tid (16-byte units)
    const int xr0 = tid >> 3, xpiece = (tid & 7) ^ ((xr0 >> 1) & 7);
    const long slab_bytes = a.xrows * 128;
    const char *xsrc = reinterpret_cast<const char *>(a.X) + (long)(n0 + xr0) * 128 + xpiece * 16;
    const uint32_t xdst_s = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(abuf + wave * 1024));
    auto x_issue = [&](int i, int st, int buf) {
        const char *gp = xsrc + (long)st * slab_bytes + i * (64 * 128);
        const uint32_t l = __builtin_amdgcn_readfirstlane(xdst_s + buf * WLDS_STAGE + i * 8192); uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gp), "s"(l) : "memory");
    };
#ifdef GEMM_PP_KO_DMA
#define PP_X_ISSUE(I_, ST_, BUF_) { (void)(ST_); }
#elif defined(GEMM_PP_FAKE_B_DMA)      /* timing experiment: the weight image filled by DMA too (from the activation slab: wrong data, the traffic of an f16 weight image) */
#define PP_X_ISSUE(I_, ST_, BUF_) { x_issue((I_), (ST_), (BUF_)); x_issue((I_), (ST_), (BUF_) + 2); }
#else
#define PP_X_ISSUE(I_, ST_, BUF_) x_issue((I_), (ST_), (BUF_))
#endif
    // ---- consumer role: A rows 128 grp + 32 t + (lane & 31), t = 0 .. 3; B rows (plain) 64 wr + 32 r + (lane & 31), (fused) 32 wr + 128 r + (lane & 31)
    const int lsw = ((lane & 31) >> 1) & 7, bsw = lsw ^ ((lane & 1) << 2);
    const uint8_t *ard = abuf + (128 * grp + (lane & 31)) * 128;
    const uint8_t *brd = bbuf + ((UPGATE ? 32 * wr : 64 * wr) + (lane & 31)) * 128;
    constexpr int BRT = (UPGATE ? 128 : 32) * 128;

    floatx16 acc[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t) { for (int r = 0; r < 16; ++r) { acc[t][0][r] = 0.f; acc[t][1][r] = 0.f; } }
#ifdef GEMM_PP_TIMELINE
    typedef __attribute__((address_space(3))) unsigned long long lds_u64_t;      // (an LDS pointer type: as a generic pointer the stamps leave as flat stores + vmcnt(0))
    lds_u64_t *tl_lds = (lds_u64_t *)(smem + 4 * WLDS_STAGE + gemm_grid_lds_bytes(TYPE)) + wave * 512;
    const bool tl_on = blockIdx.x == 300; int tl_n = 0;
    unsigned long long tl_t0 = 0, tl_t1 = 0, tl_t2 = 0, tl_t3 = 0, tl_t4 = 0, tl_t5 = 0;
#endif

    const int KT = a.K >> 7, NS = 2 * KT;                    // 64-wide stages
    WTile<TYPE> w0, w1;
#ifdef GEMM_PP_KO_DEQUANT
#define PP_FRAG(S_) half8{(_Float16)(float)(S_), 0, 0, 0, 0, 0, 0, 0}
#else
#define PP_FRAG(S_) w0.frag((S_), dh)
#endif
#define B_PUT(J_, HH_, BUF_) { const int s_ = 4 * (HH_) + (J_); const half8 f_ = PP_FRAG(s_); \
        *reinterpret_cast<half8 *>(bdst + (BUF_) * WLDS_STAGE + ((((WTile<TYPE>::kpiece(s_) + HB * dh) & 7) ^ dsw) << 4)) = f_; }
    // ---- prologue: stage 0 complete in buffers 0 (all waves), the raw weights of tile 1 on their way
#pragma unroll
    for (int i = 0; i < 4; ++i) x_issue(i, 0, 0);
    w0.load(wsrc, 0, dh);
    w1.load(wsrc, min(1, KT - 1), dh);
    __syncthreads();                                         // (codebook expansion visible before prepare() of the grid types reads it)
    w0.prepare(dh, grid_lds);
#pragma unroll
    for (int j = 0; j < 4; ++j) B_PUT(j, 0, 0)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();

    half2v dq_x0, dq_x1, dq_x2, dq_x3;                       // (GEMM_PP_DQ_IN_COMP: a weight fragment in the making)
    half8 af[2][4], bf[2][4];                                // fragments of the wave's next matrix interval: [tile of the 64-token half][k-step], [row tile][k-step]
#ifdef GEMM_PP_KO_READS
#pragma unroll
    for (int j = 0; j < 4; ++j) { af[0][j] = af[1][j] = bf[0][j] = bf[1][j] = half8{(_Float16)(float)lane, 1, 0, 0, 0, 0, 0, 0}; }
#endif
#define PQ(J_) ((WTile<TYPE>::kpiece(J_) & 7) ^ (HB * h))
#ifdef GEMM_PP_KO_READS
#define RD_A(SUB_, P_) { _Pragma("unroll") for (int j = 0; j < 4; ++j) { asm volatile("" : "+v"(af[0][j]), "+v"(af[1][j])); } }
#define RD_B(P_) { _Pragma("unroll") for (int j = 0; j < 4; ++j) { asm volatile("" : "+v"(bf[0][j]), "+v"(bf[1][j])); } }
#else
#define RD_A(SUB_, P_) { const uint8_t *ap_ = ard + (P_) * WLDS_STAGE + (SUB_) * 8192;                                                             \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) { const int po_ = (PQ(j) ^ lsw) << 4;                                                       \
            af[0][j] = *reinterpret_cast<const half8 *>(ap_ + po_); af[1][j] = *reinterpret_cast<const half8 *>(ap_ + 4096 + po_); } }
#define RD_B(P_) { const uint8_t *bp_ = brd + (P_) * WLDS_STAGE;                                                                                   \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) { const int pb_ = (PQ(j) ^ bsw) << 4;                                                       \
            bf[0][j] = *reinterpret_cast<const half8 *>(bp_ + pb_); bf[1][j] = *reinterpret_cast<const half8 *>(bp_ + BRT + pb_); } }
#endif
    // interval boundary: every wave's LDS traffic of the interval has retired (fragment reads landed, weight-image writes visible) before the barrier releases the other role
// (the waits are the BUILTIN form: hipcc's own counter pass sees them and does not repeat them in front of the MFMAs -- as inline asm it added eight lgkmcnt(N) per matrix interval)
#define PP_BAR { PP_T(4) __builtin_amdgcn_s_waitcnt(0xc07f); /* lgkmcnt(0) */ asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); PP_TL_STORE PP_T(5) __builtin_amdgcn_sched_barrier(0); \
                 __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); asm volatile("" ::: "memory"); PP_T(0) }
    // first load interval of stage (HH_, buffer P_): fragments of the first 64 tokens + both row tiles; weight fragments 0, 1 of the next stage; the next stage's activation
    // DMA (this wave's four pieces); at odd stages the de-quantizer moves on to the next 128-wide tile and requests the raw bytes of the one after
#define PP_LOAD1(HH_, P_) {                                                                                                                        \
        RD_A(0, P_) RD_B(P_)                                                                                                                       \
        __builtin_amdgcn_sched_barrier(0); PP_T(1)                                                                                                 \
        const int stn = min(2 * kt + (HH_) + 1, NS - 1);                                                                                           \
        if ((HH_) == 1) { w0 = w1; w0.prepare(dh, grid_lds); }                                                                                     \
        if (!GEMM_PP_DQ_IN_COMP) { B_PUT(0, 1 - (HH_), (P_) ^ 1) B_PUT(1, 1 - (HH_), (P_) ^ 1) }                                                   \
        __builtin_amdgcn_sched_barrier(0); PP_T(2)                                                                                                 \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) PP_X_ISSUE(i, stn, (P_) ^ 1);                                                                \
        if ((HH_) == 1) w1.load(wsrc, min(kt + 2, KT - 1), dh);                                                                                    \
        __builtin_amdgcn_sched_barrier(0); PP_T(3)                                                                                                 \
        PP_BAR }
    // second load interval: fragments of the second 64 tokens (the row-tile fragments stay), weight fragments 2, 3 of the next stage, then this wave's DMA and raw weight
    // loads -- issued one and a half intervals ago -- are drained: the stage after this one is complete once every wave has passed the barrier
#define PP_LOAD2(HH_, P_) {                                                                                                                        \
        RD_A(1, P_)                                                                                                                                \
        __builtin_amdgcn_sched_barrier(0); PP_T(1)                                                                                                 \
        if (!GEMM_PP_DQ_IN_COMP) { B_PUT(2, 1 - (HH_), (P_) ^ 1) B_PUT(3, 1 - (HH_), (P_) ^ 1) }                                                   \
        __builtin_amdgcn_sched_barrier(0); PP_T(2)                                                                                                 \
        __builtin_amdgcn_s_waitcnt(0x0070);        /* vmcnt(0) lgkmcnt(0) */                                                                       \
        PP_BAR }
#ifdef GEMM_PP_KO_MFMA
#define PP_MFMA(A_, B_, C_) ({ asm volatile("" :: "v"(A_), "v"(B_)); (C_); })
#else
#define PP_MFMA(A_, B_, C_) __builtin_amdgcn_mfma_f32_32x32x16_f16((A_), (B_), (C_), 0, 0, 0)
#endif
// GEMM_PP_DQ_IN_COMP (experiment): the de-quantization of the NEXT stage's four weight fragments rides inside the FIRST matrix interval of the stage, one fragment behind each
// k-step's four MFMAs (its VALU in the MFMAs' shadow of the SAME wave; sched_group_barrier pins 1 MFMA : GEMM_PP_DQ_VPM VALU), the load intervals keep only reads and DMA.
// (first matrix interval only: group 1's second one already overlaps group 0's first reads of the next stage)
// Q4_K spelled out in four steps so that each sits behind ONE MFMA (same arithmetic as WTile<T_Q4_K>::frag / dequant8_pk): A = field extraction (2 shifts + 4 and_or),
// B = 4 x pk_fma with the sub-block scale, C = 4 x pk_add of the min term, W = the 16-byte store into the weight image
#define PP_Q4K_A(S_) { const int gi_ = (S_) >> 2, t_ = (S_) & 3; const uint4 &w_ = w0.q[gi_]; const uint32_t b0_ = (t_ & 1) ? w_.z : w_.x, b1_ = (t_ & 1) ? w_.w : w_.y;            \
        const uint32_t mk_ = (t_ & 2) ? 0x00f000f0u : 0x000f000fu; uint32_t mg_ = 0x64006400u; asm("" : "+v"(mg_));                                                        \
        dq_x0 = as_h2(and_or_magic(b0_, mk_, mg_)); dq_x1 = as_h2(and_or_magic(b0_ >> 8, mk_, mg_)); dq_x2 = as_h2(and_or_magic(b1_, mk_, mg_)); dq_x3 = as_h2(and_or_magic(b1_ >> 8, mk_, mg_)); }
#define PP_Q4K_B(S_) { const int jj_ = 2 * ((S_) >> 2) + (((S_) & 3) >> 1); dq_x0 = pk_fma(dq_x0, w0.S[jj_], w0.C[jj_]); dq_x1 = pk_fma(dq_x1, w0.S[jj_], w0.C[jj_]);                    \
        dq_x2 = pk_fma(dq_x2, w0.S[jj_], w0.C[jj_]); dq_x3 = pk_fma(dq_x3, w0.S[jj_], w0.C[jj_]); }
#define PP_Q4K_C(S_) { const int jj_ = 2 * ((S_) >> 2) + (((S_) & 3) >> 1); dq_x0 = dq_x0 + w0.M[jj_]; dq_x1 = dq_x1 + w0.M[jj_]; dq_x2 = dq_x2 + w0.M[jj_]; dq_x3 = dq_x3 + w0.M[jj_]; }
#define PP_Q4K_W(S_, BUF_) { half8 f_; f_[0] = dq_x0[0]; f_[1] = dq_x0[1]; f_[2] = dq_x1[0]; f_[3] = dq_x1[1]; f_[4] = dq_x2[0]; f_[5] = dq_x2[1]; f_[6] = dq_x3[0]; f_[7] = dq_x3[1];        \
        *reinterpret_cast<half8 *>(bdst + (BUF_) * WLDS_STAGE + ((((WTile<TYPE>::kpiece(S_) + HB * dh) & 7) ^ dsw) << 4)) = f_; }
#define PP_SB __builtin_amdgcn_sched_barrier(0);
#define PP_PIN(X_) asm volatile("" : "+v"(X_));
#define PP_PINX asm volatile("" : "+v"(dq_x0), "+v"(dq_x1), "+v"(dq_x2), "+v"(dq_x3));
// GEMM_PP_DQ_IN_COMP (experiment): the de-quantization of the NEXT stage's four weight fragments rides inside the FIRST matrix interval of the stage, a quarter of a fragment
// behind each MFMA (its VALU in the MFMAs' shadow of the SAME wave), the load intervals keep only reads and DMA.
// (first matrix interval only: group 1's second one already overlaps group 0's first reads of the next stage)
#define PP_COMP(SUB_, HH_, P_) {                                                                                                                   \
        if (GEMM_PP_PRIO) __builtin_amdgcn_s_setprio(1);                                                                                           \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                                            \
            if constexpr (GEMM_PP_DQ_IN_COMP && (SUB_) == 0 && TYPE == T_Q4_K) {                                                                   \
                const int s_ = 4 * (1 - (HH_)) + j;                                                                                                \
                /* (an MFMA is a pure register operation: only a volatile asm that "uses" its result keeps it in front of the steps behind it -- sched_barrier alone let */ \
                /*  instruction selection sink all sixteen behind the interval's closing barrier)                                                                        */ \
                acc[0][0] = PP_MFMA(af[0][j], bf[0][j], acc[0][0]); PP_PIN(acc[0][0]) PP_SB PP_Q4K_A(s_) PP_PINX PP_SB                              \
                acc[0][1] = PP_MFMA(af[0][j], bf[1][j], acc[0][1]); PP_PIN(acc[0][1]) PP_SB PP_Q4K_B(s_) PP_PINX PP_SB                              \
                acc[1][0] = PP_MFMA(af[1][j], bf[0][j], acc[1][0]); PP_PIN(acc[1][0]) PP_SB PP_Q4K_C(s_) PP_PINX PP_SB                              \
                acc[1][1] = PP_MFMA(af[1][j], bf[1][j], acc[1][1]); PP_PIN(acc[1][1]) PP_SB PP_Q4K_W(s_, (P_) ^ 1) PP_SB                            \
            } else {                                                                                                                               \
            _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                                                                        \
                acc[2 * (SUB_) + t][0] = PP_MFMA(af[t][j], bf[0][j], acc[2 * (SUB_) + t][0]);                                                       \
                acc[2 * (SUB_) + t][1] = PP_MFMA(af[t][j], bf[1][j], acc[2 * (SUB_) + t][1]);                                                       \
            }                                                                                                                                      \
            if (GEMM_PP_DQ_IN_COMP && (SUB_) == 0) { B_PUT(j, 1 - (HH_), (P_) ^ 1) }                                                               \
            }                                                                                                                                      \
        }                                                                                                                                          \
        if (GEMM_PP_PRIO) __builtin_amdgcn_s_setprio(0);                                                                                           \
        PP_BAR }
#define PP_STAGE(HH_, P_) { PP_LOAD1(HH_, P_) PP_COMP(0, HH_, P_) PP_LOAD2(HH_, P_) PP_COMP(1, HH_, P_) }
    if (grp == 0) {
        for (int kt = 0; kt < KT; ++kt) { PP_STAGE(0, 0) PP_STAGE(1, 1) }
        PP_BAR                                               // (group 1's last matrix interval)
    } else {
        PP_BAR                                               // group 1 runs one interval behind
        for (int kt = 0; kt < KT; ++kt) { PP_STAGE(0, 0) PP_STAGE(1, 1) }
    }
#undef PP_STAGE
#undef PP_SB
#undef PP_PIN
#undef PP_PINX
#undef PP_Q4K_W
#undef PP_Q4K_C
#undef PP_Q4K_B
#undef PP_Q4K_A
#undef PP_MFMA
#undef PP_COMP
#undef PP_LOAD2
#undef PP_LOAD1
#undef PP_BAR
#undef RD_B
#undef RD_A
#undef PQ
#undef B_PUT
#undef PP_FRAG
#undef PP_X_ISSUE
#ifdef GEMM_PP_TIMELINE
    __syncthreads();
    if (tl_on) for (int i = tid; i < 8 * 512; i += 512) g_pp_timeline[i] = ((lds_u64_t *)(smem + 4 * WLDS_STAGE + gemm_grid_lds_bytes(TYPE)))[i];
#endif
    // ---- epilogue (as gemm_wlds_kernel): C[token][row]; the per-token range-guard scales are staged in LDS once
    float *xs_lds = reinterpret_cast<float *>(smem);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
                    const int tr = 128 * grp + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
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

// 0 = launched, 1 = not this kernel's case (the caller goes on), -2 = HIP failure
template <int TYPE>
static int launch_gemm_pp(int num_cu, const GemmArgs &a_in, hipStream_t st) {
    if constexpr (!gemm_wlds_type(TYPE)) { return 1; } else {
    const int env = cdna4_gemm_form();                       // cdna4_set_gemm_form: 3 = this kernel wherever it can run; 1 (default): where it measured faster; 0 / 2: never
    if (env != 3) return 1;                                  // (default dispatch: see below, enabled per shape by measurement)
    if (a_in.moe_tiles || a_in.nmat > 1 || (a_in.K & 127) || a_in.N < WLDS_BT / 2) return 1;
    const int rows = a_in.A2 ? 128 : 256;
    const long mt = (a_in.M + rows - 1) / rows, ntl = (a_in.N + WLDS_BT - 1) / WLDS_BT, wgs = mt * ntl;
    const double fill = (double)wgs / (double)(((wgs + num_cu - 1) / num_cu) * num_cu);
    const double ntok = (double)a_in.N / (double)(ntl * WLDS_BT);
    if (env != 3 && (wgs < (long)(0.8 * num_cu) || fill * ntok < 0.8)) return 1;
    GemmArgs a = a_in;
    { const long budget = 4L << 20, tile_bytes = (long)WLDS_BT * a.K * 2; long G = 1;
      for (long d = 1; d <= ntl; ++d) if (ntl % d == 0 && d * tile_bytes <= budget) G = d;
      a.m_major = (int)G; }
    const size_t lds = 4 * WLDS_STAGE + gemm_grid_lds_bytes(TYPE) + PP_TL_BYTES;
    if (a.A2) {
        if (cdna4_opt_in_lds((const void *)gemm_pp_kernel<TYPE, true>) != 0) return -2;
        hipLaunchKernelGGL((gemm_pp_kernel<TYPE, true>), dim3((unsigned)wgs), dim3(512), lds, st, a);
    } else {
        if (cdna4_opt_in_lds((const void *)gemm_pp_kernel<TYPE, false>) != 0) return -2;
        hipLaunchKernelGGL((gemm_pp_kernel<TYPE, false>), dim3((unsigned)wgs), dim3(512), lds, st, a);
    }
    cdna4_note_launch("gemm_pp type=%d nt=8 upgate=%d kx=64 ks=1 mw=2 xw=0 part=0 grid=%ldx1x1 ksplit=1 g=%d", TYPE, a.A2 ? 1 : 0, wgs, a.m_major);
    return 0;
    }
}

// ---- weights -> f16 image for gemm_ppf_kernel (gemm_ppf.cuh): W16[K / 64][row tiles][256 virtual rows][64] ---------------------------------------------------------------
// One pass per mat-mul over the quantized weights: thread (virtual row dv = tid & 255, half dh = tid >> 8) of a workgroup = the de-quantizer role of gemm_wlds_kernel -- it
// loads its row's 128-wide K tile (WTile<TYPE>::load), prepares the scales and writes the eight fragments WTile::frag gives it (the L0 value rounded once to f16; Q4_K / Q6_K: the
// packed-f16 form) as 16-byte pieces of the image.  A workgroup walks KCH consecutive K tiles of one row tile (the codebook types expand their tables once per workgroup).
// HBM-bound: reads M K bpw / 8, writes 2 M K bytes (Llama-3-8B up + gate, Q4_K: 66 + 235 MB).  Rows past M repeat the last row (never stored by the GEMM).
template <int TYPE, bool UPGATE>
__global__ void __launch_bounds__(512) dequant_slab_kernel(const uint8_t *A, const uint8_t *A2, long strideA, int M, int K, __half *W16, const uint16_t *grid, int kch) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int ROWS = UPGATE ? 128 : 256, HB = WTile<TYPE>::HBIT;
    const int tid = threadIdx.x, dv = tid & 255, dh = tid >> 8;
    const int MT = (M + ROWS - 1) / ROWS, KT = K >> 7, nch = (KT + kch - 1) / kch;
    const int m_tile = blockIdx.x / nch, kt0 = (blockIdx.x - m_tile * nch) * kch, kt1 = min(KT, kt0 + kch);
    // LDS: [0, 64 KiB) the f16 tile of one 128-wide K tile = [2 slabs][256 virtual rows][8 pieces of 16 B], piece' = piece ^ (row & 7) (8 consecutive rows of a ds_write_b128
    // lane group land in 8 different 16-byte bank groups); behind it the expanded codebook of the grid types
    half8 *tile = reinterpret_cast<half8 *>(smem);
    void *grid_lds = smem + 65536;
    if (TYPE == T_IQ2_S) expand_iq2s_grid(grid, grid_lds);
    if (TYPE == T_IQ3_S) expand_iq3s_grid(grid, grid_lds);
    if (TYPE == T_IQ2_XXS) expand_iq2_grid(grid, 256, grid_lds);
    if (TYPE == T_IQ2_XS) expand_iq2_grid(grid, 512, grid_lds);
    if (TYPE == T_IQ3_XXS) expand_iq3xxs_grid(grid, grid_lds);
    if (TYPE == T_IQ1_S || TYPE == T_IQ1_M) expand_iq1_grid(grid, grid_lds, false);
    if (gemm_grid_lds_bytes(TYPE) > 0) __syncthreads();
    int drow = m_tile * ROWS + (UPGATE ? (dv & 127) : dv); if (drow >= M) drow = M - 1;
    const uint8_t *wsrc = ((UPGATE && dv >= 128) ? A2 : A) + (long)drow * strideA;
    WTile<TYPE> w, wn;
    w.load(wsrc, kt0, dh);
    for (int kt = kt0; kt < kt1; ++kt) {
        wn.load(wsrc, min(kt + 1, kt1 - 1), dh);             // the next tile's raw bytes are on their way while this one is converted and written out
        w.prepare(dh, grid_lds);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int piece = WTile<TYPE>::kpiece(s) + HB * dh;
            tile[((piece >> 3) * 256 + dv) * 8 + ((piece & 7) ^ (dv & 7))] = w.frag(s, dh);
        }
        __syncthreads();
        // out: every slab's 256 rows x 128 B are CONTIGUOUS in the image (32 KiB): consecutive lanes store consecutive 16-byte pieces
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int u = tid + 512 * j, sl = u >> 11, row = (u >> 3) & 255, pc = (u & 7) ^ (row & 7);
            *reinterpret_cast<half8 *>(W16 + (((long)(2 * kt + sl) * MT + m_tile) * 256 + row) * 64 + pc * 8) = tile[u];
        }
        __syncthreads();
        w = wn;
    }
}
// the k pairing of TYPE's fused kernels for gemm_ppf_kernel (GemmArgs::pairing)
template <int TYPE> static constexpr int gemm_ppf_pairing() {
    int p = 0;
    for (int j = 0; j < 4; ++j) for (int h = 0; h < 2; ++h) p |= ((WTile<TYPE>::kpiece(j) & 7) ^ (WTile<TYPE>::HBIT * h)) << (3 * (2 * j + h));
    return p;
}
// 0 = launched (*pairing set), -2 = HIP failure.  a: A, A2 (fused), strideA, M, K, grid.
template <int TYPE>
static int launch_dequant_slab(const GemmArgs &a, void *w16, int *pairing, hipStream_t st) {
    const int rows = a.A2 ? 128 : 256, MT = (a.M + rows - 1) / rows, KT = a.K >> 7, kch = gemm_grid_lds_bytes(TYPE) > 0 ? 8 : 4, nch = (KT + kch - 1) / kch;
    const size_t lds = 65536 + gemm_grid_lds_bytes(TYPE);
    *pairing = gemm_ppf_pairing<TYPE>();
    if (a.A2) {
        if (cdna4_opt_in_lds((const void *)dequant_slab_kernel<TYPE, true>) != 0) return -2;
        hipLaunchKernelGGL((dequant_slab_kernel<TYPE, true>), dim3((unsigned)(MT * nch)), dim3(512), lds, st, a.A, a.A2, a.strideA, a.M, a.K, (__half *)w16, a.grid, kch);
    } else {
        if (cdna4_opt_in_lds((const void *)dequant_slab_kernel<TYPE, false>) != 0) return -2;
        hipLaunchKernelGGL((dequant_slab_kernel<TYPE, false>), dim3((unsigned)(MT * nch)), dim3(512), lds, st, a.A, a.A2, a.strideA, a.M, a.K, (__half *)w16, a.grid, kch);
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
