// gemm_ppf.cuh -- prompt GEMM for LARGE batches: the weights are de-quantized ONCE per mat-mul into an f16 image in HBM (dequant_slab_kernel, gemm_pp.cuh), then this kernel
// multiplies two f16 images with both operands streamed into LDS by LDS-DMA.
//
// Why (profiles/r06_notes.md): with the de-quantizer inside the GEMM -- per wave (gemm_mfma), per workgroup in lock step (gemm_wlds) or in alternating intervals (gemm_pp) --
// the fused up*gate launch of Llama-3-8B stays at 0.355 ... 0.36 of the 2.5 PFLOP/s MFMA peak at 4096 tokens: knock-out builds put the de-quantizer's VALU at 31 % of the
// launch, the weight image's LDS stores and fragment reads at 11 %, and moving that VALU into the shadow of the same wave's MFMAs changed nothing (1078 vs 1084 us).  Without
// the de-quantizer the same structure runs at 0.47.  At 4096 tokens the weights are 1 / 16 of the arithmetic intensity away from being free to pre-process: 117 M weights cost
// ~70 us to de-quantize once (read 66 MB, write 235 MB) against ~250 us saved in the GEMM.  This is also what the reference's CUDA backend does with large batches (convert +
// cuBLAS, ggml-cuda.cu:1723); below ~1500 tokens the extra pass costs more than it saves and the fused kernels keep the launch (cdna4_api.hip, mul_mat_mfma).
//
// Structure = gemm_pp_kernel without the de-quantizer (see there): 8 waves, 256 tokens x 256 virtual rows, 64-wide K stages, two LDS buffers per operand; the two wave groups
// alternate between matrix intervals (16 MFMAs) and load intervals (fragment reads), four workgroup barriers per stage; a wave's eight DMA pieces of the next stage (4 activation,
// 4 weight) are spread over three consecutive intervals -- 3 + 3 + 2, the ones in a matrix interval placed between its MFMAs -- and drained one interval later:
//     group 0, stage t:  L1(t) [3 pieces of t+1]  C1(t) [3]  L2(t) [2]  C2(t) [drain]          (intervals 4t-1 .. 4t+2)
//     group 1, stage t:  L1(t) [3 more of t+1]    C1(t) [2]  L2(t) [drain]  C2(t) [first 3 of t+2]     (intervals 4t .. 4t+3; the first 3 of stage 1 go out before the loop)
// Buffers of stage t+1 are last read (as stage t-1) in interval 4t-2 and first read in 4t+3: every piece is issued in [4t-1, 4t+1] and every wave has drained its own before the
// barrier that opens 4t+3.
// Weight image W16[K / 64][row tiles][256 virtual rows][64] f16 (fused up*gate: virtual rows 0..127 = up rows of the tile, 128..255 = the same gate rows), k order inside a row =
// the order WTile<TYPE>::frag produces = the activation image's; both images take the swizzle piece ^ ((row >> 1) & 7) on the DMA source address and on the fragment reads.
// Arithmetic: f16 weights = WTile<TYPE>::frag values (for Q4_K / Q6_K the packed-f16 form: the SAME bits the fused kernels multiply), same k-steps, same accumulation order:
// bit-identical to gemm_mfma_kernel / gemm_wlds_kernel / gemm_pp_kernel (tests/test_gpu_prefill.py).
#pragma once
#include "gemm_wlds.cuh"

#ifdef PPF_TIMELINE      /* experiment build (scripts/pp_exp.py --tus=gemm_ppf): interval stamps of one workgroup, see gemm_pp.cuh GEMM_PP_TIMELINE */
__device__ unsigned long long g_pp_timeline[8 * 512];
extern "C" __attribute__((visibility("default"))) int cdna4_exp_pp_timeline(void *dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_pp_timeline), sizeof(g_pp_timeline)); }
#define PF_TL_BYTES (8 * 512 * 8)
#define PF_T(I_) asm volatile("s_memtime %0" : "=s"(tl_t##I_));
#define PF_TL_STORE { if (tl_on && tl_n < 64) { if (lane == 0) { tl_lds[8 * tl_n] = tl_t0; tl_lds[8 * tl_n + 1] = tl_t1; tl_lds[8 * tl_n + 2] = tl_t2; tl_lds[8 * tl_n + 3] = tl_t3; tl_lds[8 * tl_n + 4] = tl_t4; tl_lds[8 * tl_n + 5] = tl_t5; } ++tl_n; } \
                      tl_t1 = tl_t2 = tl_t3 = 0; }
#else
#define PF_TL_BYTES 0
#define PF_T(I_)
#define PF_TL_STORE
#endif
template <bool UPGATE>
__global__ void __launch_bounds__(512, 2) gemm_ppf_kernel(const GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int ROWS = UPGATE ? 128 : 256;                 // weight rows per workgroup
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), h = lane >> 5;
    const int grp = wave >> 2, wr = wave & 3;
    const int MT = (a.M + ROWS - 1) / ROWS, T = gridDim.x;
    int tile;
    { const int b = blockIdx.x, xcd = b & 7, li = b >> 3, q = T >> 3, r = T & 7;
      tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + li; }
    const int G = a.m_major > 1 ? a.m_major : 1;
    const int sc = tile / (G * MT), rr = tile - sc * G * MT;
    const int m_tile = rr / G, n_tile = sc * G + (rr - m_tile * G);
    const int n0 = n_tile * WLDS_BT, n_valid = a.N - n0, m0 = m_tile * ROWS;

    uint8_t *abuf = smem, *bbuf = smem + 2 * WLDS_STAGE;
    // ---- DMA: slot L = i * 512 + tid (16-byte units) of a 32 KiB stage image: row = 64 i + (tid >> 3), physical piece tid & 7 holds logical piece (tid & 7) ^ swz(row)
    const int xr0 = tid >> 3, xpiece = (tid & 7) ^ ((xr0 >> 1) & 7);
    const long xslab = a.xrows * 128, wslab = (long)MT * 256 * 128;
    const char *xsrc = reinterpret_cast<const char *>(a.X) + (long)(n0 + xr0) * 128 + xpiece * 16;
    const char *wsrc = reinterpret_cast<const char *>(a.A) + ((long)m_tile * 256 + xr0) * 128 + xpiece * 16;
    const uint32_t adst_s = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(abuf + wave * 1024)), bdst_s = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(bbuf + wave * 1024));
    // piece q = 0 .. 7 of a wave's share: even = activation piece q / 2, odd = weight piece q / 2 (issued from inline asm: see gemm_wlds_kernel)
    auto issue = [&](int q, int st, int buf) {
#ifdef PPF_KO_DMA
        if (st > 0) return;
#endif
        const int i = q >> 1;
        const char *gp = ((q & 1) ? wsrc + (long)st * wslab : xsrc + (long)st * xslab) + i * (64 * 128);
        const uint32_t l = __builtin_amdgcn_readfirstlane(((q & 1) ? bdst_s : adst_s) + buf * WLDS_STAGE + i * 8192);
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(gp), "s"(l) : "memory", "m0");
    };
    const int lsw = ((lane & 31) >> 1) & 7;
    const uint8_t *ard = abuf + (128 * grp + (lane & 31)) * 128;
    const uint8_t *brd = bbuf + ((UPGATE ? 32 * wr : 64 * wr) + (lane & 31)) * 128;
    constexpr int BRT = (UPGATE ? 128 : 32) * 128;

    floatx16 acc[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t) { for (int r = 0; r < 16; ++r) { acc[t][0][r] = 0.f; acc[t][1][r] = 0.f; } }
    const int NS = a.K >> 6;                                 // 64-wide stages
#ifdef PPF_TIMELINE
    typedef __attribute__((address_space(3))) unsigned long long lds_u64_t;
    lds_u64_t *tl_lds = (lds_u64_t *)(smem + 4 * WLDS_STAGE) + wave * 512;
    const bool tl_on = blockIdx.x == 300; int tl_n = 0;
    unsigned long long tl_t0 = 0, tl_t1 = 0, tl_t2 = 0, tl_t3 = 0, tl_t4 = 0, tl_t5 = 0;
#endif
    // ---- prologue: stage 0 complete in buffers 0
#pragma unroll
    for (int q = 0; q < 8; ++q) issue(q, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    half8 af[2][4], bf[2][4];
#ifdef PPF_KO_READS
#pragma unroll
    for (int j = 0; j < 4; ++j) { af[0][j] = af[1][j] = bf[0][j] = bf[1][j] = half8{(_Float16)(float)lane, 1, 0, 0, 0, 0, 0, 0}; }
#endif
    // k-step j of a stage contracts the logical pieces kpiece(j) ^ (HBIT h) of the weight type the image was made from (a.pairing: 3 bits per (j, h), launch_gemm_ppf) -- the
    // pairing of the fused kernels, so that every MFMA adds the same 16 products in the same positions: bit-identical sums
#define PQ(J_) ((a.pairing >> (3 * (2 * (J_) + h))) & 7)
#ifdef PPF_KO_READS
#define RD_A(SUB_, P_) { _Pragma("unroll") for (int j = 0; j < 4; ++j) { asm volatile("" : "+v"(af[0][j]), "+v"(af[1][j])); } }
#define RD_B(P_) { _Pragma("unroll") for (int j = 0; j < 4; ++j) { asm volatile("" : "+v"(bf[0][j]), "+v"(bf[1][j])); } }
#else
#define RD_A(SUB_, P_) { const uint8_t *ap_ = ard + (P_) * WLDS_STAGE + (SUB_) * 8192;                                                             \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) { const int po_ = (PQ(j) ^ lsw) << 4;                                                       \
            af[0][j] = *reinterpret_cast<const half8 *>(ap_ + po_); af[1][j] = *reinterpret_cast<const half8 *>(ap_ + 4096 + po_); } }
#define RD_B(P_) { const uint8_t *bp_ = brd + (P_) * WLDS_STAGE;                                                                                   \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) { const int pb_ = (PQ(j) ^ lsw) << 4;                                                       \
            bf[0][j] = *reinterpret_cast<const half8 *>(bp_ + pb_); bf[1][j] = *reinterpret_cast<const half8 *>(bp_ + BRT + pb_); } }
#endif
#define PF_SB __builtin_amdgcn_sched_barrier(0);
#define PF_BAR { PF_T(4) __builtin_amdgcn_s_waitcnt(0xc07f); /* lgkmcnt(0) */ asm volatile("" ::: "memory"); PF_SB PF_TL_STORE PF_T(5) PF_SB __builtin_amdgcn_s_barrier(); PF_SB asm volatile("" ::: "memory"); PF_T(0) }
#define PF_DRAIN { __builtin_amdgcn_s_waitcnt(0x0070); /* vmcnt(0) lgkmcnt(0) */ }
    // an MFMA is a pure register operation: the volatile asm that "uses" its result keeps it in front of what follows (sched_barrier alone lets instruction selection sink it)
#define PF_MFMA(T_, R_, J_, SUB_) { acc[2 * (SUB_) + (T_)][R_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[T_][J_], bf[R_][J_], acc[2 * (SUB_) + (T_)][R_], 0, 0, 0);            \
        asm volatile("" : "+v"(acc[2 * (SUB_) + (T_)][R_])); }
#define PF_KSTEP(J_, SUB_) { PF_MFMA(0, 0, J_, SUB_) PF_MFMA(0, 1, J_, SUB_) PF_MFMA(1, 0, J_, SUB_) PF_MFMA(1, 1, J_, SUB_) }
    // matrix interval: 16 MFMAs; pieces Q0_ .. Q0_ + NQ_ - 1 of stage ST_ -> buffers BUF_ go out behind k-steps 0, 1, 2
#define PF_COMP(SUB_, NQ_, Q0_, ST_, BUF_) {                                                                                                       \
        __builtin_amdgcn_s_setprio(1);                                                                                                             \
        PF_KSTEP(0, SUB_) if ((NQ_) > 0) issue((Q0_), (ST_), (BUF_));                                                                              \
        PF_KSTEP(1, SUB_) if ((NQ_) > 1) issue((Q0_) + 1, (ST_), (BUF_)); PF_T(1)                                                                  \
        PF_KSTEP(2, SUB_) if ((NQ_) > 2) issue((Q0_) + 2, (ST_), (BUF_));                                                                          \
        PF_KSTEP(3, SUB_) PF_T(2)                                                                                                                  \
        __builtin_amdgcn_s_setprio(0); }
    if (grp == 0) {
        for (int st = 0; st < NS; st += 2) {
#define PF_STAGE0(P_, ST_) { const int stn = min((ST_) + 1, NS - 1);                                                                               \
            RD_A(0, P_) RD_B(P_) PF_SB PF_T(1) issue(0, stn, (P_) ^ 1); issue(1, stn, (P_) ^ 1); issue(2, stn, (P_) ^ 1); PF_T(2) PF_BAR                  \
            PF_COMP(0, 3, 3, stn, (P_) ^ 1) PF_BAR                                                                                                \
            RD_A(1, P_) PF_SB PF_T(1) issue(6, stn, (P_) ^ 1); issue(7, stn, (P_) ^ 1); PF_T(2) PF_BAR                                                  \
            PF_COMP(1, 0, 0, stn, (P_) ^ 1) PF_DRAIN PF_T(3) PF_BAR }
            PF_STAGE0(0, st) PF_STAGE0(1, st + 1)
#undef PF_STAGE0
        }
        PF_BAR                                               // (group 1's last matrix interval)
    } else {
        { const int st1 = min(1, NS - 1); issue(0, st1, 1); issue(1, st1, 1); issue(2, st1, 1); }
        PF_BAR                                               // group 1 runs one interval behind
        for (int st = 0; st < NS; st += 2) {
#define PF_STAGE1(P_, ST_) { const int stn = min((ST_) + 1, NS - 1), stnn = min((ST_) + 2, NS - 1);                                                \
            RD_A(0, P_) RD_B(P_) PF_SB PF_T(1) issue(3, stn, (P_) ^ 1); issue(4, stn, (P_) ^ 1); issue(5, stn, (P_) ^ 1); PF_T(2) PF_BAR                  \
            PF_COMP(0, 2, 6, stn, (P_) ^ 1) PF_BAR                                                                                                \
            RD_A(1, P_) PF_SB PF_T(1) PF_DRAIN PF_T(2) PF_BAR                                                                                      \
            PF_COMP(1, 3, 0, stnn, (P_)) PF_BAR }
            PF_STAGE1(0, st) PF_STAGE1(1, st + 1)
#undef PF_STAGE1
        }
    }
#undef PF_COMP
#undef PF_KSTEP
#undef PF_MFMA
#undef PF_DRAIN
#undef PF_BAR
#undef PF_T
#undef PF_TL_STORE
#undef PF_SB
#undef RD_B
#undef RD_A
#undef PQ
#ifdef PPF_TIMELINE
    __syncthreads();
    if (tl_on) for (int i = tid; i < 8 * 512; i += 512) g_pp_timeline[i] = ((lds_u64_t *)(smem + 4 * WLDS_STAGE))[i];
#endif
    // ---- epilogue (as gemm_wlds_kernel)
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
