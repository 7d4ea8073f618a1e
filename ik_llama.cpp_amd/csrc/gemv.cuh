// gemv.cuh -- decode-time (Ny <= 8) quantized mat-vec kernels for gfx950.
//
// What it computes: the reference's direct (non-repacked) CPU kernels, restated for a 64-wide wavefront:
//   mul_mat_qX_K_q8_2_X4_T  (iqk_gemm_kquants.cpp:782-864, Q4_K/Q5_K)   mul_mat_qY_K_q8_2_X4_T (:937-1033, Q6_K)
//   mul_mat_qX_1_q8_2_T<IQ4_NL_UnpackerU> (iqk_gemm_legacy_quants.cpp:779-786,2360-2366)
//   mul_mat_qX_K_q8_K_IQ_1/_N<DequantizerIQ2S/IQ3S> (iqk_gemm_iquants.cpp:382-492,583-689,728-833)
// i.e. activations are quantized to int8 exactly like quantize_row_q8_2_x4 / iqk_quantize_row_q8_K do,
// the int8 x int8 block sums are exact (v_dot4_i32_i8) and only the f32 scale accumulate differs in order.
//
// MI355X mapping (DESIGN.md "decode GEMV"):
//   * HBM-bound: every weight byte is read exactly once, straight into VGPRs (no LDS round trip for weights).
//   * One lane owns one 64-weight "unit" (a quarter super-block; two 32-blocks for IQ4_NL): its quant bytes
//     are 1-3 naturally contiguous 16-byte pieces, so a wave's load instruction covers a contiguous span of
//     the row and every fetched 128-B line is consumed by that wave within a few instructions.
//   * The activation vector(s) are quantized ONCE PER WORKGROUP in the kernel prologue into LDS (fused
//     quantize: no separate launch, no HBM round trip for the int8 copy); single-column launches then keep
//     each lane's slices in registers, multi-column ones re-read them from LDS per step.
//   * A ring of DEPTH steps of weight loads per lane is in flight; every load is unconditional (clamped index)
//     so that hipcc emits counted s_waitcnt vmcnt(N) instead of vmcnt(0) (profiles/r01_notes.md).
//   * Sub-4-bit codebooks (IQ2_S / IQ3_S) live expanded in LDS (8 KiB / 2 KiB), built from the 3 KiB packed
//     form by the prologue.
#pragma once
#include <type_traits>
#include "cdna4_common.cuh"



#define GEMV_MAX_MATS 4
struct GemvArgs {
    const uint8_t *A[GEMV_MAX_MATS];   // weights; several matrices of the SAME type / row length sharing the activations can be
    float         *C[GEMV_MAX_MATS];   // processed by one launch (q,k,v -- the reference fuses them too, ggml.c:17984-18000)
    int            mend[GEMV_MAX_MATS];// prefix sums of their row counts
    const uint8_t *A2;       // gate weights (fused up-gate, single matrix) or nullptr
    const uint8_t *B;        // activations: f32 rows (src_f32) or pre-quantized vec_dot_type rows
    const uint8_t *tables;   // IQ2_S / IQ3_S: expanded codebook + sign table ([8192 | 2048 B grid][4096 B signs], built once per context), else nullptr
    const int32_t *ids;      // MoE: expert id per (token, slot) pair, else nullptr
    long strideA;            // bytes between weight rows
    long strideB;            // bytes between activation rows
    long stride_C;           // elements between result rows (per activation row)
    long expert_stride;      // MoE: bytes between experts (nb02)
    long nb11, nb12;         // MoE: activation strides in bytes: slot, token (nb11 == 0 => one activation row per token)
    long nb1, nb2;           // MoE: result strides in elements: slot, token
    long ids_nb1;            // MoE: bytes between id rows
    int  M, K;               // total rows (all matrices), row length
    int  nmat;
    int  n_expert, n_used;
    int  pair0;              // MoE: index of the (token, slot) pair of blockIdx.y == 0 (launches are chunked at the 65535 limit of grid.y)
    int  unary_op;           // fused up-gate activation
    UpGateEpilogue epi;      // fused up-gate biases / limit
    int  src_f32;            // 1: B is f32 and is quantized in the prologue
    uint8_t *q8_out;         // fused up-gate, N = 1 only: ALSO emit the result row quantized to block_q8_2_x4 (the next mat-mul's input), else nullptr
    // graph-level fusions of a decoded token (kernel variants FX = 1 / 2, compiled separately so that the plain kernels stay byte-identical):
    const float *norm_w;     // FX = 1: the activation row is RMS-normed in the prologue -- x * rsqrt(mean(x^2) + norm_eps) * norm_w -- before it is quantized
    float norm_eps;
    const float *R;          // FX = 2: residual added in the epilogue, indexed like C[0] (C = W x + R: the ADD that follows attn_output / ffn_down)
    // FX = 4: FX = 1 + the q,k,v epilogue of one decoded token: rows of a kind-0 / kind-1 matrix are rotated (ROPE NORM mode: pairs (2 i, 2 i + 1) inside every head) with the
    // cached (cos, sin) of the token; kind 0 (Q) is stored as f32 to C[g], kinds 1 / 2 (K / V) as f16 to *kv_slot[g] (or C[g] when the slot is null): ROPE + ROPE + CPY + CPY
    const float2 *rope_tab;  // (cos, sin) of pair i of a head for this token (ops.hip rope cache), n_dims / 2 entries
    int rope_hd, rope_nd;    // head size, rotated dims
    int kind[GEMV_MAX_MATS];
    void *const *kv_slot[GEMV_MAX_MATS];
    // FX = 5 (gemv_attn.hip): FX = 2 whose activation row is produced by sibling workgroups of the SAME launch (the decode attention of the token, one workgroup per q head).
    // The weight ring is requested first -- a 4096 x 4096 attn_output matrix is entirely in flight at the top of the launch -- then workgroup thread 0 polls fa_sync[0] (agent
    // scope, relaxed) until the fa_expect producers have arrived; the row is then read with agent-scope (sc1) loads.  fa_sync[1] counts the consumers that are past the wait:
    // the last one re-arms both words (graph replays included).  A spin that runs out (producer never scheduled) sets fa_sync[2] and goes on with whatever the row holds.
    unsigned *fa_sync; unsigned fa_expect;
#ifdef GEMV_EXP_TIMELINE
    long long *timeline;     // [workgroups][8] wall-clock stamps (100 MHz): 0 start, 1 loads issued, 2 prologue done, 3 done, 4 work decomposition done, 5 activation (+ norm weight) loads issued  (scripts/gemv_timeline.py)
#endif
};

// ------------------------------------------------------------------------------------------------
// activation quantization of 8 consecutive floats held by one lane (a10)
// Q8_2_X4 (iqk_quantize.cpp:1072-1166): group = 4 lanes (32 values); Q8_K (:3809-3875): group = 32 lanes.
template <int GROUP>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
    for (int off = 1; off < GROUP; off <<= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
template <int GROUP>
__device__ __forceinline__ int group_sum(int v) {
#pragma unroll
    for (int off = 1; off < GROUP; off <<= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ int clamp_i8(int v) { return v > 127 ? 127 : (v < -128 ? -128 : v); }

// returns packed int8 (2 dwords) and the UNSATURATED integer sum of the lane's 8 quants
__device__ __forceinline__ uint2 quant8(const float4 a, const float4 b, float id, int &isum) {
    const int q0 = (int)rintf(a.x * id), q1 = (int)rintf(a.y * id), q2 = (int)rintf(a.z * id), q3 = (int)rintf(a.w * id);
    const int q4 = (int)rintf(b.x * id), q5 = (int)rintf(b.y * id), q6 = (int)rintf(b.z * id), q7 = (int)rintf(b.w * id);
    isum = q0 + q1 + q2 + q3 + q4 + q5 + q6 + q7;
    uint2 r;
    r.x = (clamp_i8(q0) & 255) | ((clamp_i8(q1) & 255) << 8) | ((clamp_i8(q2) & 255) << 16) | ((uint32_t)(clamp_i8(q3) & 255) << 24);
    r.y = (clamp_i8(q4) & 255) | ((clamp_i8(q5) & 255) << 8) | ((clamp_i8(q6) & 255) << 16) | ((uint32_t)(clamp_i8(q7) & 255) << 24);
    return r;
}
__device__ __forceinline__ float amax8(const float4 a, const float4 b) {
    return fmaxf(fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))),
                 fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w))));
}

// LDS image of the quantized activations of NCOLS columns:
//   yq  int8  [NCOLS][K]
//   yd  float [NCOLS][K/32]  (Q8_2_X4: bf16-rounded block scale)   | Q8_K: float [NCOLS][K/256]
//   ys  float [NCOLS][K/32]  (Q8_2_X4: d * int16 block sum, the reference's `my`, kquants.cpp:818-821)
// LDS sizes of the scale arrays per activation type:
//   Q8_2_X4 : yd[K/32] (bf16-rounded d), ys[K/32] (d * int16 sum)        Q8_K : yd[K/256]
//   Q8_K32  : yd[K/256], ys[K/32] (d * sum_32 as float: the reference stores these in the bsums field, iqk_quantize.cpp:3847-3851)
template <int VDT> __host__ __device__ constexpr int act_scale_block() { return VDT == T_Q8_2_X4 ? 32 : 256; }
template <int VDT> __host__ __device__ constexpr bool act_has_sums() { return VDT == T_Q8_2_X4 || VDT == T_Q8_K32; }

__host__ __device__ constexpr int iq_lds_bytes(int base_type);     // table region of the codebook types ("LDS tables" below)
__host__ __device__ constexpr bool type_is_iq8(int t) { return t == T_IQ2_S || t == T_IQ2_XS || t == T_IQ2_XXS; }     // codebook entry = 8 magnitudes (8 bytes)
__host__ __device__ constexpr bool type_is_iq4(int t) { return t == T_IQ3_S || t == T_IQ3_XXS; }                      // codebook entry = 4 magnitudes (4 bytes)
__host__ __device__ constexpr bool type_is_iq1(int t) { return t == T_IQ1_S || t == T_IQ1_M; }                        // 2048-entry ternary codebook, two signed 8-byte images (8 g + 1, 8 g - 1)
__host__ __device__ constexpr bool type_has_tables(int t) { return type_is_iq8(t) || type_is_iq4(t) || type_is_iq1(t); }
template <int VDT>
__host__ __device__ inline size_t gemv_lds_bytes(int ncols, int K, int base_type) {
    size_t n = (size_t)ncols * K + (size_t)ncols * (K / act_scale_block<VDT>()) * 4 + (act_has_sums<VDT>() ? (size_t)ncols * (K / 32) * 4 : 0);
    n = (n + 15) & ~(size_t)15;
    if (type_has_tables(base_type)) n = ((n + 4095) & ~(size_t)4095) + iq_lds_bytes(base_type);       // table region (4096-aligned, see "LDS tables")
    return n;
}

// quad (4-lane) reductions with DPP quad_perm swaps: no LDS traffic, 2 VALU ops each
__device__ __forceinline__ float quad_max(float v) {
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false)));   // quad_perm [1,0,3,2]
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false)));   // quad_perm [2,3,0,1]
    return v;
}
__device__ __forceinline__ int quad_sum(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);
    return v;
}

// The f32 activations are fetched in two phases so that the FIRST chunks are requested BEFORE the weight prefetch
// (vmcnt retires loads in issue order: an activation load issued behind 12 weight loads would wait for all of them),
// and are quantized while the weight loads are still in flight.
constexpr int XPRE = 4;                                   // chunks (8 floats / lane) preloaded ahead of the weights (upper bound)
// The loads of a chunk iteration are unconditional (see preload_activations_f32) -- an iteration no thread needs still costs every wave a full vector-memory instruction and
// 1 KiB of returned data IN FRONT of the weight ring.  Round 6: single-column launches of rows of <= 4096 values (every mat-vec of a decoded token of an 8B-class model: 512
// chunks over >= 256 threads) ran two (256 threads) or three (512 threads) such empty iterations, twice with a norm in the prologue; with XP = 2 for that family
// (xpre_for) tg128 566 -> 591 tok/s in one call.  Chunks beyond XP * blockDim are fetched by stage_activations_f32's tail loop.
template <int XP> struct XChunksT { float4 v[XP][2]; };
typedef XChunksT<XPRE> XChunks;
// lpr_ct = the kernel's compile-time lanes-per-row: 0 = a single-column launch on rows of <= 2048 values (launch_gemv_y: 256 chunks, one iteration of >= 256 threads)
// wide_wg = the launch always runs 512-thread workgroups (the codebook types: gemv_grid): 512 chunks over 512 threads
__host__ __device__ constexpr int xpre_for(int ncols, int yiters, int lpr_ct = 64, bool wide_wg = false) { return (ncols == 1 && yiters == 1) ? ((lpr_ct == 0 || wide_wg) ? 1 : 2) : XPRE; }

template <int NCOLS, int XP = XPRE>
__device__ __forceinline__ void preload_activations_f32(const GemvArgs &a, const uint8_t *Bbase, XChunksT<XP> &xc) {
    const int k8 = a.K >> 3;
#pragma unroll
    for (int p = 0; p < XP; ++p) {
        // UNCONDITIONAL (index clamped): with a load under a branch hipcc can no longer count outstanding loads and falls back to
        // s_waitcnt vmcnt(0) at the first use -- which made the whole prologue wait for the first weight batch as well.
        const int i = min((int)(threadIdx.x + p * blockDim.x), NCOLS * k8 - 1);
        const int col = NCOLS == 1 ? 0 : i / k8, j = i - col * k8;
        const float *x = reinterpret_cast<const float *>(Bbase + (long)col * a.strideB) + 8 * j;
        xc.v[p][0] = *reinterpret_cast<const float4 *>(x); xc.v[p][1] = *reinterpret_cast<const float4 *>(x + 4);
    }
}

template <int VDT>
__device__ __forceinline__ void quantize_chunk(const float4 v0, const float4 v1, int K, int col, int j, int8_t *yq, float *yd, float *ys) {
    int isum; uint2 q;
    if (VDT == T_Q8_2_X4) {
        const float amax = quad_max(amax8(v0, v1));
        const uint32_t db = float_to_bf16_bits(amax / 127.f);
        const float d = bf16_bits_to_float(db), id = d > 0 ? 1.f / d : 0.f;
        q = quant8(v0, v1, id, isum);
        isum = quad_sum(isum);
        if ((j & 3) == 0) { yd[col * (K >> 5) + (j >> 2)] = d; ys[col * (K >> 5) + (j >> 2)] = d * (float)(int)(short)isum; }
    } else {
        const float amax = group_max<32>(amax8(v0, v1));
        const float d = amax / 127.f, id = amax != 0.0f ? 127.f / amax : 0.0f;
        q = quant8(v0, v1, id, isum);
        if ((j & 31) == 0) yd[col * (K >> 8) + (j >> 5)] = d;
        if (VDT == T_Q8_K32) { isum = quad_sum(isum); if ((j & 3) == 0) ys[col * (K >> 5) + (j >> 2)] = d * (float)isum; }
    }
    *reinterpret_cast<uint2 *>(yq + (long)col * K + 8 * j) = q;
}

template <int VDT, int NCOLS, int XP = XPRE>
__device__ __forceinline__ void stage_activations_f32(const GemvArgs &a, const uint8_t *Bbase, const XChunksT<XP> &xc, int8_t *yq, float *yd, float *ys) {
    const int K = a.K, k8 = K >> 3;
#pragma unroll
    for (int p = 0; p < XP; ++p) {
        const int i = threadIdx.x + p * blockDim.x;
        if (i < NCOLS * k8) { const int col = NCOLS == 1 ? 0 : i / k8, j = i - col * k8; quantize_chunk<VDT>(xc.v[p][0], xc.v[p][1], K, col, j, yq, yd, ys); }
    }
    for (int i = threadIdx.x + XP * blockDim.x; i < NCOLS * k8; i += blockDim.x) {
        const int col = NCOLS == 1 ? 0 : i / k8, j = i - col * k8;
        const float *x = reinterpret_cast<const float *>(Bbase + (long)col * a.strideB) + 8 * j;
        quantize_chunk<VDT>(*reinterpret_cast<const float4 *>(x), *reinterpret_cast<const float4 *>(x + 4), K, col, j, yq, yd, ys);
    }
}

// already-quantized rows in the reference layout (block_q8_2_x4 144 B / 128 values; block_q8_K 296 B / 256).
// Q8_2_X4 (what a fused up*gate launch emits for the following mat-mul): the first QPRE 32-blocks per thread are requested -- like the
// f32 chunks, unconditionally and BEFORE the weight ring -- so that the copy into LDS overlaps the weight stream.
constexpr int QPRE = 2;
typedef unsigned int qreg_t __attribute__((ext_vector_type(4)));      // (HIP's uint4 struct assigns through memcpy and ends up in scratch)
template <int QP> struct QChunksT { qreg_t q[QP][2]; uint32_t d[QP], s[QP]; };     // 32 int8 + bf16 d + int16 sum, raw (any arithmetic on a loaded value here
typedef QChunksT<QPRE> QChunks;                                                    // would make the wave wait for it BEFORE the weight ring is issued)
__host__ __device__ constexpr int qpre_for(int ncols, int yiters) { return (ncols == 1 && yiters == 1) ? 1 : QPRE; }      // (rows of <= 4096 values: 128 blocks over >= 256 threads, see xpre_for)
template <int NCOLS, int QP = QPRE>
__device__ __forceinline__ void preload_activations_q8(const GemvArgs &a, const uint8_t *Bbase, QChunksT<QP> &qc) {
    const int nb = a.K >> 5;
#pragma unroll
    for (int p = 0; p < QP; ++p) {
        const int i = min((int)(threadIdx.x + p * blockDim.x), NCOLS * nb - 1);
        const int col = NCOLS == 1 ? 0 : i / nb, b = i - col * nb;
        const uint8_t *blk = Bbase + (long)col * a.strideB + (long)(b >> 2) * 144; const int ir = b & 3;
        qc.q[p][0] = *reinterpret_cast<const qreg_t *>(blk + 16 + 32 * ir); qc.q[p][1] = *reinterpret_cast<const qreg_t *>(blk + 32 + 32 * ir);
        qc.d[p] = ld16(blk + 2 * ir); qc.s[p] = ld16(blk + 8 + 2 * ir);
    }
}
template <int VDT, int NCOLS, int QP = QPRE>
__device__ __forceinline__ void stage_activations_q8(const GemvArgs &a, const uint8_t *Bbase, const QChunksT<QP> &qc, int8_t *yq, float *yd, float *ys) {
    const int K = a.K;
    if (VDT == T_Q8_2_X4) {
        const int nb = K >> 5;
#pragma unroll
        for (int p = 0; p < QP; ++p) {
            const int i = threadIdx.x + p * blockDim.x;
            if (i < NCOLS * nb) {
                const int col = NCOLS == 1 ? 0 : i / nb, b = i - col * nb;
                uint32_t dbits = qc.d[p], sbits = qc.s[p];
                asm volatile("" : "+v"(dbits), "+v"(sbits));      // (keeps the cheap conversions from being speculated up to the loads, see QChunks)
                const float d = bf16_bits_to_float(dbits); const int sm = (int)(short)sbits;
                yd[col * nb + b] = d; ys[col * nb + b] = d * (float)sm;
                *reinterpret_cast<qreg_t *>(yq + (long)col * K + 32 * b) = qc.q[p][0]; *reinterpret_cast<qreg_t *>(yq + (long)col * K + 32 * b + 16) = qc.q[p][1];
            }
        }
        for (int i = threadIdx.x + QP * blockDim.x; i < NCOLS * nb; i += blockDim.x) {
            const int col = i / nb, b = i - col * nb;
            const uint8_t *blk = Bbase + (long)col * a.strideB + (long)(b >> 2) * 144; const int ir = b & 3;
            const float d = bf16_bits_to_float(ld16(blk + 2 * ir)); const int s = (int)(short)ld16(blk + 8 + 2 * ir);
            yd[col * nb + b] = d; ys[col * nb + b] = d * (float)s;
            const uint4 q0 = *reinterpret_cast<const uint4 *>(blk + 16 + 32 * ir), q1 = *reinterpret_cast<const uint4 *>(blk + 32 + 32 * ir);
            *reinterpret_cast<uint4 *>(yq + (long)col * K + 32 * b) = q0; *reinterpret_cast<uint4 *>(yq + (long)col * K + 32 * b + 16) = q1;
        }
    } else {
        const int n8 = K >> 3;
        for (int i = threadIdx.x; i < NCOLS * n8; i += blockDim.x) {
            const int col = i / n8, j = i - col * n8, b = j >> 5;
            const uint8_t *blk = Bbase + (long)col * a.strideB + (long)b * 296;
            if ((j & 31) == 0) yd[col * (K >> 8) + b] = *reinterpret_cast<const float *>(blk);
            if (VDT == T_Q8_K32 && (j & 3) == 0) ys[col * (K >> 5) + (j >> 2)] = *reinterpret_cast<const float *>(blk + 264 + 4 * ((j & 31) >> 2));
            *reinterpret_cast<uint2 *>(yq + (long)col * K + 8 * j) = *reinterpret_cast<const uint2 *>(blk + 8 + 8 * (j & 31));
        }
    }
}

// expand the packed codebooks (3 KiB in global memory, scripts/gen_iq_tables.py) into LDS: IQ2_S 1024 x 8 magnitudes
// ({8,25,43} from 2-bit codes), IQ3_S 512 x 4 magnitudes (2 c + 1 from 3-bit codes).  Cooperative: all threads of the workgroup.
__device__ __forceinline__ void expand_iq2_grid(const uint16_t *packed, int n, void *lds) {      // IQ2_S (1024 entries), IQ2_XS (512), IQ2_XXS (256)
    uint2 *g = reinterpret_cast<uint2 *>(lds);
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const uint32_t p = packed[i]; uint32_t t0, t1;
        t0 = (p & 3) | ((p & 0xc) << 6) | ((p & 0x30) << 12) | ((p & 0xc0) << 18); const uint32_t ph = p >> 8;
        t1 = (ph & 3) | ((ph & 0xc) << 6) | ((ph & 0x30) << 12) | ((ph & 0xc0) << 18);
        g[i] = make_uint2(t0 * 17u + 0x08080808u + ((t0 >> 1) & 0x01010101u), t1 * 17u + 0x08080808u + ((t1 >> 1) & 0x01010101u));
    }
}
__device__ __forceinline__ void expand_iq2s_grid(const uint16_t *packed, void *lds) { expand_iq2_grid(packed, 1024, lds); }
// IQ1_S / IQ1_M: 2048 entries of 8 values g in {-1, 0, 1} (codes g + 1).  The kernels multiply the integers 8 g + 1 / 8 g - 1 (the weight is dl (g +- 1/8)):
// two images of signed bytes, [8 g + 1: 16 KiB][8 g - 1: 16 KiB], the delta bit picks the image.
__device__ __forceinline__ void expand_iq1_grid(const uint16_t *packed, void *out, bool both = true) {
    uint2 *g = reinterpret_cast<uint2 *>(out);
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) {
        const uint32_t p = packed[i]; uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int v = 8 * ((int)((p >> (2 * j)) & 3) - 1) + 1;
            w[j >> 2] |= (uint32_t)(uint8_t)v << (8 * (j & 3)); w[2 + (j >> 2)] |= (uint32_t)(uint8_t)(v - 2) << (8 * (j & 3));
        }
        g[i] = make_uint2(w[0], w[1]); if (both) g[2048 + i] = make_uint2(w[2], w[3]);
    }
}
__device__ __forceinline__ void expand_iq3s_grid(const uint16_t *packed, void *lds) {
    uint32_t *g = reinterpret_cast<uint32_t *>(lds);
    for (int i = threadIdx.x; i < 512; i += blockDim.x) {
        const uint32_t p = packed[i];
        const uint32_t t = (p & 7) | ((p & 0x38) << 5) | ((p & 0x1c0) << 10) | ((p & 0xe00) << 15);
        g[i] = 2u * t + 0x01010101u;
    }
}
__device__ __forceinline__ void expand_iq3xxs_grid(const uint16_t *packed, void *lds) {      // 256 entries, magnitude 4 + 8 c (c < 7), 62 (c = 7)
    uint32_t *g = reinterpret_cast<uint32_t *>(lds);
    for (int i = threadIdx.x; i < 256; i += blockDim.x) {
        const uint32_t p = packed[i];
        const uint32_t t = (p & 7) | ((p & 0x38) << 5) | ((p & 0x1c0) << 10) | ((p & 0xe00) << 15);
        g[i] = 8u * t + 0x04040404u + 2u * ((t & (t >> 1) & (t >> 2)) & 0x01010101u);
    }
}

// ------------------------------------------------------------------------------------------------
// per-type 64-weight units.
//   Unit<TYPE>            raw quant bytes of one unit (what the ring buffers hold while the loads are in flight)
//   Unit::Dec             the weight side decoded once per unit: int8 dwords + f32 scales (column independent)
//   YReg                  one column's activations for one unit: 16 dwords of int8 + up to 4 floats of scales
//   Unit::load_y          fetch a YReg from the LDS activation image
//   Unit::dot             exact int32 block sums (v_dot4_i32_i8) + f32 scale accumulate in the reference's operation order
struct YReg { uint32_t q[16]; float s[4]; };

__device__ __forceinline__ void ld_y64(const int8_t *p, YReg &y) {           // 64 contiguous int8
    const uint4 *v = reinterpret_cast<const uint4 *>(p);
    const uint4 a = v[0], b = v[1], c = v[2], d = v[3];
    y.q[0] = a.x; y.q[1] = a.y; y.q[2] = a.z; y.q[3] = a.w; y.q[4] = b.x; y.q[5] = b.y; y.q[6] = b.z; y.q[7] = b.w;
    y.q[8] = c.x; y.q[9] = c.y; y.q[10] = c.z; y.q[11] = c.w; y.q[12] = d.x; y.q[13] = d.y; y.q[14] = d.z; y.q[15] = d.w;
}

template <int TYPE> struct Unit;

// ---- Q4_K : lane = (super-block, 64-group g): header 16 B + qs[32g..32g+31]
template <> struct Unit<T_Q4_K> {
    uint4 h, q0, q1;
    struct Dec { uint32_t lo[8], hi[8]; float d_lo, d_hi, m_lo, m_hi; };
    __device__ __forceinline__ uint32_t checksum() const { return h.x ^ h.y ^ h.z ^ h.w ^ q0.x ^ q0.y ^ q0.z ^ q0.w ^ q1.x ^ q1.y ^ q1.z ^ q1.w; }
    __device__ __forceinline__ void zero() { h = q0 = q1 = make_uint4(0, 0, 0, 0); }
    __device__ __forceinline__ void load(const uint8_t *row, int u) {
        const uint8_t *b = row + (long)(u >> 2) * 144;
        h = ldw128(b); q0 = ldw128(b + 16 + 32 * (u & 3)); q1 = ldw128(b + 32 + 32 * (u & 3));
    }
    template <int VDT>
    static __device__ __forceinline__ void load_y(int u, int K, int c, const int8_t *yq, const float *yd, const float *ys, YReg &y) {
        ld_y64(yq + (long)c * K + 64 * u, y);
        const float2 sy = *reinterpret_cast<const float2 *>(ys + c * (K >> 5) + 2 * u);
        if (VDT == T_Q8_2_X4) { const float2 dy = *reinterpret_cast<const float2 *>(yd + c * (K >> 5) + 2 * u); y.s[0] = dy.x; y.s[1] = dy.y; }
        else { y.s[0] = y.s[1] = yd[c * (K >> 8) + (u >> 2)]; }               // Q8_K32: one scale per 256 (the _R4 kernels' activation type)
        y.s[2] = sy.x; y.s[3] = sy.y;
    }
    __device__ __forceinline__ void decode(int u, const void *, Dec &dc) const {
        const int g = u & 3;
        const float d = half_bits_to_float(h.x & 0xffff), dmin = half_bits_to_float(h.x >> 16);
        uint32_t sc03, sc47, mn03, mn47; k4_unpack_scales(h.y, h.z, h.w, sc03, sc47, mn03, mn47);
        const uint32_t scw = ((g & 2) ? sc47 : sc03) >> (16 * (g & 1)), mnw = ((g & 2) ? mn47 : mn03) >> (16 * (g & 1));
        dc.d_lo = d * (float)(scw & 0xff); dc.d_hi = d * (float)((scw >> 8) & 0xff);
        dc.m_lo = dmin * (float)(mnw & 0xff); dc.m_hi = dmin * (float)((mnw >> 8) & 0xff);
        const uint32_t q[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
        for (int i = 0; i < 8; ++i) { dc.lo[i] = q[i] & 0x0f0f0f0fu; dc.hi[i] = (q[i] >> 4) & 0x0f0f0f0fu; }
    }
    static __device__ __forceinline__ float dot(const Dec &dc, const YReg &y, float r) {
        int s_lo = 0, s_hi = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) { s_lo = dot4(dc.lo[i], y.q[i], s_lo); s_hi = dot4(dc.hi[i], y.q[8 + i], s_hi); }
        r = fmaf(dc.d_lo * y.s[0], (float)s_lo, r); r = fmaf(dc.d_hi * y.s[1], (float)s_hi, r);
        r = fmaf(y.s[2], -dc.m_lo, r);              r = fmaf(y.s[3], -dc.m_hi, r);
        return r;
    }
};

// ---- Q5_K : as Q4_K plus the 32 qh bytes (bit 2g / 2g+1 of qh[l] adds 16)
template <> struct Unit<T_Q5_K> {
    uint4 h, q0, q1, h0, h1;
    typedef Unit<T_Q4_K>::Dec Dec;
    __device__ __forceinline__ uint32_t checksum() const { return h.x ^ q0.x ^ q1.x ^ h0.x ^ h1.x; }
    __device__ __forceinline__ void zero() { h = q0 = q1 = h0 = h1 = make_uint4(0, 0, 0, 0); }
    __device__ __forceinline__ void load(const uint8_t *row, int u) {
        const uint8_t *b = row + (long)(u >> 2) * 176;
        h = ldw128(b); h0 = ldw128(b + 16); h1 = ldw128(b + 32);
        q0 = ldw128(b + 48 + 32 * (u & 3)); q1 = ldw128(b + 64 + 32 * (u & 3));
    }
    template <int VDT>
    static __device__ __forceinline__ void load_y(int u, int K, int c, const int8_t *yq, const float *yd, const float *ys, YReg &y) { Unit<T_Q4_K>::load_y<VDT>(u, K, c, yq, yd, ys, y); }
    __device__ __forceinline__ void decode(int u, const void *, Dec &dc) const {
        const int g = u & 3;
        const float d = half_bits_to_float(h.x & 0xffff), dmin = half_bits_to_float(h.x >> 16);
        uint32_t sc03, sc47, mn03, mn47; k4_unpack_scales(h.y, h.z, h.w, sc03, sc47, mn03, mn47);
        const uint32_t scw = ((g & 2) ? sc47 : sc03) >> (16 * (g & 1)), mnw = ((g & 2) ? mn47 : mn03) >> (16 * (g & 1));
        dc.d_lo = d * (float)(scw & 0xff); dc.d_hi = d * (float)((scw >> 8) & 0xff);
        dc.m_lo = dmin * (float)(mnw & 0xff); dc.m_hi = dmin * (float)((mnw >> 8) & 0xff);
        const uint32_t q[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w}, hb[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            dc.lo[i] = (q[i] & 0x0f0f0f0fu) | (((hb[i] >> (2 * g)) & 0x01010101u) << 4);
            dc.hi[i] = ((q[i] >> 4) & 0x0f0f0f0fu) | (((hb[i] >> (2 * g + 1)) & 0x01010101u) << 4);
        }
    }
    static __device__ __forceinline__ float dot(const Dec &dc, const YReg &y, float r) { return Unit<T_Q4_K>::dot(dc, y, r); }
};

// ---- Q6_K : lane = (super-block, half n, l0 in {0,16}): ql[64n+l0..+16), ql[64n+32+l0..+16), qh[32n+l0..+16)
// -> elements 128n + {0,32,64,96} + l0 + [0,16), one int8 scale per 16-element piece.
template <> struct Unit<T_Q6_K> {
    uint4 la, lb, qh; uint2 sc; uint32_t dh;
    struct Dec { uint32_t q[16]; float ds[4]; };
    __device__ __forceinline__ uint32_t checksum() const { return la.x ^ lb.x ^ qh.x ^ sc.x ^ dh; }
    __device__ __forceinline__ void zero() { la = lb = qh = make_uint4(0, 0, 0, 0); sc = make_uint2(0, 0); dh = 0; }
    __device__ __forceinline__ void load(const uint8_t *row, int u) {
        const uint8_t *b = row + (long)(u >> 2) * 210; const int n = (u >> 1) & 1, l0 = 16 * (u & 1);
        la = ld128(b + 64 * n + l0); lb = ld128(b + 64 * n + 32 + l0); qh = ld128(b + 128 + 32 * n + l0);
        sc = ld64(b + 192 + 8 * n); dh = ld16(b + 208);
    }
    template <int VDT>
    static __device__ __forceinline__ void load_y(int u, int K, int c, const int8_t *yq, const float *yd, const float *, YReg &y) {
        const int blk = u >> 2, n = (u >> 1) & 1, l0 = 16 * (u & 1);
        const int8_t *yb = yq + (long)c * K + 256 * blk + 128 * n + l0;
#pragma unroll
        for (int j = 0; j < 4; ++j) { const uint4 v = *reinterpret_cast<const uint4 *>(yb + 32 * j); y.q[4 * j] = v.x; y.q[4 * j + 1] = v.y; y.q[4 * j + 2] = v.z; y.q[4 * j + 3] = v.w; }
        if (VDT == T_Q8_2_X4) { const float4 dy = *reinterpret_cast<const float4 *>(yd + c * (K >> 5) + 8 * blk + 4 * n); y.s[0] = dy.x; y.s[1] = dy.y; y.s[2] = dy.z; y.s[3] = dy.w; }
        else { y.s[0] = y.s[1] = y.s[2] = y.s[3] = yd[c * (K >> 8) + blk]; }  // Q8_K (Q6_K_R4's activation type)
    }
    __device__ __forceinline__ void decode(int u, const void *, Dec &dc) const {
        const int h = u & 1; const float d = half_bits_to_float(dh);
        const uint32_t s01 = sc.x >> (8 * h), s23 = sc.y >> (8 * h);            // scales is = h + {0,2,4,6} of this half
        dc.ds[0] = d * (float)(int)(int8_t)(s01 & 0xff); dc.ds[1] = d * (float)(int)(int8_t)((s01 >> 16) & 0xff);
        dc.ds[2] = d * (float)(int)(int8_t)(s23 & 0xff); dc.ds[3] = d * (float)(int)(int8_t)((s23 >> 16) & 0xff);
        const uint32_t A[4] = {la.x, la.y, la.z, la.w}, Bq[4] = {lb.x, lb.y, lb.z, lb.w}, H[4] = {qh.x, qh.y, qh.z, qh.w};
#pragma unroll
        // the 6-bit fields stay UNSIGNED (0 .. 63 fits a signed byte of v_dot4_i32_i8): sum (q - 32) y = sum q y - 32 sum y, both exact in int32, and sum y of a lane's 16
        // activations does not depend on the row -- with register-resident activations hipcc hoists it out of the row loop.  Round 6: the per-byte "- 32" (or 0x80, subtract,
        // xor 0x80: three VALU per dword, 48 per 64 weights) was a third of this type's decode arithmetic; results are bit-identical (same integers, same float operations).
        for (int i = 0; i < 4; ++i) {
#ifdef GEMV_Q6K_SIGNED      /* the form of rounds 1-5, kept for A/B builds (scripts/pp_exp.py --tus=gemv_14_plain,gemv_14_upgate,gemv_dual): ((q | 0x80) - 0x20) ^ 0x80 == q - 32 per byte */
            dc.q[i]      = ((((A[i] & 0x0f0f0f0fu) | ((H[i] & 0x03030303u) << 4)) | 0x80808080u) - 0x20202020u) ^ 0x80808080u;
            dc.q[4 + i]  = ((((Bq[i] & 0x0f0f0f0fu) | (((H[i] >> 2) & 0x03030303u) << 4)) | 0x80808080u) - 0x20202020u) ^ 0x80808080u;
            dc.q[8 + i]  = (((((A[i] >> 4) & 0x0f0f0f0fu) | (((H[i] >> 4) & 0x03030303u) << 4)) | 0x80808080u) - 0x20202020u) ^ 0x80808080u;
            dc.q[12 + i] = (((((Bq[i] >> 4) & 0x0f0f0f0fu) | (((H[i] >> 6) & 0x03030303u) << 4)) | 0x80808080u) - 0x20202020u) ^ 0x80808080u;
#else
            dc.q[i]      = (A[i] & 0x0f0f0f0fu) | ((H[i] & 0x03030303u) << 4);
            dc.q[4 + i]  = (Bq[i] & 0x0f0f0f0fu) | (((H[i] >> 2) & 0x03030303u) << 4);
            dc.q[8 + i]  = ((A[i] >> 4) & 0x0f0f0f0fu) | (((H[i] >> 4) & 0x03030303u) << 4);
            dc.q[12 + i] = ((Bq[i] >> 4) & 0x0f0f0f0fu) | (((H[i] >> 6) & 0x03030303u) << 4);
#endif
        }
    }
    static __device__ __forceinline__ float dot(const Dec &dc, const YReg &y, float r) {
        int s[4] = {0, 0, 0, 0}, ysum[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                s[j] = dot4(dc.q[4 * j + i], y.q[4 * j + i], s[j]);
#ifndef GEMV_Q6K_SIGNED
                ysum[j] = dot4(0x01010101u, y.q[4 * j + i], ysum[j]);
#endif
            }
#pragma unroll
        for (int j = 0; j < 4; ++j) r = fmaf(dc.ds[j] * y.s[j], (float)(s[j] - 32 * ysum[j]), r);
        return r;
    }
};

// ---- IQ4_NL, Q4_0 : lane = two consecutive 18-byte blocks (36 B, 4-byte aligned)
template <int NT4> struct UnitNib {       // NT4 = T_IQ4_NL (codebook) or T_Q4_0 (nibble - 8): same 18-byte block {f16 d; u8 qs[16]}
    uint32_t w[9];
    struct Dec { uint32_t v[16]; float d0, d1; };
    __device__ __forceinline__ uint32_t checksum() const { return w[0] ^ w[8]; }
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int i = 0; i < 9; ++i) w[i] = 0;
    }
    __device__ __forceinline__ void load(const uint8_t *row, int u) {
        const uint32_t *p = reinterpret_cast<const uint32_t *>(row + (long)u * 36);
#pragma unroll
        for (int i = 0; i < 9; ++i) w[i] = p[i];
    }
    template <int VDT>
    static __device__ __forceinline__ void load_y(int u, int K, int c, const int8_t *yq, const float *yd, const float *, YReg &y) {
        ld_y64(yq + (long)c * K + 64 * u, y);
        const float2 dy = *reinterpret_cast<const float2 *>(yd + c * (K >> 5) + 2 * u); y.s[0] = dy.x; y.s[1] = dy.y;
    }
    __device__ __forceinline__ void decode(int, const void *, Dec &dc) const {
        dc.d0 = half_bits_to_float(w[0] & 0xffff); dc.d1 = half_bits_to_float(w[4] >> 16);
#pragma unroll
        for (int i = 0; i < 4; ++i) {     // v[0..3]: elements 0..15 (low nibbles), v[4..7]: 16..31 of block 0; v[8..15] block 1
            const uint32_t a = __builtin_amdgcn_alignbyte(w[i + 1], w[i], 2), b = w[5 + i];
            dc.v[i] = nib4_to_i8<NT4>(a & 0x0f0f0f0fu); dc.v[4 + i] = nib4_to_i8<NT4>((a >> 4) & 0x0f0f0f0fu);
            dc.v[8 + i] = nib4_to_i8<NT4>(b & 0x0f0f0f0fu); dc.v[12 + i] = nib4_to_i8<NT4>((b >> 4) & 0x0f0f0f0fu);
        }
    }
    static __device__ __forceinline__ float dot(const Dec &dc, const YReg &y, float r) {
        int s0 = 0, s1 = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) { s0 = dot4(dc.v[i], y.q[i], s0); s1 = dot4(dc.v[8 + i], y.q[8 + i], s1); }
        r = fmaf(dc.d0 * y.s[0], (float)s0, r); r = fmaf(dc.d1 * y.s[1], (float)s1, r);
        return r;
    }
};

template <> struct Unit<T_IQ4_NL> : UnitNib<T_IQ4_NL> {};
template <> struct Unit<T_Q4_0> : UnitNib<T_Q4_0> {};          // (reference: mul_mat_qX_1_q8_2_T<Q4_0_1_Unpacker>, iqk_gemm_legacy_quants.cpp:768,2338 -- unsigned nibbles + a -8 d sum(y) term; same value)

// ---- MXFP4 : lane = two consecutive 17-byte blocks {u8 e; u8 qs[16]} (34 B, 2-byte aligned); the IQ4_NL arithmetic with the e2m1 table and a power-of-two
// block scale (MXFP4_Unpacker, iqk_gemm_legacy_quants.cpp:774-779: unsigned values + a -12 d sum(y) term; same value)
template <> struct Unit<T_MXFP4> {
    uint32_t w[9];
    typedef UnitNib<T_IQ4_NL>::Dec Dec;
    __device__ __forceinline__ uint32_t checksum() const { return w[0] ^ w[8]; }
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int i = 0; i < 9; ++i) w[i] = 0;
    }
    __device__ __forceinline__ void load(const uint8_t *row, int u) {
        // the 34 bytes start 0 or 2 bytes into a dword: nine ALIGNED dwords + a byte shift instead of 2-byte-aligned 16-byte loads
        const uint8_t *b = row + (long)u * 34; const uint32_t sh = (uint32_t)(uintptr_t)b & 3u;
        const uint32_t *p = reinterpret_cast<const uint32_t *>(b - sh); uint32_t r[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) r[i] = p[i];
#pragma unroll
        for (int i = 0; i < 8; ++i) w[i] = __builtin_amdgcn_alignbyte(r[i + 1], r[i], sh);
        w[8] = __builtin_amdgcn_alignbyte(0u, r[8], sh);
    }
    template <int VDT>
    static __device__ __forceinline__ void load_y(int u, int K, int c, const int8_t *yq, const float *yd, const float *ys, YReg &y) { Unit<T_IQ4_NL>::template load_y<VDT>(u, K, c, yq, yd, ys, y); }
    __device__ __forceinline__ void decode(int, const void *, Dec &dc) const {
        dc.d0 = e8m0_half(w[0] & 0xff); dc.d1 = e8m0_half((w[4] >> 8) & 0xff);
#pragma unroll
        for (int i = 0; i < 4; ++i) {     // block 0: bytes 1..16, block 1: bytes 18..33
            const uint32_t a = __builtin_amdgcn_alignbyte(w[i + 1], w[i], 1), b = __builtin_amdgcn_alignbyte(w[i + 5], w[i + 4], 2);
            dc.v[i] = mxfp4_lookup4(a & 0x0f0f0f0fu); dc.v[4 + i] = mxfp4_lookup4((a >> 4) & 0x0f0f0f0fu);
            dc.v[8 + i] = mxfp4_lookup4(b & 0x0f0f0f0fu); dc.v[12 + i] = mxfp4_lookup4((b >> 4) & 0x0f0f0f0fu);
        }
    }
    static __device__ __forceinline__ float dot(const Dec &dc, const YReg &y, float r) { return UnitNib<T_IQ4_NL>::dot(dc, y, r); }
};

// ---- Q8_0 : lane = two consecutive 34-byte blocks {f16 d; i8 qs[32]} (68 B, 4-byte aligned)      (Q8_0_1_Unpacker, iqk_gemm_legacy_quants.cpp:753,2353)
template <> struct Unit<T_Q8_0> {
    uint32_t w[17];
    struct Dec { uint32_t v[16]; float d0, d1; };
    __device__ __forceinline__ uint32_t checksum() const { return w[0] ^ w[16]; }
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int i = 0; i < 17; ++i) w[i] = 0;
    }
    __device__ __forceinline__ void load(const uint8_t *row, int u) {
        const uint32_t *p = reinterpret_cast<const uint32_t *>(row + (long)u * 68);
#pragma unroll
        for (int i = 0; i < 17; ++i) w[i] = p[i];
    }
    template <int VDT>
    static __device__ __forceinline__ void load_y(int u, int K, int c, const int8_t *yq, const float *yd, const float *ys, YReg &y) { Unit<T_IQ4_NL>::template load_y<VDT>(u, K, c, yq, yd, ys, y); }
    __device__ __forceinline__ void decode(int, const void *, Dec &dc) const {
        dc.d0 = half_bits_to_float(w[0] & 0xffff); dc.d1 = half_bits_to_float(w[8] >> 16);
#pragma unroll
        for (int i = 0; i < 8; ++i) { dc.v[i] = __builtin_amdgcn_alignbyte(w[i + 1], w[i], 2); dc.v[8 + i] = w[9 + i]; }
    }
    static __device__ __forceinline__ float dot(const Dec &dc, const YReg &y, float r) {
        int s0 = 0, s1 = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) { s0 = dot4(dc.v[i], y.q[i], s0); s1 = dot4(dc.v[8 + i], y.q[8 + i], s1); }
        r = fmaf(dc.d0 * y.s[0], (float)s0, r); r = fmaf(dc.d1 * y.s[1], (float)s1, r);
        return r;
    }
};

// ---- IQ4_XS : 136-byte super-blocks {f16 d; u16 scales_h; u8 scales_l[4]; u8 qs[128]}: eight 32-weight sub-blocks in the IQ4_NL nibble layout, 6-bit scales (ls - 32);
// lane = sub-blocks 2g, 2g + 1 of a super-block.  Q8_K activations; sum = d dy (ls0 s0 + ls1 s1) with exact integers (mul_mat_qX_K_q8_K_T<DequantizerIQ4XS>,
// iqk_gemm_kquants.cpp:292-332,606-627: unsigned codebook + a -128 d sum(y) term there; same value).
template <> struct Unit<T_IQ4_XS> {
    uint4 q0, q1; uint32_t hdr0, hdr1;          // hdr0 = d | scales_h << 16, hdr1 = scales_l[0..3]
    struct Dec { uint32_t v[16]; int ls[2]; float d; };
    __device__ __forceinline__ uint32_t checksum() const { return q0.x ^ q1.x ^ hdr0 ^ hdr1; }
    __device__ __forceinline__ void zero() { q0 = q1 = make_uint4(0, 0, 0, 0); hdr0 = hdr1 = 0; }
    __device__ __forceinline__ void load(const uint8_t *row, int u) {
        const uint8_t *b = row + (long)(u >> 2) * 136; const int g = u & 3;
        const uint2 h = ld64(b); hdr0 = h.x; hdr1 = h.y; q0 = ld128(b + 8 + 32 * g); q1 = ld128(b + 24 + 32 * g);
    }
    template <int VDT>
    static __device__ __forceinline__ void load_y(int u, int K, int c, const int8_t *yq, const float *yd, const float *, YReg &y) {
        ld_y64(yq + (long)c * K + 64 * u, y); y.s[0] = yd[c * (K >> 8) + (u >> 2)];
    }
    __device__ __forceinline__ void decode(int u, const void *, Dec &dc) const {
        const int g = u & 3; const uint32_t sh = hdr0 >> 16, sl = (hdr1 >> (8 * g)) & 0xff;
        dc.d = half_bits_to_float(hdr0 & 0xffff);
        dc.ls[0] = (int)((sl & 0xf) | (((sh >> (4 * g)) & 3) << 4)) - 32; dc.ls[1] = (int)((sl >> 4) | (((sh >> (4 * g + 2)) & 3) << 4)) - 32;
        const uint32_t a[4] = {q0.x, q0.y, q0.z, q0.w}, b[4] = {q1.x, q1.y, q1.z, q1.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {     // v[0..3]: elements 0..15 (low nibbles), v[4..7]: 16..31 of sub-block 2g; v[8..15] sub-block 2g + 1
            dc.v[i] = iq4nl_lookup4(a[i] & 0x0f0f0f0fu); dc.v[4 + i] = iq4nl_lookup4((a[i] >> 4) & 0x0f0f0f0fu);
            dc.v[8 + i] = iq4nl_lookup4(b[i] & 0x0f0f0f0fu); dc.v[12 + i] = iq4nl_lookup4((b[i] >> 4) & 0x0f0f0f0fu);
        }
    }
    static __device__ __forceinline__ float dot(const Dec &dc, const YReg &y, float r) {
        int s0 = 0, s1 = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) { s0 = dot4(dc.v[i], y.q[i], s0); s1 = dot4(dc.v[8 + i], y.q[8 + i], s1); }
        return fmaf(dc.d * y.s[0], (float)(dc.ls[0] * s0 + dc.ls[1] * s1), r);
    }
};

// spread the low 4 bits of s into 4 bytes of 0x00 / 0xff
__device__ __forceinline__ uint32_t sign_mask4(uint32_t s4) { return (((s4 & 0xfu) * 0x00204081u) & 0x01010101u) * 0xffu; }
// negate the bytes of m selected by mask (0x00/0xff per byte); magnitudes < 128 so no inter-byte carry
__device__ __forceinline__ uint32_t apply_sign4(uint32_t m, uint32_t mask) { return (m ^ mask) + (mask & 0x01010101u); }

// ---- LDS tables of the codebook types (IQ2_S / IQ3_S) --------------------------------------------------------------------------------
// Round 1 kept a 256-entry x 16 B sign table and the plain codebooks in LDS and gathered from them with random addresses: 16-24 gathers per
// 64 weights at 5.6 bank-conflict cycles each (profiles/r01_pmc_gemv_sq.json) made these kernels LDS-bound at 0.16-0.22 of HBM peak.
// Now every table that is small enough is REPLICATED ONCE PER LDS BANK so that lane l only ever reads bank(s) l mod 32 -- conflict-free
// by construction, whatever the indices are:
//   sign LUT   16 entries (one per sign NIBBLE = 4 weights) x 32 lanes x 8 B {byte mask, mask & 0x01010101}                  4 KiB
//              (ds_read_b64: 64 banks, 32-lane groups -> lane l owns banks 2l, 2l+1);  apply = (m ^ mask) + (mask & 1s)
//   IQ3_S grid 512 entries x 32 lanes x 4 B  (ds_read_b32: 32 banks, 32-lane groups -> lane l owns bank l)                    64 KiB
//   IQ2_S grid 1024 x 8 B, NOT replicated (256 KiB would be needed): ds_read_b64 with random 8-byte slots, ~3.4 cycles per 32-lane
//              group instead of 1 -- 8 such gathers per 64 weights remain
// Layout of the table region (its start is 4096-aligned so that the sign-LUT address is an OR, not an add): [sign LUT 4096][grid].
constexpr int IQ_SIGN_LUT_BYTES = 16 * 32 * 8;
// IQ2_S (round 3): the sign table is indexed by the whole sign BYTE -- entry (byte, slot = lane % 8) = {mask, carry} of its low and of its high nibble, 16 bytes, one ds_read_b128
// per 8 weights instead of two ds_read_b64 with an address each (256 x 8 x 16 B = 32 KiB; eight consecutive lanes read eight different 16-byte slots of a 128-byte line)
constexpr int IQ2S_SIGN_LUT_BYTES = 256 * 8 * 16;
__host__ __device__ constexpr int iq_sign_lut_bytes(int t) { return t == T_IQ2_S ? IQ2S_SIGN_LUT_BYTES : IQ_SIGN_LUT_BYTES; }
constexpr int IQ2S_GRID_LDS = 1024 * 8, IQ3S_GRID_LDS = 512 * 32 * 4;
constexpr int IQ1_LDS_BYTES = 2 * 2048 * 8;           // IQ1_S / IQ1_M: the two signed images, no sign LUT
// entries of a type's codebook: IQ2_S 1024, IQ2_XS 512, IQ2_XXS 256 (8-byte entries, not replicated); IQ3_S 512, IQ3_XXS 256 (4-byte entries, one copy per bank)
__host__ __device__ constexpr int iq_grid_entries(int t) { return t == T_IQ2_S ? 1024 : (t == T_IQ2_XS || t == T_IQ3_S) ? 512 : (t == T_IQ2_XXS || t == T_IQ3_XXS) ? 256 : 0; }
__host__ __device__ constexpr int iq_lds_bytes(int base_type) {
    return type_is_iq1(base_type) ? IQ1_LDS_BYTES : type_is_iq8(base_type) ? iq_sign_lut_bytes(base_type) + iq_grid_entries(base_type) * 8 : type_is_iq4(base_type) ? IQ_SIGN_LUT_BYTES + iq_grid_entries(base_type) * 32 * 4 : 0;
}
// global (per context) source image, expanded once from the packed codebooks (iq_tables_init_kernel): [IQ2_S 8192 B][IQ3_S 2048][IQ2_XXS 2048][IQ2_XS 4096][IQ3_XXS 1024]
constexpr int IQ_TABLES_BYTES = 8192 + 2048 + 2048 + 4096 + 1024 + IQ1_LDS_BYTES;      // (+ the IQ1 images behind them)
constexpr int IQ_TABLES_IQ3S_OFFSET = 8192;
__host__ __device__ constexpr int iq_tables_offset(int t) { return t == T_IQ3_S ? 8192 : t == T_IQ2_XXS ? 10240 : t == T_IQ2_XS ? 12288 : t == T_IQ3_XXS ? 16384 : type_is_iq1(t) ? 17408 : 0; }

typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));      // (HIP's uint2 is a struct: no address-space-qualified copy)
typedef __attribute__((address_space(3))) const u32x2_t lds_cu2_t;
typedef __attribute__((address_space(3))) const uint32_t lds_cu32_t;
__device__ __forceinline__ uint2 lds_ld64(uint32_t byte_off) { const u32x2_t v = *reinterpret_cast<lds_cu2_t *>((uintptr_t)byte_off); return make_uint2(v[0], v[1]); }
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const u32x4_t lds_cu4_t;
__device__ __forceinline__ uint4 lds_ld128(uint32_t byte_off) { const u32x4_t v = *reinterpret_cast<lds_cu4_t *>((uintptr_t)byte_off); return make_uint4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ uint32_t lds_ld32(uint32_t byte_off) { return *reinterpret_cast<lds_cu32_t *>((uintptr_t)byte_off); }
// LDS byte offset of a generic pointer into dynamic LDS (the low 32 bits of a flat LDS address are the LDS offset)
__device__ __forceinline__ uint32_t lds_offset_of(const void *p) { return (uint32_t)(uintptr_t)p; }

// what a thread pre-loads (unconditionally, BEFORE the weight ring -- see the note on vmcnt counting at the ring) for the tables:
// 8-byte codebooks: 2 x 16 B of the image per thread of a >= 256-thread workgroup; 4-byte codebooks: 2 entries per thread
template <int TYPE, bool IS8 = type_is_iq8(TYPE), bool IS4 = type_is_iq4(TYPE), bool IS1 = type_is_iq1(TYPE)> struct IqPre {};
template <int TYPE> struct IqPre<TYPE, true, false, false> { qreg_t v[2]; };
template <int TYPE> struct IqPre<TYPE, false, true, false> { uint32_t v[2]; };
template <int TYPE> struct IqPre<TYPE, false, false, true> { qreg_t v[4]; };       // the 8 g + 1 image: 16 KiB / 256 threads (the 8 g - 1 image is derived while filling)
template <int TYPE>
__device__ __forceinline__ void iq_preload(const uint8_t *tables, IqPre<TYPE> &pre) {
    if constexpr (type_is_iq1(TYPE)) {
#pragma unroll
        for (int p = 0; p < 4; ++p) pre.v[p] = reinterpret_cast<const qreg_t *>(tables)[(threadIdx.x + p * blockDim.x) & (IQ1_LDS_BYTES / 32 - 1)];
    } else if constexpr (type_is_iq8(TYPE)) {
        constexpr int NP = iq_grid_entries(TYPE) / 2;          // 16-byte pieces
#pragma unroll
        for (int p = 0; p < 2; ++p) pre.v[p] = reinterpret_cast<const qreg_t *>(tables)[min((int)(threadIdx.x + p * blockDim.x), NP - 1)];
    } else if constexpr (type_is_iq4(TYPE)) {
        constexpr int NE = iq_grid_entries(TYPE);
#pragma unroll
        for (int p = 0; p < 2; ++p) pre.v[p] = reinterpret_cast<const uint32_t *>(tables)[min((int)(threadIdx.x + p * blockDim.x), NE - 1)];
    }
}
// write the table region (workgroups of >= 256 threads; the host guarantees it)
template <int TYPE>
__device__ __forceinline__ void iq_fill_lds(const IqPre<TYPE> &pre, uint8_t *region) {
    if constexpr (type_is_iq1(TYPE)) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int i = threadIdx.x + p * blockDim.x;
            if (i < IQ1_LDS_BYTES / 32) {
                qreg_t m;                    // bytes {-7, 1, 9} - 2 without a borrow between bytes: flip the sign bits around the subtraction
#pragma unroll
                for (int j = 0; j < 4; ++j) m[j] = ((pre.v[p][j] ^ 0x80808080u) - 0x02020202u) ^ 0x80808080u;
                reinterpret_cast<qreg_t *>(region)[i] = pre.v[p]; reinterpret_cast<qreg_t *>(region)[IQ1_LDS_BYTES / 32 + i] = m;
            }
        }
    } else if constexpr (type_has_tables(TYPE)) {
        if constexpr (TYPE == T_IQ2_S) {
            for (int i = threadIdx.x; i < 2048; i += blockDim.x) {          // sign LUT: entry (sign byte, slot = lane % 8)
                const uint32_t b = (uint32_t)i >> 3, lo = sign_mask4(b & 15u), hi = sign_mask4(b >> 4);
                reinterpret_cast<uint4 *>(region)[i] = make_uint4(lo, lo & 0x01010101u, hi, hi & 0x01010101u);
            }
        } else
        for (int i = threadIdx.x; i < 512; i += blockDim.x) {               // sign LUT: entry (nibble, lane slot)
            const uint32_t m = sign_mask4((uint32_t)i >> 5);
            reinterpret_cast<uint2 *>(region)[i] = make_uint2(m, m & 0x01010101u);
        }
        uint8_t *grid = region + iq_sign_lut_bytes(TYPE);
        constexpr int N = type_is_iq8(TYPE) ? iq_grid_entries(TYPE) / 2 : iq_grid_entries(TYPE);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int i = threadIdx.x + p * blockDim.x;
            if (i < N) {
                if constexpr (type_is_iq8(TYPE)) reinterpret_cast<qreg_t *>(grid)[i] = pre.v[p];
                else {                                                      // entry i -> all 32 banks
                    qreg_t r; r[0] = r[1] = r[2] = r[3] = pre.v[p];
#pragma unroll
                    for (int j = 0; j < 8; ++j) reinterpret_cast<qreg_t *>(grid + 128 * i)[j] = r;
                }
            }
        }
    }
}

// ---- IQ2_S : lane = (super-block, g) = 32-blocks 2g, 2g+1; codebook entry = 8 magnitudes (ds_read_b64)
template <> struct Unit<T_IQ2_S> {
    uint2 qs, sg; uint32_t qh, sc, dh;
    struct Dec { uint32_t v[16]; int ls[4]; float d; };
    __device__ __forceinline__ uint32_t checksum() const { return qs.x ^ sg.x ^ qh ^ sc ^ dh; }
    __device__ __forceinline__ void zero() { qs = sg = make_uint2(0, 0); qh = sc = dh = 0; }
    __device__ __forceinline__ void load(const uint8_t *row, int u) {
        const uint8_t *b = row + (long)(u >> 2) * 82; const int g = u & 3;
        dh = ld16(b); qs = ld64(b + 2 + 8 * g); sg = ld64(b + 34 + 8 * g); qh = ld16(b + 66 + 2 * g); sc = ld16(b + 74 + 2 * g);
    }
    template <int VDT>
    static __device__ __forceinline__ void load_y(int u, int K, int c, const int8_t *yq, const float *yd, const float *, YReg &y) {
        ld_y64(yq + (long)c * K + 64 * u, y); y.s[0] = yd[c * (K >> 8) + (u >> 2)];
    }
    // Two phases with a scheduling barrier between them: ALL 16 gathers of the unit (8 grid entries, 8 sign bytes; round 2: 16 sign nibbles) are issued back to
    // back, then consumed.  Left to itself hipcc issued 2-4 ds_reads, waited (lgkmcnt), used them, issued the next few: ~20 exposed LDS round
    // trips per 64 weights, which -- not the bank conflicts -- is what held these kernels at ~2100 cycles per step (r02 notes).
    __device__ __forceinline__ void decode(int, const void *tables, Dec &dc) const {
        const uint32_t tb = lds_offset_of(tables), sgb = tb + ((threadIdx.x & 7u) << 4), g2 = tb + IQ2S_SIGN_LUT_BYTES;
        dc.d = 0.125f * half_bits_to_float(dh);
        const uint32_t qsw[2] = {qs.x, qs.y}, sgw[2] = {sg.x, sg.y};
        uint2 m[8]; uint4 sv[8];
#pragma unroll
        for (int ib = 0; ib < 2; ++ib) {
            const uint32_t h = (qh >> (8 * ib)) & 0xff;
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                const uint32_t idx = ((qsw[ib] >> (8 * l)) & 0xff) | ((h << (8 - 2 * l)) & 0x300);
                m[4 * ib + l] = lds_ld64(g2 + 8 * idx);
            }
        }
#pragma unroll
        for (int ib = 0; ib < 2; ++ib)
#pragma unroll
            for (int l = 0; l < 4; ++l) sv[4 * ib + l] = lds_ld128(sgb + (((sgw[ib] >> (8 * l)) & 0xffu) << 7));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 8; ++i) { dc.v[2 * i] = (m[i].x ^ sv[i].x) + sv[i].y; dc.v[2 * i + 1] = (m[i].y ^ sv[i].z) + sv[i].w; }
#pragma unroll
        for (int j = 0; j < 4; ++j) dc.ls[j] = 2 * (int)((sc >> (4 * j)) & 0xf) + 1;
    }
    static __device__ __forceinline__ float dot(const Dec &dc, const YReg &y, float r) {
        int tot = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) { int s = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) s = dot4(dc.v[4 * j + i], y.q[4 * j + i], s);
            tot += dc.ls[j] * s; }
        return fmaf(dc.d * y.s[0], (float)tot, r);
    }
};

// ---- IQ3_S : lane = (super-block, g) = 32-blocks 2g, 2g+1; codebook entry = 4 magnitudes (ds_read_b32)
template <> struct Unit<T_IQ3_S> {
    uint4 qs; uint2 sg; uint32_t qh, sc, dh;
    struct Dec { uint32_t v[16]; int ls[2]; float d; };
    __device__ __forceinline__ uint32_t checksum() const { return qs.x ^ sg.x ^ qh ^ sc ^ dh; }
    __device__ __forceinline__ void zero() { qs = make_uint4(0, 0, 0, 0); sg = make_uint2(0, 0); qh = sc = dh = 0; }
    __device__ __forceinline__ void load(const uint8_t *row, int u) {
        const uint8_t *b = row + (long)(u >> 2) * 110; const int g = u & 3;
        dh = ld16(b); qs = ld128(b + 2 + 16 * g); qh = ld16(b + 66 + 2 * g); sg = ld64(b + 74 + 8 * g); sc = b[106 + g];
    }
    template <int VDT>
    static __device__ __forceinline__ void load_y(int u, int K, int c, const int8_t *yq, const float *yd, const float *, YReg &y) {
        ld_y64(yq + (long)c * K + 64 * u, y); y.s[0] = yd[c * (K >> 8) + (u >> 2)];
    }
    __device__ __forceinline__ void decode(int, const void *tables, Dec &dc) const {      // (two phases: see Unit<T_IQ2_S>::decode)
        const uint32_t tb = lds_offset_of(tables), sgl = tb | ((threadIdx.x & 31u) << 3), g3 = tb + IQ_SIGN_LUT_BYTES + ((threadIdx.x & 31u) << 2);
        dc.d = half_bits_to_float(dh);
        const uint32_t qsw[4] = {qs.x, qs.y, qs.z, qs.w}, sgw[2] = {sg.x, sg.y};
        uint32_t m[16]; uint2 sv[16];
#pragma unroll
        for (int ib = 0; ib < 2; ++ib) {
            const uint32_t h = (qh >> (8 * ib)) & 0xff;
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                const uint32_t pair = (qsw[2 * ib + (l >> 1)] >> (16 * (l & 1))) & 0xffff;    // qs[2l], qs[2l+1]
                const uint32_t i1 = (pair & 0xff) | ((h << (8 - 2 * l)) & 256), i2 = (pair >> 8) | ((h << (7 - 2 * l)) & 256);
                m[8 * ib + 2 * l] = lds_ld32(g3 + 128 * i1); m[8 * ib + 2 * l + 1] = lds_ld32(g3 + 128 * i2);
            }
        }
#pragma unroll
        for (int ib = 0; ib < 2; ++ib)
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                const uint32_t sb = (sgw[ib] >> (8 * l)) & 0xff;
                sv[8 * ib + 2 * l] = lds_ld64(sgl | ((sb & 15u) << 8)); sv[8 * ib + 2 * l + 1] = lds_ld64(sgl | ((sb >> 4) << 8));
            }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 16; ++i) dc.v[i] = (m[i] ^ sv[i].x) + sv[i].y;
        dc.ls[0] = 2 * (int)(sc & 0xf) + 1; dc.ls[1] = 2 * (int)((sc >> 4) & 0xf) + 1;
    }
    static __device__ __forceinline__ float dot(const Dec &dc, const YReg &y, float r) {
        int s0 = 0, s1 = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) { s0 = dot4(dc.v[i], y.q[i], s0); s1 = dot4(dc.v[8 + i], y.q[8 + i], s1); }
        return fmaf(dc.d * y.s[0], (float)(dc.ls[0] * s0 + dc.ls[1] * s1), r);
    }
};

// ---- IQ1_S : {f16 d; u8 qs[32]; u16 qh[8]}: per 32-block four 11-bit codebook indices (low byte in qs, 3 high bits in qh), a 3-bit scale and the delta bit
// in qh.  lane = 32-blocks 2g, 2g+1.  Integers 8 g +- 1 straight from the LDS image the delta bit selects; (2 s + 1) per 32, d / 8
// (mul_mat_iq1_s_q8_K, iqk_gemm_1bit.cpp:792-865: 8 (g + 1) x q8 plus the block sums x (2 s + 1)(-7 | -9) -- the same integers).
template <> struct Unit<T_IQ1_S> {
    uint2 qs; uint32_t qh, dh;
    typedef Unit<T_IQ3_S>::Dec Dec;
    __device__ __forceinline__ uint32_t checksum() const { return qs.x ^ qs.y ^ qh ^ dh; }
    __device__ __forceinline__ void zero() { qs = make_uint2(0, 0); qh = dh = 0; }
    __device__ __forceinline__ void load(const uint8_t *row, int u) {
        const uint8_t *b = row + (long)(u >> 2) * 50; const int g = u & 3;
        dh = ld16(b); qs = ld64(b + 2 + 8 * g); qh = ld32(b + 34 + 4 * g);
    }
    template <int VDT>
    static __device__ __forceinline__ void load_y(int u, int K, int c, const int8_t *yq, const float *yd, const float *, YReg &y) {
        ld_y64(yq + (long)c * K + 64 * u, y); y.s[0] = yd[c * (K >> 8) + (u >> 2)];
    }
    __device__ __forceinline__ void decode(int, const void *tables, Dec &dc) const {
        const uint32_t tb = lds_offset_of(tables);
        dc.d = 0.125f * half_bits_to_float(dh);
        const uint32_t qsw[2] = {qs.x, qs.y};
#pragma unroll
        for (int ib = 0; ib < 2; ++ib) {
            const uint32_t h = (qh >> (16 * ib)) & 0xffff, img = tb + ((h >> 15) << 14);
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                const uint2 m = lds_ld64(img + 8 * (((qsw[ib] >> (8 * l)) & 0xff) | (((h >> (3 * l)) & 7) << 8)));
                dc.v[8 * ib + 2 * l] = m.x; dc.v[8 * ib + 2 * l + 1] = m.y;
            }
            dc.ls[ib] = 2 * (int)((h >> 12) & 7) + 1;
        }
    }
    static __device__ __forceinline__ float dot(const Dec &dc, const YReg &y, float r) { return Unit<T_IQ3_S>::dot(dc, y, r); }
};

// ---- IQ1_M : {u8 qs[32]; u8 qh[16]; u8 scales[8]}: per 8 weights an 11-bit index (3 high bits + the delta bit in a qh nibble), 3-bit scales per 16 in the
// four scale words whose top nibbles form the f16 d.  lane = 32-blocks 2g, 2g+1 (mul_mat_iq1_m_q8_K, iqk_gemm_1bit.cpp:867-927: 8 (g + 1) - 7 | 9, (2 s + 1), d / 8)
template <> struct Unit<T_IQ1_M> {
    uint2 qs, sc8; uint32_t qh;
    typedef Unit<T_IQ2_S>::Dec Dec;
    __device__ __forceinline__ uint32_t checksum() const { return qs.x ^ qs.y ^ qh ^ sc8.x ^ sc8.y; }
    __device__ __forceinline__ void zero() { qs = sc8 = make_uint2(0, 0); qh = 0; }
    __device__ __forceinline__ void load(const uint8_t *row, int u) {
        const uint8_t *b = row + (long)(u >> 2) * 56; const int g = u & 3;
        qs = ld64(b + 8 * g); qh = ld32(b + 32 + 4 * g); sc8 = ld64(b + 48);
    }
    template <int VDT>
    static __device__ __forceinline__ void load_y(int u, int K, int c, const int8_t *yq, const float *yd, const float *, YReg &y) {
        ld_y64(yq + (long)c * K + 64 * u, y); y.s[0] = yd[c * (K >> 8) + (u >> 2)];
    }
    __device__ __forceinline__ void decode(int u, const void *tables, Dec &dc) const {
        const uint32_t tb = lds_offset_of(tables);
        dc.d = 0.125f * half_bits_to_float(((sc8.x >> 12) & 0xf) | ((sc8.x >> 24) & 0x00f0) | ((sc8.y >> 4) & 0x0f00) | ((sc8.y >> 16) & 0xf000));
        const uint32_t sw = ((u & 2) ? sc8.y : sc8.x) >> (16 * (u & 1));
        const uint32_t qsw[2] = {qs.x, qs.y};
#pragma unroll
        for (int ib = 0; ib < 2; ++ib)
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                const uint32_t h = qh >> (16 * ib + 4 * l);          // nibble l of the block's two qh bytes: 3 index bits + the delta bit
                const uint2 m = lds_ld64(tb + ((h & 8) << 11) + 8 * (((qsw[ib] >> (8 * l)) & 0xff) | ((h & 7) << 8)));
                dc.v[8 * ib + 2 * l] = m.x; dc.v[8 * ib + 2 * l + 1] = m.y;
            }
#pragma unroll
        for (int j = 0; j < 4; ++j) dc.ls[j] = 2 * (int)((sw >> (3 * j)) & 7) + 1;
    }
    static __device__ __forceinline__ float dot(const Dec &dc, const YReg &y, float r) { return Unit<T_IQ2_S>::dot(dc, y, r); }
};

// ---- IQ2_XXS : per 32-block two dwords {4 x 8-bit grid index | 4 x 7-bit sign index, 4-bit scale}; lane = 32-blocks 2g, 2g+1 (16 contiguous bytes).
// 8-byte codebook entries as IQ2_S; the sign byte of a 7-bit index is index | parity << 7 (ksign7).  d (2 s + 1) / 8 per 32 (DequantizerIQ2XXS, iqk_gemm_iquants.cpp:148-234)
template <> struct Unit<T_IQ2_XXS> {
    uint4 q; uint32_t dh;
    typedef Unit<T_IQ3_S>::Dec Dec;
    __device__ __forceinline__ uint32_t checksum() const { return q.x ^ q.y ^ dh; }
    __device__ __forceinline__ void zero() { q = make_uint4(0, 0, 0, 0); dh = 0; }
    __device__ __forceinline__ void load(const uint8_t *row, int u) { const uint8_t *b = row + (long)(u >> 2) * 66; dh = ld16(b); q = ld128(b + 2 + 16 * (u & 3)); }
    template <int VDT>
    static __device__ __forceinline__ void load_y(int u, int K, int c, const int8_t *yq, const float *yd, const float *ys, YReg &y) { Unit<T_IQ3_S>::template load_y<VDT>(u, K, c, yq, yd, ys, y); }
    __device__ __forceinline__ void decode(int, const void *tables, Dec &dc) const {
        const uint32_t tb = lds_offset_of(tables), sgl = tb | ((threadIdx.x & 31u) << 3), g2 = tb + IQ_SIGN_LUT_BYTES;
        dc.d = 0.125f * half_bits_to_float(dh);
        const uint32_t a0[2] = {q.x, q.z}, a1[2] = {q.y, q.w};
        uint2 m[8], slo[8], shi[8];
#pragma unroll
        for (int ib = 0; ib < 2; ++ib)
#pragma unroll
            for (int l = 0; l < 4; ++l) m[4 * ib + l] = lds_ld64(g2 + 8 * ((a0[ib] >> (8 * l)) & 0xff));
#pragma unroll
        for (int ib = 0; ib < 2; ++ib)
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                const uint32_t sb = ksign7((a1[ib] >> (7 * l)) & 127);
                slo[4 * ib + l] = lds_ld64(sgl | ((sb & 15u) << 8)); shi[4 * ib + l] = lds_ld64(sgl | ((sb >> 4) << 8));
            }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 8; ++i) { dc.v[2 * i] = (m[i].x ^ slo[i].x) + slo[i].y; dc.v[2 * i + 1] = (m[i].y ^ shi[i].x) + shi[i].y; }
        dc.ls[0] = 2 * (int)(a1[0] >> 28) + 1; dc.ls[1] = 2 * (int)(a1[1] >> 28) + 1;
    }
    static __device__ __forceinline__ float dot(const Dec &dc, const YReg &y, float r) { return Unit<T_IQ3_S>::dot(dc, y, r); }
};

// ---- IQ2_XS : u16 {9-bit grid index | 7-bit sign index << 9} per 8 weights, 4-bit scales per 16 (DequantizerIQ2XS, iqk_gemm_iquants.cpp:236-380); lane = 32-blocks 2g, 2g+1
template <> struct Unit<T_IQ2_XS> {
    uint4 q; uint32_t sc, dh;
    typedef Unit<T_IQ2_S>::Dec Dec;
    __device__ __forceinline__ uint32_t checksum() const { return q.x ^ q.y ^ sc ^ dh; }
    __device__ __forceinline__ void zero() { q = make_uint4(0, 0, 0, 0); sc = dh = 0; }
    __device__ __forceinline__ void load(const uint8_t *row, int u) { const uint8_t *b = row + (long)(u >> 2) * 74; const int g = u & 3; dh = ld16(b); q = ld128(b + 2 + 16 * g); sc = ld16(b + 66 + 2 * g); }
    template <int VDT>
    static __device__ __forceinline__ void load_y(int u, int K, int c, const int8_t *yq, const float *yd, const float *ys, YReg &y) { Unit<T_IQ2_S>::template load_y<VDT>(u, K, c, yq, yd, ys, y); }
    __device__ __forceinline__ void decode(int, const void *tables, Dec &dc) const {
        const uint32_t tb = lds_offset_of(tables), sgl = tb | ((threadIdx.x & 31u) << 3), g2 = tb + IQ_SIGN_LUT_BYTES;
        dc.d = 0.125f * half_bits_to_float(dh);
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
        uint2 m[8], slo[8], shi[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) m[i] = lds_ld64(g2 + 8 * ((w[i >> 1] >> (16 * (i & 1))) & 511));
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t sb = ksign7((w[i >> 1] >> (16 * (i & 1) + 9)) & 127);
            slo[i] = lds_ld64(sgl | ((sb & 15u) << 8)); shi[i] = lds_ld64(sgl | ((sb >> 4) << 8));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 8; ++i) { dc.v[2 * i] = (m[i].x ^ slo[i].x) + slo[i].y; dc.v[2 * i + 1] = (m[i].y ^ shi[i].x) + shi[i].y; }
#pragma unroll
        for (int j = 0; j < 4; ++j) dc.ls[j] = 2 * (int)((sc >> (4 * j)) & 0xf) + 1;
    }
    static __device__ __forceinline__ float dot(const Dec &dc, const YReg &y, float r) { return Unit<T_IQ2_S>::dot(dc, y, r); }
};

// ---- IQ3_XXS : qs[64] 8-bit grid indices (4 magnitudes each), then per 32-block a dword {4 x 7-bit sign index, 4-bit scale}; d (2 s + 1) / 4 per 32
// (DequantizerIQ3XXS, iqk_gemm_iquants.cpp:494-581); lane = 32-blocks 2g, 2g+1
template <> struct Unit<T_IQ3_XXS> {
    uint4 q; uint2 sa; uint32_t dh;
    typedef Unit<T_IQ3_S>::Dec Dec;
    __device__ __forceinline__ uint32_t checksum() const { return q.x ^ sa.x ^ dh; }
    __device__ __forceinline__ void zero() { q = make_uint4(0, 0, 0, 0); sa = make_uint2(0, 0); dh = 0; }
    __device__ __forceinline__ void load(const uint8_t *row, int u) { const uint8_t *b = row + (long)(u >> 2) * 98; const int g = u & 3; dh = ld16(b); q = ld128(b + 2 + 16 * g); sa = ld64(b + 66 + 8 * g); }
    template <int VDT>
    static __device__ __forceinline__ void load_y(int u, int K, int c, const int8_t *yq, const float *yd, const float *ys, YReg &y) { Unit<T_IQ3_S>::template load_y<VDT>(u, K, c, yq, yd, ys, y); }
    __device__ __forceinline__ void decode(int, const void *tables, Dec &dc) const {
        const uint32_t tb = lds_offset_of(tables), sgl = tb | ((threadIdx.x & 31u) << 3), g3 = tb + IQ_SIGN_LUT_BYTES + ((threadIdx.x & 31u) << 2);
        dc.d = 0.25f * half_bits_to_float(dh);
        const uint32_t w[4] = {q.x, q.y, q.z, q.w}, a[2] = {sa.x, sa.y};
        uint32_t m[16]; uint2 sv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) m[i] = lds_ld32(g3 + 128 * ((w[i >> 2] >> (8 * (i & 3))) & 0xff));
#pragma unroll
        for (int ib = 0; ib < 2; ++ib)
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                const uint32_t sb = ksign7((a[ib] >> (7 * l)) & 127);
                sv[8 * ib + 2 * l] = lds_ld64(sgl | ((sb & 15u) << 8)); sv[8 * ib + 2 * l + 1] = lds_ld64(sgl | ((sb >> 4) << 8));
            }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 16; ++i) dc.v[i] = (m[i] ^ sv[i].x) + sv[i].y;
        dc.ls[0] = 2 * (int)(a[0] >> 28) + 1; dc.ls[1] = 2 * (int)(a[1] >> 28) + 1;
    }
    static __device__ __forceinline__ float dot(const Dec &dc, const YReg &y, float r) { return Unit<T_IQ3_S>::dot(dc, y, r); }
};

// ---- Q5_0 : lane = two consecutive 22-byte blocks {f16 d; u32 qh; u8 qs[16]}; value = (nibble | bit << 4) - 16  (Q5_0_1_Unpacker, iqk_gemm_legacy_quants.cpp: unsigned 5-bit
// quants + a -16 d sum(y) term there; same value).  Q8_2_X4 activations.
template <> struct Unit<T_Q5_0> {
    uint32_t w[11];
    typedef Unit<T_Q8_0>::Dec Dec;
    __device__ __forceinline__ uint32_t checksum() const { return w[0] ^ w[10]; }
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int i = 0; i < 11; ++i) w[i] = 0;
    }
    __device__ __forceinline__ void load(const uint8_t *row, int u) {
        const uint32_t *p = reinterpret_cast<const uint32_t *>(row + (long)u * 44);
#pragma unroll
        for (int i = 0; i < 11; ++i) w[i] = p[i];
    }
    template <int VDT>
    static __device__ __forceinline__ void load_y(int u, int K, int c, const int8_t *yq, const float *yd, const float *ys, YReg &y) { Unit<T_IQ4_NL>::template load_y<VDT>(u, K, c, yq, yd, ys, y); }
    // 4 nibbles (already masked to 0x0f per byte) + 4 high bits (low 4 bits of hb) -> 4 signed bytes q - 16
    static __device__ __forceinline__ uint32_t q5_bytes(uint32_t nib, uint32_t hb) {
        const uint32_t x = nib | ((((hb & 0xfu) * 0x00204081u) & 0x01010101u) << 4);
        return ((x | 0x80808080u) - 0x10101010u) ^ 0x80808080u;
    }
    __device__ __forceinline__ void decode(int, const void *, Dec &dc) const {
        dc.d0 = half_bits_to_float(w[0] & 0xffff); dc.d1 = half_bits_to_float(w[5] >> 16);
        const uint32_t qh0 = __builtin_amdgcn_alignbyte(w[1], w[0], 2), qh1 = w[6];
#pragma unroll
        for (int i = 0; i < 4; ++i) {     // v[0..3]: elements 0..15 (low nibbles, bits j), v[4..7]: 16..31 (high nibbles, bits 16 + j)
            const uint32_t a = __builtin_amdgcn_alignbyte(w[i + 2], w[i + 1], 2), b = w[7 + i];
            dc.v[i] = q5_bytes(a & 0x0f0f0f0fu, qh0 >> (4 * i)); dc.v[4 + i] = q5_bytes((a >> 4) & 0x0f0f0f0fu, qh0 >> (16 + 4 * i));
            dc.v[8 + i] = q5_bytes(b & 0x0f0f0f0fu, qh1 >> (4 * i)); dc.v[12 + i] = q5_bytes((b >> 4) & 0x0f0f0f0fu, qh1 >> (16 + 4 * i));
        }
    }
    static __device__ __forceinline__ float dot(const Dec &dc, const YReg &y, float r) { return Unit<T_Q8_0>::dot(dc, y, r); }
};

// ---- Q4_1 / Q5_1 : lane = two consecutive blocks {f16 d, m; [u32 qh;] u8 qs[16]} (40 / 48 B); value = q d + m with UNSIGNED q (Q4_1_Unpacker / Q5_1_Unpacker,
// iqk_gemm_legacy_quants.cpp): d d_y (q . y) + m (d_y sum(y)) with the activation block's stored sum -- Unit<Q4_K>'s accumulate with the min term's sign flipped
template <int T51> struct UnitQX1 {
    static constexpr int NW = T51 ? 12 : 10, QS0 = T51 ? 2 : 1;       // dwords per lane; first qs dword of a block
    uint32_t w[NW];
    typedef Unit<T_Q4_K>::Dec Dec;
    __device__ __forceinline__ uint32_t checksum() const { return w[0] ^ w[NW - 1]; }
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int i = 0; i < NW; ++i) w[i] = 0;
    }
    __device__ __forceinline__ void load(const uint8_t *row, int u) {
        const uint32_t *p = reinterpret_cast<const uint32_t *>(row + (long)u * (4 * NW));
#pragma unroll
        for (int i = 0; i < NW; ++i) w[i] = p[i];
    }
    template <int VDT>
    static __device__ __forceinline__ void load_y(int u, int K, int c, const int8_t *yq, const float *yd, const float *ys, YReg &y) { Unit<T_Q4_K>::template load_y<VDT>(u, K, c, yq, yd, ys, y); }
    __device__ __forceinline__ void decode(int, const void *, Dec &dc) const {
        dc.d_lo = half_bits_to_float(w[0] & 0xffff); dc.m_lo = -half_bits_to_float(w[0] >> 16);
        dc.d_hi = half_bits_to_float(w[NW / 2] & 0xffff); dc.m_hi = -half_bits_to_float(w[NW / 2] >> 16);
        const uint32_t qh0 = T51 ? w[1] : 0u, qh1 = T51 ? w[NW / 2 + 1] : 0u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {     // lo[0..3]: elements 0..15 of block 0 (low nibbles), lo[4..7]: 16..31 (high nibbles); hi[]: block 1
            const uint32_t a = w[QS0 + i], b = w[NW / 2 + QS0 + i];
            dc.lo[i] = (a & 0x0f0f0f0fu) | (((((qh0 >> (4 * i)) & 0xfu) * 0x00204081u) & 0x01010101u) << 4);
            dc.lo[4 + i] = ((a >> 4) & 0x0f0f0f0fu) | (((((qh0 >> (16 + 4 * i)) & 0xfu) * 0x00204081u) & 0x01010101u) << 4);
            dc.hi[i] = (b & 0x0f0f0f0fu) | (((((qh1 >> (4 * i)) & 0xfu) * 0x00204081u) & 0x01010101u) << 4);
            dc.hi[4 + i] = ((b >> 4) & 0x0f0f0f0fu) | (((((qh1 >> (16 + 4 * i)) & 0xfu) * 0x00204081u) & 0x01010101u) << 4);
        }
    }
    static __device__ __forceinline__ float dot(const Dec &dc, const YReg &y, float r) { return Unit<T_Q4_K>::dot(dc, y, r); }
};
template <> struct Unit<T_Q4_1> : UnitQX1<0> {};
template <> struct Unit<T_Q5_1> : UnitQX1<1> {};

// ---- Q6_0 : lane = two consecutive 26-byte blocks {f16 d; u8 qh[8]; u8 qs[16]}; value = (nibble | 2 bits << 4) - 32  (Q6_0_1_Unpacker: unsigned + a -32 d sum(y) term; same value)
template <> struct Unit<T_Q6_0> {
    uint32_t w[13];
    typedef Unit<T_Q8_0>::Dec Dec;
    __device__ __forceinline__ uint32_t checksum() const { return w[0] ^ w[12]; }
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int i = 0; i < 13; ++i) w[i] = 0;
    }
    __device__ __forceinline__ void load(const uint8_t *row, int u) {
        const uint32_t *p = reinterpret_cast<const uint32_t *>(row + (long)u * 52);
#pragma unroll
        for (int i = 0; i < 13; ++i) w[i] = p[i];
    }
    template <int VDT>
    static __device__ __forceinline__ void load_y(int u, int K, int c, const int8_t *yq, const float *yd, const float *ys, YReg &y) { Unit<T_IQ4_NL>::template load_y<VDT>(u, K, c, yq, yd, ys, y); }
    static __device__ __forceinline__ uint32_t q6_bytes(uint32_t nib, uint32_t hb2) { const uint32_t x = nib | ((hb2 & 0x03030303u) << 4); return ((x | 0x80808080u) - 0x20202020u) ^ 0x80808080u; }
    __device__ __forceinline__ void decode(int, const void *, Dec &dc) const {
        dc.d0 = half_bits_to_float(w[0] & 0xffff); dc.d1 = half_bits_to_float(w[6] >> 16);
        // block 0 occupies bytes 0..25 (qh at 2, qs at 10), block 1 bytes 26..51 (qh at 28 = dword 7, qs at 36 = dword 9)
        const uint32_t qh0[2] = {__builtin_amdgcn_alignbyte(w[1], w[0], 2), __builtin_amdgcn_alignbyte(w[2], w[1], 2)}, qh1[2] = {w[7], w[8]};
#pragma unroll
        for (int i = 0; i < 4; ++i) {     // element j = 4 i + byte: high bits from qh[j % 8] >> 4 (j / 8) (low nibble elements) / + 2 (high nibble elements)
            const uint32_t a = __builtin_amdgcn_alignbyte(w[3 + i], w[2 + i], 2), b = w[9 + i];
            const uint32_t h0 = qh0[i & 1] >> (4 * (i >> 1)), h1 = qh1[i & 1] >> (4 * (i >> 1));
            dc.v[i] = q6_bytes(a & 0x0f0f0f0fu, h0); dc.v[4 + i] = q6_bytes((a >> 4) & 0x0f0f0f0fu, h0 >> 2);
            dc.v[8 + i] = q6_bytes(b & 0x0f0f0f0fu, h1); dc.v[12 + i] = q6_bytes((b >> 4) & 0x0f0f0f0fu, h1 >> 2);
        }
    }
    static __device__ __forceinline__ float dot(const Dec &dc, const YReg &y, float r) { return Unit<T_Q8_0>::dot(dc, y, r); }
};

// ---- Q2_K : 84-byte super-blocks {u8 scales[16]; u8 qs[64]; f16 d, dmin}; element 128 n + 32 j + l = (qs[32 n + l] >> 2 j) & 3, scale / min nibbles per 16.
// lane = 64 elements (n, j = 2 jj, 2 jj + 1): the 32 qs bytes of half n at shifts 4 jj, 4 jj + 2.  Q8_K activations: d d_y sum_k sc_k (q . y)_k - dmin d_y sum_k m_k sum(y)_k
// in exact integers (set_functions<DequantizerQ2K>, iqk_gemm_kquants.cpp:192-209); the sums of y come from v_dot4 with the min replicated into 4 bytes
template <> struct Unit<T_Q2_K> {
    uint4 q0, q1; uint32_t sc, dd;
    struct Dec { uint32_t v[16]; uint32_t scm; float d, dmin; };
    __device__ __forceinline__ uint32_t checksum() const { return q0.x ^ q1.x ^ sc ^ dd; }
    __device__ __forceinline__ void zero() { q0 = q1 = make_uint4(0, 0, 0, 0); sc = dd = 0; }
    __device__ __forceinline__ void load(const uint8_t *row, int u) {
        const uint8_t *b = row + (long)(u >> 2) * 84; const int n = (u >> 1) & 1, jj = u & 1;
        q0 = ld128(b + 16 + 32 * n); q1 = ld128(b + 32 + 32 * n); sc = ld32(b + 8 * n + 4 * jj); dd = ld32(b + 80);
    }
    template <int VDT>
    static __device__ __forceinline__ void load_y(int u, int K, int c, const int8_t *yq, const float *yd, const float *ys, YReg &y) { Unit<T_IQ2_S>::template load_y<VDT>(u, K, c, yq, yd, ys, y); }
    __device__ __forceinline__ void decode(int u, const void *, Dec &dc) const {
        const int s0 = 4 * (u & 1);
        dc.d = half_bits_to_float(dd & 0xffff); dc.dmin = half_bits_to_float(dd >> 16); dc.scm = sc;
        const uint32_t q[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
        for (int i = 0; i < 8; ++i) { dc.v[i] = (q[i] >> s0) & 0x03030303u; dc.v[8 + i] = (q[i] >> (s0 + 2)) & 0x03030303u; }
    }
    static __device__ __forceinline__ float dot(const Dec &dc, const YReg &y, float r) {
        int tot = 0, mt = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t b = (dc.scm >> (8 * k)) & 0xff; int s = 0;
            const uint32_t m4 = (b >> 4) * 0x01010101u;
#pragma unroll
            for (int i = 0; i < 4; ++i) { s = dot4(dc.v[4 * k + i], y.q[4 * k + i], s); mt = dot4(m4, y.q[4 * k + i], mt); }
            tot += (int)(b & 15) * s;
        }
        r = fmaf(dc.d * y.s[0], (float)tot, r);
        return fmaf(-(dc.dmin * y.s[0]), (float)mt, r);
    }
};

// ---- Q3_K : 110-byte super-blocks {u8 hmask[32]; u8 qs[64]; u8 scales[12]; f16 d}; low 2 bits as Q2_K, - 4 unless bit 4 n + j of hmask[l] is set; sixteen 6-bit scales - 32.
// lane = 64 elements (n, j = 2 jj, 2 jj + 1).  d d_y sum_k (sc_k - 32) (q . y)_k in exact integers (set_functions<DequantizerQ3K>, iqk_gemm_kquants.cpp:210-260)
template <> struct Unit<T_Q3_K> {
    uint4 q0, q1, h0, h1; uint32_t s0w, s1w, s2w, dh;
    struct Dec { uint32_t v[16]; int ls[4]; float d; };
    __device__ __forceinline__ uint32_t checksum() const { return q0.x ^ q1.x ^ h0.x ^ h1.x ^ s0w ^ dh; }
    __device__ __forceinline__ void zero() { q0 = q1 = h0 = h1 = make_uint4(0, 0, 0, 0); s0w = s1w = s2w = dh = 0; }
    __device__ __forceinline__ void load(const uint8_t *row, int u) {
        const uint8_t *b = row + (long)(u >> 2) * 110; const int n = (u >> 1) & 1;
        h0 = ld128(b); h1 = ld128(b + 16); q0 = ld128(b + 32 + 32 * n); q1 = ld128(b + 48 + 32 * n); s0w = ld32(b + 96); s1w = ld32(b + 100); s2w = ld32(b + 104); dh = ld16(b + 108);
    }
    template <int VDT>
    static __device__ __forceinline__ void load_y(int u, int K, int c, const int8_t *yq, const float *yd, const float *ys, YReg &y) { Unit<T_IQ2_S>::template load_y<VDT>(u, K, c, yq, yd, ys, y); }
    static __device__ __forceinline__ uint32_t q3_bytes(uint32_t lo2, uint32_t hb) { const uint32_t x = lo2 | ((hb & 0x01010101u) << 2); return ((x | 0x80808080u) - 0x04040404u) ^ 0x80808080u; }
    __device__ __forceinline__ void decode(int u, const void *, Dec &dc) const {
        const int n = (u >> 1) & 1, jj = u & 1, s0 = 4 * jj, hbit = 4 * n + 2 * jj;
        dc.d = half_bits_to_float(dh);
        const uint32_t q[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w}, hm[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
        for (int i = 0; i < 8; ++i) { dc.v[i] = q3_bytes((q[i] >> s0) & 0x03030303u, hm[i] >> hbit); dc.v[8 + i] = q3_bytes((q[i] >> (s0 + 2)) & 0x03030303u, hm[i] >> (hbit + 1)); }
        // scales is = 8 n + 4 jj + k: low 4 bits from byte is (is < 8: low nibble) or is - 8 (high nibble) of scales[0..7], bits 4..5 from scales[8 + (is & 3)] >> 2 (is >> 2)
        const uint32_t lo = jj ? s1w : s0w, lo4 = (n ? (lo >> 4) : lo) & 0x0f0f0f0fu, hi2 = (s2w >> (2 * (2 * n + jj))) & 0x03030303u, sc6 = lo4 | (hi2 << 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) dc.ls[k] = (int)((sc6 >> (8 * k)) & 0xff) - 32;
    }
    static __device__ __forceinline__ float dot(const Dec &dc, const YReg &y, float r) {
        int tot = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { int s = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) s = dot4(dc.v[4 * k + i], y.q[4 * k + i], s);
            tot += dc.ls[k] * s; }
        return fmaf(dc.d * y.s[0], (float)tot, r);
    }
};

// ---- ik's non-linear K types (iqk_gemm_iqk_quants.cpp: set_functions<DequantizerIQ2K ... IQ5KS>): int8 table values (an `extra` / scale bit adds a constant per 16 or 32
// weights), integer scales, Q8_K activations; sum = d d_y sum_k ls_k (v . y)_k in exact integers.  lane = 64 weights; IQ2_K / IQ3_K share Q2_K's 2-bit packing.
template <> struct Unit<T_IQ2_K> {          // 76 bytes {f16 d; u16 extra; u8 scales[8]; u8 qs[64]}
    uint4 q0, q1; uint32_t hdr, sc;
    typedef Unit<T_Q3_K>::Dec Dec;
    __device__ __forceinline__ uint32_t checksum() const { return q0.x ^ q1.x ^ hdr ^ sc; }
    __device__ __forceinline__ void zero() { q0 = q1 = make_uint4(0, 0, 0, 0); hdr = sc = 0; }
    __device__ __forceinline__ void load(const uint8_t *row, int u) {
        const uint8_t *b = row + (long)(u >> 2) * 76; const int n = (u >> 1) & 1;
        hdr = ld32(b); sc = ld16(b + 4 + 2 * (u & 3)); q0 = ld128(b + 12 + 32 * n); q1 = ld128(b + 28 + 32 * n);
    }
    template <int VDT>
    static __device__ __forceinline__ void load_y(int u, int K, int c, const int8_t *yq, const float *yd, const float *ys, YReg &y) { Unit<T_IQ2_S>::template load_y<VDT>(u, K, c, yq, yd, ys, y); }
    __device__ __forceinline__ void decode(int u, const void *, Dec &dc) const {
        const int s0 = 4 * (u & 1); const uint32_t ex = (hdr >> 16) >> (4 * (u & 3));
        dc.d = half_bits_to_float(hdr & 0xffff);
        const uint32_t q[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t t = k_iq2nl_packed[(ex >> k) & 1]; const int sh = (k & 2) ? s0 + 2 : s0;
#pragma unroll
            for (int i = 0; i < 4; ++i) dc.v[4 * k + i] = __builtin_amdgcn_perm(t, t, (q[4 * (k & 1) + i] >> sh) & 0x03030303u);
            dc.ls[k] = (int)((sc >> (4 * k)) & 15) - 8;
        }
    }
    static __device__ __forceinline__ float dot(const Dec &dc, const YReg &y, float r) { return Unit<T_Q3_K>::dot(dc, y, r); }
};
template <> struct Unit<T_IQ3_K> {          // 110 bytes {f16 d; u16 extra; u16 scales_h; u8 scales_l[8]; u8 qs[64]; u8 qh[32]}
    uint4 q0, q1, h0, h1; uint32_t hdr, sh, sl;
    typedef Unit<T_Q3_K>::Dec Dec;
    __device__ __forceinline__ uint32_t checksum() const { return q0.x ^ q1.x ^ h0.x ^ h1.x ^ hdr ^ sl; }
    __device__ __forceinline__ void zero() { q0 = q1 = h0 = h1 = make_uint4(0, 0, 0, 0); hdr = sh = sl = 0; }
    __device__ __forceinline__ void load(const uint8_t *row, int u) {
        const uint8_t *b = row + (long)(u >> 2) * 110; const int n = (u >> 1) & 1;
        hdr = ld32(b); sh = ld16(b + 4); sl = ld16(b + 6 + 2 * (u & 3)); q0 = ld128(b + 14 + 32 * n); q1 = ld128(b + 30 + 32 * n); h0 = ld128(b + 78); h1 = ld128(b + 94);
    }
    template <int VDT>
    static __device__ __forceinline__ void load_y(int u, int K, int c, const int8_t *yq, const float *yd, const float *ys, YReg &y) { Unit<T_IQ2_S>::template load_y<VDT>(u, K, c, yq, yd, ys, y); }
    __device__ __forceinline__ void decode(int u, const void *, Dec &dc) const {
        const int uu = u & 3, s0 = 4 * (u & 1); const uint32_t ex = (hdr >> 16) >> (4 * uu), sg = sh >> (4 * uu);
        dc.d = half_bits_to_float(hdr & 0xffff);
        const uint32_t q[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w}, hb[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = (ex >> k) & 1; const uint32_t t0 = k_iq3nl_packed[2 * e], t1 = k_iq3nl_packed[2 * e + 1];
            const int shl = (k & 2) ? s0 + 2 : s0, shh = 2 * uu + (k >> 1);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int w = 4 * (k & 1) + i;
                dc.v[4 * k + i] = __builtin_amdgcn_perm(t1, t0, ((q[w] >> shl) & 0x03030303u) | (((hb[w] >> shh) & 0x01010101u) << 2));
            }
            const int m = 2 * (int)((sl >> (4 * k)) & 15) + 1;
            dc.ls[k] = ((sg >> k) & 1) ? -m : m;
        }
    }
    static __device__ __forceinline__ float dot(const Dec &dc, const YReg &y, float r) { return Unit<T_Q3_K>::dot(dc, y, r); }
};
// scale k of a 64-weight group of IQ4_K / IQ5_K: low 4 bits = nibble k of `sl` (two scales_l bytes), bits 4..5 = bits 2k, 2k+1 of the group's scales_h byte; - 32
__device__ __forceinline__ int iqk_scale6(uint32_t sl, uint32_t shb, int k) { return (int)(((sl >> (4 * k)) & 15) | (((shb >> (2 * k)) & 3) << 4)) - 32; }
template <> struct Unit<T_IQ4_K> {          // 144 bytes {f16 d; u16 extra; u8 scales_h[4]; u8 scales_l[8]; u8 qs[128]}: 32-blocks in the IQ4_NL nibble layout
    uint4 q0, q1; uint32_t hdr, shw, sl;
    typedef Unit<T_Q3_K>::Dec Dec;
    __device__ __forceinline__ uint32_t checksum() const { return q0.x ^ q1.x ^ hdr ^ shw ^ sl; }
    __device__ __forceinline__ void zero() { q0 = q1 = make_uint4(0, 0, 0, 0); hdr = shw = sl = 0; }
    __device__ __forceinline__ void load(const uint8_t *row, int u) {
        const uint8_t *b = row + (long)(u >> 2) * 144; const int uu = u & 3;
        hdr = ld32(b); shw = ld32(b + 4); sl = ld16(b + 8 + 2 * uu); q0 = ld128(b + 16 + 32 * uu); q1 = ld128(b + 32 + 32 * uu);
    }
    template <int VDT>
    static __device__ __forceinline__ void load_y(int u, int K, int c, const int8_t *yq, const float *yd, const float *ys, YReg &y) { Unit<T_IQ2_S>::template load_y<VDT>(u, K, c, yq, yd, ys, y); }
    __device__ __forceinline__ void decode(int u, const void *, Dec &dc) const {
        const int uu = u & 3; const uint32_t ex = (hdr >> 16) >> (4 * uu), shb = (shw >> (8 * uu)) & 0xff;
        dc.d = half_bits_to_float(hdr & 0xffff);
        const uint32_t q[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {             // k = 2 p + h: 32-block p of the pair, low (h = 0) / high (h = 1) nibbles of its 16 bytes
            const uint32_t add = ((ex >> k) & 1) ? 0x04040404u : 0u;
#pragma unroll
            for (int i = 0; i < 4; ++i) { const uint32_t w = q[4 * (k >> 1) + i]; dc.v[4 * k + i] = add_bytes(iq4nl_lookup4(((k & 1) ? (w >> 4) : w) & 0x0f0f0f0fu), add); }
            dc.ls[k] = iqk_scale6(sl, shb, k);
        }
    }
    static __device__ __forceinline__ float dot(const Dec &dc, const YReg &y, float r) { return Unit<T_Q3_K>::dot(dc, y, r); }
};
template <> struct Unit<T_IQ5_K> {          // 176 bytes {f16 d; u16 extra; u8 scales_h[4]; u8 scales_l[8]; u8 qs[128]; u8 qh[32]}: per 64: qs[0..15] / qs[16..31] low, then high nibbles
    uint4 q0, q1, h0, h1; uint32_t hdr, shw, sl;
    typedef Unit<T_Q3_K>::Dec Dec;
    __device__ __forceinline__ uint32_t checksum() const { return q0.x ^ q1.x ^ h0.x ^ h1.x ^ hdr ^ sl; }
    __device__ __forceinline__ void zero() { q0 = q1 = h0 = h1 = make_uint4(0, 0, 0, 0); hdr = shw = sl = 0; }
    __device__ __forceinline__ void load(const uint8_t *row, int u) {
        const uint8_t *b = row + (long)(u >> 2) * 176; const int uu = u & 3;
        hdr = ld32(b); shw = ld32(b + 4); sl = ld16(b + 8 + 2 * uu); q0 = ld128(b + 16 + 32 * uu); q1 = ld128(b + 32 + 32 * uu); h0 = ld128(b + 144); h1 = ld128(b + 160);
    }
    template <int VDT>
    static __device__ __forceinline__ void load_y(int u, int K, int c, const int8_t *yq, const float *yd, const float *ys, YReg &y) { Unit<T_IQ2_S>::template load_y<VDT>(u, K, c, yq, yd, ys, y); }
    __device__ __forceinline__ void decode(int u, const void *, Dec &dc) const {
        const int uu = u & 3; const uint32_t ex = (hdr >> 16) >> (4 * uu), shb = (shw >> (8 * uu)) & 0xff;
        dc.d = half_bits_to_float(hdr & 0xffff);
        const uint32_t q[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w}, hb[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {             // k = 0, 1: low nibbles of qs[0..15], qs[16..31] + qh bit 2 uu;  k = 2, 3: high nibbles + qh bit 2 uu + 1
            const uint32_t add = ((ex >> k) & 1) ? 0x02020202u : 0u;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int w = 4 * (k & 1) + i;
                const uint32_t idx = (((k & 2) ? (q[w] >> 4) : q[w]) & 0x0f0f0f0fu) | (((hb[w] >> (2 * uu + (k >> 1))) & 0x01010101u) << 4);
                dc.v[4 * k + i] = add_bytes(iq5nl_lookup4(idx), add);
            }
            dc.ls[k] = iqk_scale6(sl, shb, k);
        }
    }
    static __device__ __forceinline__ float dot(const Dec &dc, const YReg &y, float r) { return Unit<T_Q3_K>::dot(dc, y, r); }
};
// the _KS types: an f32 scale in front of the row's blocks; per 32 weights one byte {7-bit scale (& 254) - 127, bit 0 = shifted value table}
template <> struct Unit<T_IQ4_KS> {         // blocks of 136 bytes {u8 scales[8]; u8 qs[128]} in the IQ4_NL nibble layout
    uint4 q0, q1; uint32_t sc; float drow;
    typedef Unit<T_IQ3_S>::Dec Dec;
    __device__ __forceinline__ uint32_t checksum() const { return q0.x ^ q1.x ^ sc; }
    __device__ __forceinline__ void zero() { q0 = q1 = make_uint4(0, 0, 0, 0); sc = 0; drow = 0.f; }
    __device__ __forceinline__ void load(const uint8_t *row, int u) {
        const uint8_t *b = row + 4 + (long)(u >> 2) * 136; const int uu = u & 3;
        drow = *reinterpret_cast<const float *>(row); sc = ld16(b + 2 * uu); q0 = ld128(b + 8 + 32 * uu); q1 = ld128(b + 24 + 32 * uu);
    }
    template <int VDT>
    static __device__ __forceinline__ void load_y(int u, int K, int c, const int8_t *yq, const float *yd, const float *ys, YReg &y) { Unit<T_IQ3_S>::template load_y<VDT>(u, K, c, yq, yd, ys, y); }
    __device__ __forceinline__ void decode(int, const void *, Dec &dc) const {
        dc.d = drow;
        const uint32_t q[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const uint32_t s = (sc >> (8 * p)) & 0xff, add = (s & 1) ? 0x04040404u : 0u;
#pragma unroll
            for (int i = 0; i < 4; ++i) { const uint32_t w = q[4 * p + i]; dc.v[8 * p + i] = add_bytes(iq4nl_lookup4(w & 0x0f0f0f0fu), add); dc.v[8 * p + 4 + i] = add_bytes(iq4nl_lookup4((w >> 4) & 0x0f0f0f0fu), add); }
            dc.ls[p] = (int)(s & 254) - 127;
        }
    }
    static __device__ __forceinline__ float dot(const Dec &dc, const YReg &y, float r) { return Unit<T_IQ3_S>::dot(dc, y, r); }
};
template <> struct Unit<T_IQ5_KS> {         // blocks of 168 bytes {u8 scales[8]; u8 qs[128]; u8 qh[32]}: per 64: 32 low nibbles + qh bit 2 i, 32 high nibbles + qh bit 2 i + 1
    uint4 q0, q1, h0, h1; uint32_t sc; float drow;
    typedef Unit<T_IQ3_S>::Dec Dec;
    __device__ __forceinline__ uint32_t checksum() const { return q0.x ^ q1.x ^ h0.x ^ h1.x ^ sc; }
    __device__ __forceinline__ void zero() { q0 = q1 = h0 = h1 = make_uint4(0, 0, 0, 0); sc = 0; drow = 0.f; }
    __device__ __forceinline__ void load(const uint8_t *row, int u) {
        const uint8_t *b = row + 4 + (long)(u >> 2) * 168; const int uu = u & 3;
        drow = *reinterpret_cast<const float *>(row); sc = ld16(b + 2 * uu); q0 = ld128(b + 8 + 32 * uu); q1 = ld128(b + 24 + 32 * uu); h0 = ld128(b + 136); h1 = ld128(b + 152);
    }
    template <int VDT>
    static __device__ __forceinline__ void load_y(int u, int K, int c, const int8_t *yq, const float *yd, const float *ys, YReg &y) { Unit<T_IQ3_S>::template load_y<VDT>(u, K, c, yq, yd, ys, y); }
    __device__ __forceinline__ void decode(int u, const void *, Dec &dc) const {
        const int uu = u & 3;
        dc.d = drow;
        const uint32_t q[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w}, hb[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const uint32_t s = (sc >> (8 * p)) & 0xff, add = (s & 1) ? 0x02020202u : 0u;
#pragma unroll
            for (int i = 0; i < 8; ++i) dc.v[8 * p + i] = add_bytes(iq5nl_lookup4(((p ? (q[i] >> 4) : q[i]) & 0x0f0f0f0fu) | (((hb[i] >> (2 * uu + p)) & 0x01010101u) << 4)), add);
            dc.ls[p] = (int)(s & 254) - 127;
        }
    }
    static __device__ __forceinline__ float dot(const Dec &dc, const YReg &y, float r) { return Unit<T_IQ3_S>::dot(dc, y, r); }
};

// IQ2_KS / IQ3_KS: f16 row scale, blocks {u16 extra; u8 scales[4]; u8 qs[64]; [u8 qh[32]]} = the IQ2_K / IQ3_K packing with 5-bit scales per 32 weights
template <> struct Unit<T_IQ2_KS> {
    uint4 q0, q1; uint32_t ex, sc; uint32_t drow;
    typedef Unit<T_IQ3_S>::Dec Dec;
    __device__ __forceinline__ uint32_t checksum() const { return q0.x ^ q1.x ^ ex ^ sc; }
    __device__ __forceinline__ void zero() { q0 = q1 = make_uint4(0, 0, 0, 0); ex = sc = drow = 0; }
    __device__ __forceinline__ void load(const uint8_t *row, int u) {
        const uint8_t *b = row + 2 + (long)(u >> 2) * 70; const int n = (u >> 1) & 1;
        drow = ld16(row); ex = ld16(b); sc = b[2 + (u & 3)]; q0 = ld128(b + 6 + 32 * n); q1 = ld128(b + 22 + 32 * n);
    }
    template <int VDT>
    static __device__ __forceinline__ void load_y(int u, int K, int c, const int8_t *yq, const float *yd, const float *ys, YReg &y) { Unit<T_IQ3_S>::template load_y<VDT>(u, K, c, yq, yd, ys, y); }
    __device__ __forceinline__ void decode(int u, const void *, Dec &dc) const {
        const int uu = u & 3, s0 = 4 * (u & 1);
        dc.d = half_bits_to_float(drow);
        const uint32_t q[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
        for (int p = 0; p < 2; ++p) {             // 32-block ib = 2 uu + p
            const int ib = 2 * uu + p; const uint32_t t = k_iq2nl_packed[(ex >> ib) & 1];
#pragma unroll
            for (int i = 0; i < 8; ++i) dc.v[8 * p + i] = __builtin_amdgcn_perm(t, t, (q[i] >> (s0 + 2 * p)) & 0x03030303u);
            dc.ls[p] = (int)(((sc >> (4 * p)) & 15) | (((ex >> (8 + ib)) & 1) << 4)) - 16;
        }
    }
    static __device__ __forceinline__ float dot(const Dec &dc, const YReg &y, float r) { return Unit<T_IQ3_S>::dot(dc, y, r); }
};
template <> struct Unit<T_IQ3_KS> {
    uint4 q0, q1, h0, h1; uint32_t ex, scw; uint32_t drow;
    typedef Unit<T_IQ3_S>::Dec Dec;
    __device__ __forceinline__ uint32_t checksum() const { return q0.x ^ q1.x ^ h0.x ^ h1.x ^ ex ^ scw; }
    __device__ __forceinline__ void zero() { q0 = q1 = h0 = h1 = make_uint4(0, 0, 0, 0); ex = scw = drow = 0; }
    __device__ __forceinline__ void load(const uint8_t *row, int u) {
        const uint8_t *b = row + 2 + (long)(u >> 2) * 102; const int n = (u >> 1) & 1;
        drow = ld16(row); ex = ld16(b); scw = ld32(b + 2); q0 = ld128(b + 6 + 32 * n); q1 = ld128(b + 22 + 32 * n); h0 = ld128(b + 70); h1 = ld128(b + 86);
    }
    template <int VDT>
    static __device__ __forceinline__ void load_y(int u, int K, int c, const int8_t *yq, const float *yd, const float *ys, YReg &y) { Unit<T_IQ3_S>::template load_y<VDT>(u, K, c, yq, yd, ys, y); }
    __device__ __forceinline__ void decode(int u, const void *, Dec &dc) const {
        const int uu = u & 3, s0 = 4 * (u & 1);
        dc.d = half_bits_to_float(drow);
        const uint32_t q[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w}, hb[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int ib = 2 * uu + p, e = (ex >> (8 + ib)) & 1; const uint32_t t0 = k_iq3nl_packed[2 * e], t1 = k_iq3nl_packed[2 * e + 1];
#pragma unroll
            for (int i = 0; i < 8; ++i) dc.v[8 * p + i] = __builtin_amdgcn_perm(t1, t0, ((q[i] >> (s0 + 2 * p)) & 0x03030303u) | (((hb[i] >> ib) & 0x01010101u) << 2));
            dc.ls[p] = (int)(((scw >> (8 * (ib & 3) + 4 * (ib >> 2))) & 15) | (((ex >> ib) & 1) << 4)) - 16;
        }
    }
    static __device__ __forceinline__ float dot(const Dec &dc, const YReg &y, float r) { return Unit<T_IQ3_S>::dot(dc, y, r); }
};

// IQ4_KSS: f32 row scale, blocks of 128 bytes {u32 qs[32]}: per 32 weights eight 16-bit words whose LOW bits spell the IQ4_KS scale byte; the other 15 bits, after
// a ^= a >> 1, are the 4 nibbles of the IQ4_NL layout (dequantize_row_iq4_kss)
template <> struct Unit<T_IQ4_KSS> {
    uint4 q0, q1; float drow;
    typedef Unit<T_IQ3_S>::Dec Dec;
    __device__ __forceinline__ uint32_t checksum() const { return q0.x ^ q1.x; }
    __device__ __forceinline__ void zero() { q0 = q1 = make_uint4(0, 0, 0, 0); drow = 0.f; }
    __device__ __forceinline__ void load(const uint8_t *row, int u) {
        const uint8_t *b = row + 4 + (long)(u >> 2) * 128 + 32 * (u & 3);
        drow = *reinterpret_cast<const float *>(row); q0 = ld128(b); q1 = ld128(b + 16);
    }
    template <int VDT>
    static __device__ __forceinline__ void load_y(int u, int K, int c, const int8_t *yq, const float *yd, const float *ys, YReg &y) { Unit<T_IQ3_S>::template load_y<VDT>(u, K, c, yq, yd, ys, y); }
    __device__ __forceinline__ void decode(int, const void *, Dec &dc) const {
        dc.d = drow;
        const uint32_t q[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            uint32_t ls = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) { const uint32_t w = q[4 * p + i]; ls |= ((w & 1u) | ((w >> 15) & 2u)) << (2 * i); }
            const uint32_t add = (ls & 1) ? 0x04040404u : 0u;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint32_t a = q[4 * p + i] & 0xfffefffeu; a ^= (a >> 1) & 0x7fff7fffu;
                dc.v[8 * p + i] = add_bytes(iq4nl_lookup4(a & 0x0f0f0f0fu), add); dc.v[8 * p + 4 + i] = add_bytes(iq4nl_lookup4((a >> 4) & 0x0f0f0f0fu), add);
            }
            dc.ls[p] = (int)(ls & 254) - 127;
        }
    }
    static __device__ __forceinline__ float dot(const Dec &dc, const YReg &y, float r) { return Unit<T_IQ3_S>::dot(dc, y, r); }
};
// IQ2_KL: f16 row scale, blocks of 86 bytes {u16 scales_h; u8 scales_l[4]; u8 qs[64]; u8 qh[16]}: a 5-bit index (nibble + one qh bit) names a PAIR of values; per 64 weights
// the low nibbles of 16 bytes are elements 0..31 (two per byte), the high nibbles 32..63; 6-bit scales per 32
template <> struct Unit<T_IQ2_KL> {
    uint4 q, qh; uint32_t sl, sh, drow;
    typedef Unit<T_IQ3_S>::Dec Dec;
    __device__ __forceinline__ uint32_t checksum() const { return q.x ^ qh.x ^ sl ^ sh; }
    __device__ __forceinline__ void zero() { q = qh = make_uint4(0, 0, 0, 0); sl = sh = drow = 0; }
    __device__ __forceinline__ void load(const uint8_t *row, int u) {
        const uint8_t *b = row + 2 + (long)(u >> 2) * 86;
        drow = ld16(row); sh = ld16(b); sl = ld32(b + 2); q = ld128(b + 6 + 16 * (u & 3)); qh = ld128(b + 70);
    }
    template <int VDT>
    static __device__ __forceinline__ void load_y(int u, int K, int c, const int8_t *yq, const float *yd, const float *ys, YReg &y) { Unit<T_IQ3_S>::template load_y<VDT>(u, K, c, yq, yd, ys, y); }
    __device__ __forceinline__ void decode(int u, const void *, Dec &dc) const {
        const int i = u & 3;
        dc.d = half_bits_to_float(drow);
        const uint32_t w[4] = {q.x, q.y, q.z, q.w}, hb[4] = {qh.x, qh.y, qh.z, qh.w};
#pragma unroll
        for (int p = 0; p < 2; ++p) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {             // bytes 4 k .. 4 k + 3 -> 8 weights (two per byte)
                const uint32_t idx = ((p ? (w[k] >> 4) : w[k]) & 0x0f0f0f0fu) | (((hb[k] >> (2 * i + p)) & 0x01010101u) << 4);
                const uint32_t a = lookup32x4(k_iq2kl_v0, idx), b2 = lookup32x4(k_iq2kl_v1, idx);
                dc.v[8 * p + 2 * k] = __builtin_amdgcn_perm(b2, a, 0x05010400u); dc.v[8 * p + 2 * k + 1] = __builtin_amdgcn_perm(b2, a, 0x07030602u);
            }
            dc.ls[p] = (int)(((sl >> (8 * ((2 * i + p) & 3) + 4 * (i >> 1))) & 15) | (((sh >> (4 * i + 2 * p)) & 3) << 4)) - 32;
        }
    }
    static __device__ __forceinline__ float dot(const Dec &dc, const YReg &y, float r) { return Unit<T_IQ3_S>::dot(dc, y, r); }
};

// IQ6_K: 212 bytes {f16 d; u16 extra; i8 scales[16]; u8 qs[128]; u8 qh[64]}: 6-bit index = nibble | 2 qh bits, int8 table iq6nl_values (+ 1 where the extra bit says so), int8 scale per
// 16 weights (DequantizerIQ6K, iqk_gemm_iqk_quants.cpp:692-750).  lane = the 64-group i: qs[32 i ..], qh[32 (i / 2) ..] >> 4 (i % 2)
template <> struct Unit<T_IQ6_K> {
    uint4 q0, q1, h0, h1; uint32_t hdr, scw;
    typedef Unit<T_Q3_K>::Dec Dec;
    __device__ __forceinline__ uint32_t checksum() const { return q0.x ^ q1.x ^ h0.x ^ h1.x ^ hdr ^ scw; }
    __device__ __forceinline__ void zero() { q0 = q1 = h0 = h1 = make_uint4(0, 0, 0, 0); hdr = scw = 0; }
    __device__ __forceinline__ void load(const uint8_t *row, int u) {
        const uint8_t *b = row + (long)(u >> 2) * 212; const int i = u & 3;
        hdr = ld32(b); scw = ld32(b + 4 + 4 * i); q0 = ld128(b + 20 + 32 * i); q1 = ld128(b + 36 + 32 * i); h0 = ld128(b + 148 + 32 * (i >> 1)); h1 = ld128(b + 164 + 32 * (i >> 1));
    }
    template <int VDT>
    static __device__ __forceinline__ void load_y(int u, int K, int c, const int8_t *yq, const float *yd, const float *ys, YReg &y) { Unit<T_IQ2_S>::template load_y<VDT>(u, K, c, yq, yd, ys, y); }
    static __device__ __forceinline__ uint32_t lookup64x4(uint32_t idx) {
        const uint32_t lo = lookup32x4(k_iq6nl_packed, idx), hi = lookup32x4(k_iq6nl_packed + 8, idx), m5 = ((idx >> 5) & 0x01010101u) * 0xffu;
        return (hi & m5) | (lo & ~m5);
    }
    __device__ __forceinline__ void decode(int u, const void *, Dec &dc) const {
        const int i = u & 3, sh = 4 * (i & 1); const uint32_t ex = (hdr >> 16) >> (4 * i);
        dc.d = half_bits_to_float(hdr & 0xffff);
        const uint32_t q[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w}, hb[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {             // k = 0, 1: low nibbles of qs[0..15], qs[16..31] + qh bits 0..1;  k = 2, 3: high nibbles + qh bits 2..3
            const uint32_t add = ((ex >> k) & 1) ? 0x01010101u : 0u;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int w = 4 * (k & 1) + j;
                const uint32_t idx = (((k & 2) ? (q[w] >> 4) : q[w]) & 0x0f0f0f0fu) | (((hb[w] >> (sh + (k & 2))) & 0x03030303u) << 4);
                dc.v[4 * k + j] = add_bytes(lookup64x4(idx), add);
            }
            dc.ls[k] = (int)(int8_t)((scw >> (8 * k)) & 0xff);
        }
    }
    static __device__ __forceinline__ float dot(const Dec &dc, const YReg &y, float r) { return Unit<T_Q3_K>::dot(dc, y, r); }
};

#ifndef GEMV_DEPTH
#define GEMV_DEPTH 4
#endif

// sum over aligned groups of `width` (16 / 32 / 64) lanes with DPP only (no LDS traffic); the total lands in the LAST lane
// of each group.  Sequence: row_shr 1,2,3 / row_shr 4 / row_shr 8 inside 16-lane rows, then row_bcast15, row_bcast31.
// ---- trellis types (IQ1_KT / IQ2_KT / IQ3_KT / IQ4_KT): an f32 scale in front of the row, then 256-blocks of generator seeds.  One generator for all four
// (Trellis3, iqk_gemm_ktquants.cpp:101-172; trellis_next_int, convert.cu:342-346): value j of a seed = (sum of the four 6-bit fields of (seed + offset) * ka^(j+1) mod 2^32) - 126,
// ka = 0xCBAC1FED -- an int8 in [-126, 126]; 8 values per seed (IQ4_KT: 4).  Nothing to look up: the decode is ALU work (per weight: the product, a mask, a byte sum;
// per four weights a pack and a dot) and these kernels are issue-bound far below the HBM roof.  Activations: block_q8_2_x4 (ggml.c:1634-1694), exact int32 block sums,
// f32 accumulate fma((d f s_b) dy_b, sum, .) as mul_mat_iqX_kt_q8_2_x4_T (iqk_gemm_ktquants.cpp:568-657,658-738,803-893,1102-1199) -- f = the 1.05 / 1.01 the reference's
// mat-mul kernels put on the row scale of IQ2_KT / IQ3_KT (kt_matmul_factor).
constexpr uint32_t KT_KA = 0xCBAC1FEDu;
constexpr uint32_t kt_pow(int n) { uint32_t r = 1; for (int i = 0; i < n; ++i) r *= KT_KA; return r; }
// seed * ka^(J+1) mod 2^32 for a seed below 2^17 (16-bit index + 4096 [+ 32768]): two 24-bit multiplies (full rate) instead of v_mul_lo_u32
template <int J> __device__ __forceinline__ uint32_t kt_x(uint32_t seed) { constexpr uint32_t K = kt_pow(J + 1); return __umul24(seed, K & 0xffffu) + (__umul24(seed, K >> 16) << 16); }
template <int J> __device__ __forceinline__ int kt_val(uint32_t seed) { return (int)__builtin_amdgcn_udot4(kt_x<J>(seed) & 0x3f3f3f3fu, 0x01010101u, (uint32_t)-126, false); }
// the low bytes of four int32 in [-128, 127] -> one dword (element 0 in byte 0)
__device__ __forceinline__ uint32_t kt_pack4(int a, int b, int c, int d) {
    return __builtin_amdgcn_perm((uint32_t)b, (uint32_t)a, 0x0c0c0400u) | __builtin_amdgcn_perm((uint32_t)d, (uint32_t)c, 0x04000c0cu);
}
template <bool ABS> __device__ __forceinline__ void kt_group8(uint32_t seed, uint32_t &lo, uint32_t &hi) {       // the 8 values of one seed, packed
    int v0 = kt_val<0>(seed), v1 = kt_val<1>(seed), v2 = kt_val<2>(seed), v3 = kt_val<3>(seed), v4 = kt_val<4>(seed), v5 = kt_val<5>(seed), v6 = kt_val<6>(seed), v7 = kt_val<7>(seed);
    if (ABS) { v0 = abs(v0); v1 = abs(v1); v2 = abs(v2); v3 = abs(v3); v4 = abs(v4); v5 = abs(v5); v6 = abs(v6); v7 = abs(v7); }
    lo = kt_pack4(v0, v1, v2, v3); hi = kt_pack4(v4, v5, v6, v7);
}
__device__ __forceinline__ int iq4k_value(uint32_t i) { return (int)(int8_t)((k_iq4nl_packed[i >> 2] >> (8 * (i & 3))) & 0xff); }       // iq4k_values[0..15] = the IQ4_NL table
template <> struct Unit<T_IQ2_KT> {         // blocks of 68 bytes {u8 scales[4]; u16 ql[32]}: seed g (weights 8 g .. 8 g + 7) = ql[g] + 4096; scale of 32-block ib = iq4k_values[nibble ib / 4 of scales[ib % 4]]
    uint4 q; uint32_t sc; float drow;
    typedef UnitNib<T_IQ4_NL>::Dec Dec;
    __device__ __forceinline__ uint32_t checksum() const { return q.x ^ q.y ^ q.z ^ q.w ^ sc; }
    __device__ __forceinline__ void zero() { q = make_uint4(0, 0, 0, 0); sc = 0; drow = 0.f; }
    __device__ __forceinline__ void load(const uint8_t *row, int u) {
        const uint8_t *b = row + 4 + (long)(u >> 2) * 68;
        drow = *reinterpret_cast<const float *>(row); sc = ld32(b); q = ld128(b + 4 + 16 * (u & 3));
    }
    template <int VDT>
    static __device__ __forceinline__ void load_y(int u, int K, int c, const int8_t *yq, const float *yd, const float *ys, YReg &y) { Unit<T_IQ4_NL>::template load_y<VDT>(u, K, c, yq, yd, ys, y); }
    __device__ __forceinline__ void decode(int u, const void *, Dec &dc) const {
        const int uu = u & 3; const float d = drow * kt_matmul_factor(T_IQ2_KT);
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int g = 0; g < 8; ++g) kt_group8<false>(((w[g >> 1] >> (16 * (g & 1))) & 0xffffu) + 4096u, dc.v[2 * g], dc.v[2 * g + 1]);
        const uint32_t s = sc >> (4 * (uu >> 1));                        // blocks 2 uu, 2 uu + 1: bytes (2 uu) % 4 and + 1, nibble uu / 2
        dc.d0 = d * (float)iq4k_value((s >> (8 * ((2 * uu) & 3))) & 15u); dc.d1 = d * (float)iq4k_value((s >> (8 * ((2 * uu + 1) & 3))) & 15u);
    }
    static __device__ __forceinline__ float dot(const Dec &dc, const YReg &y, float r) { return Unit<T_IQ4_NL>::dot(dc, y, r); }
};
template <> struct Unit<T_IQ3_KT> {         // blocks of 100 bytes {u8 scales[4]; u16 ql[32]; u8 qh[32]}: the IQ2_KT seeds, values |.|, sign of weight j of 32-block ib = bit ib of qh[j], scale = the nibble
    uint4 q, h0, h1; uint32_t sc; float drow;
    typedef UnitNib<T_IQ4_NL>::Dec Dec;
    __device__ __forceinline__ uint32_t checksum() const { return q.x ^ q.y ^ q.z ^ q.w ^ h0.x ^ h1.x ^ sc; }
    __device__ __forceinline__ void zero() { q = h0 = h1 = make_uint4(0, 0, 0, 0); sc = 0; drow = 0.f; }
    __device__ __forceinline__ void load(const uint8_t *row, int u) {
        const uint8_t *b = row + 4 + (long)(u >> 2) * 100;
        drow = *reinterpret_cast<const float *>(row); sc = ld32(b); q = ld128(b + 4 + 16 * (u & 3)); h0 = ld128(b + 68); h1 = ld128(b + 84);
    }
    template <int VDT>
    static __device__ __forceinline__ void load_y(int u, int K, int c, const int8_t *yq, const float *yd, const float *ys, YReg &y) { Unit<T_IQ4_NL>::template load_y<VDT>(u, K, c, yq, yd, ys, y); }
    __device__ __forceinline__ void decode(int u, const void *, Dec &dc) const {
        const int uu = u & 3; const float d = drow * kt_matmul_factor(T_IQ3_KT);
        const uint32_t w[4] = {q.x, q.y, q.z, q.w}, hb[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
        for (int g = 0; g < 8; ++g) kt_group8<true>(((w[g >> 1] >> (16 * (g & 1))) & 0xffffu) + 4096u, dc.v[2 * g], dc.v[2 * g + 1]);
#pragma unroll
        for (int p = 0; p < 2; ++p)                                      // 32-block 2 uu + p: dword i of it = weights 4 i .. 4 i + 3 = bit (2 uu + p) of qh bytes 4 i .. 4 i + 3
#pragma unroll
            for (int i = 0; i < 8; ++i) {        // byte-wise negation of magnitudes 0 .. 126 without a carry between bytes: 7-bit two's complement ((m ^ 0x7f) + 1 <= 0x80), then the sign bit
                const uint32_t neg = (hb[i] >> (2 * uu + p)) & 0x01010101u;
                dc.v[8 * p + i] = ((dc.v[8 * p + i] ^ (neg * 0x7fu)) + neg) ^ (neg << 7);
            }
        const uint32_t s = sc >> (4 * (uu >> 1));
        dc.d0 = d * (float)((s >> (8 * ((2 * uu) & 3))) & 15u); dc.d1 = d * (float)((s >> (8 * ((2 * uu + 1) & 3))) & 15u);
    }
    static __device__ __forceinline__ float dot(const Dec &dc, const YReg &y, float r) { return Unit<T_IQ4_NL>::dot(dc, y, r); }
};
template <> struct Unit<T_IQ4_KT> {         // blocks of 128 bytes {u32 shb[8]; u8 ql[64]; u8 qh[32]}: 64 seeds of FOUR weights: ql[jj] | nibble (jj / 32) of qh[jj % 32] << 8 | 3 bits of shb[ib] << 12,
                                            // offset 4096 (+ 32768 if shb[ib] & 1), scale of 32-block ib = ((shb[ib] & 0xff) >> 1) - 64
    uint4 ql, qh; uint2 sh; float drow;
    typedef UnitNib<T_IQ4_NL>::Dec Dec;
    __device__ __forceinline__ uint32_t checksum() const { return ql.x ^ ql.y ^ ql.z ^ ql.w ^ qh.x ^ sh.x ^ sh.y; }
    __device__ __forceinline__ void zero() { ql = qh = make_uint4(0, 0, 0, 0); sh = make_uint2(0, 0); drow = 0.f; }
    __device__ __forceinline__ void load(const uint8_t *row, int u) {
        const uint8_t *b = row + 4 + (long)(u >> 2) * 128; const int uu = u & 3;
        drow = *reinterpret_cast<const float *>(row); sh = ld64(b + 8 * uu); ql = ld128(b + 32 + 16 * uu); qh = ld128(b + 96 + 16 * (uu & 1));
    }
    template <int VDT>
    static __device__ __forceinline__ void load_y(int u, int K, int c, const int8_t *yq, const float *yd, const float *ys, YReg &y) { Unit<T_IQ4_NL>::template load_y<VDT>(u, K, c, yq, yd, ys, y); }
    __device__ __forceinline__ void decode(int u, const void *, Dec &dc) const {
        const int hs = 4 * ((u & 3) >> 1);                               // groups 16 uu .. 16 uu + 15: jj / 32 = uu / 2 picks the nibble of qh
        const uint32_t lw[4] = {ql.x, ql.y, ql.z, ql.w}, hw[4] = {qh.x, qh.y, qh.z, qh.w}, sw[2] = {sh.x, sh.y};
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const uint32_t s = sw[p], offset = 4096u + ((s & 1u) << 15);
#pragma unroll
            for (int ig = 0; ig < 8; ++ig) {
                const int j = 8 * p + ig;                                // group within the unit
                const uint32_t seed = (((lw[j >> 2] >> (8 * (j & 3))) & 0xffu) | ((((hw[j >> 2] >> (8 * (j & 3))) >> hs) & 15u) << 8) | (((s >> (8 + 3 * ig)) & 7u) << 12)) + offset;
                dc.v[j] = kt_pack4(kt_val<0>(seed), kt_val<1>(seed), kt_val<2>(seed), kt_val<3>(seed));
            }
        }
        dc.d0 = drow * (float)((int)((sw[0] & 0xffu) >> 1) - 64); dc.d1 = drow * (float)((int)((sw[1] & 0xffu) >> 1) - 64);
    }
    static __device__ __forceinline__ float dot(const Dec &dc, const YReg &y, float r) { return Unit<T_IQ4_NL>::dot(dc, y, r); }
};
template <> struct Unit<T_IQ1_KT> {         // blocks of 56 bytes {u8 sh[8]; u8 ql[32]; u8 qh[16]}: seed of group ib (8 weights) = ql[ib] | nibble (ib / 16) of qh[ib % 16] << 8 | bit 4 + ib % 4 of sh[ib / 4] << 12, + 4096;
                                            // scale of 32-block ib = iq4k_values[sh[ib] & 15]
    uint2 ql, qh; uint32_t sh; float drow;
    typedef UnitNib<T_IQ4_NL>::Dec Dec;
    __device__ __forceinline__ uint32_t checksum() const { return ql.x ^ ql.y ^ qh.x ^ qh.y ^ sh; }
    __device__ __forceinline__ void zero() { ql = qh = make_uint2(0, 0); sh = 0; drow = 0.f; }
    __device__ __forceinline__ void load(const uint8_t *row, int u) {
        const uint8_t *b = row + 4 + (long)(u >> 2) * 56; const int uu = u & 3;
        drow = *reinterpret_cast<const float *>(row); sh = ld16(b + 2 * uu); ql = ld64(b + 8 + 8 * uu); qh = ld64(b + 40 + 8 * (uu & 1));
    }
    template <int VDT>
    static __device__ __forceinline__ void load_y(int u, int K, int c, const int8_t *yq, const float *yd, const float *ys, YReg &y) { Unit<T_IQ4_NL>::template load_y<VDT>(u, K, c, yq, yd, ys, y); }
    __device__ __forceinline__ void decode(int u, const void *, Dec &dc) const {
        const int hs = 4 * ((u & 3) >> 1);                               // groups 8 uu .. 8 uu + 7: ib / 16 = uu / 2
        const uint32_t lw[2] = {ql.x, ql.y}, hw[2] = {qh.x, qh.y};
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const uint32_t shb = (sh >> (8 * (g >> 2))) & 0xffu;         // sh[(8 uu + g) / 4]
            const uint32_t seed = (((lw[g >> 2] >> (8 * (g & 3))) & 0xffu) | ((((hw[g >> 2] >> (8 * (g & 3))) >> hs) & 15u) << 8) | (((shb >> (4 + (g & 3))) & 1u) << 12)) + 4096u;
            kt_group8<false>(seed, dc.v[2 * g], dc.v[2 * g + 1]);
        }
        dc.d0 = drow * (float)iq4k_value(sh & 15u); dc.d1 = drow * (float)iq4k_value((sh >> 8) & 15u);
    }
    static __device__ __forceinline__ float dot(const Dec &dc, const YReg &y, float r) { return Unit<T_IQ4_NL>::dot(dc, y, r); }
};

template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ float dpp_mov(float src) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(src), CTRL, ROW_MASK, BANK_MASK, false));
}
// Sum over `width` (16 / 32 / 64) consecutive lanes, valid in the LAST lane of each group.  Every step runs with full row / bank masks
// (lanes that do not feed the last lane just accumulate values nobody reads; invalid sources read as 0), which lets the compiler fold
// each DPP move into its add (v_add_f32_dpp) instead of emitting v_mov_dpp + s_nop + v_add.
__device__ __forceinline__ float dpp_row_sum(float v, int width) {
    v = v + dpp_mov<0x111, 0xf, 0xf>(v) + dpp_mov<0x112, 0xf, 0xf>(v) + dpp_mov<0x113, 0xf, 0xf>(v);   // row_shr:1,2,3
    v += dpp_mov<0x114, 0xf, 0xf>(v);                                                                   // row_shr:4
    v += dpp_mov<0x118, 0xf, 0xf>(v);                                                                   // row_shr:8 -> lane 15 of each row
    if (width >= 32) v += dpp_mov<0x142, 0xf, 0xf>(v);                                                  // row_bcast:15 -> lanes 31, 63 (row r += row r-1's total)
    if (width >= 64) v += dpp_mov<0x143, 0xf, 0xf>(v);                                                  // row_bcast:31 -> lane 63 (rows 2, 3 += lane 31)
    return v;
}

// the waves' partial sums of squares (one float per wave, 16 slots) added in wave order.  All four 16-byte reads are issued before the first value is used: as a loop over a
// run-time wave count this was a chain of eight dependent LDS round trips (~0.3 us) in the prologue of every norm-carrying launch (round 6, scripts/mb_norm.py)
__device__ __forceinline__ float norm_partials_sum(const float *nred, int nwaves) {
    const float4 *p = reinterpret_cast<const float4 *>(nred);
    const float4 v0 = p[0], v1 = p[1], v2 = p[2], v3 = p[3];
    const float v[16] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) tot += i < nwaves ? v[i] : 0.f;
    return tot;
}

// ------------------------------------------------------------------------------------------------
// the kernel.  grid.x = workgroups striding over row groups; grid.y = MoE (token, slot) pair or 1.
// YITERS > 0 (NCOLS == 1 only): the lane's activation slices live in registers for the whole kernel (each row has exactly
// YITERS K-slices of 64 lanes; DEPTH % YITERS == 0 makes the slice index of a ring slot a compile-time constant).
// YITERS == 0: activations are re-read from the LDS image per step (any K, NCOLS up to 4).
// VDT = the activation quantization the CPU path pairs with the tensor's type: type_vec_dot(TYPE) for base types; for weights that
// arrived row-interleaved (_R4, un-interleaved at upload) it is the _R4 kernels' type (Q8_K32 for Q4_K/Q5_K, Q8_K for Q6_K).
// NR = weight rows a lane works on per step (same 64-weight column slice, so one set of activation registers serves them all and the
// per-step bookkeeping -- ~80 of ~170 instructions at NR = 1 -- is shared; the loop is VALU-bound, see profiles/r01_notes.md)
// bx / gx: this workgroup's index and the number of workgroups working on `a` (= blockIdx.x / gridDim.x except in gemv_dual_kernel)
// LPR = lanes per row as a compile-time constant (64: every K >= 2112, i.e. all of a 4096-wide model) or 0 = decided at run time
// (16 / 32 / 64 by K).  With a run-time value the row-end code (reduction width, which lane parks which sum) is a chain of a dozen
// uniform branches -- a third of the per-step instructions of the fused kernel.
template <int TYPE, int NCOLS, bool UPGATE, int YITERS, int VDT, int DEPTH, bool MULTI, int NR, int LPR, int FX = 0>
static __device__ __forceinline__ void gemv_body(const GemvArgs &a, const int bx, const int gx) {
    static_assert(FX == 0 || (NCOLS == 1 && YITERS == 1), "fused norm / residual variants exist for single-column, single-slice launches");
    static_assert(FX != 5 || (!UPGATE && !MULTI && NR == 1), "attention-fed variant: one plain matrix");
    // FX = 4: the q,k,v epilogue (round 3's first form, FX = 3 -- four dependent memory round trips at the tail of every wave -- was retired in round 5)
    constexpr bool QKV = FX == 4;
    static_assert(!QKV || (NR == 1 && LPR == 64 && !UPGATE), "q,k,v epilogue: one row per wave step");
    constexpr bool NORM = FX == 1 || QKV;
    constexpr bool RES = FX == 2 || FX == 5;          // residual added in the epilogue
    constexpr bool WAITX = FX == 5;                   // the activation row comes from sibling workgroups of this launch
    static_assert(YITERS == 0 || (DEPTH % YITERS == 0 && (NCOLS == 1 || YITERS == 1)), "register-resident activations: one column, or several columns of a single K-slice");
    static_assert(NR == 1 || NCOLS == 1, "several rows per step: single column only");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int K = a.K;
    int8_t *yq = reinterpret_cast<int8_t *>(smem);
    float  *yd = reinterpret_cast<float *>(smem + (size_t)NCOLS * K);
    float  *ys = yd + (size_t)NCOLS * (K / act_scale_block<VDT>());
    constexpr size_t TAB_ALIGN = type_has_tables(TYPE) ? 4095 : 15;       // (as gemv_lds_bytes)
    const size_t grid_off = (((size_t)NCOLS * K + (size_t)NCOLS * (K / act_scale_block<VDT>()) * 4 + (act_has_sums<VDT>() ? (size_t)NCOLS * (K / 32) * 4 : 0)) + TAB_ALIGN) & ~TAB_ALIGN;
    uint8_t *grid_lds = smem + grid_off;                       // IQ2_S / IQ3_S: [sign LUT][codebook] ("LDS tables" above)

#ifdef GEMV_EXP_TIMELINE
#define TL_STAMP(I_) if (a.timeline && threadIdx.x == 0 && blockIdx.y == 0) a.timeline[8 * bx + (I_)] = wall_clock64()
#else
#define TL_STAMP(I_)
#endif
    TL_STAMP(0);
    const uint8_t *A0 = a.A[0], *A2 = a.A2, *Bbase = a.B; float *C0 = a.C[0];
    long expert = 0;
    if (a.ids) {                                 // MoE: one (token, slot) pair per blockIdx.y
        const int pair = blockIdx.y + a.pair0, tok = pair / a.n_used, slot = pair - tok * a.n_used;
        const int e = reinterpret_cast<const int32_t *>(reinterpret_cast<const uint8_t *>(a.ids) + (long)tok * a.ids_nb1)[slot];
        C0 += (long)tok * a.nb2 + (long)slot * a.nb1;
        if (e < 0 || e >= a.n_expert) {          // invalid id -> zero row (ggml.c:18178-18187)
            for (int i = bx * blockDim.x + threadIdx.x; i < a.M; i += gx * blockDim.x) C0[i] = 0.f;
            return;
        }
        A0 += (long)e * a.expert_stride; if (UPGATE) A2 += (long)e * a.expert_stride;
        expert = e;
        Bbase += (long)tok * a.nb12 + (long)slot * a.nb11;
    }

    // ---- work decomposition
    const int U = K >> 6;                                    // 64-weight units per row
    int lpr_rt = 64; if (U <= 16) lpr_rt = 16; else if (U <= 32) lpr_rt = 32;
    const int lpr = LPR ? LPR : lpr_rt;
    const int rpi = 64 / lpr;                                // rows per wave-iteration
    const int iters = YITERS > 0 ? YITERS : (U + lpr - 1) / lpr;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int sub = lane / lpr, u0 = lane - sub * lpr;
    // 32-bit bookkeeping throughout (rows < 2^31): the GEMV main loop is VALU-bound (~250 instructions per 64-weight step before this
    // was trimmed, of which ~90 are decode + dot), so every 64-bit compare / select in the per-step code costs bandwidth.
    const int wave_id = bx * nwaves + wave, wave_stride = gx * nwaves;
    const int rpg = rpi * NR;                                // rows per group: NR sets of rpi rows (row = grp * rpg + r * rpi + sub)
    const int ngroups = (a.M + rpg - 1) / rpg;
    // group index of this wave's i-th group: g0 + i * gstep.  Default: strided over the whole grid.  Emit mode (a.q8_out): workgroup bx
    // owns the 64 consecutive rows [64 bx, 64 bx + 64) so that it can quantize them as two 32-blocks when they are finished.
    const bool emit = UPGATE && NR == 2 && NCOLS == 1 && a.q8_out != nullptr;
    int g0 = wave_id, gstep = wave_stride;
    int my_groups = wave_id < ngroups ? (ngroups - wave_id + wave_stride - 1) / wave_stride : 0;
    if (emit) {                                             // (rpi == 1: K >= 4096; the host checks)
        constexpr int GPW = 64 / 2;                         // groups (row pairs) per workgroup
        g0 = bx * GPW + wave; gstep = nwaves;
        const int gend = min(ngroups, (bx + 1) * GPW);
        my_groups = g0 < gend ? (gend - g0 + gstep - 1) / gstep : 0;
    }
    // FX = 4 (rows = groups, M even): a wave takes row PAIRS (2 p, 2 p + 1), p = wave_id + j * wave_stride, so that the two rows of a rotation finish in
    // neighbouring result lanes
    if (QKV) { const int npairs = a.M >> 1; my_groups = wave_id < npairs ? 2 * ((npairs - wave_id + wave_stride - 1) / wave_stride) : 0; }
    auto grp_of = [&](int i) { return QKV ? 2 * (g0 + (i >> 1) * gstep) + (i & 1) : g0 + i * gstep; };
    const int nsteps = my_groups * iters;

    // global row -> (matrix, local row); a single matrix (everything but the fused q,k,v launch, MULTI) needs no lookup --
    // as a run-time test the select chain (64-bit pointers x 3 matrices, twice per step) was a fifth of the loop's instructions
    auto locate = [&](int row, const uint8_t *&Ap, float *&Cp, int &lrow) {
        Ap = A0; Cp = C0; lrow = row;
        if (MULTI) {
#pragma unroll
            for (int i = 1; i < GEMV_MAX_MATS; ++i) if (i < a.nmat && row >= a.mend[i - 1]) { Ap = a.A[i]; Cp = a.C[i]; lrow = row - a.mend[i - 1]; }
        }
    };

    const int stride32 = (int)a.strideA;         // (row strides are far below 2 GiB; checked by the host)
    Unit<TYPE> ring[DEPTH][NR], ring2[DEPTH][UPGATE ? NR : 1];
    // FX = 2: the residual value of a step's row travels WITH the step's weights (one more load per ring slot).  vmcnt retires in order: a residual read at the store -- or
    // prefetched at the top of the step -- is the NEWEST load when it is needed, and waiting for it drains the whole ring (the round-2/3 kernels did exactly that per row).
    float rring[DEPTH];
    int is_gi = 0, is_it = 0;                                // running (group index, K-slice) of the next step to ISSUE
    auto issue = [&](Unit<TYPE> (&w)[NR], Unit<TYPE> (&w2)[UPGATE ? NR : 1], float &rr) {
        // Always load (steps past the end / lanes past the row re-read unit 0 of row 0 -- one cached line -- and are skipped at
        // compute time): unconditional loads let the compiler emit exact s_waitcnt vmcnt(N) for the ring instead of vmcnt(0).
        const int row0 = grp_of(is_gi) * rpg + sub; int u = is_it * lpr + u0;
        const bool live = is_gi < my_groups && u < U;
        if (!live) u = 0;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            int row = row0 + r * rpi; if (!(live && row < a.M)) row = 0;
            const uint8_t *Ap; float *Cp; int lrow; locate(row, Ap, Cp, lrow);
            const long roff = (long)lrow * stride32;      // 32 x 32 -> 64-bit multiply (one v_mad_i64_i32; the 64 x 64 form costs four instructions per row)
            w[r].load(Ap + roff, u); if (UPGATE) w2[r].load(A2 + roff, u);
            // (FX = 2: NR = 1, one matrix, one slice per row.  The row of the STEP, not of the lane's unit: the lane that stores -- the last of its lpr lanes -- has no unit of its
            //  own when a row is shorter than lpr units (K = 512: 8 units on 16 lanes) and would otherwise pick up row 0's residual)
            if constexpr (RES) { if (r == 0) { const int rrow = row0 + r * rpi; rr = a.R[(is_gi < my_groups && rrow < a.M) ? rrow : 0]; } }
        }
        if (++is_it == iters) { is_it = 0; ++is_gi; }
    };
    // request the first activation chunks, THEN the first DEPTH weight steps; both are in flight during the prologue
    // IQ2_S / IQ3_S: codebook + sign table are COPIED from their expanded global image (12 / 6 KiB, L2-resident) rather than expanded
    // from the packed form by every workgroup (that expansion cost ~2.5 us of prologue); requested before the weight ring like the
    // activations, unconditionally, written to LDS in the prologue.
    TL_STAMP(4);
    IqPre<TYPE> iqpre; iq_preload<TYPE>(a.tables, iqpre);
    constexpr int XP = xpre_for(NCOLS, YITERS, LPR, type_has_tables(TYPE)), QP = qpre_for(NCOLS, YITERS);
    XChunksT<XP> xc; QChunksT<QP> qc;
    if constexpr (!WAITX) {
    if (a.src_f32) preload_activations_f32<NCOLS, XP>(a, Bbase, xc);
    else if (VDT == T_Q8_2_X4) preload_activations_q8<NCOLS, QP>(a, Bbase, qc);
    }
    XChunksT<XP> wc;                                   // FX = 1 / 3: the norm weights of this thread's chunks, requested with the activations (ahead of the weight ring)
    if (NORM) {
        const int k8n = a.K >> 3;
#pragma unroll
        for (int p = 0; p < XP; ++p) {
            const int i = min((int)(threadIdx.x + p * blockDim.x), k8n - 1); const float *x = a.norm_w + 8 * i;
            wc.v[p][0] = *reinterpret_cast<const float4 *>(x); wc.v[p][1] = *reinterpret_cast<const float4 *>(x + 4);
        }
    }
    TL_STAMP(5);
#pragma unroll
    for (int dslot = 0; dslot < DEPTH; ++dslot) issue(ring[dslot], ring2[dslot], rring[dslot]);

    TL_STAMP(1);
    if constexpr (WAITX) {
        // the weights are on their way; now wait for the producers of the activation row (see GemvArgs::fa_sync), then fetch it with agent-scope loads (8-byte granules: what
        // a relaxed agent-scope atomic load lowers to, `global_load_dwordx2 ... sc1`: served past the L1, coherent with the producers' write-through stores)
        __builtin_amdgcn_sched_barrier(0);
        if (threadIdx.x == 0) {
            unsigned spins = 0;
            while (__hip_atomic_load(a.fa_sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < a.fa_expect) {
                if (++spins > (1u << 22)) { __hip_atomic_store(a.fa_sync + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                __builtin_amdgcn_s_sleep(2);
            }
            const unsigned seen = __hip_atomic_fetch_add(a.fa_sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (seen == (unsigned)gx - 1) { __hip_atomic_store(a.fa_sync, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(a.fa_sync + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        }
        __syncthreads();
        const int k8 = a.K >> 3;
#pragma unroll
        for (int p = 0; p < XP; ++p) {
            const int i = min((int)(threadIdx.x + p * blockDim.x), k8 - 1);
            const unsigned long long *x = reinterpret_cast<const unsigned long long *>(Bbase) + 4 * i;
            const unsigned long long g0 = __hip_atomic_load(x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), g1 = __hip_atomic_load(x + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                                     g2 = __hip_atomic_load(x + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), g3 = __hip_atomic_load(x + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            xc.v[p][0] = make_float4(__uint_as_float((unsigned)g0), __uint_as_float((unsigned)(g0 >> 32)), __uint_as_float((unsigned)g1), __uint_as_float((unsigned)(g1 >> 32)));
            xc.v[p][1] = make_float4(__uint_as_float((unsigned)g2), __uint_as_float((unsigned)(g2 >> 32)), __uint_as_float((unsigned)g3), __uint_as_float((unsigned)(g3 >> 32)));
        }
    }
    // ---- prologue: codebook + quantized activations into LDS
    __builtin_amdgcn_sched_barrier(0);           // nothing that consumes a pre-loaded activation may be scheduled above the ring issue
    iq_fill_lds<TYPE>(iqpre, grid_lds);
    // RMS norm of the row (every workgroup holds the whole row in its pre-loaded chunks: K <= 8 * XP * blockDim, host-checked).  Activation types with an f32 block scale
    // (the K-quants' Q8_K forms) take the scale AFTER the quantization: rsqrt(mean(x^2) + eps) is one positive factor of the whole row, so the int8 values of x * w_norm ARE the
    // int8 values of the normed row and the factor goes onto the lane's block scales behind the staging barrier -- the partial sums ride on that barrier instead of one of
    // their own (round 6: with the reduction knocked out tg128 565 -> 579 tok/s; the extra barrier sat between the arrival of the row and the first quantized value of EVERY
    // norm-carrying launch, 64 per token).  bf16 block scales (Q8_2_X4, the reference's activation type for IQ4_NL and the legacy quants) keep the scale in front: bf16(s * d)
    // is not s * bf16(d) and the reference rounds the former.
#ifdef GEMV_EXP_NORM_EARLY           /* A/B build: the round-5 order (scale in front of the quantization, a barrier of its own) */
    constexpr bool NORM_LATE = false;
#else
    constexpr bool NORM_LATE = NORM && VDT != T_Q8_2_X4;
#endif
    float *nred = reinterpret_cast<float *>(smem + gemv_lds_bytes<VDT>(NCOLS, K, TYPE));        // 16 floats behind the activation image (host adds them)
    if (NORM) {
        const int k8n = K >> 3; float ss = 0.f;
#pragma unroll
        for (int p = 0; p < XP; ++p) if ((int)(threadIdx.x + p * blockDim.x) < k8n) {
            const float4 u = xc.v[p][0], v = xc.v[p][1];
            ss += u.x * u.x + u.y * u.y + u.z * u.z + u.w * u.w + v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
#ifdef GEMV_EXP_NO_NORM_REDUCE      /* knock-out (results wrong): no cross-wave reduction, no barrier */
        const float sc = NORM_LATE ? 1.f : 1.0f / sqrtf(1.f + 1e-30f * ss + a.norm_eps);
#else
        ss = dpp_row_sum(ss, 64);
        if (lane == 63) nred[wave] = ss;
        float sc = 1.f;
        if constexpr (!NORM_LATE) {
            __syncthreads();
            sc = 1.0f / sqrtf(norm_partials_sum(nred, nwaves) / (float)K + a.norm_eps);
        }
#endif
#pragma unroll
        for (int p = 0; p < XP; ++p) {
            float4 &u = xc.v[p][0], &v = xc.v[p][1]; const float4 cu = wc.v[p][0], cv = wc.v[p][1];
            if constexpr (NORM_LATE) {
                u.x = cu.x * u.x; u.y = cu.y * u.y; u.z = cu.z * u.z; u.w = cu.w * u.w; v.x = cv.x * v.x; v.y = cv.y * v.y; v.z = cv.z * v.z; v.w = cv.w * v.w;
            } else {
                u.x = sc * cu.x * u.x; u.y = sc * cu.y * u.y; u.z = sc * cu.z * u.z; u.w = sc * cu.w * u.w;
                v.x = sc * cv.x * v.x; v.y = sc * cv.y * v.y; v.z = sc * cv.z * v.z; v.w = sc * cv.w * v.w;
            }
        }
    }
#ifndef GEMV_EXP_NO_PROLOGUE
    if (a.src_f32) stage_activations_f32<VDT, NCOLS, XP>(a, Bbase, xc, yq, yd, ys);
    else           stage_activations_q8<VDT, NCOLS, QP>(a, Bbase, qc, yq, yd, ys);
#endif
    __syncthreads();
    TL_STAMP(2);
    float nsc = 1.f;                              // NORM_LATE: the row's norm factor, applied to the lane's block scales below
    if constexpr (NORM_LATE) {
#ifndef GEMV_EXP_NO_NORM_REDUCE
        nsc = 1.0f / sqrtf(norm_partials_sum(nred, nwaves) / (float)K + a.norm_eps);
#endif
    }

    // register-resident activations: slice `it` of this lane
    YReg yreg[YITERS > 0 ? YITERS : 1][YITERS > 0 ? NCOLS : 1];
    if (YITERS > 0) {
#pragma unroll
        for (int it = 0; it < (YITERS > 0 ? YITERS : 1); ++it) {
            const int u = it * lpr + u0;
#pragma unroll
            for (int c = 0; c < NCOLS; ++c) {
                if (u < U) {
                    Unit<TYPE>::template load_y<VDT>(u, K, c, yq, yd, ys, yreg[it][c]);
                    if constexpr (NORM_LATE) { yreg[it][c].s[0] *= nsc; yreg[it][c].s[1] *= nsc; yreg[it][c].s[2] *= nsc; yreg[it][c].s[3] *= nsc; }
                } else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) yreg[it][c].q[i] = 0;
                    yreg[it][c].s[0] = yreg[it][c].s[1] = yreg[it][c].s[2] = yreg[it][c].s[3] = 0.f;
                }
            }
        }
    }

    // ---- main loop: a wave walks "steps" = (row group, K-slice) pairs; a ring of DEPTH units keeps DEPTH-1..DEPTH
    // weight loads in flight per lane (Little's law: ~50 KB per CU must be outstanding to saturate HBM3E).
    float acc[NR][NCOLS], acc2[NR][NCOLS];
#pragma unroll
    for (int r = 0; r < NR; ++r) { for (int c = 0; c < NCOLS; ++c) { acc[r][c] = 0.f; acc2[r][c] = 0.f; } }
    // Finished rows are PARKED, one per lane in completion order (slot = ((gi - res_gi0) * NR + r) * rpi + sub), and written out up to
    // 64 at a time: locate + epilogue + store then cost one pass per 64 rows instead of one per row (the fused up*gate epilogue alone is
    // ~100 instructions -- exp, two divides -- which every row used to pay with a single lane active).
    // Short row lists keep the immediate store: there the final flush sits on the critical path (measured +0.2-0.4 us on 4096-row matrices).
    const bool park = UPGATE || NR > 1 || my_groups * rpi >= 16 || QKV;
    float res[NCOLS], res2[NCOLS]; int nres = 0, res_gi0 = 0;
#pragma unroll
    for (int c = 0; c < NCOLS; ++c) { res[c] = 0.f; res2[c] = 0.f; }
    const int rpi_sh = rpi == 1 ? 0 : (rpi == 2 ? 1 : 2);
    float *wg_out = reinterpret_cast<float *>(smem + gemv_lds_bytes<VDT>(NCOLS, K, TYPE));      // emit mode: the workgroup's 64 results
    auto flush = [&]() {
        float partner = 0.f;
        if (QKV) partner = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(res[0]), 0xb1, 0xf, 0xf, false));      // quad_perm [1,0,3,2]: the other row of the pair
        if (lane < nres) {
            const int sb = lane & (rpi - 1), t = lane >> rpi_sh, r = NR == 1 ? 0 : (t & (NR - 1)), g = res_gi0 + (NR == 1 ? t : t / NR);
            const int row = grp_of(g) * rpg + r * rpi + sb;
            if (row < a.M) {
                const uint8_t *Ap; float *Cp; int lrow; locate(row, Ap, Cp, lrow);
                if constexpr (FX == 4) {
                    // The first form of this flush made FOUR dependent memory round trips at the tail of every wave (ISA: a lane-indexed load of kind[mi] from the argument block, wait;
                    // rope_tab[d >> 1], wait; a lane-indexed load of kv_slot[mi], wait; the slot's pointer, wait; then the store).  Here the per-matrix values are picked by a select
                    // chain over the (wave-uniform, scalar-loaded) entries, so the only loads left are the slot's pointer and the table entry -- independent of each other, issued
                    // together, one wait.
                    int kind = a.kind[0]; void *const *slot = a.kv_slot[0];
#pragma unroll
                    for (int i = 1; i < GEMV_MAX_MATS; ++i) if (i < a.nmat && row >= a.mend[i - 1]) { kind = a.kind[i]; slot = a.kv_slot[i]; }
                    __half *kv = reinterpret_cast<__half *>(Cp);
                    if (kind != 0 && slot) kv = static_cast<__half *>(*slot);
                    const int d = lrow % a.rope_hd;
                    const bool rot = kind < 2 && d < a.rope_nd;
                    const float2 cs = a.rope_tab[rot ? (d >> 1) : 0];
                    float v = res[0];
                    if (rot) v = (lane & 1) ? __fmaf_rn(partner, cs.y, __fmul_rn(v, cs.x)) : __fmaf_rn(v, cs.x, -__fmul_rn(partner, cs.y));      // (the roundings of rope_rot, ops.hip)
                    if (kind == 0) Cp[lrow] = v;
                    else kv[lrow] = __float2half_rn(v);
                } else {
#pragma unroll
                for (int c = 0; c < NCOLS; ++c) {
                    float v = UPGATE ? up_gate_combine(a.unary_op, res[c], res2[c], a.epi, lrow, expert) : res[c];
                    if (RES) v += a.R[(long)c * a.stride_C + lrow];
                    Cp[(long)c * a.stride_C + lrow] = v;
                    if (emit) wg_out[row - bx * 64] = v;
                }
                }
            }
        }
        nres = 0;
    };
    int gi = 0, it = 0;                                          // counters of the step being COMPUTED
    // One outer iteration = DEPTH steps.  REFILL 0: every slot is refilled (its step s + dslot + DEPTH exists: the main loop); 1: a slot is refilled if its step exists (the
    // iteration the row list ends in); 2: no refills (the last DEPTH steps).  Rounds 1-5 refilled unconditionally -- steps past the end re-read one cached line so that the
    // waits stay counted -- which made 22 % (fused up*gate: 7 steps per wave, ring of 2) to 50 % (attn_output: 4 steps, ring of 4) of a wave's load instructions dummies, each
    // returning 1 KiB through the CU's vector-memory path.  Three instances of the loop body keep the counts exact where they matter: the main loop's refills are all real, and
    // the conditional refills of the second instance only make the waits of the LAST instance conservative, where the ring drains anyway.
    auto outer = [&](const int s, auto refill_mode) __attribute__((always_inline)) {
        constexpr int REFILL = decltype(refill_mode)::value;
#pragma unroll
        for (int dslot = 0; dslot < DEPTH; ++dslot) {
            if (s + dslot < nsteps) {
                const int grp = grp_of(gi);
                const int u = it * lpr + u0;
                const float rpre = RES ? rring[dslot] : 0.f;        // (the residual of this step's row, loaded with its weights)
                if (u < U) {
#ifdef GEMV_EXP_NO_COMPUTE
                    acc[0][0] += __uint_as_float(ring[dslot][0].checksum());
#else
                    YReg ystep;                                  // activations of this K-slice: shared by the NR rows (and by up / gate)
                    if (YITERS == 0 && NCOLS == 1) Unit<TYPE>::template load_y<VDT>(u, K, 0, yq, yd, ys, ystep);
#pragma unroll
                    for (int r = 0; r < NR; ++r) {
                        typename Unit<TYPE>::Dec dc, dc2;
                        ring[dslot][r].decode(u, grid_lds, dc); if (UPGATE) ring2[dslot][r].decode(u, grid_lds, dc2);
#pragma unroll
                        for (int c = 0; c < NCOLS; ++c) {
                            if (YITERS > 0) {
                                const YReg &y = yreg[YITERS > 0 ? dslot % (YITERS > 0 ? YITERS : 1) : 0][YITERS > 0 ? c : 0];
                                acc[r][c] = Unit<TYPE>::dot(dc, y, acc[r][c]); if (UPGATE) acc2[r][c] = Unit<TYPE>::dot(dc2, y, acc2[r][c]);
                            } else if (NCOLS == 1) {
                                acc[r][c] = Unit<TYPE>::dot(dc, ystep, acc[r][c]); if (UPGATE) acc2[r][c] = Unit<TYPE>::dot(dc2, ystep, acc2[r][c]);
                            } else {
                                YReg y; Unit<TYPE>::template load_y<VDT>(u, K, c, yq, yd, ys, y);
                                acc[r][c] = Unit<TYPE>::dot(dc, y, acc[r][c]); if (UPGATE) acc2[r][c] = Unit<TYPE>::dot(dc2, y, acc2[r][c]);
                            }
                        }
                    }
#endif
                }
                if (++it == iters) {                             // row group finished: reduce over the lpr lanes, park the sums
                    it = 0;
                    if (nres == 0) res_gi0 = gi;
                    ++gi;
                    if (!park) {                                 // (NR == 1, not fused): reduce and store right away
                        const int row = grp * rpi + sub;
                        const uint8_t *Ap; float *Cp; int lrow; locate(row < a.M ? row : 0, Ap, Cp, lrow);
#pragma unroll
                        for (int c = 0; c < NCOLS; ++c) {
                            float v = dpp_row_sum(acc[0][c], lpr);
                            if (RES) v += rpre;                    // (FX = 2: single column)
                            if (u0 == lpr - 1 && row < a.M) Cp[(long)c * a.stride_C + lrow] = v;
                            acc[0][c] = 0.f;
                        }
                    } else
#pragma unroll
                    for (int r = 0; r < NR; ++r) {
                        const int j = lane - nres;                   // this lane takes the sum of sub-row j (source lane j * lpr + lpr - 1)
                        const bool take = j >= 0 && j < rpi;
#pragma unroll
                        for (int c = 0; c < NCOLS; ++c) {
                            const float v = dpp_row_sum(acc[r][c], lpr), v2 = UPGATE ? dpp_row_sum(acc2[r][c], lpr) : 0.f;
                            float g1, g2 = 0.f;
                            if (lpr == 64) { g1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63)); if (UPGATE) g2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v2), 63)); }
                            else { const int src = (take ? j : 0) * lpr + lpr - 1; g1 = __shfl(v, src, 64); if (UPGATE) g2 = __shfl(v2, src, 64); }
                            res[c] = take ? g1 : res[c]; if (UPGATE) res2[c] = take ? g2 : res2[c];
                            acc[r][c] = 0.f; acc2[r][c] = 0.f;
                        }
                        nres += rpi;
                    }
                }
            }
            // refill this slot with step s + dslot + DEPTH
            if constexpr (REFILL == 0) issue(ring[dslot], ring2[dslot], rring[dslot]);
            else if constexpr (REFILL == 1) { if (s + dslot + DEPTH < nsteps) issue(ring[dslot], ring2[dslot], rring[dslot]); }
        }
        if (nres + DEPTH * NR * rpi > 64) flush();               // (at most DEPTH * NR * rpi <= 32 new sums per outer iteration)
    };
    {
        int s = 0;
#ifdef GEMV_EXP_DUMMY_REFILLS          /* A/B build: the unconditional refills of rounds 1-5 */
        constexpr bool SPLIT_TAIL = false;
#else
        constexpr bool SPLIT_TAIL = NCOLS == 1;       // (the 2 ... 4-column kernels keep the single loop with dummy refills: three instances of their body would only grow the library)
#endif
        if constexpr (SPLIT_TAIL) {
            for (; s + 2 * DEPTH <= nsteps; s += DEPTH) outer(s, std::integral_constant<int, 0>());
            if (s < nsteps) { outer(s, std::integral_constant<int, 1>()); s += DEPTH; }
            if (s < nsteps) outer(s, std::integral_constant<int, 2>());
        } else {
            for (; s < nsteps; s += DEPTH) outer(s, std::integral_constant<int, 0>());
        }
    }
    flush();
    if (emit) {      // quantize the workgroup's 64 finished rows exactly like quantize_row_q8_2_x4 (iqk_quantize.cpp:1072-1166) would from the f32 row
        __syncthreads();
        if (wave == 0) {
            const int row = bx * 64 + lane;                  // two 32-blocks: lanes 0..31, 32..63
            const float v = wg_out[lane];
            const float amax = group_max<32>(fabsf(v));
            const uint32_t db = float_to_bf16_bits(amax / 127.f);
            const float d = bf16_bits_to_float(db), id = d > 0 ? 1.f / d : 0.f;
            const int qv = (int)rintf(v * id);
            const int isum = group_sum<32>(qv);              // (sum before saturation, as the reference)
            const int b = row >> 5, ir = b & 3;
            uint8_t *blk = a.q8_out + (long)(b >> 2) * 144;
            if ((lane & 31) == 0) { *reinterpret_cast<uint16_t *>(blk + 2 * ir) = (uint16_t)db; *reinterpret_cast<int16_t *>(blk + 8 + 2 * ir) = (int16_t)isum; }
            blk[16 + 32 * ir + (lane & 31)] = (uint8_t)(clamp_i8(qv) & 255);
        }
    }
#ifdef GEMV_EXP_TIMELINE
    __syncthreads();
#endif
    TL_STAMP(3);
#undef TL_STAMP
}

// (round 5: requesting the kernel arguments of a dense launch at the entry -- one batch of scalar loads instead of the four dependent kernarg round trips hipcc emits -- took 0.25 us
//  off the decode attention but does nothing here: llama-bench tg128 567 vs 565-572 tok/s, interleaved; the weight ring is requested before the late fields are needed)
#ifndef GEMV_MAX_THREADS
#define GEMV_MAX_THREADS 512      // (experiment builds: 768 = 12 waves per workgroup, scripts/iq_exp.py)
#endif
template <int TYPE, int NCOLS, bool UPGATE, int YITERS, int VDT, int DEPTH = GEMV_DEPTH, bool MULTI = false, int NR = 1, int LPR = 0, int FX = 0>
__global__ void __launch_bounds__(GEMV_MAX_THREADS) gemv_kernel(const GemvArgs a) {
    gemv_body<TYPE, NCOLS, UPGATE, YITERS, VDT, DEPTH, MULTI, NR, LPR, FX>(a, blockIdx.x, gridDim.x);
}

// Two differently-typed groups of matrices sharing the activations in ONE launch (Q4_K_M / Q5_K_M layers: q,k in Q4_K / Q5_K next to a
// Q6_K attn_v): workgroups [0, split) run group A, the rest group B.  A second launch would cost ~5 us for a 3 MB matrix.
template <int TYPE_A, int VDT_A, bool MULTI_A, int TYPE_B, int VDT_B, int YITERS, int LPR = 0, int FX = 0>
__global__ void __launch_bounds__(512) gemv_dual_kernel(const GemvArgs a, const GemvArgs b, const int split) {
    if ((int)blockIdx.x < split) gemv_body<TYPE_A, 1, false, YITERS, VDT_A, GEMV_DEPTH, MULTI_A, 1, LPR, FX>(a, blockIdx.x, split);
    else                         gemv_body<TYPE_B, 1, false, YITERS, VDT_B, GEMV_DEPTH, false, 1, LPR, FX>(b, blockIdx.x - split, gridDim.x - split);
}

// ---- long rows, slice-major -------------------------------------------------------------------------------
// ffn_down of a 14336-wide FFN: one row is 2..4 K-slices of 4096 weights (64 lanes x 64).  gemv_body quantizes the WHOLE activation
// vector (4 chunks per thread, ~2 us of VALU) before the first weight is consumed; the ring is full long before that and HBM idles.
// Here every wave owns exactly two rows and walks them slice by slice: (slice 0: row a, row b), (slice 1: row a, row b), ...; slice i of
// the activations is quantized right before its two steps, so the first weights are consumed after a quarter of the quantize work and
// the ring refills run under the rest of it.  Chunk p of a 512-thread workgroup (8 floats x 512) IS slice p, so the pre-loaded
// f32 chunks are used as they are.  Host side: M == 2 x waves of the grid, f32 activations, one column, 64 < K / 64 <= 256.
template <int TYPE, int VDT, int NW, int RD, int FX = 0>
__global__ void __launch_bounds__(64 * NW) gemv_sliced_kernel(const GemvArgs a) {
    // NW = 8: two rows per wave, chunk p == slice p.   NW = 16: one row per wave (twice the waves per SIMD to hide the waits; same
    // quantize work per SIMD), chunk p == slices 2p, 2p + 1.  16 rows per workgroup either way.  RD = ring depth in steps.
    // Measured at 4096 x 14336 Q4_K: (NW, RD) = (8, 4) 9.7 us, (16, 4) 10.6 us, (8, 8: the wave's whole share up front) 11.7 us.
    constexpr int ROWS = 16 / NW, SPC = NW / 8, NT = 64 * NW;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int K = a.K, k8 = K >> 3;
    int8_t *yq = reinterpret_cast<int8_t *>(smem);
    float  *yd = reinterpret_cast<float *>(smem + (size_t)K);
    float  *ys = yd + (size_t)(K / act_scale_block<VDT>());
    constexpr size_t TAB_ALIGN = type_has_tables(TYPE) ? 4095 : 15;
    const size_t grid_off = (((size_t)K + (size_t)(K / act_scale_block<VDT>()) * 4 + (act_has_sums<VDT>() ? (size_t)(K / 32) * 4 : 0)) + TAB_ALIGN) & ~TAB_ALIGN;
    uint8_t *grid_lds = smem + grid_off;

    const int U = K >> 6, iters = (U + 63) >> 6;             // 2..4 slices
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int W = gridDim.x * NW, r0 = blockIdx.x * NW + wave;  // rows r0 (and r0 + W)
    const int stride32 = (int)a.strideA;
    const uint8_t *rowp[2] = { a.A[0] + (long)r0 * stride32, a.A[0] + (long)(r0 + (ROWS - 1) * W) * stride32 };

    IqPre<TYPE> iqpre; iq_preload<TYPE>(a.tables, iqpre);
    XChunks xc;
    preload_activations_f32<1>(a, a.B, xc);
    Unit<TYPE> ring[RD];
    // step s = (slice s / ROWS, row s % ROWS); steps past the end re-read unit 0 (loads stay unconditional: exact vmcnt)
    auto issue = [&](Unit<TYPE> &w, const int s) {
        int u = (s / ROWS) * 64 + lane; if (!(s < ROWS * iters && u < U)) u = 0;
        w.load(rowp[s % ROWS], u);
    };
#pragma unroll
    for (int s = 0; s < RD; ++s) issue(ring[s], s);
    // FX = 2: the residual values of the wave's rows are requested HERE, behind the first ring loads: read right before the stores they cost two dependent memory round trips at
    // the end of every wave (the second load waited for the first STORE: vmcnt counts both)
    float rres[ROWS];
#pragma unroll
    for (int g = 0; g < ROWS; ++g) rres[g] = (FX == 2 && lane == 63) ? a.R[r0 + g * W] : 0.f;
    __builtin_amdgcn_sched_barrier(0);
    iq_fill_lds<TYPE>(iqpre, grid_lds);
    float acc[ROWS];
#pragma unroll
    for (int g = 0; g < ROWS; ++g) acc[g] = 0.f;
#pragma unroll
    for (int c = 0; c < 4 / SPC; ++c) {
        if (c * SPC < iters) {
            const int i = threadIdx.x + c * NT;
            if (i < k8) quantize_chunk<VDT>(xc.v[c][0], xc.v[c][1], K, 0, i, yq, yd, ys);
            __syncthreads();
#pragma unroll
            for (int sl = 0; sl < SPC; ++sl) {
                const int it = c * SPC + sl;
                if (it < iters) {
                    const int u = it * 64 + lane;
                    YReg y;
                    if (u < U) Unit<TYPE>::template load_y<VDT>(u, K, 0, yq, yd, ys, y);
#pragma unroll
                    for (int g = 0; g < ROWS; ++g) {
                        const int slot = (ROWS * it + g) % RD;
                        if (u < U) {
                            typename Unit<TYPE>::Dec dc;
                            ring[slot].decode(u, grid_lds, dc);
                            acc[g] = Unit<TYPE>::dot(dc, y, acc[g]);
                        }
                        if (ROWS * 4 > RD) issue(ring[slot], ROWS * it + g + RD);   // (not when the ring holds the wave's whole share)
                    }
                }
            }
        }
    }
#pragma unroll
    for (int g = 0; g < ROWS; ++g) {
        float v = dpp_row_sum(acc[g], 64);
        if (FX == 2) v += rres[g];
        if (lane == 63) a.C[0][r0 + g * W] = v;
    }
}
