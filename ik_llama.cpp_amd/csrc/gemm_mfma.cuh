// gemm_mfma.cuh -- prompt-processing (Ny > 8) dequant -> f16 MFMA GEMM for gfx950.
//
// What it replaces: the reference's repack-then-int8-GEMM prompt path (iqk_mul_mat.cpp:537-571 iqk_convert_repack +
// mul_mat_q8_1_r8_q8_2 / mul_mat_q8_k_r8_q8_k) and ggml-cuda's mul_mat_q (mmq.cuh:3938).  Per north_star the
// contraction runs on MFMA f16 tiles with f32 accumulate; weights are de-quantized exactly like the L0 dequantizer
// (f32 fma of the integer quant with its f32 scale / min) and rounded ONCE to f16; activations are rounded once to f16.
//
// MI355X mapping (DESIGN.md "prefill GEMM"):
//   * MFMA v_mfma_f32_32x32x16_f16 with A = activations (32 tokens x 16 k), B = weights (16 k x 32 rows): the 32 result
//     columns of a lane group are 32 consecutive weight rows => 128-byte coalesced C stores.
//   * A wave owns 32 weight rows x (32*NT) tokens.  Its B fragments never touch LDS: lane (row = lane&31, h = lane>>5)
//     loads that row's raw quant bytes straight from HBM (read exactly once per token tile) and de-quantizes 8
//     consecutive weights into its B registers; every B fragment is reused for NT token tiles, so the dequant VALU work
//     (~20 ops / fragment) hides under NT x 32-cycle MFMAs.
//   * MFMA does not care WHICH 16 k-indices form a k-step as long as A and B agree, so the k order inside a 128-wide
//     K tile is permuted per type to whatever makes a lane's 8 weights come out of one 8-byte piece of its quant data
//     (kpiece()).  No cross-lane shuffles.
//   * The activation tile (32*NT tokens x 128 k, f16) is shared by the 4 waves through LDS, filled by LDS-DMA
//     (global_load_lds_dwordx4, no VGPR round trip), double buffered, one barrier per K tile.  ds_read_b128 of a lane
//     group hits 16 distinct rows: 16-byte pieces are XOR-swizzled with (row & 15) -> conflict free; the swizzle is
//     applied to the DMA *source* address (LDS side of the DMA is lane-linear) and to the read.
#pragma once
#include <cstring>
#include "cdna4_common.cuh"
#include "api_internal.h"
#include "gemv.cuh"      // expand_iq2s_grid / expand_iq3s_grid, sign_mask4 / apply_sign4

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

#define GEMM_MAX_MATS 4
struct GemmArgs {
    const uint8_t *A, *A2;     // weights (A2 = gate for fused up*gate)
    // several matrices of the same type sharing the activations (q,k,v): rows are concatenated, matrix i covers [mend[i-1], mend[i])
    const uint8_t *Am[GEMM_MAX_MATS]; float *Cm[GEMM_MAX_MATS]; int mend[GEMM_MAX_MATS]; int nmat;
    long stride_Cm[GEMM_MAX_MATS];          // (nmat > 1) result row stride of every matrix: q [4096 x n] and k [1024 x n] go out in one launch
    const __half  *X;          // activations f16 in the slab layout X16[K / 64][xrows][64] (convert.cuh)
    const float   *xscale;     // per activation row: the power of two it was divided by before the f16 rounding (range guard, convert.cuh); nullptr = all 1
    long xrows;                // rows per slab (>= every row a tile can touch; rows past the data are zero)
    float         *C;
    const uint16_t *grid;
    long strideA, stride_C;
    int  M, N, K;
    int  unary_op;
    UpGateEpilogue epi;        // fused up*gate biases / limit
    // grouped (MUL_MAT_ID) form: token tiles are (expert, row range) pairs produced on the device by moe_sort_kernel
    const int *moe_tiles;      // [max_tiles][3] = {expert, first row in the sorted activation matrix, valid rows}; expert < 0 => unused tile
    const int *moe_pairs;      // sorted position -> (token * n_used + slot)
    long expert_stride;        // bytes between experts (nb02)
    long nb1, nb2;             // result strides in elements: slot, token
    int  n_used;
    int  moe_rb;               // grouped launches: bands of weight-row tiles over the XCDs (1, 2, 4 or 8; set by launch_gemm_ks)
    int  expert_lo, expert_hi; // (expert_hi > 0) the weight buffer holds experts [expert_lo, expert_hi) only -- the f16 route of the decode-only types de-quantizes them in chunks; other tiles exit
    int  m_major;              // token tiles per super-column of the tile order (see kernel); set by launch_gemm_ks
    // split-K launches (gridDim.z > 1): every K slice stores its partial tile to ks_ws[z][N][M]; the workgroup that arrives LAST at the tile's counter adds the slices in
    // slice order and writes C -- deterministic (the round-1/2 kernels accumulated with f32 atomics into a zero-filled C: run-to-run different prompts), no zero-fill
    float *ks_ws; size_t ks_ws_bytes; unsigned *ks_cnt;
    int ks_fence;              // the slab hand-off with an agent-scope release before the ticket and an acquire behind it (cdna4_context::handoff >= 1: requested, or the start-up self-test failed)
    int no_ksplit;             // (self-test reference) never split K over grid.z
    int pairing;               // gemm_ppf_kernel: which 16-byte pieces of a 64-wide stage k-step j contracts, 3 bits per (j, half) -- the weight type's kpiece() / HBIT (gemm_ppf.cuh)
};

// The 8 k-values of a fragment are ordered (0,2,1,3,4,6,5,7): the f16 activations are stored in that order (convert.cuh,
// x16_slab_index), which lets the packed de-quantizer below produce the pairs (k0,k2), (k1,k3) of a dword with one and_or each.
// f0..f7 are the weights of k = 0..7 in natural order.
__device__ __forceinline__ half8 pack8(float f0, float f1, float f2, float f3, float f4, float f5, float f6, float f7) {
    half8 r; r[0] = (_Float16)f0; r[1] = (_Float16)f2; r[2] = (_Float16)f1; r[3] = (_Float16)f3;
    r[4] = (_Float16)f4; r[5] = (_Float16)f6; r[6] = (_Float16)f5; r[7] = (_Float16)f7; return r;
}
// ---- packed f16 de-quantization of 4-bit fields (Q4_K / Q5_K) ------------------------------------------------------------------
// (w & 0x000f000f) | 0x64006400 is the f16 pair (1024 + q(byte 0), 1024 + q(byte 2)) -- the field sits in the low mantissa bits of
// 1024.0 -- and with mask 0x00f000f0 it is (1024 + 16 q_hi(byte 0), ...).  fma(x, S, -1024 S) = q * S with ONE rounding (-1024 S is
// exact), then the min term is added: 2 packed ops per 2 weights instead of cvt + fma + pack per weight.  S = f16(d * sc) (or / 16 for
// the high nibbles, exact), i.e. the scale is rounded to f16 BEFORE the product (2^-11 relative, as large as the f16 rounding of the
// activations); the f32 path rounds once after the product.  CDNA4_GEMM_DEQUANT_F32 restores the f32 path for comparisons.
__device__ __forceinline__ half2v pk_fma(half2v a, half2v b, half2v c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ half2v as_h2(uint32_t u) { union { uint32_t u; half2v h; } c; c.u = u; return c.h; }
__device__ __forceinline__ half2v h2_dup(float f) { half2v r; r[0] = (_Float16)f; r[1] = r[0]; return r; }
// 8 weights of two dwords (b0: k 0..3, b1: k 4..7), field selected by `mask` (0x000f000f low nibbles, 0x00f000f0 high nibbles)
// (b & mask) | magic as ONE v_and_or_b32: VOP3 takes no literals on gfx9 and only one SGPR, so hipcc splits it into v_and + v_or with two
// literals; with the mask in an SGPR and the magic number in a VGPR the fused form is encodable.
__device__ __forceinline__ uint32_t and_or_magic(uint32_t b, uint32_t mask, uint32_t magic_vgpr) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r; asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(b), "s"(mask), "v"(magic_vgpr)); return r;
#else
    return (b & mask) | magic_vgpr;
#endif
}
__device__ __forceinline__ half8 dequant8_pk(uint32_t b0, uint32_t b1, uint32_t mask, half2v S, half2v C, half2v M) {
    uint32_t magic = 0x64006400u;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+v"(magic));                      // keep it in a VGPR (not re-materialised as a literal)
#endif
    const half2v x0 = as_h2(and_or_magic(b0, mask, magic)), x1 = as_h2(and_or_magic(b0 >> 8, mask, magic));
    const half2v x2 = as_h2(and_or_magic(b1, mask, magic)), x3 = as_h2(and_or_magic(b1 >> 8, mask, magic));
    const half2v r0 = pk_fma(x0, S, C) + M, r1 = pk_fma(x1, S, C) + M, r2 = pk_fma(x2, S, C) + M, r3 = pk_fma(x3, S, C) + M;
    half8 r; r[0] = r0[0]; r[1] = r0[1]; r[2] = r1[0]; r[3] = r1[1]; r[4] = r2[0]; r[5] = r2[1]; r[6] = r3[0]; r[7] = r3[1]; return r;
}
// byte k of a dword -> f32 in ONE instruction (v_cvt_f32_ubyteK).  hipcc otherwise merges the nibble shift into the byte
// select and emits lshr + and + cvt_ubyte0 per element (measured: 8.1 VALU per MFMA instead of 5.8).
__device__ __forceinline__ float ubyte0(uint32_t b) { float f; asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(f) : "v"(b)); return f; }
__device__ __forceinline__ float ubyte1(uint32_t b) { float f; asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(f) : "v"(b)); return f; }
__device__ __forceinline__ float ubyte2(uint32_t b) { float f; asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(f) : "v"(b)); return f; }
__device__ __forceinline__ float ubyte3(uint32_t b) { float f; asm("v_cvt_f32_ubyte3 %0, %1" : "=v"(f) : "v"(b)); return f; }
// 4 bytes of `b` (each 0..255) -> a*q + c for bytes 0..3
__device__ __forceinline__ void fma4_ubytes(uint32_t b, float a, float c, float &f0, float &f1, float &f2, float &f3) {
    f0 = fmaf(a, ubyte0(b), c); f1 = fmaf(a, ubyte1(b), c); f2 = fmaf(a, ubyte2(b), c); f3 = fmaf(a, ubyte3(b), c);
}

// ---- per-type weight tiles: one lane's share (row, half h) of a 128-element K tile -----------------------
// interface:  load(row_ptr, kt, h)   issue the global loads
//             prepare(h)             decode scales (once per tile)
//             frag(s, h)             B fragment (8 f16) of k-step s (0..7)
//             kpiece(s)              16-byte piece index (k/8 inside the tile) supplied by half h=0 at step s; half 1 adds HBIT
template <int TYPE> struct WTile;

template <> struct WTile<T_Q4_K> {
    static constexpr int HBIT = 2;                       // half h supplies pieces +2 (16 elements further)
    uint4 hdr, q[2];
#ifdef CDNA4_GEMM_DEQUANT_F32
    float dsc[4], dmn[4];
#else
    half2v S[4], C[4], M[4];                             // per sub-block: f16 scale (x 1/16 for high nibbles), -1024 * S, -dmin * m
#endif
    __device__ __forceinline__ void load(const uint8_t *row, int kt, int h) {
        const uint8_t *b = row + (long)(kt >> 1) * 144;
        hdr = *reinterpret_cast<const uint4 *>(b);
        q[0] = *reinterpret_cast<const uint4 *>(b + 16 + 64 * (kt & 1) + 16 * h);
        q[1] = *reinterpret_cast<const uint4 *>(b + 48 + 64 * (kt & 1) + 16 * h);
        n_ = kt & 1;
    }
    int n_;
    __device__ __forceinline__ void prepare(int, const void *) {
        const float d = half_bits_to_float(hdr.x & 0xffff), dmin = half_bits_to_float(hdr.x >> 16);
        uint32_t sc03, sc47, mn03, mn47; k4_unpack_scales(hdr.y, hdr.z, hdr.w, sc03, sc47, mn03, mn47);
        const uint32_t sc = n_ ? sc47 : sc03, mn = n_ ? mn47 : mn03;       // sub-blocks 4n .. 4n+3
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float ds = d * (float)((sc >> (8 * j)) & 0xff), dm = -(dmin * (float)((mn >> (8 * j)) & 0xff));
#ifdef CDNA4_GEMM_DEQUANT_F32
            dsc[j] = ds; dmn[j] = dm;
#else
            S[j] = h2_dup((j & 1) ? ds * 0.0625f : ds); C[j] = S[j] * (_Float16)(-1024.f); M[j] = h2_dup(dm);
#endif
        }
    }
    static __host__ __device__ __forceinline__ constexpr int kpiece(int s) { return 8 * (s >> 2) + 4 * ((s >> 1) & 1) + (s & 1); }
    __device__ __forceinline__ half8 frag(int s, int) const {
        const int gi = s >> 2, t = s & 3, j = 2 * gi + (t >> 1);            // sub-block within the tile (odd j = high nibbles)
        const uint4 &w = q[gi];
        uint32_t b0 = (t & 1) ? w.z : w.x, b1 = (t & 1) ? w.w : w.y;
#ifdef CDNA4_GEMM_DEQUANT_F32
        if (t & 2) { b0 >>= 4; b1 >>= 4; }
        b0 &= 0x0f0f0f0fu; b1 &= 0x0f0f0f0fu;
        float f[8]; fma4_ubytes(b0, dsc[j], dmn[j], f[0], f[1], f[2], f[3]); fma4_ubytes(b1, dsc[j], dmn[j], f[4], f[5], f[6], f[7]);
        return pack8(f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7]);
#else
        return dequant8_pk(b0, b1, (t & 2) ? 0x00f000f0u : 0x000f000fu, S[j], C[j], M[j]);
#endif
    }
};

template <> struct WTile<T_Q5_K> {
    static constexpr int HBIT = 2;
    uint4 hdr, q[2], qh; float dsc[4], dmn[4]; int n_;
    __device__ __forceinline__ void load(const uint8_t *row, int kt, int h) {
        const uint8_t *b = row + (long)(kt >> 1) * 176;
        hdr = *reinterpret_cast<const uint4 *>(b);
        qh = *reinterpret_cast<const uint4 *>(b + 16 + 16 * h);             // qh[16h .. 16h+15]: high bits of l = 16h + [0,16) for all 8 sub-blocks
        q[0] = *reinterpret_cast<const uint4 *>(b + 48 + 64 * (kt & 1) + 16 * h);
        q[1] = *reinterpret_cast<const uint4 *>(b + 80 + 64 * (kt & 1) + 16 * h);
        n_ = kt & 1;
    }
    __device__ __forceinline__ void prepare(int, const void *) {
        const float d = half_bits_to_float(hdr.x & 0xffff), dmin = half_bits_to_float(hdr.x >> 16);
        uint32_t sc03, sc47, mn03, mn47; k4_unpack_scales(hdr.y, hdr.z, hdr.w, sc03, sc47, mn03, mn47);
        const uint32_t sc = n_ ? sc47 : sc03, mn = n_ ? mn47 : mn03;
#pragma unroll
        for (int j = 0; j < 4; ++j) { dsc[j] = d * (float)((sc >> (8 * j)) & 0xff); dmn[j] = -(dmin * (float)((mn >> (8 * j)) & 0xff)); }
    }
    static __host__ __device__ __forceinline__ constexpr int kpiece(int s) { return 8 * (s >> 2) + 4 * ((s >> 1) & 1) + (s & 1); }
    __device__ __forceinline__ half8 frag(int s, int) const {
        const int gi = s >> 2, t = s & 3, j = 2 * gi + (t >> 1);
        const uint4 &w = q[gi];
        uint32_t b0 = (t & 1) ? w.z : w.x, b1 = (t & 1) ? w.w : w.y;
        uint32_t h0 = (t & 1) ? qh.z : qh.x, h1 = (t & 1) ? qh.w : qh.y;
        if (t & 2) { b0 >>= 4; b1 >>= 4; }
        const int bit = 4 * n_ + j;                                        // sub-block index in the super-block = qh bit
        b0 = (b0 & 0x0f0f0f0fu) | (((h0 >> bit) & 0x01010101u) << 4); b1 = (b1 & 0x0f0f0f0fu) | (((h1 >> bit) & 0x01010101u) << 4);
        float f[8]; fma4_ubytes(b0, dsc[j], dmn[j], f[0], f[1], f[2], f[3]); fma4_ubytes(b1, dsc[j], dmn[j], f[4], f[5], f[6], f[7]);
        return pack8(f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7]);
    }
};

#ifdef GEMM_Q6K_NARROW_LOADS          /* A/B build: the layout of rounds 1-5 -- eight 8-byte loads per lane and tile, half h = the odd 8-element pieces */
template <> struct WTile<T_Q6_K> {
    static constexpr int HBIT = 1;                       // half h supplies the next 8 elements
    uint2 la[2], lb[2], qh[2], sc; uint32_t dh; float ds[8];
    __device__ __forceinline__ void load(const uint8_t *row, int kt, int h) {
        const uint8_t *b = row + (long)(kt >> 1) * 210; const int n = kt & 1;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            la[c] = ld64(b + 64 * n + 16 * c + 8 * h); lb[c] = ld64(b + 64 * n + 32 + 16 * c + 8 * h); qh[c] = ld64(b + 128 + 32 * n + 16 * c + 8 * h);
        }
        sc = ld64(b + 192 + 8 * n); dh = ld16(b + 208);
    }
    __device__ __forceinline__ void prepare(int, const void *) {
        const float d = half_bits_to_float(dh);
#pragma unroll
        for (int i = 0; i < 4; ++i) { ds[i] = d * (float)(int)(int8_t)((sc.x >> (8 * i)) & 0xff); ds[4 + i] = d * (float)(int)(int8_t)((sc.y >> (8 * i)) & 0xff); }
    }
    // step s -> (j = 2 (s>>2) + (s&1), c = (s>>1)&1) : elements 32 j + 16 c + 8 h + [0,8); steps 0..3 cover k < 64
    static __host__ __device__ __forceinline__ constexpr int kpiece(int s) { return 8 * (s >> 2) + 4 * (s & 1) + 2 * ((s >> 1) & 1); }
    __device__ __forceinline__ half8 frag(int s, int) const {
        const int c = (s >> 1) & 1, j = 2 * (s >> 2) + (s & 1);
        const uint2 L = (j & 1) ? lb[c] : la[c]; const uint2 H = qh[c];
        uint32_t b0 = L.x, b1 = L.y; if (j & 2) { b0 >>= 4; b1 >>= 4; }
        b0 = (b0 & 0x0f0f0f0fu) | (((H.x >> (2 * j)) & 0x03030303u) << 4); b1 = (b1 & 0x0f0f0f0fu) | (((H.y >> (2 * j)) & 0x03030303u) << 4);
#ifdef CDNA4_GEMM_DEQUANT_F32
        const float a = ds[c + 2 * j], m32 = -32.f * a;                     // (d*sc)*(q-32) == fma(d*sc, q, -32*d*sc) exactly (q-32 is exact)
        float f[8]; fma4_ubytes(b0, a, m32, f[0], f[1], f[2], f[3]); fma4_ubytes(b1, a, m32, f[4], f[5], f[6], f[7]);
        return pack8(f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7]);
#else
        // packed f16: (byte & 0x00ff00ff) | 0x64006400 = (1024 + q(k0), 1024 + q(k2)); - 1056 is exact (q - 32), then ONE rounding in the
        // product with S = f16(d * sc) -- the same error model as the Q4_K packed path (scale rounded to f16 before the product)
        uint32_t magic = 0x64006400u;
#if defined(__HIP_DEVICE_COMPILE__)
        asm("" : "+v"(magic));
#endif
        const half2v S = h2_dup(ds[c + 2 * j]), off = h2_dup(-1056.f);
        const half2v r0 = (as_h2(and_or_magic(b0, 0x00ff00ffu, magic)) + off) * S, r1 = (as_h2(and_or_magic(b0 >> 8, 0x00ff00ffu, magic)) + off) * S;
        const half2v r2 = (as_h2(and_or_magic(b1, 0x00ff00ffu, magic)) + off) * S, r3 = (as_h2(and_or_magic(b1 >> 8, 0x00ff00ffu, magic)) + off) * S;
        half8 r; r[0] = r0[0]; r[1] = r0[1]; r[2] = r1[0]; r[3] = r1[1]; r[4] = r2[0]; r[5] = r2[1]; r[6] = r3[0]; r[7] = r3[1]; return r;
#endif
    }
};
#else
// Round 6: half h owns the 16-element halves of every 32-group (elements 32 j + 16 h + 8 c + [0,8)) instead of its odd / even 8-element pieces: a lane's share of a tile is THREE
// 16-byte loads (ql low half, ql high half, qh) + scales + d instead of six 8-byte loads -- 210-byte blocks are 2-byte aligned, every load of a wave touches 32 different rows, and
// the CU's address unit serves the weight loads of all eight waves (phase timeline of the 4096 x 14336 x 512 launch: 3.72 us per K tile against 2.44 for Q4_K with 19 % more VALU).
// The lane needs only the four 16-element scales of its own half.
template <> struct WTile<T_Q6_K> {
    static constexpr int HBIT = 2;                       // half h supplies pieces +2 (16 elements further)
    uint4 la, lb, qh; uint2 sc; uint32_t dh; float ds[4];
    __device__ __forceinline__ void load(const uint8_t *row, int kt, int h) {
        const uint8_t *b = row + (long)(kt >> 1) * 210; const int n = kt & 1;
        la = ld128(b + 64 * n + 16 * h); lb = ld128(b + 64 * n + 32 + 16 * h); qh = ld128(b + 128 + 32 * n + 16 * h);
        sc = ld64(b + 192 + 8 * n); dh = ld16(b + 208);
    }
    __device__ __forceinline__ void prepare(int h, const void *) {
        const float d = half_bits_to_float(dh);
        const uint32_t sx = sc.x >> (8 * h), sy = sc.y >> (8 * h);      // 16-element group 2 j + h of the tile
        ds[0] = d * (float)(int)(int8_t)(sx & 0xff); ds[1] = d * (float)(int)(int8_t)((sx >> 16) & 0xff);
        ds[2] = d * (float)(int)(int8_t)(sy & 0xff); ds[3] = d * (float)(int)(int8_t)((sy >> 16) & 0xff);
    }
    // step s -> (j = 2 (s>>2) + (s&1), c = (s>>1)&1) : elements 32 j + 16 h + 8 c + [0,8)
    static __host__ __device__ __forceinline__ constexpr int kpiece(int s) { return 8 * (s >> 2) + 4 * (s & 1) + ((s >> 1) & 1); }
    __device__ __forceinline__ half8 frag(int s, int) const {
        const int c = (s >> 1) & 1, j = 2 * (s >> 2) + (s & 1);
        const uint4 L = (j & 1) ? lb : la;
        uint32_t b0 = c ? L.z : L.x, b1 = c ? L.w : L.y; const uint32_t h0 = c ? qh.z : qh.x, h1 = c ? qh.w : qh.y;
        if (j & 2) { b0 >>= 4; b1 >>= 4; }
        b0 = (b0 & 0x0f0f0f0fu) | (((h0 >> (2 * j)) & 0x03030303u) << 4); b1 = (b1 & 0x0f0f0f0fu) | (((h1 >> (2 * j)) & 0x03030303u) << 4);
#ifdef CDNA4_GEMM_DEQUANT_F32
        const float a = ds[j], m32 = -32.f * a;
        float f[8]; fma4_ubytes(b0, a, m32, f[0], f[1], f[2], f[3]); fma4_ubytes(b1, a, m32, f[4], f[5], f[6], f[7]);
        return pack8(f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7]);
#else
        // packed f16: (byte & 0x00ff00ff) | 0x64006400 = (1024 + q(k0), 1024 + q(k2)); - 1056 is exact (q - 32), then ONE rounding in the
        // product with S = f16(d * sc) -- the same error model as the Q4_K packed path (scale rounded to f16 before the product)
        uint32_t magic = 0x64006400u;
#if defined(__HIP_DEVICE_COMPILE__)
        asm("" : "+v"(magic));
#endif
        const half2v S = h2_dup(ds[j]), off = h2_dup(-1056.f);
        const half2v r0 = (as_h2(and_or_magic(b0, 0x00ff00ffu, magic)) + off) * S, r1 = (as_h2(and_or_magic(b0 >> 8, 0x00ff00ffu, magic)) + off) * S;
        const half2v r2 = (as_h2(and_or_magic(b1, 0x00ff00ffu, magic)) + off) * S, r3 = (as_h2(and_or_magic(b1 >> 8, 0x00ff00ffu, magic)) + off) * S;
        half8 r; r[0] = r0[0]; r[1] = r0[1]; r[2] = r1[0]; r[3] = r1[1]; r[4] = r2[0]; r[5] = r2[1]; r[6] = r3[0]; r[7] = r3[1]; return r;
#endif
    }
};
#endif

template <int NT4> struct WTileNib {        // IQ4_NL (codebook) / Q4_0 (nibble - 8): 18-byte blocks, four per 128-wide K tile
    static constexpr int HBIT = 1;
    uint2 q[4]; float d[4];
    uint32_t dh[4];
    __device__ __forceinline__ void load(const uint8_t *row, int kt, int h) {
        const uint8_t *b = row + (long)kt * 72;
#pragma unroll
        for (int i = 0; i < 4; ++i) { dh[i] = ld16(b + 18 * i); q[i] = ld64(b + 18 * i + 2 + 8 * h); }
    }
    __device__ __forceinline__ void prepare(int, const void *) {
#pragma unroll
        for (int i = 0; i < 4; ++i) d[i] = half_bits_to_float(dh[i]);
    }
    // step s = 2 b + hi : elements 32 b + 16 hi + 8 h + [0,8)
    static __host__ __device__ __forceinline__ constexpr int kpiece(int s) { return 4 * (s >> 1) + 2 * (s & 1); }
    __device__ __forceinline__ half8 frag(int s, int) const {
        const int b = s >> 1, hi = s & 1;
        uint32_t n0 = q[b].x, n1 = q[b].y; if (hi) { n0 >>= 4; n1 >>= 4; }
        const uint32_t v0 = nib4_to_i8<NT4>(n0 & 0x0f0f0f0fu), v1 = nib4_to_i8<NT4>(n1 & 0x0f0f0f0fu);
        const float a = d[b];
        return pack8(a * (float)(int)(int8_t)(v0 & 0xff), a * (float)(int)(int8_t)((v0 >> 8) & 0xff), a * (float)(int)(int8_t)((v0 >> 16) & 0xff), a * (float)((int)v0 >> 24),
                     a * (float)(int)(int8_t)(v1 & 0xff), a * (float)(int)(int8_t)((v1 >> 8) & 0xff), a * (float)(int)(int8_t)((v1 >> 16) & 0xff), a * (float)((int)v1 >> 24));
    }
};

template <> struct WTile<T_IQ4_NL> : WTileNib<T_IQ4_NL> {};

template <> struct WTile<T_Q4_0> : WTileNib<T_Q4_0> {};

// Q8_0: four 34-byte blocks {f16 d; i8 qs[32]} per K tile; half h owns bytes 16 hi + 8 h + [0, 8) of every block (same element -> step map as the nibble types)
template <> struct WTile<T_Q8_0> {
    static constexpr int HBIT = 1;
    uint2 q[4][2]; float d[4];
    uint32_t dh[4];
    __device__ __forceinline__ void load(const uint8_t *row, int kt, int h) {
        const uint8_t *b = row + (long)kt * 136;
#pragma unroll
        for (int i = 0; i < 4; ++i) { dh[i] = ld16(b + 34 * i); q[i][0] = ld64(b + 34 * i + 2 + 8 * h); q[i][1] = ld64(b + 34 * i + 18 + 8 * h); }
    }
    __device__ __forceinline__ void prepare(int, const void *) {
#pragma unroll
        for (int i = 0; i < 4; ++i) d[i] = half_bits_to_float(dh[i]);
    }
    static __host__ __device__ __forceinline__ constexpr int kpiece(int s) { return 4 * (s >> 1) + 2 * (s & 1); }
    __device__ __forceinline__ half8 frag(int s, int) const {
        const int b = s >> 1, hi = s & 1;
        const uint32_t v0 = q[b][hi].x, v1 = q[b][hi].y; const float a = d[b];
        return pack8(a * (float)(int)(int8_t)(v0 & 0xff), a * (float)(int)(int8_t)((v0 >> 8) & 0xff), a * (float)(int)(int8_t)((v0 >> 16) & 0xff), a * (float)((int)v0 >> 24),
                     a * (float)(int)(int8_t)(v1 & 0xff), a * (float)(int)(int8_t)((v1 >> 8) & 0xff), a * (float)(int)(int8_t)((v1 >> 16) & 0xff), a * (float)((int)v1 >> 24));
    }
};

// f16 weights (row-major [M][K]): the tile is 256 bytes of the row; half h owns the 16-byte pieces 2 s + h.  Serves every quant type that has no tile of its own
// (cdna4_api.hip de-quantizes a row chunk into the workspace first -- the reference's own route for large batches: convert.cu + cuBLAS, ggml-cuda.cu:1723)
template <> struct WTile<T_F16> {
    static constexpr int HBIT = 1;
    uint4 q[8];
    __device__ __forceinline__ void load(const uint8_t *row, int kt, int h) {
        const uint8_t *b = row + (long)kt * 256 + 16 * h;
#pragma unroll
        for (int s = 0; s < 8; ++s) q[s] = ld128(b + 32 * s);
    }
    __device__ __forceinline__ void prepare(int, const void *) {}
    static __host__ __device__ __forceinline__ constexpr int kpiece(int s) { return 2 * s; }
    __device__ __forceinline__ half8 frag(int s, int) const {          // fragment order (0,2,1,3,4,6,5,7), as pack8
        union { uint32_t u[4]; half8 h; } c;
        c.u[0] = __builtin_amdgcn_perm(q[s].y, q[s].x, 0x05040100u); c.u[1] = __builtin_amdgcn_perm(q[s].y, q[s].x, 0x07060302u);
        c.u[2] = __builtin_amdgcn_perm(q[s].w, q[s].z, 0x05040100u); c.u[3] = __builtin_amdgcn_perm(q[s].w, q[s].z, 0x07060302u);
        return c.h;
    }
};

// IQ4_XS: K tile kt = half n = kt & 1 of super-block kt >> 1: its four sub-blocks ib = 4 n + i in the IQ4_NL nibble layout, scale d (ls_ib - 32)
template <> struct WTile<T_IQ4_XS> {
    static constexpr int HBIT = 1;
    uint2 q[4]; float d[4]; uint32_t hdr0, hdr1; int n4;
    __device__ __forceinline__ void load(const uint8_t *row, int kt, int h) {
        const uint8_t *b = row + (long)(kt >> 1) * 136; n4 = 4 * (kt & 1);
        const uint2 hd = ld64(b); hdr0 = hd.x; hdr1 = hd.y;
#pragma unroll
        for (int i = 0; i < 4; ++i) q[i] = ld64(b + 8 + 16 * (n4 + i) + 8 * h);
    }
    __device__ __forceinline__ void prepare(int, const void *) {
        const float dd = half_bits_to_float(hdr0 & 0xffff); const uint32_t sh = hdr0 >> 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) { const int ib = n4 + i; const int ls = (int)(((hdr1 >> (4 * ib)) & 0xf) | (((sh >> (2 * ib)) & 3) << 4)) - 32; d[i] = dd * (float)ls; }
    }
    static __host__ __device__ __forceinline__ constexpr int kpiece(int s) { return 4 * (s >> 1) + 2 * (s & 1); }
    __device__ __forceinline__ half8 frag(int s, int) const {
        const int b = s >> 1, hi = s & 1;
        uint32_t n0 = q[b].x, n1 = q[b].y; if (hi) { n0 >>= 4; n1 >>= 4; }
        const uint32_t v0 = iq4nl_lookup4(n0 & 0x0f0f0f0fu), v1 = iq4nl_lookup4(n1 & 0x0f0f0f0fu);
        const float a = d[b];
        return pack8(a * (float)(int)(int8_t)(v0 & 0xff), a * (float)(int)(int8_t)((v0 >> 8) & 0xff), a * (float)(int)(int8_t)((v0 >> 16) & 0xff), a * (float)((int)v0 >> 24),
                     a * (float)(int)(int8_t)(v1 & 0xff), a * (float)(int)(int8_t)((v1 >> 8) & 0xff), a * (float)(int)(int8_t)((v1 >> 16) & 0xff), a * (float)((int)v1 >> 24));
    }
};

// signed int8 x 4 (one dword) -> a * v for bytes 0..3
__device__ __forceinline__ void mul4_sbytes(uint32_t v, float a, float &f0, float &f1, float &f2, float &f3) {
    f0 = a * (float)(int)(int8_t)(v & 0xff); f1 = a * (float)(int)(int8_t)((v >> 8) & 0xff);
    f2 = a * (float)(int)(int8_t)((v >> 16) & 0xff); f3 = a * (float)((int)v >> 24);
}
__device__ __forceinline__ half8 frag_sbytes(uint32_t v0, uint32_t v1, float a) {
    float f[8]; mul4_sbytes(v0, a, f[0], f[1], f[2], f[3]); mul4_sbytes(v1, a, f[4], f[5], f[6], f[7]);
    return pack8(f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7]);
}
// 4 bits of `bits` (bit i -> byte i, value 0 / 1)
__device__ __forceinline__ uint32_t spread4(uint32_t bits) { return ((bits & 0xfu) * 0x00204081u) & 0x01010101u; }

// The other legacy 32-block types -- Q5_0 {f16 d; u32 qh; u8 qs[16]}, Q4_1 {f16 d, m; qs}, Q5_1 {f16 d, m; u32 qh; qs}, Q6_0 {f16 d; u8 qh[8]; qs}: four blocks per K tile, the
// nibble layout and step map of WTileNib; the fifth (and sixth) bits come from qh, Q4_1 / Q5_1 are q d + m (one fma, as the L0 value), Q5_0 / Q6_0 (q - 16 | 32) d
template <int TYPE> struct WTileLeg {
    static constexpr int HBIT = 1;
    static constexpr int BB = type_block_bytes(TYPE), QO = TYPE == T_Q5_0 ? 6 : TYPE == T_Q4_1 ? 4 : TYPE == T_Q5_1 ? 8 : 10;
    static constexpr bool HASM = TYPE == T_Q4_1 || TYPE == T_Q5_1;
    uint2 q[4]; uint32_t hd[4], qh[4]; uint2 qh6[4]; float d[4], m[4];
    __device__ __forceinline__ void load(const uint8_t *row, int kt, int h) {
        const uint8_t *b = row + (long)kt * (4 * BB);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            hd[i] = HASM ? ld32(b + BB * i) : ld16(b + BB * i); q[i] = ld64(b + BB * i + QO + 8 * h);
            if (TYPE == T_Q5_0) qh[i] = ld32(b + BB * i + 2);
            if (TYPE == T_Q5_1) qh[i] = ld32(b + BB * i + 4);
            if (TYPE == T_Q6_0) qh6[i] = ld64(b + BB * i + 2);
        }
    }
    __device__ __forceinline__ void prepare(int, const void *) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { d[i] = half_bits_to_float(hd[i] & 0xffff); m[i] = HASM ? half_bits_to_float(hd[i] >> 16) : 0.f; }
    }
    static __host__ __device__ __forceinline__ constexpr int kpiece(int s) { return 4 * (s >> 1) + 2 * (s & 1); }
    __device__ __forceinline__ half8 frag(int s, int h) const {          // elements 32 b + 16 hi + 8 h + [0, 8)
        const int b = s >> 1, hi = s & 1;
        uint32_t n0 = q[b].x, n1 = q[b].y; if (hi) { n0 >>= 4; n1 >>= 4; }
        n0 &= 0x0f0f0f0fu; n1 &= 0x0f0f0f0fu;
        if (TYPE == T_Q5_0 || TYPE == T_Q5_1) { const uint32_t hb = qh[b] >> (16 * hi + 8 * h); n0 |= spread4(hb) << 4; n1 |= spread4(hb >> 4) << 4; }
        if (TYPE == T_Q6_0) { const int sh = 4 * h + 2 * hi; n0 |= ((qh6[b].x >> sh) & 0x03030303u) << 4; n1 |= ((qh6[b].y >> sh) & 0x03030303u) << 4; }
        const float a = d[b];
        if (HASM) {
            const float c = m[b];
            return pack8(fmaf(ubyte0(n0), a, c), fmaf(ubyte1(n0), a, c), fmaf(ubyte2(n0), a, c), fmaf(ubyte3(n0), a, c), fmaf(ubyte0(n1), a, c), fmaf(ubyte1(n1), a, c), fmaf(ubyte2(n1), a, c), fmaf(ubyte3(n1), a, c));
        }
        const float o = TYPE == T_Q5_0 ? 16.f : 32.f;
        return pack8((ubyte0(n0) - o) * a, (ubyte1(n0) - o) * a, (ubyte2(n0) - o) * a, (ubyte3(n0) - o) * a, (ubyte0(n1) - o) * a, (ubyte1(n1) - o) * a, (ubyte2(n1) - o) * a, (ubyte3(n1) - o) * a);
    }
};
template <> struct WTile<T_Q5_0> : WTileLeg<T_Q5_0> {};
template <> struct WTile<T_Q4_1> : WTileLeg<T_Q4_1> {};
template <> struct WTile<T_Q5_1> : WTileLeg<T_Q5_1> {};
template <> struct WTile<T_Q6_0> : WTileLeg<T_Q6_0> {};
// MXFP4: four 17-byte blocks {u8 e; u8 qs[16]} per K tile (byte-aligned 8-byte loads), the nibble layout and step map of WTileNib, e2m1 table, 2^(e - 128) scale
template <> struct WTile<T_MXFP4> {
    static constexpr int HBIT = 1;
    uint2 q[4]; float d[4]; uint32_t eb[4];
    __device__ __forceinline__ void load(const uint8_t *row, int kt, int h) {
        const uint8_t *b = row + (long)kt * 68;
#pragma unroll
        for (int i = 0; i < 4; ++i) { eb[i] = b[17 * i]; q[i] = ld64_a1(b + 17 * i + 1 + 8 * h); }
    }
    __device__ __forceinline__ void prepare(int, const void *) {
#pragma unroll
        for (int i = 0; i < 4; ++i) d[i] = e8m0_half(eb[i]);
    }
    static __host__ __device__ __forceinline__ constexpr int kpiece(int s) { return 4 * (s >> 1) + 2 * (s & 1); }
    __device__ __forceinline__ half8 frag(int s, int) const {
        const int b = s >> 1, hi = s & 1;
        uint32_t n0 = q[b].x, n1 = q[b].y; if (hi) { n0 >>= 4; n1 >>= 4; }
        return frag_sbytes(mxfp4_lookup4(n0 & 0x0f0f0f0fu), mxfp4_lookup4(n1 & 0x0f0f0f0fu), d[b]);
    }
};

// IQ4_K {f16 d; u16 extra; u8 scales_h[4]; u8 scales_l[8]; u8 qs[128]} and IQ4_KS (f32 row scale, then {u8 scales[8]; u8 qs[128]}): the IQ4_XS tile with a scale per 16 (IQ4_K)
// or 32 (IQ4_KS) weights and the value table shifted by 4 where the block's bit says so
template <int TYPE> struct WTileIq4k {
    static constexpr int HBIT = 1;
    uint2 q[4]; uint32_t hdr, shw, slw; float drow; int n4; float d[4][2], add[4][2];
    __device__ __forceinline__ void load(const uint8_t *row, int kt, int h) {
        n4 = 4 * (kt & 1);
        if (TYPE == T_IQ4_K) {
            const uint8_t *b = row + (long)(kt >> 1) * 144;
            hdr = ld32(b); shw = ld16(b + 4 + 2 * (kt & 1)); slw = ld32(b + 8 + n4);
#pragma unroll
            for (int i = 0; i < 4; ++i) q[i] = ld64(b + 16 + 16 * (n4 + i) + 8 * h);
        } else if (TYPE == T_IQ4_KS) {
            const uint8_t *b = row + 4 + (long)(kt >> 1) * 136;
            drow = *reinterpret_cast<const float *>(row); slw = ld32(b + n4);
#pragma unroll
            for (int i = 0; i < 4; ++i) q[i] = ld64(b + 8 + 16 * (n4 + i) + 8 * h);
        } else {                        // IQ4_KSS: the scale byte of a 32-block = the low bits of its eight 16-bit words -> every lane needs the whole 16 bytes
            const uint8_t *b = row + 4 + (long)(kt >> 1) * 128 + 16 * n4;
            drow = *reinterpret_cast<const float *>(row); slw = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint4 w = ld128(b + 16 * i); const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
                uint32_t ls = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) ls |= ((ww[j] & 1u) | ((ww[j] >> 15) & 2u)) << (2 * j);
                slw |= ls << (8 * i);
                uint32_t a0 = ww[2 * h] & 0xfffefffeu, a1 = ww[2 * h + 1] & 0xfffefffeu;
                a0 ^= (a0 >> 1) & 0x7fff7fffu; a1 ^= (a1 >> 1) & 0x7fff7fffu;
                q[i] = make_uint2(a0, a1);
            }
        }
    }
    __device__ __forceinline__ void prepare(int, const void *) {
        if (TYPE == T_IQ4_K) {
            const float dd = half_bits_to_float(hdr & 0xffff); const uint32_t ex = (hdr >> 16) >> (2 * n4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {       // sub-block ib = n4 + i: scales_h byte (ib / 2) >> 4 (ib % 2) = byte (i / 2) of shw >> 4 (i % 2)
                const uint32_t sl = (slw >> (8 * i)) & 0xff, sh = (shw >> (8 * (i >> 1) + 4 * (i & 1))) & 0xf;
                d[i][0] = dd * (float)((int)((sl & 15) | ((sh << 4) & 0x30)) - 32); d[i][1] = dd * (float)((int)((sl >> 4) | ((sh << 2) & 0x30)) - 32);
                add[i][0] = ((ex >> (2 * i)) & 1) ? 4.f : 0.f; add[i][1] = ((ex >> (2 * i + 1)) & 1) ? 4.f : 0.f;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) { const uint32_t sb = (slw >> (8 * i)) & 0xff; d[i][0] = d[i][1] = drow * (float)((int)(sb & 254) - 127); add[i][0] = add[i][1] = (sb & 1) ? 4.f : 0.f; }
        }
    }
    static __host__ __device__ __forceinline__ constexpr int kpiece(int s) { return 4 * (s >> 1) + 2 * (s & 1); }
    __device__ __forceinline__ half8 frag(int s, int) const {
        const int b = s >> 1, hi = s & 1;
        uint32_t n0 = q[b].x, n1 = q[b].y; if (hi) { n0 >>= 4; n1 >>= 4; }
        const uint32_t v0 = iq4nl_lookup4(n0 & 0x0f0f0f0fu), v1 = iq4nl_lookup4(n1 & 0x0f0f0f0fu);
        const float a = d[b][hi], c = add[b][hi];
        return pack8(a * ((float)(int)(int8_t)(v0 & 0xff) + c), a * ((float)(int)(int8_t)((v0 >> 8) & 0xff) + c), a * ((float)(int)(int8_t)((v0 >> 16) & 0xff) + c), a * ((float)((int)v0 >> 24) + c),
                     a * ((float)(int)(int8_t)(v1 & 0xff) + c), a * ((float)(int)(int8_t)((v1 >> 8) & 0xff) + c), a * ((float)(int)(int8_t)((v1 >> 16) & 0xff) + c), a * ((float)((int)v1 >> 24) + c));
    }
};
template <> struct WTile<T_IQ4_K> : WTileIq4k<T_IQ4_K> {};
template <> struct WTile<T_IQ4_KS> : WTileIq4k<T_IQ4_KS> {};
template <> struct WTile<T_IQ4_KSS> : WTileIq4k<T_IQ4_KSS> {};

// IQ5_K {f16 d; u16 extra; u8 scales_h[4]; u8 scales_l[8]; u8 qs[128]; u8 qh[32]} and IQ5_KS (f32 row scale, {u8 scales[8]; u8 qs[128]; u8 qh[32]}): K tile = the 64-groups
// i = 2 n, 2 n + 1; group elements 16 c + j (c = 0..3) = nibble (c >> 1) of qs[32 i + 16 (c & 1) + j] with bit 2 i + (c >> 1) of qh[16 (c & 1) + j] as fifth bit.
// step s = 4 gi + c; half h owns j = 8 h + [0, 8): piece 8 gi + 2 c + h
template <int TYPE> struct WTileIq5k {
    static constexpr int HBIT = 1;
    uint2 q[2][2], qh[2]; uint32_t hdr, shw, slw; float drow; int n2; float d[2][4], add[2][4];
    __device__ __forceinline__ void load(const uint8_t *row, int kt, int h) {
        n2 = 2 * (kt & 1);
        const uint8_t *b = TYPE == T_IQ5_K ? row + (long)(kt >> 1) * 176 : row + 4 + (long)(kt >> 1) * 168;
        constexpr int QS = TYPE == T_IQ5_K ? 16 : 8, QH = TYPE == T_IQ5_K ? 144 : 136;
        if (TYPE == T_IQ5_K) { hdr = ld32(b); shw = ld16(b + 4 + n2); slw = ld32(b + 8 + 2 * n2); }
        else { drow = *reinterpret_cast<const float *>(row); slw = ld32(b + 2 * n2); }
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) { q[gi][0] = ld64(b + QS + 32 * (n2 + gi) + 8 * h); q[gi][1] = ld64(b + QS + 32 * (n2 + gi) + 16 + 8 * h); }
        qh[0] = ld64(b + QH + 8 * h); qh[1] = ld64(b + QH + 16 + 8 * h);
    }
    __device__ __forceinline__ void prepare(int, const void *) {
#pragma unroll
        for (int gi = 0; gi < 2; ++gi)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (TYPE == T_IQ5_K) {
                    const float dd = half_bits_to_float(hdr & 0xffff); const uint32_t ex = (hdr >> 16) >> (4 * (n2 + gi)), shb = (shw >> (8 * gi)) & 0xff, sl = (slw >> (16 * gi)) & 0xffff;
                    d[gi][c] = dd * (float)((int)(((sl >> (4 * c)) & 15) | (((shb >> (2 * c)) & 3) << 4)) - 32); add[gi][c] = ((ex >> c) & 1) ? 2.f : 0.f;
                } else {
                    const uint32_t sb = (slw >> (16 * gi + 8 * (c >> 1))) & 0xff;
                    d[gi][c] = drow * (float)((int)(sb & 254) - 127); add[gi][c] = (sb & 1) ? 2.f : 0.f;
                }
            }
    }
    static __host__ __device__ __forceinline__ constexpr int kpiece(int s) { return 8 * (s >> 2) + 2 * (s & 3); }
    __device__ __forceinline__ half8 frag(int s, int) const {
        const int gi = s >> 2, c = s & 3, hb = 2 * (n2 + gi) + (c >> 1);
        uint32_t n0 = q[gi][c & 1].x, n1 = q[gi][c & 1].y; if (c & 2) { n0 >>= 4; n1 >>= 4; }
        const uint32_t i0 = (n0 & 0x0f0f0f0fu) | (((qh[c & 1].x >> hb) & 0x01010101u) << 4), i1 = (n1 & 0x0f0f0f0fu) | (((qh[c & 1].y >> hb) & 0x01010101u) << 4);
        const uint32_t v0 = iq5nl_lookup4(i0), v1 = iq5nl_lookup4(i1);
        const float a = d[gi][c], o = add[gi][c];
        return pack8(a * ((float)(int)(int8_t)(v0 & 0xff) + o), a * ((float)(int)(int8_t)((v0 >> 8) & 0xff) + o), a * ((float)(int)(int8_t)((v0 >> 16) & 0xff) + o), a * ((float)((int)v0 >> 24) + o),
                     a * ((float)(int)(int8_t)(v1 & 0xff) + o), a * ((float)(int)(int8_t)((v1 >> 8) & 0xff) + o), a * ((float)(int)(int8_t)((v1 >> 16) & 0xff) + o), a * ((float)((int)v1 >> 24) + o));
    }
};
template <> struct WTile<T_IQ5_K> : WTileIq5k<T_IQ5_K> {};
template <> struct WTile<T_IQ5_KS> : WTileIq5k<T_IQ5_KS> {};

// The 2-bit-packed super-block types: Q2_K {u8 scales[16]; u8 qs[64]; f16 d, dmin}, Q3_K {u8 hmask[32]; u8 qs[64]; u8 scales[12]; f16 d}, IQ2_K {f16 d; u16 extra; u8 scales[8]; u8 qs[64]},
// IQ3_K {f16 d; u16 extra; u16 scales_h; u8 scales_l[8]; u8 qs[64]; u8 qh[32]}.  K tile = half n of a super-block = the 32 qs bytes 32 n .. at shifts 0, 2, 4, 6: element 32 j + l =
// (qs[l] >> 2 j) & 3; step s = 16 elements (j = s / 2, l = 16 (s & 1) + 8 h + [0, 8)) = exactly one 16-weight scale (index 8 n + s): natural k order, piece 2 s + h
template <int TYPE> struct WTile2b {
    static constexpr int HBIT = 1;
    uint2 q[2], hb[2]; uint32_t w0, w1, w2, w3; int n; float a[8], c[8];          // a: scale per step; c: Q2_K -dmin m | IQ2_K / IQ3_K value shift
    __device__ __forceinline__ void load(const uint8_t *row, int kt, int h) {
        n = kt & 1;
        const uint8_t *b = row + type_row_meta(TYPE) + (long)(kt >> 1) * type_block_bytes(TYPE);
        constexpr int QS = TYPE == T_Q2_K ? 16 : TYPE == T_Q3_K ? 32 : TYPE == T_IQ2_K ? 12 : TYPE == T_IQ3_K ? 14 : 6;
        q[0] = ld64(b + QS + 32 * n + 8 * h); q[1] = ld64(b + QS + 32 * n + 16 + 8 * h);
        if (TYPE == T_Q2_K) { w0 = ld32(b + 8 * n); w1 = ld32(b + 8 * n + 4); w2 = ld32(b + 80); }
        if (TYPE == T_Q3_K) { hb[0] = ld64(b + 8 * h); hb[1] = ld64(b + 16 + 8 * h); w0 = ld32(b + 96); w1 = ld32(b + 100); w2 = ld32(b + 104); w3 = ld16(b + 108); }
        if (TYPE == T_IQ2_K) { w0 = ld32(b); w1 = ld32(b + 4 + 4 * n); }
        if (TYPE == T_IQ3_K) { hb[0] = ld64(b + 78 + 8 * h); hb[1] = ld64(b + 78 + 16 + 8 * h); w0 = ld32(b); w1 = ld16(b + 4); w2 = ld32(b + 6 + 4 * n); }
        if (TYPE == T_IQ2_KS) { w0 = ld16(b); w1 = ld32(b + 2); w3 = ld16(row); }
        if (TYPE == T_IQ3_KS) { hb[0] = ld64(b + 70 + 8 * h); hb[1] = ld64(b + 70 + 16 + 8 * h); w0 = ld16(b); w1 = ld32(b + 2); w3 = ld16(row); }
    }
    __device__ __forceinline__ void prepare(int, const void *) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (TYPE == T_Q2_K) {
                const uint32_t sb = ((s < 4 ? w0 : w1) >> (8 * (s & 3))) & 0xff;
                a[s] = half_bits_to_float(w2 & 0xffff) * (float)(sb & 15); c[s] = -(half_bits_to_float(w2 >> 16) * (float)(sb >> 4));
            } else if (TYPE == T_Q3_K) {
                const int is = 8 * n + s; const uint32_t lo = ((s < 4 ? w0 : w1) >> (8 * (s & 3))) & 0xff, hi = (w2 >> (8 * (is & 3) + 2 * (is >> 2))) & 3;
                a[s] = half_bits_to_float(w3) * (float)((int)((n ? lo >> 4 : lo & 15) | (hi << 4)) - 32); c[s] = 0.f;
            } else if (TYPE == T_IQ2_K) {
                const int is = 8 * n + s; const uint32_t nib = (w1 >> (4 * s)) & 15;
                a[s] = half_bits_to_float(w0 & 0xffff) * (float)((int)nib - 8); c[s] = (((w0 >> 16) >> is) & 1) ? 5.f : 0.f;
            } else if (TYPE == T_IQ3_K) {
                const int is = 8 * n + s; const int m = 2 * (int)((w2 >> (4 * s)) & 15) + 1;
                a[s] = half_bits_to_float(w0 & 0xffff) * (float)(((w1 >> is) & 1) ? -m : m); c[s] = (((w0 >> 16) >> is) & 1) ? 4.f : 0.f;
            } else if (TYPE == T_IQ2_KS) {        // 32-block ib = 4 n + s / 2: nibble (ib & 1) of scales[ib / 2] | extra bit 8 + ib << 4, - 16; value shift: extra bit ib
                const int ib = 4 * n + (s >> 1);
                a[s] = half_bits_to_float(w3) * (float)((int)(((w1 >> (4 * ib)) & 15) | (((w0 >> (8 + ib)) & 1) << 4)) - 16); c[s] = ((w0 >> ib) & 1) ? 5.f : 0.f;
            } else {                               // IQ3_KS: nibble (ib / 4) of scales[ib % 4] | extra bit ib << 4, - 16; value shift: extra bit 8 + ib
                const int ib = 4 * n + (s >> 1);
                a[s] = half_bits_to_float(w3) * (float)((int)(((w1 >> (8 * (ib & 3) + 4 * (ib >> 2))) & 15) | (((w0 >> ib) & 1) << 4)) - 16); c[s] = ((w0 >> (8 + ib)) & 1) ? 4.f : 0.f;
            }
        }
    }
    static __host__ __device__ __forceinline__ constexpr int kpiece(int s) { return 2 * s; }
    __device__ __forceinline__ half8 frag(int s, int) const {
        const int sh = 2 * (s >> 1);
        uint32_t n0 = (q[s & 1].x >> sh) & 0x03030303u, n1 = (q[s & 1].y >> sh) & 0x03030303u;
        float f[8];
        if (TYPE == T_Q2_K) { fma4_ubytes(n0, a[s], c[s], f[0], f[1], f[2], f[3]); fma4_ubytes(n1, a[s], c[s], f[4], f[5], f[6], f[7]); }
        else if (TYPE == T_Q3_K) {
            const int hs = 4 * n + (s >> 1);
            n0 |= ((hb[s & 1].x >> hs) & 0x01010101u) << 2; n1 |= ((hb[s & 1].y >> hs) & 0x01010101u) << 2;
            f[0] = a[s] * (ubyte0(n0) - 4.f); f[1] = a[s] * (ubyte1(n0) - 4.f); f[2] = a[s] * (ubyte2(n0) - 4.f); f[3] = a[s] * (ubyte3(n0) - 4.f);
            f[4] = a[s] * (ubyte0(n1) - 4.f); f[5] = a[s] * (ubyte1(n1) - 4.f); f[6] = a[s] * (ubyte2(n1) - 4.f); f[7] = a[s] * (ubyte3(n1) - 4.f);
        } else {
            uint32_t v0, v1;
            if (TYPE == T_IQ2_K || TYPE == T_IQ2_KS) { const uint32_t t = k_iq2nl_packed[0]; v0 = __builtin_amdgcn_perm(t, t, n0); v1 = __builtin_amdgcn_perm(t, t, n1); }
            else {
                const int hs = 4 * n + (s >> 1);
                n0 |= ((hb[s & 1].x >> hs) & 0x01010101u) << 2; n1 |= ((hb[s & 1].y >> hs) & 0x01010101u) << 2;
                v0 = __builtin_amdgcn_perm(k_iq3nl_packed[1], k_iq3nl_packed[0], n0); v1 = __builtin_amdgcn_perm(k_iq3nl_packed[1], k_iq3nl_packed[0], n1);
            }
            const float o = c[s];
            f[0] = a[s] * ((float)(int)(int8_t)(v0 & 0xff) + o); f[1] = a[s] * ((float)(int)(int8_t)((v0 >> 8) & 0xff) + o); f[2] = a[s] * ((float)(int)(int8_t)((v0 >> 16) & 0xff) + o); f[3] = a[s] * ((float)((int)v0 >> 24) + o);
            f[4] = a[s] * ((float)(int)(int8_t)(v1 & 0xff) + o); f[5] = a[s] * ((float)(int)(int8_t)((v1 >> 8) & 0xff) + o); f[6] = a[s] * ((float)(int)(int8_t)((v1 >> 16) & 0xff) + o); f[7] = a[s] * ((float)((int)v1 >> 24) + o);
        }
        return pack8(f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7]);
    }
};
template <> struct WTile<T_Q2_K> : WTile2b<T_Q2_K> {};
template <> struct WTile<T_Q3_K> : WTile2b<T_Q3_K> {};
template <> struct WTile<T_IQ2_K> : WTile2b<T_IQ2_K> {};
template <> struct WTile<T_IQ3_K> : WTile2b<T_IQ3_K> {};
template <> struct WTile<T_IQ2_KS> : WTile2b<T_IQ2_KS> {};
template <> struct WTile<T_IQ3_KS> : WTile2b<T_IQ3_KS> {};

// IQ2_XXS / IQ2_XS / IQ3_XXS: the IQ2_S / IQ3_S tiles with the sign byte derived from a 7-bit index (ksign7) and the scales of those formats
template <> struct WTile<T_IQ2_XXS> {      // per 32-block two dwords {4 x u8 grid index | 4 x 7-bit sign index, 4-bit scale}
    static constexpr int HBIT = 2;
    uint4 q0, q1; uint32_t dh; float db[4]; const uint2 *grid;
    __device__ __forceinline__ void load(const uint8_t *row, int kt, int) { const uint8_t *b = row + (long)(kt >> 1) * 66; dh = ld16(b); q0 = ld128(b + 2 + 32 * (kt & 1)); q1 = ld128(b + 18 + 32 * (kt & 1)); }
    __device__ __forceinline__ void prepare(int, const void *g) {
        grid = reinterpret_cast<const uint2 *>(g);
        const float d = half_bits_to_float(dh); const uint32_t a1[4] = {q0.y, q0.w, q1.y, q1.w};
#pragma unroll
        for (int b = 0; b < 4; ++b) db[b] = d * (0.5f + (float)(a1[b] >> 28)) * 0.25f;                       // ggml-quants.c:3688
    }
    static __host__ __device__ __forceinline__ constexpr int kpiece(int s) { return 4 * (s >> 1) + (s & 1); }
    __device__ __forceinline__ half8 frag(int s, int h) const {
        const int b = s >> 1, l = 2 * h + (s & 1);
        const uint32_t a0 = b == 0 ? q0.x : b == 1 ? q0.z : b == 2 ? q1.x : q1.z, a1 = b == 0 ? q0.y : b == 1 ? q0.w : b == 2 ? q1.y : q1.w;
        const uint2 m = grid[(a0 >> (8 * l)) & 0xff]; const uint32_t sgn = ksign7((a1 >> (7 * l)) & 127);
        return frag_sbytes(apply_sign4(m.x, sign_mask4(sgn)), apply_sign4(m.y, sign_mask4(sgn >> 4)), db[b]);
    }
};
template <> struct WTile<T_IQ2_XS> {       // u16 {9-bit grid index | 7-bit sign index << 9} per 8 weights, 4-bit scales per 16
    static constexpr int HBIT = 2;
    uint4 q0, q1; uint32_t sc, dh; float db[4]; const uint2 *grid;
    __device__ __forceinline__ void load(const uint8_t *row, int kt, int) {
        const uint8_t *b = row + (long)(kt >> 1) * 74; const int n = kt & 1;
        dh = ld16(b); q0 = ld128(b + 2 + 32 * n); q1 = ld128(b + 18 + 32 * n); sc = ld32(b + 66 + 4 * n);
    }
    __device__ __forceinline__ void prepare(int h, const void *g) {
        grid = reinterpret_cast<const uint2 *>(g);
        const float d = half_bits_to_float(dh);
#pragma unroll
        for (int b = 0; b < 4; ++b) db[b] = d * (0.5f + (float)((sc >> (8 * b + 4 * h)) & 0xf)) * 0.25f;      // ggml-quants.c:3714-3715 (l / 2 = h)
    }
    static __host__ __device__ __forceinline__ constexpr int kpiece(int s) { return 4 * (s >> 1) + (s & 1); }
    __device__ __forceinline__ half8 frag(int s, int h) const {
        const int b = s >> 1;                                        // u16 l = 2 h + (s & 1) of block b: dword l / 2 = h, halfword s & 1
        const uint32_t w = b == 0 ? (h ? q0.y : q0.x) : b == 1 ? (h ? q0.w : q0.z) : b == 2 ? (h ? q1.y : q1.x) : (h ? q1.w : q1.z);
        const uint32_t v = (w >> (16 * (s & 1))) & 0xffff;
        const uint2 m = grid[v & 511]; const uint32_t sgn = ksign7(v >> 9);
        return frag_sbytes(apply_sign4(m.x, sign_mask4(sgn)), apply_sign4(m.y, sign_mask4(sgn >> 4)), db[b]);
    }
};
// IQ1_S {f16 d; u8 qs[32]; u16 qh[8]} / IQ1_M {u8 qs[32]; u8 qh[16]; u8 scales[8]}: 11-bit index into the ternary codebook per 8 weights; LDS image = signed bytes
// 8 g + 1 (16 KiB).  weight = dl (g +- 1/8) = (dl / 8) (8 g + 1) [- 2 (dl / 8) when the delta bit is set]: every product and the correction are exact in f32,
// so the f16 value is the rounded L0 value.  Step map of the IQ2_XS tile (group l = 2 h + (s & 1) of 32-block s / 2).
template <int TYPE> struct WTileIq1 {
    static constexpr int HBIT = 2;
    uint4 qs; uint2 qh, sc8; uint32_t dh; int n; float db[4], cn[4]; const uint2 *grid;
    __device__ __forceinline__ void load(const uint8_t *row, int kt, int) {
        n = kt & 1;
        if (TYPE == T_IQ1_S) { const uint8_t *b = row + (long)(kt >> 1) * 50; dh = ld16(b); qs = ld128(b + 2 + 16 * n); qh = ld64(b + 34 + 8 * n); }
        else { const uint8_t *b = row + (long)(kt >> 1) * 56; qs = ld128(b + 16 * n); qh = ld64(b + 32 + 8 * n); sc8 = ld64(b + 48); }
    }
    __device__ __forceinline__ uint32_t qh16(int b) const { return ((b < 2 ? qh.x : qh.y) >> (16 * (b & 1))) & 0xffff; }
    __device__ __forceinline__ void prepare(int h, const void *g) {
        grid = reinterpret_cast<const uint2 *>(g);
        if (TYPE == T_IQ1_S) {
            const float d = 0.125f * half_bits_to_float(dh);
#pragma unroll
            for (int b = 0; b < 4; ++b) { const uint32_t q = qh16(b); db[b] = d * (float)(2 * (int)((q >> 12) & 7) + 1); cn[b] = (q & 0x8000) ? -2.f * db[b] : 0.f; }
        } else {
            const float d = 0.125f * half_bits_to_float(((sc8.x >> 12) & 0xf) | ((sc8.x >> 24) & 0x00f0) | ((sc8.y >> 4) & 0x0f00) | ((sc8.y >> 16) & 0xf000));
            const uint32_t w2 = n ? sc8.y : sc8.x;
#pragma unroll
            for (int b = 0; b < 4; ++b) db[b] = d * (float)(2 * (int)((w2 >> (16 * (b >> 1) + 6 * (b & 1) + 3 * h)) & 7) + 1);
        }
    }
    static __host__ __device__ __forceinline__ constexpr int kpiece(int s) { return 4 * (s >> 1) + (s & 1); }
    __device__ __forceinline__ half8 frag(int s, int h) const {
        const int b = s >> 1, j = s & 1, l = 2 * h + j;
        const uint32_t qsb = ((b == 0 ? qs.x : b == 1 ? qs.y : b == 2 ? qs.z : qs.w) >> (8 * l)) & 0xff;
        uint32_t idx; float c;
        if (TYPE == T_IQ1_S) { idx = qsb | (((qh16(b) >> (3 * l)) & 7) << 8); c = cn[b]; }
        else { const uint32_t nib = (qh16(b) >> (8 * h + 4 * j)) & 0xf; idx = qsb | ((nib & 7) << 8); c = (nib & 8) ? -2.f * db[b] : 0.f; }
        const uint2 m = grid[idx]; float f[8];
        mul4_sbytes(m.x, db[b], f[0], f[1], f[2], f[3]); mul4_sbytes(m.y, db[b], f[4], f[5], f[6], f[7]);
        return pack8(f[0] + c, f[1] + c, f[2] + c, f[3] + c, f[4] + c, f[5] + c, f[6] + c, f[7] + c);
    }
};
template <> struct WTile<T_IQ1_S> : WTileIq1<T_IQ1_S> {};
template <> struct WTile<T_IQ1_M> : WTileIq1<T_IQ1_M> {};
template <> struct WTile<T_IQ3_XXS> {      // qs[64] 8-bit grid indices (4 magnitudes each), then per 32-block a dword {4 x 7-bit sign index, 4-bit scale}
    static constexpr int HBIT = 2;
    uint4 q0, q1, sa; uint32_t dh; float db[4]; const uint32_t *grid;
    __device__ __forceinline__ void load(const uint8_t *row, int kt, int) {
        const uint8_t *b = row + (long)(kt >> 1) * 98; const int n = kt & 1;
        dh = ld16(b); q0 = ld128(b + 2 + 32 * n); q1 = ld128(b + 18 + 32 * n); sa = ld128(b + 66 + 16 * n);
    }
    __device__ __forceinline__ void prepare(int, const void *g) {
        grid = reinterpret_cast<const uint32_t *>(g);
        const float d = half_bits_to_float(dh); const uint32_t a[4] = {sa.x, sa.y, sa.z, sa.w};
#pragma unroll
        for (int b = 0; b < 4; ++b) db[b] = d * (0.5f + (float)(a[b] >> 28)) * 0.5f;                         // ggml-quants.c:3776
    }
    static __host__ __device__ __forceinline__ constexpr int kpiece(int s) { return 4 * (s >> 1) + (s & 1); }
    __device__ __forceinline__ half8 frag(int s, int h) const {
        const int b = s >> 1, l = 2 * h + (s & 1);
        const uint32_t w = b == 0 ? (h ? q0.y : q0.x) : b == 1 ? (h ? q0.w : q0.z) : b == 2 ? (h ? q1.y : q1.x) : (h ? q1.w : q1.z);      // qs[8 b + 2 l], qs[8 b + 2 l + 1]: halfword l of the block's 8 bytes
        const uint32_t pair = (w >> (16 * (s & 1))) & 0xffff, a = b == 0 ? sa.x : b == 1 ? sa.y : b == 2 ? sa.z : sa.w;
        const uint32_t sgn = ksign7((a >> (7 * l)) & 127);
        return frag_sbytes(apply_sign4(grid[pair & 0xff], sign_mask4(sgn)), apply_sign4(grid[pair >> 8], sign_mask4(sgn >> 4)), db[b]);
    }
};

// IQ6_K {f16 d; u16 extra; i8 scales[16]; u8 qs[128]; u8 qh[64]}: the nibble layout of IQ5_K with TWO qh bits (6-bit index) and an int8 scale per 16 weights
template <> struct WTile<T_IQ6_K> {
    static constexpr int HBIT = 1;
    uint2 q[2][2], qh[2]; uint32_t hdr, scw0, scw1; int n2; float d[2][4], add[2][4];
    __device__ __forceinline__ void load(const uint8_t *row, int kt, int h) {
        n2 = 2 * (kt & 1);
        const uint8_t *b = row + (long)(kt >> 1) * 212;
        hdr = ld32(b); scw0 = ld32(b + 4 + 4 * n2); scw1 = ld32(b + 8 + 4 * n2);
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) { q[gi][0] = ld64(b + 20 + 32 * (n2 + gi) + 8 * h); q[gi][1] = ld64(b + 20 + 32 * (n2 + gi) + 16 + 8 * h); }
        qh[0] = ld64(b + 148 + 32 * (kt & 1) + 8 * h); qh[1] = ld64(b + 148 + 32 * (kt & 1) + 16 + 8 * h);
    }
    __device__ __forceinline__ void prepare(int, const void *) {
        const float dd = half_bits_to_float(hdr & 0xffff);
#pragma unroll
        for (int gi = 0; gi < 2; ++gi)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint32_t ex = (hdr >> 16) >> (4 * (n2 + gi));
                d[gi][c] = dd * (float)(int)(int8_t)(((gi ? scw1 : scw0) >> (8 * c)) & 0xff); add[gi][c] = ((ex >> c) & 1) ? 1.f : 0.f;
            }
    }
    static __host__ __device__ __forceinline__ constexpr int kpiece(int s) { return 8 * (s >> 2) + 2 * (s & 3); }
    __device__ __forceinline__ half8 frag(int s, int) const {
        const int gi = s >> 2, c = s & 3, sh = 4 * gi + (c & 2);          // group i = n2 + gi: qh bits at 4 (i & 1) = 4 gi (n2 is even), + 2 for the high nibbles
        uint32_t n0 = q[gi][c & 1].x, n1 = q[gi][c & 1].y; if (c & 2) { n0 >>= 4; n1 >>= 4; }
        const uint32_t i0 = (n0 & 0x0f0f0f0fu) | (((qh[c & 1].x >> sh) & 0x03030303u) << 4), i1 = (n1 & 0x0f0f0f0fu) | (((qh[c & 1].y >> sh) & 0x03030303u) << 4);
        // the L0 value (to_float): a cubic in the index, d * scale * (A + q (B + q (-C + q D)) + m) as the reference build's fma chain -- the prompt path's contract is "L0 weights rounded
        // once to f16"; the decode unit uses the int8 table like the reference's mat-mul kernels
        const float a = d[gi][c], o = add[gi][c];
        auto cub = [&](float qf) { return a * (fmaf(qf, fmaf(qf, fmaf(qf, 0.0011972f, -0.11218f), 6.2568f), -127.f) + o); };
        return pack8(cub(ubyte0(i0)), cub(ubyte1(i0)), cub(ubyte2(i0)), cub(ubyte3(i0)), cub(ubyte0(i1)), cub(ubyte1(i1)), cub(ubyte2(i1)), cub(ubyte3(i1)));
    }
};
// IQ2_KL (f16 row scale; {u16 scales_h; u8 scales_l[4]; u8 qs[64]; u8 qh[16]}): group i of 64 = 16 bytes; byte j's low nibble (+ qh[j] bit 2 i) names the PAIR (2 j, 2 j + 1), its
// high nibble (+ bit 2 i + 1) the pair (32 + 2 j, 33 + 2 j).  step s = 4 gi + c: elements 16 c + 8 h + [0, 8) = nibble (c >> 1) of bytes 8 (c & 1) + 4 h + [0, 4)
template <> struct WTile<T_IQ2_KL> {
    static constexpr int HBIT = 1;
    uint32_t q[2][2], qh[2], sl, sh, dr; int n2; float d[2][2];
    __device__ __forceinline__ void load(const uint8_t *row, int kt, int h) {
        n2 = 2 * (kt & 1);
        const uint8_t *b = row + 2 + (long)(kt >> 1) * 86;
        dr = ld16(row); sh = ld16(b); sl = ld32(b + 2);
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) { q[gi][0] = ld32(b + 6 + 16 * (n2 + gi) + 4 * h); q[gi][1] = ld32(b + 6 + 16 * (n2 + gi) + 8 + 4 * h); }
        qh[0] = ld32(b + 70 + 4 * h); qh[1] = ld32(b + 70 + 8 + 4 * h);
    }
    __device__ __forceinline__ void prepare(int, const void *) {
        const float dd = half_bits_to_float(dr);
#pragma unroll
        for (int gi = 0; gi < 2; ++gi)
#pragma unroll
            for (int p = 0; p < 2; ++p) { const int i = n2 + gi; d[gi][p] = dd * (float)((int)(((sl >> (8 * ((2 * i + p) & 3) + 4 * (i >> 1))) & 15) | (((sh >> (4 * i + 2 * p)) & 3) << 4)) - 32); }
    }
    static __host__ __device__ __forceinline__ constexpr int kpiece(int s) { return 8 * (s >> 2) + 2 * (s & 3); }
    __device__ __forceinline__ half8 frag(int s, int) const {
        const int gi = s >> 2, c = s & 3, p = c >> 1;
        const uint32_t w = q[gi][c & 1], idx = ((p ? (w >> 4) : w) & 0x0f0f0f0fu) | (((qh[c & 1] >> (2 * (n2 + gi) + p)) & 0x01010101u) << 4);
        const uint32_t a0 = lookup32x4(k_iq2kl_v0, idx), a1 = lookup32x4(k_iq2kl_v1, idx);
        return frag_sbytes(__builtin_amdgcn_perm(a1, a0, 0x05010400u), __builtin_amdgcn_perm(a1, a0, 0x07030602u), d[gi][p]);
    }
};

// IQ2_S: tile = 32-blocks 4n..4n+3; half h owns grid entries l = 2h, 2h+1 of every 32-block (8 elements each)
template <> struct WTile<T_IQ2_S> {
    static constexpr int HBIT = 2;
    uint4 qs, sg; uint32_t qh, sc, dh; float db[4]; const uint2 *grid;
    __device__ __forceinline__ void load(const uint8_t *row, int kt, int) {
        const uint8_t *b = row + (long)(kt >> 1) * 82; const int n = kt & 1;
        dh = ld16(b); qs = ld128(b + 2 + 16 * n); sg = ld128(b + 34 + 16 * n); qh = ld32(b + 66 + 4 * n); sc = ld32(b + 74 + 4 * n);
    }
    __device__ __forceinline__ void prepare(int h, const void *g) {
        grid = reinterpret_cast<const uint2 *>(g);
        const float d = half_bits_to_float(dh);
#pragma unroll
        for (int b = 0; b < 4; ++b) db[b] = d * (0.5f + (float)((sc >> (8 * b + 4 * h)) & 0xf)) * 0.25f;      // ggml-quants.c:3744-3745
    }
    // step s = 2 b + j : grid entry l = 2h + j of 32-block b : elements 32 b + 16 h + 8 j + [0,8)
    static __host__ __device__ __forceinline__ constexpr int kpiece(int s) { return 4 * (s >> 1) + (s & 1); }
    __device__ __forceinline__ half8 frag(int s, int h) const {
        const int b = s >> 1, l = 2 * h + (s & 1);
        const uint32_t qw = b == 0 ? qs.x : b == 1 ? qs.y : b == 2 ? qs.z : qs.w, sw = b == 0 ? sg.x : b == 1 ? sg.y : b == 2 ? sg.z : sg.w;
        const uint32_t hb = (qh >> (8 * b)) & 0xff;
        const uint32_t idx = ((qw >> (8 * l)) & 0xff) | ((hb << (8 - 2 * l)) & 0x300);
        const uint2 m = grid[idx]; const uint32_t sgn = (sw >> (8 * l)) & 0xff;
        const uint32_t v0 = apply_sign4(m.x, sign_mask4(sgn)), v1 = apply_sign4(m.y, sign_mask4(sgn >> 4));
        float f[8]; mul4_sbytes(v0, db[b], f[0], f[1], f[2], f[3]); mul4_sbytes(v1, db[b], f[4], f[5], f[6], f[7]);
        return pack8(f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7]);
    }
};

// IQ3_S: tile = 32-blocks 4n..4n+3; half h owns l = 2h, 2h+1 (grid1 + grid2 = 8 elements per l)
template <> struct WTile<T_IQ3_S> {
    static constexpr int HBIT = 2;
    uint4 q0, q1, sg; uint32_t qh, sc, dh; float db[4]; const uint32_t *grid;
    __device__ __forceinline__ void load(const uint8_t *row, int kt, int) {
        const uint8_t *b = row + (long)(kt >> 1) * 110; const int n = kt & 1;
        dh = ld16(b); q0 = ld128(b + 2 + 32 * n); q1 = ld128(b + 18 + 32 * n); qh = ld32(b + 66 + 4 * n); sg = ld128(b + 74 + 16 * n); sc = ld16(b + 106 + 2 * n);
    }
    __device__ __forceinline__ void prepare(int, const void *g) {
        grid = reinterpret_cast<const uint32_t *>(g);
        const float d = half_bits_to_float(dh);
#pragma unroll
        for (int b = 0; b < 4; ++b) db[b] = d * (float)(1 + 2 * (int)((sc >> (4 * b)) & 0xf));                   // ggml-quants.c:3807-3808
    }
    static __host__ __device__ __forceinline__ constexpr int kpiece(int s) { return 4 * (s >> 1) + (s & 1); }
    __device__ __forceinline__ half8 frag(int s, int h) const {
        const int b = s >> 1, l = 2 * h + (s & 1);
        // qs bytes 8b .. 8b+7 of the tile = dwords (2b, 2b+1) of {q0,q1}; pair (qs[2l], qs[2l+1]) = halfword l
        const uint32_t w0 = b == 0 ? q0.x : b == 1 ? q0.z : b == 2 ? q1.x : q1.z, w1 = b == 0 ? q0.y : b == 1 ? q0.w : b == 2 ? q1.y : q1.w;
        const uint32_t pair = ((l & 2) ? w1 : w0) >> (16 * (l & 1)) & 0xffff;
        const uint32_t hb = (qh >> (8 * b)) & 0xff;
        const uint32_t i1 = (pair & 0xff) | ((hb << (8 - 2 * l)) & 256), i2 = (pair >> 8) | ((hb << (7 - 2 * l)) & 256);
        const uint32_t sw = b == 0 ? sg.x : b == 1 ? sg.y : b == 2 ? sg.z : sg.w; const uint32_t sgn = (sw >> (8 * l)) & 0xff;
        const uint32_t v0 = apply_sign4(grid[i1], sign_mask4(sgn)), v1 = apply_sign4(grid[i2], sign_mask4(sgn >> 4));
        float f[8]; mul4_sbytes(v0, db[b], f[0], f[1], f[2], f[3]); mul4_sbytes(v1, db[b], f[4], f[5], f[6], f[7]);
        return pack8(f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7]);
    }
};

static inline bool gemm_mfma_supported(int t) { return t == T_MXFP4 || t == T_IQ1_S || t == T_IQ1_M || t == T_Q4_K || t == T_Q5_K || t == T_Q6_K || t == T_IQ4_NL || t == T_IQ2_S || t == T_IQ3_S || t == T_Q4_0 || t == T_Q8_0 || t == T_IQ4_XS ||
                                                        t == T_Q5_0 || t == T_Q4_1 || t == T_Q5_1 || t == T_Q6_0 || t == T_IQ4_K || t == T_IQ4_KS || t == T_IQ5_K || t == T_IQ5_KS ||
                                                        t == T_Q2_K || t == T_Q3_K || t == T_IQ2_K || t == T_IQ3_K || t == T_IQ2_XXS || t == T_IQ2_XS || t == T_IQ3_XXS || t == T_IQ2_KS || t == T_IQ3_KS || t == T_IQ4_KSS || t == T_IQ6_K || t == T_IQ2_KL; }
__host__ __device__ constexpr size_t gemm_grid_lds_bytes(int t) { return t == T_IQ2_S ? 8192 : t == T_IQ3_S ? 2048 : t == T_IQ2_XXS ? 2048 : t == T_IQ2_XS ? 4096 : t == T_IQ3_XXS ? 1024 : (t == T_IQ1_S || t == T_IQ1_M) ? 16384 : 0; }

// grid: x = (128*MW-row weight tile, (32*NT)-token tile) pairs in XCD-aware order, z = K split.  256*MW threads per K-group = 4*MW
// waves, wave w owns rows [32w, 32w+32).
//
// Pipeline: weights advance in 128-wide K tiles (the natural half super-block), activations in KX-wide tiles (KX = 64 at NT = 8,
// 128 otherwise; one barrier per activation tile, >= 32 MFMAs between barriers):
//     barrier                       (carries vmcnt(0)) activation tile t is complete in LDS buffer p; buffer p^1 is free
//     (first sub-tile) weights      raw quant bytes of K tile kt+1 -> registers ; scales of tile kt prepared
//     KX/16 k-steps x NT MFMAs      B fragments de-quantized from registers, A fragments ds_read_b128 one k-step ahead;
//                                   between the MFMAs, one global_load_lds piece of tile t+1 -> buffer p^1 per 4*MW MFMAs
// LDS image of a tile: [32*NT rows][KX/8 pieces of 16 B]; piece' = piece ^ ((row >> 1) & 7) (KX = 64: 128-byte rows alias every 2
// rows on the 64 banks) or piece ^ (row & 15) (KX = 128) => any 16 consecutive rows cover all banks: conflict-free ds_read_b128.
// KS = 2: the workgroup has 8 waves = two groups of 4; group g contracts K-half g of the SAME (128 rows x 32*NT tokens) tile with its
// own activation buffers, and the two partial accumulators are added through LDS at the end.  This keeps 2 waves per SIMD
// resident with 256-token tiles when the grid has fewer workgroups than 2 per CU (prompt of 512 tokens), without the global
// atomics / zero-fill of a grid-level K split.
#ifdef GEMM_EXP_NO_DEQUANT
template <class W> static __device__ __forceinline__ half8 exp_raw_frag(const W &w, int s) {      // timing experiments only
    union { uint32_t u[4]; half8 h; } c; __builtin_memcpy(&c, &w, 16); c.u[0] ^= (uint32_t)s; return c.h;
}
#endif

// MW = 2: 8 waves = 256 weight rows share ONE activation tile (half the LDS-DMA bytes and issues per MFMA -- the activation path is the
// largest non-MFMA cost, profiles/r01_notes.md); used when the 256-row grid still fills the chip (4k-token prefill).
// XW = 3 (KS = MW = 1): SEVEN waves = 224 weight rows per workgroup -- 14336-row matrices at 512 tokens are 64 x 4 = 256 such tiles, one per CU, where 128-row tiles give
// 448 workgroups for 512 slots (a quarter of the CUs runs one workgroup while the others run two).  The three extra waves issue no activation pieces.
// PART = true (the grouped MoE launches with 128-token tiles): a partly filled token tile -- the last tile of every expert -- only reads and multiplies its populated
// 32-token sub-tiles (a second instance of the K loop with wave-uniform guards; the dense launches keep the single branch-free loop)
#ifdef GEMM_EXP_TIMELINE      /* experiment build of ONE translation unit (scripts/mfma_timeline.py): 100 MHz wall-clock stamps of every workgroup's phases, wave 0 of each K-group */
__device__ unsigned long long g_mfma_timeline[2048 * 2 * 32];
extern "C" __attribute__((visibility("default"))) int cdna4_exp_mfma_timeline(void *dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_mfma_timeline), sizeof(g_mfma_timeline)); }
extern "C" __attribute__((visibility("default"))) int cdna4_exp_mfma_timeline_clear() { void *p = nullptr; if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_mfma_timeline)) != hipSuccess) return -1; return (int)hipMemset(p, 0, sizeof(g_mfma_timeline)); }
#define MT_STAMP(I_) { if (tl_rec && (I_) < 32) tl_p[(I_)] = wall_clock64(); }
#else
#define MT_STAMP(I_)
#endif
template <int TYPE, int NT, bool UPGATE, int KX, int KS, int MW = 1, int XW = 0, bool PART = false>
__global__ void __launch_bounds__(256 * KS * MW + 64 * XW, XW ? 1 : 2) gemm_mfma_kernel(const GemmArgs a) {
    static_assert(KS == 1 || MW == 1, "K-split workgroups are 128 rows tall");
    static_assert(XW == 0 || (KS == 1 && MW == 1), "extra waves: plain 4-wave DMA layout");
    // XW = 4: the seven compute waves of XW = 3 plus ONE PRODUCER wave (wave 7) that issues every LDS-DMA piece of the workgroup into a ring of THREE activation buffers, two
    // tiles ahead, and waits for them with a COUNTED vmcnt in front of a raw s_barrier (guide, "pipelining across barriers").  The compute waves issue no DMA: their per-tile
    // barrier no longer drains anything (with a DMA of their own in flight, __syncthreads' fence waits vmcnt(0) -- half a tile of MFMAs after the piece was requested), and
    // the ~100-clk issue stalls of the pieces leave their instruction streams.  One workgroup per CU (XW > 0), so the third 32 KiB buffer costs nothing.
    constexpr bool PROD = XW == 4; constexpr int XC = PROD ? 3 : XW, NBUF = PROD ? 3 : 2;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
#ifdef GEMM_EXP_TIMELINE
    const bool tl_rec = (threadIdx.x & 255) == 0 && threadIdx.x < 512 && (blockIdx.z * gridDim.x + blockIdx.x) < 2048;
    unsigned long long *tl_p = g_mfma_timeline + ((blockIdx.z * gridDim.x + blockIdx.x) * 2 + (threadIdx.x >> 8)) * 32;
    int tl_i = 8;
#endif
    MT_STAMP(0)
    // KX = k-width of the activation tile in LDS (64 or 128): LDS image [32*NT rows][KX/8 pieces of 16 B]
    constexpr int BN = 32 * NT, ROWB = KX * 2, PIECES = KX / 8, XT_BYTES = BN * ROWB, NXR = NT * KX / 64 / MW, NSUB = 128 / KX, SPS = 8 / NSUB;
    constexpr int WGT = 256 * MW, MROWS = 128 * MW + 32 * XC;           // threads per K-group (that stage activations), weight rows per workgroup
    static_assert(NXR >= 1, "tile too small for this many waves");
    const int lane = threadIdx.x & 63, wave = XW ? (int)(threadIdx.x >> 6) : (int)((threadIdx.x >> 6) & (4 * MW - 1)), kg = XW ? 0 : threadIdx.x / WGT, tg = threadIdx.x & (WGT - 1), h = lane >> 5;
    const bool stager = !PROD && (XW == 0 || wave < 4);       // waves that issue LDS-DMA pieces
    const bool producer = PROD && wave == 4 + XC;
    // XCD-aware tile order (guide T1): workgroup b runs on XCD b % 8 and XCDs have private L2s.  Tiles are ordered n-major
    // (all 128-row tiles of one token tile, then the next token tile) and every XCD gets a CONTIGUOUS chunk of that order, so
    // the workgroups resident on an XCD share one activation tile (L2-resident) instead of streaming several through 4 MB of L2.
    const int MT = (a.M + MROWS - 1) / MROWS, T = gridDim.x;
    int tile;
    { const int b = blockIdx.x, xcd = b & 7, li = b >> 3, q = T >> 3, r = T & 7;
      tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + li; }
    // Tile order: super-columns of G = a.m_major token tiles (a divisor of the token-tile count chosen by the host so that their
    // activations fit an XCD's L2); inside a super-column the weight tile is the outer index.  The workgroups resident on an XCD then
    // share G activation tiles AND run the same few weight tiles G times in a row: a weight tile is fetched once per super-column
    // instead of once per token tile (G = 1: n-major; G = all token tiles: m-major, the pp512 case).
    const int G = a.m_major > 1 ? a.m_major : 1;
    const int sc = tile / (G * MT), rr = tile - sc * G * MT;
    int m_tile = rr / G, n_tile = sc * G + (rr - m_tile * G);
    if (a.moe_tiles) {
        // Grouped form: the tile table is sized for the worst case and its unused entries sit at the END (Mixtral, 512 tokens: 17 entries, ~12 used), so contiguous chunks
        // of the n-major order would leave the last XCDs without work (2 of 8 idle).  Instead every XCD takes a BAND of weight-row tiles through ALL token tiles (token
        // tile outer, row tile inner: the workgroups resident on an XCD share a few activation tiles, every weight tile is still read once); grid = 8 * band * tiles per phase.
        // When the row tiles do not split into 8 equal bands (Qwen3-30B-A3B experts: 6 row tiles), RB < 8 bands x 8 / RB token phases: XCD (rb, tp) takes band rb of the
        // token tiles n = tp mod 8 / RB -- used tiles come first in the table, so the phases get equal shares of them (+- 1).
        const int RB = a.moe_rb, TP = 8 / RB, xcd = blockIdx.x & 7, rb = xcd % RB, tp = xcd / RB;
        const int band = (MT + RB - 1) / RB, li = blockIdx.x >> 3, nl = li / band;
        n_tile = nl * TP + tp; m_tile = rb * band + (li - nl * band);
        if (m_tile >= MT || n_tile >= a.N) return;
    }
    int n0 = n_tile * BN, n_valid = a.N - n0; long eoff = 0, expert = 0;
    if (a.moe_tiles) {                                   // grouped form: this token tile belongs to one expert
        const int e = a.moe_tiles[3 * n_tile];
        if (e < 0) return;
        if (a.expert_hi > 0 && (e < a.expert_lo || e >= a.expert_hi)) return;
        n0 = a.moe_tiles[3 * n_tile + 1]; n_valid = a.moe_tiles[3 * n_tile + 2]; eoff = (long)(e - a.expert_lo) * a.expert_stride; expert = e;
    }
    const int m0 = m_tile * MROWS + wave * 32;
    int mrow = m0 + (lane & 31); const bool m_ok = mrow < a.M && !producer; if (mrow >= a.M) mrow = a.M - 1;
#ifdef GEMM_EXP_SAME_ROWS                  /* timing experiment: every workgroup streams the same 32 rows (weights always cache-resident) */
    mrow = lane & 31;
#endif
    const uint8_t *Abase = a.A; float *Cbase = a.C; long ldc = a.stride_C;
    if (a.nmat > 1) {                                    // per-lane (matrix, local row)
        Abase = a.Am[0]; Cbase = a.Cm[0]; ldc = a.stride_Cm[0]; int lrow = mrow;
#pragma unroll
        for (int i = 1; i < GEMM_MAX_MATS; ++i) if (i < a.nmat && mrow >= a.mend[i - 1]) { Abase = a.Am[i]; Cbase = a.Cm[i]; ldc = a.stride_Cm[i]; lrow = mrow - a.mend[i - 1]; }
        mrow = lrow;
    }
    const uint8_t *wrow = Abase + eoff + (long)mrow * a.strideA, *wrow2 = UPGATE ? a.A2 + eoff + (long)mrow * a.strideA : nullptr;
    const int KT_all = a.K >> 7, kt_per = (KT_all + gridDim.z * KS - 1) / (gridDim.z * KS);
    const int kt_begin = (blockIdx.z * KS + kg) * kt_per, kt_end = min(KT_all, kt_begin + kt_per);
    if (KS == 1 && kt_begin >= kt_end) return;                    // (KS == 2: the host guarantees both halves are non-empty and equal)

    floatx16 acc[NT], acc2[UPGATE ? NT : 1];
#pragma unroll
    for (int t = 0; t < NT; ++t) { for (int r = 0; r < 16; ++r) { acc[t][r] = 0.f; if (UPGATE) acc2[t][r] = 0.f; } }

    void *grid_lds = smem + NBUF * KS * XT_BYTES;      // expanded IQ2_S / IQ3_S codebook behind the activation buffers
    if (TYPE == T_IQ2_S) expand_iq2s_grid(a.grid, grid_lds);
    if (TYPE == T_IQ3_S) expand_iq3s_grid(a.grid, grid_lds);
    if (TYPE == T_IQ2_XXS) expand_iq2_grid(a.grid, 256, grid_lds);
    if (TYPE == T_IQ2_XS) expand_iq2_grid(a.grid, 512, grid_lds);
    if (TYPE == T_IQ3_XXS) expand_iq3xxs_grid(a.grid, grid_lds);
    if (TYPE == T_IQ1_S || TYPE == T_IQ1_M) expand_iq1_grid(a.grid, grid_lds, false);

    // activation staging: LDS slot L (16-byte units) = i*256 + tid ; row = L / PIECES ; the slot's piece index is XOR-swizzled:
    //   KX = 128 (256-byte rows, all rows alias the same banks):      piece' = piece ^ (row & 15)
    //   KX =  64 (128-byte rows alias every 2 rows on the 64 banks):  piece' = piece ^ ((row >> 1) & 7)
    // => any 16 consecutive rows hit 16 distinct 16-byte bank groups: conflict-free ds_read_b128 (PMC: SQ_LDS_BANK_CONFLICT = 0).
    // (i*256 + tid) / PIECES = i * (256 / PIECES) + tid / PIECES, and 256 / PIECES is a multiple of 16 => the swizzle is the same for every i.
    const int xrow0 = tg / PIECES;
    const int xsw = KX == 128 ? (xrow0 & 15) : ((xrow0 >> 1) & 7);
    const int xpiece = (tg & (PIECES - 1)) ^ xsw;
    uint8_t *xbuf = smem + kg * NBUF * XT_BYTES;                  // this K-group's activation buffers
    // global side: slab layout X16[k / 64][row][64] (convert.cuh) -- the tile rows of one slab are contiguous
    const long slab_bytes = a.xrows * 128, xtile_step = (KX / 64) * slab_bytes;
    const char *xthread = reinterpret_cast<const char *>(a.X) + (xpiece >> 3) * slab_bytes + (long)(n0 + xrow0) * 128 + (xpiece & 7) * 16;
    constexpr long xstep = (WGT / PIECES) * 128;
    // Tiles go global -> LDS directly (global_load_lds_dwordx4: LDS address = wave-uniform base + 16 * lane, which is exactly the slot
    // order above; the swizzle sits on the per-lane SOURCE address).  Staging through VGPRs + ds_write_b128 cost a third of the
    // kernel: the store path moves <= 79 B/clk/CU (MI355X guide, LDS table) against 256 B/clk for the reads.
    typedef __attribute__((address_space(3))) void lds_void_t;
    typedef const __attribute__((address_space(1))) void glb_void_t;
    uint8_t *xwave = xbuf + wave * 1024;
    // LDS byte offset of this wave's first slot as an SGPR (low 32 bits of the flat address): the DMA destination is wave-uniform, but
    // derived from threadIdx the compiler re-derives it with v_readfirstlane for every piece
    const uint32_t xwave_s = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)xwave);
#ifdef GEMM_EXP_NO_XSTORE
#define X_ISSUE1(I_, XT_, BUF_) (void)xwave
#else
#define X_ISSUE1(I_, XT_, BUF_) __builtin_amdgcn_global_load_lds((glb_void_t *)(xthread + (I_) * xstep + (long)(XT_) * xtile_step),              \
                                                                 (lds_void_t *)(uintptr_t)(xwave_s + (BUF_) * XT_BYTES + (I_) * (4096 * MW)), 16, 0, 0)
#endif

    const int xt_last = NSUB * kt_end - 1;
    const uint8_t *xlane = xbuf + (lane & 31) * ROWB;
    const int hx = (WTile<TYPE>::HBIT * h) ^ (KX == 128 ? (lane & 15) : ((lane >> 1) & 7));     // lane-constant part of the swizzled piece index

    // Experiment knobs (scripts/gemm_exp.py builds variants with -D...; results are WRONG with any of them set):
    //   GEMM_EXP_NO_DEQUANT  B fragments are raw bits (no VALU de-quantization)      GEMM_EXP_NO_AREAD  one A-fragment read per tile
    //   GEMM_EXP_NO_XSTORE   activation tiles are never fetched / written to LDS
#ifdef GEMM_EXP_NO_DEQUANT
#define W_FRAG(W_, S_) exp_raw_frag((W_), (S_))
#else
#define W_FRAG(W_, S_) (W_).frag((S_), h)
#endif
#ifdef GEMM_EXP_NO_AREAD
#define A_PIECE(S_) 0
#else
#define A_PIECE(S_) (WTile<TYPE>::kpiece(S_))
#endif
    // LDS-DMA cadence: one piece per ISSUE_EVERY MFMAs, all NXR pieces within the FIRST HALF of the tile's MFMAs -- the barrier that ends
    // the tile waits for the last piece to land (vmcnt(0)), so a piece issued next to the barrier exposes its whole latency.
    // (measured: plain GEMMs +1-3 %; the fused kernel -- two MFMAs per iteration, i.e. already one piece per 8 MFMAs -- LOSES 10-20 % with
    //  the denser issue and keeps the even spread)
    constexpr int TM_ = SPS * NT, ISSUE_EVERY = UPGATE ? TM_ / NXR : ((TM_ / NXR) >= 4 ? (TM_ / NXR) / 2 : 1);
    // one activation tile worth of MFMAs: B fragments de-quantized from registers, A fragments ds_read_b128 one k-step ahead.
    // The NXR pieces of the NEXT tile are issued one per 4 (fused: 8) MFMAs: eight global_load_lds back to back stall the wave's
    // issue for ~100 clk each (MI355X guide, LDS-DMA issue cost), spread out they ride under the matrix pipe.
#define COMPUTE_TILE(W0_, V0_, XB_, S_BASE_, XTN_, XBN_, FETCH_)                                                                                          \
    {   half8 af[2][NT];                                                                                                              \
        { const int poff0 = (((A_PIECE(S_BASE_)) & (PIECES - 1)) ^ hx) << 4;                                              \
          _Pragma("unroll") for (int t = 0; t < NT; ++t) af[0][t] = *reinterpret_cast<const half8 *>((XB_) + t * (32 * ROWB) + poff0); } \
        _Pragma("unroll") for (int s4 = 0; s4 < SPS; ++s4) {                                                                          \
            const int s = (S_BASE_) + s4;                                                                                             \
            if (s4 < SPS - 1) {                                                                                                       \
                const int poffn = (((A_PIECE(s + 1)) & (PIECES - 1)) ^ hx) << 4;                                          \
                _Pragma("unroll") for (int t = 0; t < NT; ++t) af[(s4 + 1) & 1][t] = *reinterpret_cast<const half8 *>((XB_) + t * (32 * ROWB) + poffn); \
            }                                                                                                                         \
            const half8 bf = W_FRAG(W0_, s);                                                                                        \
            half8 bf2; if (UPGATE) bf2 = W_FRAG(V0_, s);                                                                            \
            _Pragma("unroll") for (int t = 0; t < NT; ++t) {                                                                          \
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[s4 & 1][t], bf, acc[t], 0, 0, 0);                                  \
                if (UPGATE) acc2[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[s4 & 1][t], bf2, acc2[t], 0, 0, 0);                   \
                if ((((s4 * NT + t) & (ISSUE_EVERY - 1)) == (ISSUE_EVERY > 1 ? 1 : 0)) && ((s4 * NT + t) / ISSUE_EVERY < NXR) && (FETCH_)) { X_ISSUE1((s4 * NT + t) / ISSUE_EVERY, XTN_, XBN_); } \
            }                                                                                                                         \
        }                                                                                                                             \
    }
    // Partly filled token tiles (PART): only the first nt_live 32-token sub-tiles are read and multiplied (wave-uniform branches).  A loop of its own, entered per
    // workgroup.  Plain launches keep the one-k-step look-ahead of the A fragments; the fused instance (at the register limit) reads them right before their MFMAs.
#define COMPUTE_TILE_PART(W0_, V0_, XB_, S_BASE_, XTN_, XBN_, FETCH_)                                                                                     \
    {   half8 afp[2][UPGATE ? 1 : NT];                                                                                                \
        if (!UPGATE) { const int poff0 = (((A_PIECE(S_BASE_)) & (PIECES - 1)) ^ hx) << 4;                                 \
          _Pragma("unroll") for (int t = 0; t < NT; ++t) if (t < nt_live) afp[0][UPGATE ? 0 : t] = *reinterpret_cast<const half8 *>((XB_) + t * (32 * ROWB) + poff0); } \
        _Pragma("unroll") for (int s4 = 0; s4 < SPS; ++s4) {                                                                          \
            const int s = (S_BASE_) + s4;                                                                                             \
            const int poff = (((A_PIECE(s)) & (PIECES - 1)) ^ hx) << 4;                                                   \
            if (!UPGATE && s4 < SPS - 1) {                                                                                            \
                const int poffn = (((A_PIECE(s + 1)) & (PIECES - 1)) ^ hx) << 4;                                          \
                _Pragma("unroll") for (int t = 0; t < NT; ++t) if (t < nt_live) afp[(s4 + 1) & 1][UPGATE ? 0 : t] = *reinterpret_cast<const half8 *>((XB_) + t * (32 * ROWB) + poffn); \
            }                                                                                                                         \
            const half8 bf = W_FRAG(W0_, s);                                                                                        \
            half8 bf2; if (UPGATE) bf2 = W_FRAG(V0_, s);                                                                            \
            _Pragma("unroll") for (int t = 0; t < NT; ++t) {                                                                          \
                if (t < nt_live) {                                                                                                    \
                    half8 a1; if (UPGATE) a1 = *reinterpret_cast<const half8 *>((XB_) + t * (32 * ROWB) + poff); else a1 = afp[s4 & 1][UPGATE ? 0 : t]; \
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, bf, acc[t], 0, 0, 0);                                         \
                    if (UPGATE) acc2[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, bf2, acc2[t], 0, 0, 0);                          \
                }                                                                                                                     \
                if ((((s4 * NT + t) & (ISSUE_EVERY - 1)) == (ISSUE_EVERY > 1 ? 1 : 0)) && ((s4 * NT + t) / ISSUE_EVERY < NXR) && (FETCH_)) { X_ISSUE1((s4 * NT + t) / ISSUE_EVERY, XTN_, XBN_); } \
            }                                                                                                                         \
        }                                                                                                                             \
    }
    const int nt_live = __builtin_amdgcn_readfirstlane(min(NT, (n_valid + 31) >> 5));

    if (PROD && producer) {
        // the source addresses of the four staging waves this wave stands in for (same slot order and swizzle as above)
        const char *xsrc[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int tgv = 64 * v + lane, r0 = tgv / PIECES, sw = KX == 128 ? (r0 & 15) : ((r0 >> 1) & 7), pc = (tgv & (PIECES - 1)) ^ sw;
            xsrc[v] = reinterpret_cast<const char *>(a.X) + (pc >> 3) * slab_bytes + (long)(n0 + r0) * 128 + (pc & 7) * 16;
        }
        const uint32_t xb_s = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)xbuf);
        auto issue = [&](int xt, int buf) {
#pragma unroll
            for (int i = 0; i < NXR; ++i)
#pragma unroll
                for (int v = 0; v < 4; ++v)
                    __builtin_amdgcn_global_load_lds((glb_void_t *)(xsrc[v] + i * xstep + (long)xt * xtile_step), (lds_void_t *)(uintptr_t)(xb_s + v * 1024 + buf * XT_BYTES + i * 4096), 16, 0, 0);
        };
        constexpr int VMN = 4 * NXR;                              // pieces of ONE tile: what may stay in flight across a barrier
        static_assert(VMN <= 32, "two tiles of pieces must fit the 6-bit vmcnt");
        constexpr int WAIT_ONE = (VMN & 15) | ((VMN >> 4) << 14) | (7 << 4) | (15 << 8), WAIT_ALL = (7 << 4) | (15 << 8);
        const int xt0 = NSUB * kt_begin, n_tiles = xt_last - xt0 + 1;
        issue(xt0, 0);
        if (n_tiles > 1) issue(xt0 + 1, 1);
        int buf = 2;
        for (int t = 0; t < n_tiles; ++t) {
            if (t + 1 < n_tiles) __builtin_amdgcn_s_waitcnt(WAIT_ONE); else __builtin_amdgcn_s_waitcnt(WAIT_ALL);      // tile t has landed (tile t + 1 may be in flight)
            __builtin_amdgcn_s_barrier();                         // the compute waves start tile t; they are done with tile t - 1 = the buffer tile t + 2 goes to
            if (t + 2 < n_tiles) { issue(xt0 + t + 2, buf); buf = buf == 2 ? 0 : buf + 1; }
        }
    } else {
    WTile<TYPE> w0, w1, v0, v1;               // weight tiles kt, kt+1 ; v* = gate weights for fused up*gate
    if (stager) {
#pragma unroll
        for (int i_ = 0; i_ < NXR; ++i_) { X_ISSUE1(i_, NSUB * kt_begin, 0); }
    }
    w0.load(wrow, kt_begin, h); if (UPGATE) v0.load(wrow2, kt_begin, h);
    int p = 0;
#ifdef GEMM_EXP_NO_WLOAD                     /* timing experiment: one weight tile for the whole K loop */
#define W_NEXT(KTN_) w1 = w0; if (UPGATE) v1 = v0; (void)(KTN_)
#else
#define W_NEXT(KTN_) w1.load(wrow, (KTN_), h); if (UPGATE) v1.load(wrow2, (KTN_), h)
#endif
#ifdef GEMM_EXP_TIMELINE
#define MT_TILE { MT_STAMP(tl_i) ++tl_i; }
#else
#define MT_TILE
#endif
#define K_LOOP(COMPUTE_)                                                                                                              \
    for (int kt = kt_begin; kt < kt_end; ++kt) {                                                                                      \
        _Pragma("unroll") for (int hh = 0; hh < NSUB; ++hh) {                                                                         \
            __syncthreads();                   /* (carries vmcnt(0)) tile in buffer p has landed for every wave; nobody reads buffer p^1 any more */ \
            MT_TILE                                                                                                                   \
            const int xtn = NSUB * kt + hh + 1; const bool fetch = stager && xtn <= xt_last;                                          \
            if (hh == 0) {                                                                                                            \
                const int ktn = min(kt + 1, kt_end - 1);                                                                              \
                W_NEXT(ktn);                                                                                                          \
                w0.prepare(h, grid_lds); if (UPGATE) v0.prepare(h, grid_lds);                                                         \
            }                                                                                                                         \
            COMPUTE_(w0, v0, xlane + p * XT_BYTES, SPS * hh, xtn, p ^ 1, fetch)                                                       \
            p = NBUF == 3 ? (p == 2 ? 0 : p + 1) : (p ^ 1);                                                                           \
        }                                                                                                                             \
        w0 = w1; if (UPGATE) v0 = v1;                                                                                                 \
    }
    if (!PART || nt_live == NT) { K_LOOP(COMPUTE_TILE) } else { K_LOOP(COMPUTE_TILE_PART) }
    }
    MT_STAMP(1)
#undef K_LOOP
#undef W_NEXT
#undef COMPUTE_TILE_PART
#undef COMPUTE_TILE
#undef W_FRAG
#undef A_PIECE
#undef X_ISSUE1
    if (KS == 2) {             // add the second K-half's accumulators through LDS (the activation buffers are free now: 2*KS*XT_BYTES >= 128 KiB at NT = 8)
        float *red = reinterpret_cast<float *>(smem);
        constexpr int CH = (2 * KS * XT_BYTES) / (4 * 64 * 4 * 16);          // token tiles that fit per pass: [4 waves][CH][16][64 lanes] floats
        static_assert(CH >= 1, "reduction scratch");
        for (int t0 = 0; t0 < NT; t0 += CH) {
            __syncthreads();
            if (kg == 1) {
#pragma unroll
                for (int t = 0; t < CH; ++t) if (t0 + t < NT) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[((wave * CH + t) * 16 + r) * 64 + lane] = acc[t0 + t][r];
                }
            }
            __syncthreads();
            if (kg == 0) {
#pragma unroll
                for (int t = 0; t < CH; ++t) if (t0 + t < NT) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t0 + t][r] += red[((wave * CH + t) * 16 + r) * 64 + lane];
                }
            }
        }
    }
    MT_STAMP(2)
    // epilogue: C[token][row]; lanes 0..31 of a register hold 32 consecutive weight rows -> 128-byte stores.
    // The per-token values of the tile -- the range-guard scale and, grouped, the result row of the pair -- are staged in LDS ONCE.  Read from global memory per element
    // (round 1 - 3: a global_load_dword + s_waitcnt vmcnt(0) in front of each of the 16 x NT stores of a lane) they serialize the epilogue: on gfx9 vmcnt also counts the
    // STORES, so every element waited for the previous element's store (or atomic) to complete -- 64 dependent memory round trips per lane at the end of every workgroup.
    float *xs_lds = reinterpret_cast<float *>(smem); int *pr_lds = reinterpret_cast<int *>(smem) + BN;
    __syncthreads();                                          // (every wave is past its last read of the activation buffers / the K-half reduction scratch)
    for (int i = threadIdx.x; i < BN; i += blockDim.x) {
        xs_lds[i] = (a.xscale && i < n_valid) ? a.xscale[n0 + i] : 1.f;
        if (a.moe_pairs) pr_lds[i] = i < n_valid ? a.moe_pairs[n0 + i] : 0;
    }
    __syncthreads();
    const bool live = !(KS == 2 && kg == 1);                  // (the second K-half's waves have handed their accumulators over; they stay for the barriers below)
    if (!UPGATE && gridDim.z > 1 && a.ks_ws) {
        // K split over grid.z, deterministic and without fences: every slice stores its partial tile WRITE-THROUGH (16-byte sc1 stores, fragment order: the reader has the
        // same lane mapping, so the slab needs no C layout and every store instruction covers 1 KiB contiguous), drains them (vmcnt(0): they have left the XCD), then ONE relaxed
        // agent-scope ticket per workgroup; the workgroup that draws the last ticket of the tile reads all slices back with sc1 loads, adds them in SLICE ORDER -- the same
        // order whoever arrives last: bit-reproducible prompts -- and writes C.  The counter re-arms itself (graph replays included).  MI355X guide, "in-launch split-K
        // reduction": the fenced form of rounds 2-3 (release fence + acquire fence: two whole-L2 operations per workgroup) cost 25-30 us per launch, f32 atomics into a
        // zero-filled C cost a 5.6 us fill launch per mat-mul and were not reproducible.
        constexpr int WPK = 4 * MW;                           // waves that hold accumulators
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const long slab = (long)MROWS * BN;                   // floats per (slice, tile)
        float *part = a.ks_ws + ((long)blockIdx.z * gridDim.x + tile) * slab;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(part, 0, (int)(slab * 4), 0x00020000);
        if (live) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    u32x4 v; v[0] = __float_as_uint(acc[t][4 * j]); v[1] = __float_as_uint(acc[t][4 * j + 1]); v[2] = __float_as_uint(acc[t][4 * j + 2]); v[3] = __float_as_uint(acc[t][4 * j + 3]);
                    __builtin_amdgcn_raw_buffer_store_b128(v, rs, (((t * 4 + j) * WPK + wave) * 64 + lane) * 16, 0, 16);      // aux 16 = sc1
                }
            }
        }
        int *s_last = reinterpret_cast<int *>(smem) + 2 * BN; // (behind the staged per-token values)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        MT_STAMP(3)
        __syncthreads();
        if (threadIdx.x == 0) {
            // fenced fallback (a.ks_fence; guide: release, THEN the ticket, with the wait behind buffer_wbl2 restated in asm -- hipcc drops its own when the scoreboard is empty)
            if (a.ks_fence) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            const unsigned old = __hip_atomic_fetch_add(a.ks_cnt + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = old == gridDim.z - 1;
            if (last) __hip_atomic_store(a.ks_cnt + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (last && a.ks_fence) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            *s_last = last;
        }
        __syncthreads();
        MT_STAMP(4)
        if (!*s_last || !live) return;
        if constexpr (NT <= 4) {
            // two slices (every split launch of an 8B-class model at 512 tokens): the last arriver still HOLDS its own slice -- only the other slab is read back, every load of it
            // issued before the first is consumed (round 6 timeline, 4096 x 4096 x 512: the generic loop below -- both slabs, four loads per round trip -- took 7.4 us of a 35 us
            // launch).  Same sums in the same order as the generic loop: (0 + slice 0) + slice 1.
            if (gridDim.z == 2) {
                const unsigned zo = 1u - blockIdx.z;
                const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(a.ks_ws + ((long)zo * gridDim.x + tile) * slab, 0, (int)(slab * 4), 0x00020000);
                u32x4 p[NT][4];
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int j = 0; j < 4; ++j) p[t][j] = __builtin_amdgcn_raw_buffer_load_b128(rz, (((t * 4 + j) * WPK + wave) * 64 + lane) * 16, 0, 16);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float o = __uint_as_float(p[t][r >> 2][r & 3]), s0 = blockIdx.z == 0 ? acc[t][r] : o, s1 = blockIdx.z == 0 ? o : acc[t][r];
                        const int tr = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
                        if (m_ok && tr < n_valid) Cbase[(long)(n0 + tr) * ldc + mrow] = ((0.f + s0) + s1) * xs_lds[tr];
                    }
                }
#ifdef GEMM_EXP_TIMELINE
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                MT_STAMP(5)
#endif
                return;
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = 0.f;
            for (unsigned z = 0; z < gridDim.z; ++z) {         // slice order
                const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(a.ks_ws + ((long)z * gridDim.x + tile) * slab, 0, (int)(slab * 4), 0x00020000);
                u32x4 p[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) p[j] = __builtin_amdgcn_raw_buffer_load_b128(rz, (((t * 4 + j) * WPK + wave) * 64 + lane) * 16, 0, 16);
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[4 * j] += __uint_as_float(p[j][0]); v[4 * j + 1] += __uint_as_float(p[j][1]); v[4 * j + 2] += __uint_as_float(p[j][2]); v[4 * j + 3] += __uint_as_float(p[j][3]); }
            }
            if (m_ok) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int tr = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (tr < n_valid) Cbase[(long)(n0 + tr) * ldc + mrow] = v[r] * xs_lds[tr];
                }
            }
        }
        #ifdef GEMM_EXP_TIMELINE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        MT_STAMP(5)
#endif
        return;
    }
    if (!live) return;
    if (m_ok) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int tr = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;          // row inside the token tile
                if (tr < n_valid) {
                    const float xs = xs_lds[tr];                                      // undo the f16 range-guard scale of this token (exact: a power of two)
                    acc[t][r] *= xs; if (UPGATE) acc2[t][r] *= xs;
                    float *dst;
                    if (a.moe_pairs) { const int pr = pr_lds[tr]; const int tk = pr / a.n_used; dst = Cbase + (long)tk * a.nb2 + (long)(pr - tk * a.n_used) * a.nb1 + mrow; }
                    else dst = Cbase + (long)(n0 + tr) * ldc + mrow;
                    if (UPGATE) *dst = up_gate_combine(a.unary_op, acc[t][r], acc2[t][r], a.epi, mrow, expert);
                    else if (gridDim.z > 1) unsafeAtomicAdd(dst, acc[t][r]);          // (default K-split form: hardware f32 atomics into a zero-filled C)
                    else *dst = acc[t][r];
                }
            }
        }
    }
#ifdef GEMM_EXP_TIMELINE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MT_STAMP(5)
#endif
}

template <int TYPE, int NT, bool UPGATE, int KS, int MW = 1, bool PART = false>
static int launch_gemm_ks(const GemmArgs &a_in, int ksplit, hipStream_t st) {
    // 64 KiB of activation buffers per K-group (2 buffers): 256-token tiles stage 64 k at a time, narrower ones 128 k
    // (one barrier per >= 32 MFMAs either way; measured: 16 MFMAs per barrier costs ~20 %)
    constexpr int KX = NT >= 8 ? 64 : 128;
    const size_t lds = (size_t)KS * 2 * 32 * NT * KX * 2 + gemm_grid_lds_bytes(TYPE);
    if (lds > 64 * 1024 && cdna4_opt_in_lds((const void *)gemm_mfma_kernel<TYPE, NT, UPGATE, KX, KS, MW, 0, PART>) != 0) return -2;
    GemmArgs a = a_in;
    const long ntl = a.moe_tiles ? a.N : (a.N + 32 * NT - 1) / (32 * NT);      // grouped form: a.N carries the (worst-case) tile count
    {   // super-column width: the largest divisor of the token-tile count whose activations (G x 32*NT tokens x K f16) fit the budget
        static const long budget = (getenv("CDNA4_GEMM_XBUDGET_MB") ? atol(getenv("CDNA4_GEMM_XBUDGET_MB")) : 4) << 20;
        const long tile_bytes = 32L * NT * a.K * 2; long G = 1;
        if (!a.moe_tiles) for (long d = 1; d <= ntl; ++d) if (ntl % d == 0 && d * tile_bytes <= budget) G = d;
        a.m_major = (int)G;
    }
    { const int KTa = a.K >> 7; while (ksplit > 1 && (ksplit - 1) * ((KTa + ksplit - 1) / ksplit) >= KTa) ksplit >>= 1; }      // every K slice must own at least one tile (an empty slice would never arrive at the tile's counter)
    // two ways to add the K slices (a.ks_ws set by the host = deterministic mode, cdna4_set_deterministic): partial tiles through the workspace, added in slice order by
    // the last arriver -- bit-reproducible, but the release / acquire fences and the extra pass cost 25-30 us per launch on the 4096-row matrices at 512 tokens (58.9 ->
    // 88.1 us, profiles/r03_notes.md); or f32 hardware atomics into a zero-filled C (default: sums of the same terms in arrival order).
    if (ksplit > 1 && !a.ks_ws) {
        if (a.stride_C != a.M) ksplit = 1;
        else if (hipMemsetAsync(a.C, 0, (size_t)a.N * a.M * sizeof(float), st) != hipSuccess) return -2;
    } else
    if (ksplit > 1 && (!a.ks_ws || (size_t)ksplit * (size_t)(((a.M + 128 * MW - 1) / (128 * MW)) * ntl) * (128 * MW) * (32 * NT) * sizeof(float) > a.ks_ws_bytes || ((a.M + 128 * MW - 1) / (128 * MW)) * ntl > CDNA4_KS_MAX_TILES)) ksplit = 1;      // (no room for the tile-padded partial slabs: unsplit)
    const long mt_wg = (a.M + 128 * MW - 1) / (128 * MW);
    long grid_x = mt_wg * ntl;
    if (a.moe_tiles) {      // grouped: RB bands of row tiles x 8 / RB token phases over the 8 XCDs (see the kernel): the most bands that divide the row tiles evenly
        int rb = 8; while (rb > 1 && mt_wg % rb) rb >>= 1;
        static const int env_rb = getenv("CDNA4_MOE_RB") ? atoi(getenv("CDNA4_MOE_RB")) : 0;      // (developer A/B knob)
        if (env_rb) { rb = env_rb; while (rb > 1 && mt_wg % rb) rb >>= 1; }
        a.moe_rb = rb; grid_x = 8L * ((mt_wg + rb - 1) / rb) * ((ntl + 8 / rb - 1) / (8 / rb));
    }
    const dim3 grid((unsigned)grid_x, 1, (unsigned)ksplit);
    auto note = [&](int xw, long gx, int gz) { cdna4_note_launch("gemm_mfma type=%d nt=%d upgate=%d kx=%d ks=%d mw=%d xw=%d part=%d grid=%ldx1x%d ksplit=%d g=%d", TYPE, NT, (int)UPGATE, KX, KS, MW, xw, (int)PART, gx, gz, ksplit, a.m_major); };
    if constexpr (KS == 1 && MW == 1 && NT == 4) {
        // 224-row tiles when they cover the matrix exactly and give every CU exactly one workgroup per round (14336 rows x 512 tokens: 64 x 4 = 256)
        constexpr int env_xw = 1;
        const long wg7 = (a.M / 224) * ntl;
        int ncu = 0; (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
        if (env_xw && !a.moe_tiles && a.nmat <= 1 && ksplit == 1 && a.M % 224 == 0 && ncu > 0 && wg7 % ncu == 0 && (((a.M + 127) / 128) * ntl) % ncu != 0) {
            // (measured at 14336 x 4096 x 512 fused: Q4_K 167.9 -> 160.7 us, IQ4_NL 217.1 -> 214.1, Q6_K 211.9 -> 215.4: used where it won)
            // (instantiated for those two types only)
            constexpr int env_prod = 1;
            if constexpr (TYPE == T_Q4_K || TYPE == T_IQ4_NL) if (env_prod) {
                const size_t lds3 = (size_t)3 * 32 * NT * KX * 2 + gemm_grid_lds_bytes(TYPE);
                if (cdna4_opt_in_lds((const void *)gemm_mfma_kernel<TYPE, NT, UPGATE, KX, KS, MW, 4>) != 0) return -2;
                hipLaunchKernelGGL((gemm_mfma_kernel<TYPE, NT, UPGATE, KX, KS, MW, 4>), dim3((unsigned)wg7, 1, 1), dim3(512), lds3, st, a);
                note(4, wg7, 1);
                return 0;
            }
            if (lds > 64 * 1024 && cdna4_opt_in_lds((const void *)gemm_mfma_kernel<TYPE, NT, UPGATE, KX, KS, MW, 3>) != 0) return -2;
            hipLaunchKernelGGL((gemm_mfma_kernel<TYPE, NT, UPGATE, KX, KS, MW, 3>), dim3((unsigned)wg7, 1, 1), dim3(256 + 192), lds, st, a);
            note(3, wg7, 1);
            return 0;
        }
    }
    hipLaunchKernelGGL((gemm_mfma_kernel<TYPE, NT, UPGATE, KX, KS, MW, 0, PART>), grid, dim3(256 * KS * MW), lds, st, a);
    note(0, grid_x, ksplit);
    return 0;
}
template <int TYPE, int NT, bool UPGATE>
static int launch_gemm_nt(const GemmArgs &a, int ksplit, hipStream_t st) { return launch_gemm_ks<TYPE, NT, UPGATE, 1>(a, ksplit, st); }

// padded token count the activation workspace must hold for a given N
static inline long gemm_mfma_npad(long N) { return (N + 255) & ~255L; }


template <int TYPE>
static int launch_gemm_type(int num_cu, const GemmArgs &a, hipStream_t st) {
    // Token tile as wide as possible: every B fragment (~26 VALU ops of dequant) is reused by NT MFMAs.  When the
    // (rows x tokens) grid cannot fill the chip, split K over grid.z (atomic f32 accumulate) rather than shrinking tiles.
    const long mt = (a.M + 127) / 128; const int KT = a.K >> 7;
    int nt = a.A2 ? 4 : 8;                                         // fused up*gate keeps two accumulator sets
    while (nt > 1 && a.N <= 16 * nt) nt >>= 1;
    // measured on MI355X (profiles/r01_microbench.md): the 256-token tile wins only when it still yields ~2 workgroups
    // per CU (2 waves / SIMD); otherwise the 128-token tile with twice the workgroups is faster.
    auto n_wgs = [&](int t) { return mt * ((a.N + 32 * t - 1) / (32 * t)); };
    // 256-row workgroup tiles (8 waves on one activation tile) once that grid still gives every CU a workgroup (4k-token prompts)
    static const int env_mw = getenv("CDNA4_GEMM_MW") ? atoi(getenv("CDNA4_GEMM_MW")) : 0;
    if (!a.moe_tiles && (env_mw ? env_mw == 2 : true)) {
        const int t = a.A2 ? 4 : 8;
        const long wg2 = ((a.M + 255) / 256) * ((a.N + 32 * t - 1) / (32 * t));
        const double eff = (double)wg2 / (double)(((wg2 + num_cu - 1) / num_cu) * num_cu);       // fill of the last round of workgroups
        // measured (Q4_K, N = 4096): 4096 x 4096 777 -> 850 TF, 4096 x 14336 831 -> 862 TF (256 workgroups = one full round);
        // 14336 x 4096 798 -> 776 TF (896 workgroups = 3.5 rounds): only taken when the rounds come out even
        if (a.N >= 32 * t && (env_mw == 2 || (wg2 >= num_cu && eff >= 0.95))) {
            if (a.A2) return launch_gemm_ks<TYPE, 4, true, 1, 2>(a, 1, st);
            return launch_gemm_ks<TYPE, 8, false, 1, 2>(a, 1, st);
        }
    }
    // 256-token tiles with an intra-workgroup K split (8 waves): same waves per CU as two 4-wave workgroups, half the dequant work
    if (!a.A2 && nt == 8 && a.nmat <= 1 && !a.moe_tiles && n_wgs(8) < (long)(1.75 * num_cu) && n_wgs(8) >= num_cu / 2 && (KT % 2) == 0 && KT >= 8 && a.N > 128)
        return launch_gemm_ks<TYPE, 8, false, 2>(a, 1, st);
    while (nt > 1 && n_wgs(nt) < (long)(1.75 * num_cu) && a.N > 16 * nt) nt >>= 1;
    constexpr int env_nt_min = 0;
    // never below 128 tokens when the batch has them (dequant-bound) -- unless that grid leaves most of the chip idle (tensor-parallel shards: 3584 x 8192 fused = 112 workgroups)
    // (fewer than half a workgroup per CU; at exactly num_cu / 2 -- 4096-row matrices of an 8B model at 512 tokens -- the K split below is the better remedy: Q6_K 88 vs 113 us)
    const int nt_min = env_nt_min ? env_nt_min : (n_wgs(4) * 2 < (long)num_cu ? 2 : 4);
    if (nt < nt_min && a.N > 16 * nt_min) nt = nt_min;
    while (nt > 1 && a.N <= 16 * nt) nt >>= 1;
    if (a.A2 && nt > 4) nt = 4;
    const long wgs = n_wgs(nt);
    int ksplit = 1;
    static const int ks_mult = getenv("CDNA4_GEMM_KSPLIT_MULT") ? atoi(getenv("CDNA4_GEMM_KSPLIT_MULT")) : 1;
    if (!a.A2 && a.nmat <= 1 && !a.moe_tiles && !a.no_ksplit) { while (ksplit < 8 && wgs * ksplit < (long)num_cu * ks_mult && KT / (ksplit * 2) >= 4) ksplit *= 2; }
    if (a.A2) { switch (nt) { case 4: return launch_gemm_nt<TYPE, 4, true>(a, 1, st); case 2: return launch_gemm_nt<TYPE, 2, true>(a, 1, st); default: return launch_gemm_nt<TYPE, 1, true>(a, 1, st); } }
    // developer A/B knobs (scripts/mfma_timeline.py): token-tile width and grid-level K split of the plain single-matrix launches
    static const int env_nt = getenv("CDNA4_GEMM_NT") ? atoi(getenv("CDNA4_GEMM_NT")) : 0, env_ksplit = getenv("CDNA4_GEMM_KSPLIT") ? atoi(getenv("CDNA4_GEMM_KSPLIT")) : 0;
    if ((env_nt || env_ksplit) && a.nmat <= 1 && !a.moe_tiles) {
        if (env_nt == 8 || env_nt == 4 || env_nt == 2 || env_nt == 1) nt = env_nt;
        if (env_ksplit >= 1 && env_ksplit <= 8 && !a.no_ksplit) ksplit = env_ksplit;
        switch (nt) { case 8: return launch_gemm_nt<TYPE, 8, false>(a, ksplit, st); case 4: return launch_gemm_nt<TYPE, 4, false>(a, ksplit, st);
                      case 2: return launch_gemm_nt<TYPE, 2, false>(a, ksplit, st); default: return launch_gemm_nt<TYPE, 1, false>(a, ksplit, st); }
    }
    // 128-token tiles on a grid that gives every CU at most ONE workgroup (4096 x 4096 at 512 tokens: 128 tiles, K split in two = 256 workgroups of one wave per SIMD): the
    // 8-wave form -- two groups of four waves contract the two halves of the workgroup's K range with their own activation buffers, partial tiles added through LDS -- puts two
    // waves on every SIMD without further atomics
    // (measured, kernel times at 512 tokens: 4096 x 14336 Q6_K 128.2 -> 115.3 us, Q4_K 94.1 -> 87.8 us, 4096 x 4096 35.2 -> 34.9 us; CDNA4_GEMM_KS2_NT4=0 turns it off)
    static const int env_ks2 = getenv("CDNA4_GEMM_KS2_NT4") ? atoi(getenv("CDNA4_GEMM_KS2_NT4")) : 1;
    if (env_ks2 && nt == 4 && (a.nmat <= 1 || ksplit == 1) && !a.moe_tiles && wgs * ksplit <= (long)num_cu && (KT % (2 * ksplit)) == 0 && KT / (2 * ksplit) >= 2)
        return launch_gemm_ks<TYPE, 4, false, 2>(a, ksplit, st);
    switch (nt) { case 8: return launch_gemm_nt<TYPE, 8, false>(a, ksplit, st); case 4: return launch_gemm_nt<TYPE, 4, false>(a, ksplit, st);
                  case 2: return launch_gemm_nt<TYPE, 2, false>(a, ksplit, st); default: return launch_gemm_nt<TYPE, 1, false>(a, ksplit, st); }
}

// grouped launch for MUL_MAT_ID: NT fixed by the caller (it sized the tile table with it); a.N = number of token tiles
template <int TYPE>
static int launch_gemm_grouped(int nt, const GemmArgs &a, hipStream_t st) {
    // 128-token tiles: the instance that skips the unpopulated sub-tiles of an expert's last tile (PART)
    if (a.A2) { switch (nt) { case 4: return launch_gemm_ks<TYPE, 4, true, 1, 1, true>(a, 1, st); case 2: return launch_gemm_nt<TYPE, 2, true>(a, 1, st); default: return launch_gemm_nt<TYPE, 1, true>(a, 1, st); } }
    switch (nt) { case 4: return launch_gemm_ks<TYPE, 4, false, 1, 1, true>(a, 1, st); case 2: return launch_gemm_nt<TYPE, 2, false>(a, 1, st); default: return launch_gemm_nt<TYPE, 1, false>(a, 1, st); }
}
