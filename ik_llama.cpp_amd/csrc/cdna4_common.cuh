// cdna4_common.cuh -- shared device helpers for the gfx950 quantized mat-mul kernels.
// Block formats follow the reference's on-disk layouts (ggml/src/ggml-common.h:348-353 Q4_K, :367-373 Q5_K,
// :388-394 Q6_K, :468-474 IQ2_S, :503-510 IQ3_S, :586-590 IQ4_NL); they are addressed here as raw bytes.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#define CDNA4_WAVE 64

enum : int {
    T_F32 = 0, T_F16 = 1, T_Q4_0 = 2, T_Q4_1 = 3, T_Q5_0 = 6, T_Q5_1 = 7, T_Q8_0 = 8, T_Q2_K = 10, T_Q3_K = 11, T_Q4_K = 12, T_Q5_K = 13, T_Q6_K = 14, T_Q8_K = 15, T_IQ2_XXS = 16, T_IQ2_XS = 17, T_IQ3_XXS = 18, T_IQ1_S = 19, T_IQ4_NL = 20, T_IQ3_S = 21, T_IQ2_S = 22, T_IQ4_XS = 23, T_IQ1_M = 29, T_MXFP4 = 39,
    T_BF16 = 30, T_Q8_2_X4 = 99, T_Q6_0 = 133, T_IQ1_BN = 134, T_IQ2_BN = 135, T_Q8_K64 = 136,      /* BitNet: 13 / 16 bytes per 64 ternary weights behind an f16 / f32 ROW scale (ggml-common.h:561-576) */ T_IQ2_K = 137, T_IQ3_K = 138, T_IQ4_K = 139, T_IQ5_K = 140, T_IQ6_K = 141, T_IQ4_KS = 144, T_IQ2_KS = 145, T_IQ4_KSS = 146, T_IQ5_KS = 152, T_IQ3_KS = 156, T_IQ2_KL = 157, T_Q8_K32 = 148,
    T_IQ2_KT = 153, T_IQ3_KT = 154, T_IQ4_KT = 155, T_IQ1_KT = 158,      /* trellis types: an f32 ROW scale, then 256-blocks of 16-bit (IQ4_KT: 15-bit per 4) generator seeds (ggml-common.h:664-689) */
    T_Q4_K_R4 = 212, T_Q5_K_R4 = 213, T_Q6_K_R4 = 214, T_IQ4_NL_R4 = 220, T_IQ3_S_R4 = 221, T_IQ2_S_R4 = 222,
    T_PRETILED = 1000,       // _R4 id + 1000: an _R4 tensor whose bytes were un-interleaved to the base tiling at upload (CDNA4_TYPE_PRETILED)
};
// an _R4 tensor already stored in the base tiling: base-layout kernels, the _R4 kernels' activation arithmetic, no shadow copy
__host__ __device__ constexpr bool type_is_pretiled(int t) { return t >= 1200 && t < 1300; }

__host__ __device__ constexpr int type_block_bytes(int t) {
    if (t >= 1200 && t < 1300) t -= 1000;
    return (t == T_Q4_K || t == T_Q4_K_R4) ? 144 : (t == T_Q5_K || t == T_Q5_K_R4) ? 176 : (t == T_Q6_K || t == T_Q6_K_R4) ? 210
         : (t == T_IQ2_S || t == T_IQ2_S_R4) ? 82 : (t == T_IQ3_S || t == T_IQ3_S_R4) ? 110 : (t == T_IQ4_NL || t == T_IQ4_NL_R4 || t == T_Q4_0) ? 18 : (t == T_Q8_0) ? 34 : (t == T_IQ4_XS) ? 136
         : (t == T_IQ2_K) ? 76 : (t == T_IQ3_K) ? 110 : (t == T_IQ4_K) ? 144 : (t == T_IQ5_K) ? 176 : (t == T_IQ4_KS) ? 136 : (t == T_IQ5_KS) ? 168 : (t == T_IQ2_KS) ? 70 : (t == T_IQ3_KS) ? 102 : (t == T_IQ4_KSS) ? 128 : (t == T_IQ2_KL) ? 86 : (t == T_IQ6_K) ? 212
         : (t == T_Q4_1) ? 20 : (t == T_Q5_1) ? 24 : (t == T_Q6_0) ? 26 : (t == T_Q2_K) ? 84 : (t == T_Q3_K) ? 110
         : (t == T_IQ2_KT) ? 68 : (t == T_IQ3_KT) ? 100 : (t == T_IQ4_KT) ? 128 : (t == T_IQ1_KT) ? 56
         : (t == T_IQ1_S) ? 50 : (t == T_IQ1_M) ? 56 : (t == T_MXFP4) ? 17 : (t == T_IQ1_BN) ? 13 : (t == T_IQ2_BN) ? 16
         : (t == T_Q5_0) ? 22 : (t == T_IQ2_XXS) ? 66 : (t == T_IQ2_XS) ? 74 : (t == T_IQ3_XXS) ? 98
         : (t == T_Q8_K || t == T_Q8_K32) ? 296 : (t == T_Q8_2_X4) ? 36 : 0;
}
__host__ __device__ constexpr int type_block_elems(int t) { return (t == T_IQ1_BN || t == T_IQ2_BN) ? 64 : (t == T_IQ4_NL || t == T_IQ4_NL_R4 || t == T_IQ4_NL_R4 + 1000 || t == T_Q8_2_X4 || t == T_Q4_0 || t == T_Q8_0 || t == T_Q5_0 || t == T_Q4_1 || t == T_Q5_1 || t == T_Q6_0 || t == T_MXFP4) ? 32 : 256; }
// bytes in front of a row's blocks (type traits row_meta_size): the _KS types keep an f32 row scale there
__host__ __device__ constexpr int type_row_meta(int t) { return (t == T_IQ4_KS || t == T_IQ5_KS || t == T_IQ4_KSS || t == T_IQ2_BN || t == T_IQ2_KT || t == T_IQ3_KT || t == T_IQ4_KT || t == T_IQ1_KT) ? 4 : (t == T_IQ2_KS || t == T_IQ3_KS || t == T_IQ2_KL || t == T_IQ1_BN) ? 2 : 0; }
__host__ __device__ constexpr bool type_is_bitnet(int t) { return t == T_IQ1_BN || t == T_IQ2_BN; }
__host__ __device__ constexpr bool type_is_kt(int t) { return t == T_IQ2_KT || t == T_IQ3_KT || t == T_IQ4_KT || t == T_IQ1_KT; }
// what the reference's MAT-MUL kernels multiply the row scale of a trellis type by (iqk_gemm_ktquants.cpp:705,849 dptr[0] * 1.05f / 1.01f; mmq.cuh:3117,3194; convert.cu:396,417) --
// its scalar to_float (dequantize_row_iq2_kt, iqk_quantize.cpp:9751) leaves the factor out, so de-quantization and get_rows do too
__host__ __device__ constexpr float kt_matmul_factor(int t) { return t == T_IQ2_KT ? 1.05f : t == T_IQ3_KT ? 1.01f : 1.0f; }
__host__ __device__ constexpr bool type_is_r4(int t) { return t >= 200 && t < 300; }      // row-interleaved bytes (needs un-interleaving before the kernels)
__host__ __device__ constexpr int type_base(int t) {
    if (t >= 1200 && t < 1300) t -= 1000;
    return t == T_Q4_K_R4 ? T_Q4_K : t == T_Q5_K_R4 ? T_Q5_K : t == T_Q6_K_R4 ? T_Q6_K : t == T_IQ4_NL_R4 ? T_IQ4_NL
         : t == T_IQ2_S_R4 ? T_IQ2_S : t == T_IQ3_S_R4 ? T_IQ3_S : t;
}
// activation quant type of the CPU path (ggml.c type_traits vec_dot_type)
__host__ __device__ constexpr int type_vec_dot(int t) {
    if (t >= 1200 && t < 1300) t -= 1000;
    return (t == T_Q4_K || t == T_Q5_K || t == T_Q6_K || t == T_IQ4_NL || t == T_IQ4_NL_R4 || t == T_Q4_0 || t == T_Q8_0 || t == T_Q5_0 || t == T_Q4_1 || t == T_Q5_1 || t == T_Q6_0 || t == T_MXFP4 ||
            t == T_IQ2_KT || t == T_IQ3_KT || t == T_IQ4_KT || t == T_IQ1_KT) ? T_Q8_2_X4
         : (t == T_Q4_K_R4 || t == T_Q5_K_R4) ? T_Q8_K32 : (t == T_IQ1_BN || t == T_IQ2_BN) ? T_Q8_K64 : T_Q8_K;
}

// packed codebooks in the context's grid image (uint16 units; iq_grids_packed.inc, scripts/gen_iq_tables.py)
constexpr int GRID_IQ2S = 0, GRID_IQ3S = 1024, GRID_IQ2XXS = 1536, GRID_IQ2XS = 1792, GRID_IQ3XXS = 2304, GRID_IQ1S = 2560, GRID_U16_TOTAL = 4608;
__host__ __device__ constexpr int grid_offset_of(int t) { return t == T_IQ3_S ? GRID_IQ3S : t == T_IQ2_XXS ? GRID_IQ2XXS : t == T_IQ2_XS ? GRID_IQ2XS : t == T_IQ3_XXS ? GRID_IQ3XXS : (t == T_IQ1_S || t == T_IQ1_M) ? GRID_IQ1S : GRID_IQ2S; }
// sign byte of a 7-bit sign index (IQ2_XXS / IQ2_XS / IQ3_XXS; ggml-common.h ksigns_iq2xs): bit 7 = parity of the other seven
__device__ __forceinline__ uint32_t ksign7(uint32_t i) { return i | ((uint32_t)(__builtin_popcount(i) & 1) << 7); }

// ---- unaligned-safe raw loads.  Block bases are only 2-byte aligned for Q6_K / IQ2_S / IQ3_S / IQ4_NL;
// gfx950 under HSA runs with unaligned access mode, so these lower to single global_load_dword[xN].
struct __attribute__((packed, aligned(2))) u32_a2 { uint32_t v; };
struct __attribute__((packed, aligned(2))) u64_a2 { uint32_t x, y; };
struct __attribute__((packed, aligned(2))) u128_a2 { uint32_t x, y, z, w; };
struct __attribute__((packed, aligned(1))) u64_a1 { uint32_t x, y; };           // MXFP4: 17-byte blocks
// streamed-once weight loads: non-temporal hint (MI355X guide "nt-weights": weights that one CU reads once)
#ifdef CDNA4_USE_NT      /* measured on MI355X: nt on plain VGPR loads LOSES 15-20% (profiles/r01_notes.md) */
#define WLOAD(p) __builtin_nontemporal_load(p)
#else
#define WLOAD(p) (*(p))
#endif
// ---- fused up*gate epilogue (a12): mul_mat_up_gate_NxM, iqk_mul_mat.cpp:146-175 --------------------------------------------
struct UpGateEpilogue {                  // plain data: lives inside the kernel argument structs
    const float *up_b, *gate_b;          // optional f32 biases, one per weight row (nullptr = none)
    long up_b_stride, gate_b_stride;     // MoE: elements between experts' bias vectors (ggml nb41 / nb51 in floats)
    float limit;                         // op_params[1]: > 1e-6 clamps act(gate) from above and up to [-limit, limit]
};
// unary ops by this fork's enum ggml_unary_op (ggml.h:721-743): RELU 6, SILU 10, SWIGLU_OAI 14, GELU 15
__device__ __forceinline__ float unary_apply(int op, float g) {
    switch (op) {
        case 6:  return g > 0.f ? g : 0.f;
        case 15: { const float a = 0.797884560802865f, c = 0.044715f; return 0.5f * g * (1.0f + tanhf(a * g * (1.0f + c * g * g))); }
        case 10: return g / (1.0f + expf(-g));
        case 14: { const float xi = fminf(g, 7.f); return xi / (1.0f + expf(-xi * 1.702f)); }       // swiglu_oai, iqk_mul_mat.cpp:1059-1086
    }
    return g;
}
//   t = act(gate + b_g), limit > 1e-6 => min(t, limit);  u = up + b_u, SWIGLU_OAI => 1 + clamp(u, -7, 7) (clamp_oai :1088-1090),
//   else limit > 1e-6 => clamp(u, -limit, limit);  result = u * t
__device__ __forceinline__ float up_gate_combine(int op, float up, float gate, const UpGateEpilogue &e, long row, long expert) {
    if (e.gate_b) gate += e.gate_b[expert * e.gate_b_stride + row];
    if (e.up_b) up += e.up_b[expert * e.up_b_stride + row];
    float t = unary_apply(op, gate);
    if (e.limit > 1e-6f) t = fminf(t, e.limit);
    if (op == 14) up = 1.f + fmaxf(fminf(up, 7.f), -7.f);
    else if (e.limit > 1e-6f) up = fmaxf(fminf(up, e.limit), -e.limit);
    return up * t;
}

__device__ __forceinline__ uint4 ldw128(const uint8_t *p) {   // 16-byte aligned weight piece
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = WLOAD(reinterpret_cast<const u32x4 *>(p)); return make_uint4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ uint32_t ld32(const uint8_t *p) { return reinterpret_cast<const u32_a2 *>(p)->v; }
__device__ __forceinline__ uint2 ld64(const uint8_t *p) { const u64_a2 *q = reinterpret_cast<const u64_a2 *>(p); return make_uint2(q->x, q->y); }
__device__ __forceinline__ uint4 ld128(const uint8_t *p) { const u128_a2 *q = reinterpret_cast<const u128_a2 *>(p); return make_uint4(q->x, q->y, q->z, q->w); }
__device__ __forceinline__ uint2 ld64_a1(const uint8_t *p) { const u64_a1 *q = reinterpret_cast<const u64_a1 *>(p); return make_uint2(q->x, q->y); }
__device__ __forceinline__ uint32_t ld16(const uint8_t *p) { return *reinterpret_cast<const uint16_t *>(p); }

__device__ __forceinline__ float half_bits_to_float(uint32_t h) { return __half2float(__ushort_as_half((unsigned short)h)); }
__device__ __forceinline__ float bf16_bits_to_float(uint32_t b) { return __uint_as_float(b << 16); }
// fp32 -> bf16 bits, RNE, quiet NaN: same integer formula as the reference (ggml-impl.h:106-119)
__device__ __forceinline__ uint32_t float_to_bf16_bits(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 64;
    return (u + (0x7fffu + ((u >> 16) & 1))) >> 16;
}

// 4 x int8 dot, exact int32 accumulate (v_dot4_i32_i8)
__device__ __forceinline__ int dot4(uint32_t a, uint32_t b, int c) { return __builtin_amdgcn_sdot4((int)a, (int)b, c, false); }

// wave-wide sum (all 64 lanes end with the total)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// IQ4_NL codebook {-127,...,113} (ggml-quants.c:3911) packed little-endian, 4 entries per dword
__device__ __constant__ static const uint32_t k_iq4nl_packed[4] = {0xbfad9881u, 0xf6eaddcfu, 0x26190d01u, 0x71594535u};

// 8 nibbles of `q` (low nibbles if HI == 0 else high) -> 8 codebook bytes, as two dwords:
// lo4 = entries for nibbles of bytes 0..3, v_perm_b32 based 16-entry table lookup.
__device__ __forceinline__ uint32_t iq4nl_lookup4(uint32_t nib /* 4 nibbles, one per byte, values 0..15 */) {
    // select from the low 8 entries and the high 8 entries, then merge on bit 3 of every nibble
    const uint32_t t0 = k_iq4nl_packed[0], t1 = k_iq4nl_packed[1], t2 = k_iq4nl_packed[2], t3 = k_iq4nl_packed[3];
    const uint32_t sel = nib & 0x07070707u;
    const uint32_t lo = __builtin_amdgcn_perm(t1, t0, sel);     // entries 0..7
    const uint32_t hi = __builtin_amdgcn_perm(t3, t2, sel);     // entries 8..15
    const uint32_t m = ((nib >> 3) & 0x01010101u) * 0xffu;      // 0xff in bytes whose nibble >= 8
    return (hi & m) | (lo & ~m);
}

// MXFP4: e2m1 values doubled {0, 1, 2, 3, 4, 6, 8, 12, -0, -1, ...} (ggml-common.h:2250-2252), packed like the IQ4_NL codebook; the block scale is an E8M0
// exponent byte taken at HALF its value (ggml-impl.h:40-45)
__device__ __constant__ static const uint32_t k_mxfp4_packed[4] = {0x03020100u, 0x0c080604u, 0xfdfeff00u, 0xf4f8fafcu};
__device__ __forceinline__ uint32_t mxfp4_lookup4(uint32_t nib) {
    const uint32_t sel = nib & 0x07070707u;
    const uint32_t lo = __builtin_amdgcn_perm(k_mxfp4_packed[1], k_mxfp4_packed[0], sel), hi = __builtin_amdgcn_perm(k_mxfp4_packed[3], k_mxfp4_packed[2], sel);
    const uint32_t m = ((nib >> 3) & 0x01010101u) * 0xffu;
    return (hi & m) | (lo & ~m);
}
__device__ __forceinline__ float e8m0_half(uint32_t x) { return __uint_as_float(x >= 2 ? (x - 1) << 23 : (x ? 0x00400000u : 0x00200000u)); }

// 4 nibbles (one per byte, 0..15) -> the 4 signed weights as bytes: IQ4_NL through its codebook, Q4_0 as nibble - 8 (per byte, no cross-byte borrow)
template <int TYPE> __device__ __forceinline__ uint32_t nib4_to_i8(uint32_t nib) {
    if (TYPE == T_Q4_0) return ((nib | 0x80808080u) - 0x08080808u) ^ 0x80808080u;
    if (TYPE == T_MXFP4) return mxfp4_lookup4(nib);
    return iq4nl_lookup4(nib);
}

// ---- ik's non-linear value tables (ggml-common.h iq2nl_values, iq3nl_values, iq5nl_values; iq4k_values[0..15] = the IQ4_NL codebook): packed 4 per dword.
// The second half of every table is the first + a constant (5 / 4 / 4 / 2), selected per 16 or 32 weights by a bit of the block.
__device__ __constant__ static const uint32_t k_iq2nl_packed[2] = {0x1101f3e1u, 0x1606f8e6u};                      // {-31,-13,1,17}, + 5
__device__ __constant__ static const uint32_t k_iq3nl_packed[4] = {0xf6e9d8c1u, 0x2f1c0d01u, 0xfaeddcc5u, 0x33201105u};   // 8 values, + 4
__device__ __constant__ static const uint32_t k_iq5nl_packed[8] = {0xa4998e82u, 0xc7bfb6adu, 0xe2dcd5ceu, 0xfaf4eee8u, 0x110b05ffu, 0x2b241d17u, 0x4d443b33u, 0x796d6157u};
// iq2kl_values (32 PAIRS of values, ggml-common.h): first and second value of every pair as two 32-entry byte tables
__device__ __constant__ static const uint32_t k_iq2kl_v0[8] = {0xd8d8c1c1u, 0xe9e9d8d8u, 0xf6e9e9e9u, 0x01f6f6f6u, 0x01010101u, 0x0d0d0d0du, 0x1c1c1c0du, 0x2f2f1c1cu};
__device__ __constant__ static const uint32_t k_iq2kl_v1[8] = {0xf6c10de9u, 0xe9d82f0du, 0xc11c0d01u, 0xe92f0d01u, 0x1c0d01f6u, 0x01f6e9d8u, 0x01e9c10du, 0x0de92f1cu};
__device__ __constant__ static const uint32_t k_iq6nl_packed[16] = {0x938d8781u, 0xa8a39e98u, 0xbab6b1acu, 0xcac6c2beu, 0xd8d4d1cdu, 0xe4e1dedbu, 0xf0edeae7u, 0xfbf8f5f3u, 0x060300feu, 0x110e0c09u, 0x1e1b1714u, 0x2c282421u, 0x3b37332fu, 0x4d48443fu, 0x625c5752u, 0x79736d67u};      // iq6nl_values[0..63] (second half = + 1)
// 4 indices 0..31 (one per byte) -> 4 bytes of a 32-entry table
__device__ __forceinline__ uint32_t lookup32x4(const uint32_t *t, uint32_t idx) {
    const uint32_t sel = idx & 0x07070707u;
    const uint32_t c0 = __builtin_amdgcn_perm(t[1], t[0], sel), c1 = __builtin_amdgcn_perm(t[3], t[2], sel);
    const uint32_t c2 = __builtin_amdgcn_perm(t[5], t[4], sel), c3 = __builtin_amdgcn_perm(t[7], t[6], sel);
    const uint32_t m3 = ((idx >> 3) & 0x01010101u) * 0xffu, m4 = ((idx >> 4) & 0x01010101u) * 0xffu;
    const uint32_t r01 = (c1 & m3) | (c0 & ~m3), r23 = (c3 & m3) | (c2 & ~m3);
    return (r23 & m4) | (r01 & ~m4);
}
__device__ __forceinline__ uint32_t iq5nl_lookup4(uint32_t idx /* 4 indices 0..31, one per byte */) {
    const uint32_t sel = idx & 0x07070707u;
    const uint32_t c0 = __builtin_amdgcn_perm(k_iq5nl_packed[1], k_iq5nl_packed[0], sel), c1 = __builtin_amdgcn_perm(k_iq5nl_packed[3], k_iq5nl_packed[2], sel);
    const uint32_t c2 = __builtin_amdgcn_perm(k_iq5nl_packed[5], k_iq5nl_packed[4], sel), c3 = __builtin_amdgcn_perm(k_iq5nl_packed[7], k_iq5nl_packed[6], sel);
    const uint32_t m3 = ((idx >> 3) & 0x01010101u) * 0xffu, m4 = ((idx >> 4) & 0x01010101u) * 0xffu;
    const uint32_t r01 = (c1 & m3) | (c0 & ~m3), r23 = (c3 & m3) | (c2 & ~m3);
    return (r23 & m4) | (r01 & ~m4);
}
// bytewise a + c for c < 0x80 per byte (no carry across bytes)
__device__ __forceinline__ uint32_t add_bytes(uint32_t a, uint32_t c) { return ((a & 0x7f7f7f7fu) + c) ^ (a & 0x80808080u); }

// all 8 (scale, min) pairs of a Q4_K / Q5_K super-block from its 12 scale bytes (as 3 dwords):
// sc[j], mn[j] are byte j of {sc03, sc47}, {mn03, mn47} (6-bit packing of ggml-quants.c:2036-2043)
__device__ __forceinline__ void k4_unpack_scales(uint32_t s0, uint32_t s1, uint32_t s2,
                                                 uint32_t &sc03, uint32_t &sc47, uint32_t &mn03, uint32_t &mn47) {
    sc03 = s0 & 0x3f3f3f3fu;
    mn03 = s1 & 0x3f3f3f3fu;
    sc47 = (s2 & 0x0f0f0f0fu) | (((s0 >> 6) & 0x03030303u) << 4);
    mn47 = ((s2 >> 4) & 0x0f0f0f0fu) | (((s1 >> 6) & 0x03030303u) << 4);
}
