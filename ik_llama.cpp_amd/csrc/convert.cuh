// convert.cuh -- L0 (bit-exact) dequantizers, activation quantizers writing the reference byte layouts,
// and the device-side run-time repack to the row-interleaved (_R4) layouts.
#pragma once
#include "cdna4_common.cuh"
#include "gemv.cuh"   // quant8 / group_max / group_sum

// ------------------------------------------------------------------------------------------------
// integer decode of ONE element e (0..255, or 0..31 for IQ4_NL) of a block: returns the integer the reference
// derives and the index of the f32 scale / min it is combined with.  Used by the L0 dequantizer; the hot
// kernels (gemv.cuh, gemm_mfma.cuh) use vectorised forms of the same formulas.

struct Elem { int q; float scale, minv; };   // value = f(scale, q, minv) in the reference's operation order

// grid tables for the L0 kernel live in global memory (packed, 3 KiB, L1/L2 resident)
__device__ __forceinline__ int iq2s_mag(const uint16_t *grid2, int idx, int j) { const int c = (grid2[idx] >> (2 * j)) & 3; return 8 + 17 * c + (c >> 1); }
__device__ __forceinline__ int iq3s_mag(const uint16_t *grid3, int idx, int j) { return 2 * ((grid3[idx] >> (3 * j)) & 7) + 1; }
__device__ __forceinline__ int iq3xxs_mag(const uint16_t *grid3, int idx, int j) { const int c = (grid3[idx] >> (3 * j)) & 7; return c < 7 ? 8 * c + 4 : 62; }

__device__ __forceinline__ int tab_byte(const uint32_t *t, int i) { return (int)(int8_t)((t[i >> 2] >> (8 * (i & 3))) & 0xff); }
// `b` = the block, `rowp` = the row (its first bytes are the row scale of the _KS types)
// `rs` multiplies the ROW scale of the trellis types: 1 = the reference's scalar to_float; kt_matmul_factor(type) = the value its mat-mul kernels use (the f16 prompt route)
template <int BASE>
__device__ __forceinline__ float dequant_base_elem(const uint8_t *b, int e, const uint16_t *grid, const uint8_t *rowp = nullptr, float rs = 1.0f) {
    if (type_is_kt(BASE)) {                        // dequantize_row_iq1_kt / iq2_kt / iq3_kt / iq4_kt (iqk_quantize.cpp:9470-9489,9751-9779,10021-10057,10286-10314): y = (d * scale) * value
        const uint32_t pw[8] = {kt_pow(1), kt_pow(2), kt_pow(3), kt_pow(4), kt_pow(5), kt_pow(6), kt_pow(7), kt_pow(8)};
        const float d = __uint_as_float(ld32(rowp)) * rs;
        const int ib = e >> 5; uint32_t seed; int sc, j;
        if (BASE == T_IQ2_KT || BASE == T_IQ3_KT) {
            seed = ld16(b + 4 + 2 * (e >> 3)) + 4096u; j = e & 7;
            const uint32_t nib = (b[ib & 3] >> (4 * (ib >> 2))) & 15u; sc = BASE == T_IQ2_KT ? iq4k_value(nib) : (int)nib;
        } else if (BASE == T_IQ4_KT) {
            const int jj = e >> 2, ig = jj & 7; const uint32_t sh = ld32(b + 4 * ib);
            seed = ((uint32_t)b[32 + jj] | ((((uint32_t)b[96 + (jj & 31)] >> (4 * (jj >> 5))) & 15u) << 8) | (((sh >> (8 + 3 * ig)) & 7u) << 12)) + 4096u + ((sh & 1u) << 15); j = e & 3;
            sc = (int)((sh & 0xffu) >> 1) - 64;
        } else {
            const int g = e >> 3; const uint32_t shb = b[g >> 2];
            seed = ((uint32_t)b[8 + g] | ((((uint32_t)b[40 + (g & 15)] >> (4 * (g >> 4))) & 15u) << 8) | (((shb >> (4 + (g & 3))) & 1u) << 12)) + 4096u; j = e & 7;
            sc = iq4k_value(b[ib] & 15u);
        }
        const uint32_t x = (seed * pw[j]) & 0x3f3f3f3fu;
        int v = (int)((x & 0xff) + ((x >> 8) & 0xff) + ((x >> 16) & 0xff) + (x >> 24)) - 126;
        if (BASE == T_IQ3_KT) { const float y = (d * (float)sc) * (float)abs(v); return (b[68 + (e & 31)] & (1u << ib)) ? -y : y; }      // (the sign of sl |v| is flipped: a zero magnitude gives -0.0f, as the reference)
        return (d * (float)sc) * (float)v;
    }
    if (BASE == T_IQ1_BN || BASE == T_IQ2_BN) {    // BitNet: the VALUE the mat-mul kernels give a weight, row scale x (u - 1) (mul_mat_iq1bn / iq2bn_q8_K64, iqk_gemm_1bit.cpp:1247-1447;
                                                   // the reference's to_float of these types leaves the row scale out -- the oracle documents the choice, oracle/iqk_oracle.c:415-419)
        int u;
        if (BASE == T_IQ2_BN) u = (b[e & 15] >> (2 * (e >> 4))) & 3;
        else { const int i16 = e >> 4, r = e & 15; const uint32_t km[5] = {81, 27, 9, 3, 1};
               const uint32_t v = r < 15 ? ((uint32_t)b[3 * i16 + r / 5] * km[r % 5]) & 255u : ((uint32_t)b[12] * km[i16]) & 255u; u = (int)((3u * v) >> 8); }
        const float d = BASE == T_IQ2_BN ? __uint_as_float(reinterpret_cast<const u32_a2 *>(rowp)->v) : half_bits_to_float(ld16(rowp));
        return d * (float)(u - 1);
    }
    if (BASE == T_IQ2_K) {                         // dequantize_row_iq2_k (iqk_quantize.cpp:1356-1387) ; y = (d * (nibble - 8)) * value
        const int ib = e >> 5, j = e & 31, is = 2 * ib + (j >> 4); const uint32_t extra = ld16(b + 2);
        const int sc = (int)((b[4 + ib] >> (4 * (j >> 4))) & 15) - 8;
        const int v = tab_byte(k_iq2nl_packed, (b[12 + 32 * (ib >> 2) + j] >> (2 * (ib & 3))) & 3) + (((extra >> is) & 1) ? 5 : 0);
        return (half_bits_to_float(ld16(b)) * (float)sc) * (float)v;
    }
    if (BASE == T_IQ3_K) {                         // dequantize_row_iq3_k (:2534-2567)
        const int ib = e >> 5, j = e & 31, is = 2 * ib + (j >> 4); const uint32_t extra = ld16(b + 2), sh = ld16(b + 4);
        const int sc = (2 * (int)((b[6 + ib] >> (4 * (j >> 4))) & 15) + 1) * (((sh >> is) & 1) ? -1 : 1);
        const int idx = ((b[14 + 32 * (ib >> 2) + j] >> (2 * (ib & 3))) & 3) | (((b[78 + j] >> ib) & 1) << 2);
        const int v = tab_byte(k_iq3nl_packed, idx) + (((extra >> is) & 1) ? 4 : 0);
        return (half_bits_to_float(ld16(b)) * (float)sc) * (float)v;
    }
    if (BASE == T_IQ4_K) {                         // dequantize_row_iq4_k (:2822-2850)
        const int ib = e >> 5, j = e & 15, h = (e >> 4) & 1, is = 2 * ib + h; const uint32_t extra = ld16(b + 2), sh = b[4 + (ib >> 1)] >> (4 * (ib & 1));
        const int sc = (int)(h ? ((b[8 + ib] >> 4) | ((sh << 2) & 0x30)) : ((b[8 + ib] & 15) | ((sh << 4) & 0x30))) - 32;
        const int nib = h ? (b[16 + 16 * ib + j] >> 4) : (b[16 + 16 * ib + j] & 15);
        const int v = tab_byte(k_iq4nl_packed, nib) + (((extra >> is) & 1) ? 4 : 0);
        return (half_bits_to_float(ld16(b)) * (float)sc) * (float)v;
    }
    if (BASE == T_IQ5_K) {                         // dequantize_row_iq5_k (:3112-3152)
        const int i = e >> 6, k = (e >> 4) & 3, j = e & 15; const uint32_t extra = ld16(b + 2), sh = b[4 + i], slb = b[8 + 2 * i + (k >> 1)];
        const int sc = (int)(((k & 1) ? (slb >> 4) : (slb & 15)) | (((sh >> (2 * k)) & 3) << 4)) - 32;
        const uint32_t qb = b[16 + 32 * i + j + 16 * (k & 1)], hb = b[144 + j + 16 * (k & 1)] >> (2 * i);
        const int idx = (int)((k & 2) ? ((qb >> 4) | ((hb & 2) << 3)) : ((qb & 15) | ((hb & 1) << 4)));
        const int v = tab_byte(k_iq5nl_packed, idx) + (((extra >> (4 * i + k)) & 1) ? 2 : 0);
        return (half_bits_to_float(ld16(b)) * (float)sc) * (float)v;
    }
    if (BASE == T_IQ6_K) {                         // dequantize_row_iq6_k (iqk_quantize.cpp:3442-3490): a CUBIC in the 6-bit index (fma chain as the reference build contracts it), not the int8 table
        const int i = e >> 6, k = (e >> 4) & 3, j = e & 15; const uint32_t extra = ld16(b + 2);
        const uint32_t ql = b[20 + 32 * i + 16 * (k & 1) + j], h = b[148 + 32 * (i >> 1) + 16 * (k & 1) + j] >> (4 * (i & 1));
        const float q = (float)(int)((k & 2) ? ((ql >> 4) | ((h & 0x0c) << 2)) : ((ql & 15) | ((h & 3) << 4)));
        const float dl = half_bits_to_float(ld16(b)) * (float)(int)(int8_t)b[4 + (e >> 4)], m = ((extra >> (e >> 4)) & 1) ? 1.f : 0.f;
        return dl * (fmaf(q, fmaf(q, fmaf(q, 0.0011972f, -0.11218f), 6.2568f), -127.f) + m);
    }
    if (BASE == T_IQ4_KSS) {                       // dequantize_row_iq4_kss ; f32 row scale; the scale byte = the low bits of the block's eight 16-bit words
        const int ib = e >> 5, j = e & 15, h = (e >> 4) & 1; uint32_t ls = 0;
        for (int k = 0; k < 8; ++k) ls |= (ld16(b + 16 * ib + 2 * k) & 1u) << k;
        uint32_t a = ld16(b + 16 * ib + 2 * (j >> 1)) & 0xfffeu; a ^= a >> 1;
        const uint32_t byte = (a >> (8 * (j & 1))) & 0xff;
        const int v = tab_byte(k_iq4nl_packed, h ? (byte >> 4) : (byte & 15)) + ((ls & 1) ? 4 : 0);
        return (*reinterpret_cast<const float *>(rowp) * (float)((int)(ls & 254) - 127)) * (float)v;
    }
    if (BASE == T_IQ2_KL) {                        // dequantize_row_iq2_kl ; f16 row scale; 5-bit index -> a pair of values
        const int i = e >> 6, h = (e >> 5) & 1, j = (e & 31) >> 1, w = e & 1; const uint32_t sh = ld16(b);
        const int sc = (int)(((b[2 + ((2 * i + h) & 3)] >> (4 * (i >> 1))) & 15) | (((sh >> (4 * i + 2 * h)) & 3) << 4)) - 32;
        const uint32_t qb = b[6 + 16 * i + j], idx = (h ? (qb >> 4) : (qb & 15)) | (((b[70 + j] >> (2 * i + h)) & 1) << 4);
        const int v = tab_byte(w ? k_iq2kl_v1 : k_iq2kl_v0, idx);
        return (half_bits_to_float(ld16(rowp)) * (float)sc) * (float)v;
    }
    if (BASE == T_IQ2_KS) {                        // dequantize_row_iq2_ks ; f16 row scale, 5-bit scales per 32
        const int ib = e >> 5, j = e & 31; const uint32_t extra = ld16(b);
        const int sc = (int)(((b[2 + (ib >> 1)] >> (4 * (ib & 1))) & 15) | (((extra >> (8 + ib)) & 1) << 4)) - 16;
        const int v = tab_byte(k_iq2nl_packed, (b[6 + 32 * (ib >> 2) + j] >> (2 * (ib & 3))) & 3) + (((extra >> ib) & 1) ? 5 : 0);
        return (half_bits_to_float(ld16(rowp)) * (float)sc) * (float)v;
    }
    if (BASE == T_IQ3_KS) {                        // dequantize_row_iq3_ks
        const int ib = e >> 5, j = e & 31; const uint32_t extra = ld16(b);
        const int sc = (int)(((b[2 + (ib & 3)] >> (4 * (ib >> 2))) & 15) | (((extra >> ib) & 1) << 4)) - 16;
        const int idx = ((b[6 + 32 * (ib >> 2) + j] >> (2 * (ib & 3))) & 3) | (((b[70 + j] >> ib) & 1) << 2);
        const int v = tab_byte(k_iq3nl_packed, idx) + (((extra >> (8 + ib)) & 1) ? 4 : 0);
        return (half_bits_to_float(ld16(rowp)) * (float)sc) * (float)v;
    }
    if (BASE == T_IQ4_KS) {                        // dequantize_row_iq4_ks (:4555-4575) ; row scale d in front of the blocks
        const int ib = e >> 5, j = e & 15, h = (e >> 4) & 1; const uint32_t s = b[ib];
        const int nib = h ? (b[8 + 16 * ib + j] >> 4) : (b[8 + 16 * ib + j] & 15);
        const int v = tab_byte(k_iq4nl_packed, nib) + ((s & 1) ? 4 : 0);
        return (*reinterpret_cast<const float *>(rowp) * (float)((int)(s & 254) - 127)) * (float)v;
    }
    if (BASE == T_IQ5_KS) {                        // dequantize_row_iq5_ks
        const int i = e >> 6, h = (e >> 5) & 1, j = e & 31; const uint32_t s = b[2 * i + h], qb = b[8 + 32 * i + j], hb = b[136 + j] >> (2 * i + h);
        const int idx = (int)((h ? (qb >> 4) : (qb & 15)) | ((hb & 1) << 4));
        const int v = tab_byte(k_iq5nl_packed, idx) + ((s & 1) ? 2 : 0);
        return (*reinterpret_cast<const float *>(rowp) * (float)((int)(s & 254) - 127)) * (float)v;
    }
    if (BASE == T_Q4_K || BASE == T_Q5_K) {        // ggml-quants.c:2797-2819, 3015-3038 ; y = fma(d*sc, q, -(dmin*m))
        const int j = e >> 5, g = e >> 6, l = e & 31; const uint8_t *s = b + 4;
        int sc, mn;
        if (j < 4) { sc = s[j] & 63; mn = s[j + 4] & 63; }
        else { sc = (s[j + 4] & 15) | ((s[j - 4] >> 6) << 4); mn = (s[j + 4] >> 4) | ((s[j] >> 6) << 4); }
        const uint8_t *qs = b + (BASE == T_Q5_K ? 48 : 16);
        int q = (e & 32) ? (qs[32 * g + l] >> 4) : (qs[32 * g + l] & 15);
        if (BASE == T_Q5_K) q += ((b[16 + l] >> (2 * g + ((e >> 5) & 1))) & 1) << 4;
        const float d = half_bits_to_float(ld16(b)), dmin = half_bits_to_float(ld16(b + 2));
        return fmaf(d * (float)sc, (float)q, -(dmin * (float)mn));
    }
    if (BASE == T_Q6_K) {                          // ggml-quants.c:3231-3258 ; y = (d*sc)*q
        const int n = e >> 7, r = e & 127, l = r & 31, k = r >> 5; const uint8_t *ql = b + 64 * n, *qh = b + 128 + 32 * n;
        const int lo = (k & 2) ? (ql[l + 32 * (k & 1)] >> 4) : (ql[l + 32 * (k & 1)] & 15);
        const int q = (lo | (((qh[l] >> (2 * k)) & 3) << 4)) - 32;
        const int sc = (int)(int8_t)b[192 + 8 * n + (l >> 4) + 2 * k];
        return half_bits_to_float(ld16(b + 208)) * (float)sc * (float)q;
    }
    if (BASE == T_Q4_0) {                          // dequantize_row_q4_0 (ggml-quants.c): y = (nibble - 8) * d
        const int nib = e < 16 ? (b[2 + e] & 15) : (b[2 + e - 16] >> 4);
        return (float)(nib - 8) * half_bits_to_float(ld16(b));
    }
    if (BASE == T_Q8_0) return (float)(int)(int8_t)b[2 + e] * half_bits_to_float(ld16(b));      // dequantize_row_q8_0
    if (BASE == T_Q4_1 || BASE == T_Q5_1) {        // dequantize_row_q4_1 / q5_1 ; y = fma(q, d, m)  (the reference build contracts x * d + m)
        const int j = e & 15; const uint32_t qh = BASE == T_Q5_1 ? ld32(b + 4) : 0u; const uint8_t *qs = b + (BASE == T_Q5_1 ? 8 : 4);
        const int q = e < 16 ? (int)((qs[j] & 15) | (((qh >> j) << 4) & 0x10)) : (int)((qs[j] >> 4) | ((qh >> (j + 12)) & 0x10));
        return fmaf((float)q, half_bits_to_float(ld16(b)), half_bits_to_float(ld16(b + 2)));
    }
    if (BASE == T_Q6_0) {                          // ggml-quants.c:1675-1695 ; y = ((nibble | 2 bits << 4) - 32) * d
        const int j = e & 15; const uint32_t h = b[2 + (j & 7)] >> (4 * (j >> 3));
        const int q = e < 16 ? (int)((b[10 + j] & 15) | ((h << 4) & 0x30)) : (int)((b[10 + j] >> 4) | ((h << 2) & 0x30));
        return (float)(q - 32) * half_bits_to_float(ld16(b));
    }
    if (BASE == T_Q2_K) {                          // dequantize_row_q2_K ; y = fma(d * sc, q, -(dmin * m))
        const int n = e >> 7, j = (e >> 5) & 3, l = e & 31, is = 8 * n + 2 * j + (l >> 4);
        const int q = (b[16 + 32 * n + l] >> (2 * j)) & 3;
        return fmaf(half_bits_to_float(ld16(b + 80)) * (float)(b[is] & 15), (float)q, -(half_bits_to_float(ld16(b + 82)) * (float)(b[is] >> 4)));
    }
    if (BASE == T_Q3_K) {                          // dequantize_row_q3_K ; y = (d * (sc - 32)) * (q2 - (hbit ? 0 : 4))
        const int n = e >> 7, j = (e >> 5) & 3, l = e & 31, is = 8 * n + 2 * j + (l >> 4); const uint8_t *scl = b + 96;
        const int sc = (int)((is < 8 ? scl[is] & 15 : scl[is - 8] >> 4) | (((scl[8 + (is & 3)] >> (2 * (is >> 2))) & 3) << 4)) - 32;
        const int q = (int)((b[32 + 32 * n + l] >> (2 * j)) & 3) - (((b[l] >> (4 * n + j)) & 1) ? 0 : 4);
        return (half_bits_to_float(ld16(b + 108)) * (float)sc) * (float)q;
    }
    if (BASE == T_Q5_0) {                          // ggml-quants.c:1622-1646 ; y = ((nibble | bit << 4) - 16) * d
        const int j = e & 15; const uint32_t qh = ld32(b + 2);
        const int q = e < 16 ? (int)((b[6 + j] & 15) | (((qh >> j) << 4) & 0x10)) : (int)((b[6 + j] >> 4) | ((qh >> (j + 12)) & 0x10));
        return (float)(q - 16) * half_bits_to_float(ld16(b));
    }
    if (BASE == T_MXFP4) {                         // iqk_quantize.cpp:4224-4236 ; y = (2^(e - 128)) * kvalue
        const int j = e & 15; const int nib = e < 16 ? (b[1 + j] & 15) : (b[1 + j] >> 4);
        return e8m0_half(b[0]) * (float)(int)(int8_t)((k_mxfp4_packed[nib >> 2] >> (8 * (nib & 3))) & 0xff);
    }
    if (BASE == T_IQ1_S) {                         // ggml-quants.c:3836-3859 ; y = (d (2 s + 1)) * (grid + delta), grid in {-1, 0, 1} (2-bit codes + 1), delta = +-1/8
        const int ib = e >> 5, l = (e >> 3) & 3, j = e & 7; const uint32_t qh = ld16(b + 34 + 2 * ib);
        const int g = (int)((grid[GRID_IQ1S + (b[2 + 4 * ib + l] | (((qh >> (3 * l)) & 7) << 8))] >> (2 * j)) & 3) - 1;
        return (half_bits_to_float(ld16(b)) * (float)(2 * (int)((qh >> 12) & 7) + 1)) * ((float)g + ((qh & 0x8000) ? -0.125f : 0.125f));
    }
    if (BASE == T_IQ1_M) {                         // ggml-quants.c:3861-3908 ; the f16 d sits in the top nibbles of the four scale words, 3-bit scales per 16, delta bit per 8
        const int ib = e >> 5, l = (e >> 3) & 3, j = e & 7; const uint32_t h = (uint32_t)b[32 + 2 * ib + (l >> 1)] >> (4 * (l & 1));
        const int g = (int)((grid[GRID_IQ1S + (b[4 * ib + l] | ((h & 7) << 8))] >> (2 * j)) & 3) - 1;
        const uint32_t s0 = ld16(b + 48), s1 = ld16(b + 50), s2 = ld16(b + 52), s3 = ld16(b + 54), sw = (ib >> 1) == 0 ? s0 : (ib >> 1) == 1 ? s1 : (ib >> 1) == 2 ? s2 : s3;
        const float d = half_bits_to_float((s0 >> 12) | ((s1 >> 8) & 0x00f0) | ((s2 >> 4) & 0x0f00) | (s3 & 0xf000));
        return (d * (float)(2 * (int)((sw >> (6 * (ib & 1) + 3 * (l >> 1))) & 7) + 1)) * ((float)g + ((h & 8) ? -0.125f : 0.125f));
    }
    if (BASE == T_IQ2_XXS) {                       // ggml-quants.c:3674-3700 ; y = (d*(0.5+s)*0.25) * grid * sign
        const int ib = e >> 5, l = (e >> 3) & 3, j = e & 7; const uint32_t a0 = ld32(b + 2 + 8 * ib), a1 = ld32(b + 6 + 8 * ib);
        const float db = half_bits_to_float(ld16(b)) * (0.5f + (float)(a1 >> 28)) * 0.25f;
        return db * (float)iq2s_mag(grid + GRID_IQ2XXS, (a0 >> (8 * l)) & 255, j) * (((ksign7((a1 >> (7 * l)) & 127) >> j) & 1) ? -1.f : 1.f);
    }
    if (BASE == T_IQ2_XS) {                        // ggml-quants.c:3702-3727
        const int ib = e >> 5, l = (e >> 3) & 3, j = e & 7; const uint32_t v = ld16(b + 2 + 2 * (4 * ib + l));
        const int s4 = (l < 2) ? (b[66 + ib] & 15) : (b[66 + ib] >> 4);
        const float db = half_bits_to_float(ld16(b)) * (0.5f + (float)s4) * 0.25f;
        return db * (float)iq2s_mag(grid + GRID_IQ2XS, v & 511, j) * (((ksign7(v >> 9) >> j) & 1) ? -1.f : 1.f);
    }
    if (BASE == T_IQ3_XXS) {                       // ggml-quants.c:3761-3791 ; y = (d*(0.5+s)*0.5) * grid * sign
        const int ib = e >> 5, l = (e >> 3) & 3, j = e & 7; const uint32_t a = ld32(b + 66 + 4 * ib);
        const float db = half_bits_to_float(ld16(b)) * (0.5f + (float)(a >> 28)) * 0.5f;
        return db * (float)iq3xxs_mag(grid + GRID_IQ3XXS, b[2 + 8 * ib + 2 * l + (j >> 2)], j & 3) * (((ksign7((a >> (7 * l)) & 127) >> j) & 1) ? -1.f : 1.f);
    }
    if (BASE == T_IQ4_XS) {                        // ggml-quants.c:3931-3952 ; y = (d * (ls - 32)) * kvalue
        const int ib = e >> 5, j = e & 31; const uint32_t sh = ld16(b + 2);
        const int ls = (int)(((b[4 + ib / 2] >> (4 * (ib & 1))) & 0xf) | (((sh >> (2 * ib)) & 3) << 4)) - 32;
        const int nib = j < 16 ? (b[8 + 16 * ib + j] & 15) : (b[8 + 16 * ib + j - 16] >> 4);
        const int kv = (int)(int8_t)((k_iq4nl_packed[nib >> 2] >> (8 * (nib & 3))) & 0xff);
        return (half_bits_to_float(ld16(b)) * (float)ls) * (float)kv;
    }
    if (BASE == T_IQ4_NL) {                        // ggml-quants.c:3913-3929
        const int nib = e < 16 ? (b[2 + e] & 15) : (b[2 + e - 16] >> 4);
        const int kv = (int)(int8_t)((k_iq4nl_packed[nib >> 2] >> (8 * (nib & 3))) & 0xff);
        return half_bits_to_float(ld16(b)) * (float)kv;
    }
    if (BASE == T_IQ2_S) {                         // ggml-quants.c:3729-3757 ; y = (d*(0.5+s)*0.25) * grid * sign
        const int ib = e >> 5, l = (e >> 3) & 3, j = e & 7;
        const int idx = b[2 + 4 * ib + l] | ((b[66 + ib] << (8 - 2 * l)) & 0x300);
        const int s4 = (l < 2) ? (b[74 + ib] & 15) : (b[74 + ib] >> 4);
        const float db = half_bits_to_float(ld16(b)) * (0.5f + (float)s4) * 0.25f;
        return db * (float)iq2s_mag(grid, idx, j) * (((b[34 + 4 * ib + l] >> j) & 1) ? -1.f : 1.f);
    }
    if (BASE == T_IQ3_S) {                         // ggml-quants.c:3793-3838 ; y = (d*(1+2s)) * grid * sign
        const int ib = e >> 5, l = (e >> 3) & 3, j = e & 7;
        const int idx = b[2 + 8 * ib + 2 * l + (j >> 2)] | ((b[66 + ib] << (8 - 2 * l - (j >> 2))) & 256);
        const int s4 = (b[106 + (ib >> 1)] >> (4 * (ib & 1))) & 15;
        const float db = half_bits_to_float(ld16(b)) * (float)(1 + 2 * s4);
        return db * (float)iq3s_mag(grid + GRID_IQ3S, idx, j & 3) * (((b[74 + 4 * ib + l] >> j) & 1) ? -1.f : 1.f);
    }
    return 0.f;
}

// position maps of the _R4 layouts (defined operationally by the reference repackers, iqk_quantize.cpp:6075 q4_k,
// :6297 q5_k, :6188 q6_k, :5214 iq4_nl, :7835 iq2_s, :8015 iq3_s).  Shared nibble position for the 4 nibble types:
__host__ __device__ __forceinline__ int r4_nib_byte(int ib, int r, int e) { const int i = e & 3, g = e >> 2; return 64 * ib + 4 * r + i + 16 * (g >> 2) + 32 * (g & 1); }
__host__ __device__ __forceinline__ int r4_nib_shift(int e) { return 4 * ((e >> 3) & 1); }

// element e of row r (0..3) of interleaved block `b` (4*type_size bytes; 72 for IQ4_NL_R4)
template <int BASE>
__device__ __forceinline__ float dequant_r4_elem(const uint8_t *b, int r, int e, const uint16_t *grid) {
    if (BASE == T_IQ4_NL) {                        // iqk_quantize.cpp:5255-5276
        const int nib = (b[8 + r4_nib_byte(0, r, e)] >> r4_nib_shift(e)) & 15;
        const int kv = (int)(int8_t)((k_iq4nl_packed[nib >> 2] >> (8 * (nib & 3))) & 0xff);
        return half_bits_to_float(ld16(b + 2 * r)) * (float)kv;
    }
    const int ib = e >> 5, el = e & 31;
    if (BASE == T_Q4_K || BASE == T_Q5_K) {        // :6118-6143, :6342-6372
        const uint8_t *sh = b + 16, *sl = b + 32, *qh = b + 64, *qs = b + (BASE == T_Q5_K ? 192 : 64);
        const int is = 4 * ib + r, h = (sh[is & 15] >> (4 * (is >> 4))) & 15;
        const int sc = (sl[is] & 15) | ((h & 3) << 4), mn = (sl[is] >> 4) | ((h & 12) << 2);
        int q = (qs[r4_nib_byte(ib, r, el)] >> r4_nib_shift(el)) & 15;
        if (BASE == T_Q5_K) { const int i = el & 3, g = el >> 2; const int bit = (g & 4) | ((g & 1) << 1) | ((g >> 1) & 1); q |= ((qh[16 * ib + 4 * r + i] >> bit) & 1) << 4; }
        const float d = half_bits_to_float(ld16(b + 2 * r)), dmin = half_bits_to_float(ld16(b + 2 * (r + 4)));
        return fmaf(d * (float)sc, (float)q, -(dmin * (float)mn));
    }
    if (BASE == T_Q6_K) {                          // :6229-6257
        const uint8_t *scales = b + 8, *qh = b + 72, *ql = b + 328;
        const int i = el & 3, g = el >> 2; const int shq = ((g & 1) << 2) | (g & 2);   // {0,4,2,6}[g&3]
        const int lo = (ql[r4_nib_byte(ib, r, el)] >> r4_nib_shift(el)) & 15;
        const int hi = (qh[32 * ib + 4 * r + i + 16 * (g >> 2)] >> shq) & 3;
        const int sc = (int)(int8_t)scales[8 * ib + r + 4 * (el >> 4)];
        return (half_bits_to_float(ld16(b + 2 * r)) * (float)sc) * (float)((lo | (hi << 4)) - 32);
    }
    if (BASE == T_IQ2_S) {                         // :7871-7892 ; (0.125f*d)*(2s+1)
        const uint8_t *qs = b + 8, *qh = b + 136, *sg = b + 168, *scl = b + 296;
        const int i = el >> 3, j = el & 7;
        const int idx = qs[16 * ib + 4 * r + i] | ((qh[4 * ib + r] << (8 - 2 * i)) & 0x300);
        const int s4 = (i < 2) ? (scl[4 * ib + r] & 15) : (scl[4 * ib + r] >> 4);
        const float dl = (0.125f * half_bits_to_float(ld16(b + 2 * r))) * (float)(2 * s4 + 1);
        return dl * (float)iq2s_mag(grid, idx, j) * (((sg[16 * ib + 4 * r + i] >> j) & 1) ? -1.f : 1.f);
    }
    if (BASE == T_IQ3_S) {                         // :8063-8088
        const uint8_t *qs = b + 8, *qh = b + 264, *sg = b + 296, *scl = b + 424;
        const int l = 4 * ib + r, s4 = (scl[l & 15] >> (4 * (l >> 4))) & 15;
        const int half = el >> 4, i = (el >> 2) & 3, j = el & 3;     // el = 16*half + 4*i + j
        const int idx = qs[32 * ib + r + 8 * i + 4 * half] + ((qh[4 * ib + r] << (8 - i - 4 * half)) & 0x100);
        const float dl = half_bits_to_float(ld16(b + 2 * r)) * (float)(1 + 2 * s4);
        return dl * (float)iq3s_mag(grid + GRID_IQ3S, idx, j) * (((sg[16 * ib + 4 * r + j] >> (i + 4 * half)) & 1) ? -1.f : 1.f);
    }
    return 0.f;
}

template <typename T> __device__ __forceinline__ void store_out(T *p, float v);
template <> __device__ __forceinline__ void store_out<float>(float *p, float v) { *p = v; }
template <> __device__ __forceinline__ void store_out<__half>(__half *p, float v) { *p = __float2half_rn(v); }

// one thread per output element (utility path: to_float / get_rows / parity; not a hot kernel)
template <int TYPE, typename OUT>
__global__ void dequantize_kernel(const uint8_t *A, long strideA, long nrows, long K, OUT *dst, long dst_stride, const uint16_t *grid, float rs) {
    constexpr int BASE = type_base(TYPE), BS = type_block_elems(BASE), TS = type_block_bytes(BASE);
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nrows * K) return;
    const long row = idx / K, k = idx - row * K, blk = k / BS; const int e = (int)(k - blk * BS);
    float v;
    if (type_is_r4(TYPE)) v = dequant_r4_elem<BASE>(A + (row >> 2) * 4 * strideA + blk * (4 * TS), (int)(row & 3), e, grid);
    else                  v = dequant_base_elem<BASE>(A + row * strideA + type_row_meta(BASE) + blk * TS, e, grid, A + row * strideA, rs);
    asm volatile("" : "+v"(v));        // the f32 L0 value first, THEN one rounding to the output type (hipcc otherwise folds the last multiply into v_fma_mixlo_f16: one rounding of the exact product)
    store_out<OUT>(dst + row * dst_stride + k, v);
}

// 8 consecutive elements per thread (base types, dst rows 16-byte aligned): the block header / scale work is shared by the 8 decodes and the result leaves as ONE 16-byte
// (f16) or two 16-byte (f32) stores -- the f16 prompt route of the decode-only types spends its time here (element-per-thread: 0.3 T elements / s)
template <int TYPE, typename OUT>
__global__ void dequantize8_kernel(const uint8_t *A, long strideA, long nrows, long K, OUT *dst, long dst_stride, const uint16_t *grid, float rs) {
    constexpr int BS = type_block_elems(TYPE), TS = type_block_bytes(TYPE);
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x, k8 = K >> 3;
    if (idx >= nrows * k8) return;
    const long row = idx / k8, k = (idx - row * k8) << 3, blk = k / BS; const int e = (int)(k - blk * BS);
    const uint8_t *rowp = A + row * strideA, *b = rowp + type_row_meta(TYPE) + blk * TS;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { v[j] = dequant_base_elem<TYPE>(b, e + j, grid, rowp, rs); asm volatile("" : "+v"(v[j])); }
    OUT *o = dst + row * dst_stride + k;
    if constexpr (sizeof(OUT) == 2) {
        union { __half h[8]; uint4 u; } c;
#pragma unroll
        for (int j = 0; j < 8; ++j) c.h[j] = __float2half_rn(v[j]);
        *reinterpret_cast<uint4 *>(o) = c.u;
    } else {
        reinterpret_cast<float4 *>(o)[0] = make_float4(v[0], v[1], v[2], v[3]); reinterpret_cast<float4 *>(o)[1] = make_float4(v[4], v[5], v[6], v[7]);
    }
}

// ------------------------------------------------------------------------------------------------
// activation quantizers writing the reference's block_q8_2_x4 / block_q8_K byte layouts (a10)
// one lane per 8 consecutive floats; grid.y = row.
template <int VDT>
__global__ void quantize_rows_kernel(const uint8_t *B, long strideB, long K, long nrows, uint8_t *dst, long dst_row_bytes) {
    const long row = blockIdx.y + (long)gridDim.y * blockIdx.z; const int k8 = (int)(K >> 3);      // (grid.y <= 65535: taller batches continue in grid.z)
    if (row >= nrows) return;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = j < k8;
    float4 v0 = make_float4(0, 0, 0, 0), v1 = v0;
    if (active) { const float *x = reinterpret_cast<const float *>(B + row * strideB) + 8 * j; v0 = *reinterpret_cast<const float4 *>(x); v1 = *reinterpret_cast<const float4 *>(x + 4); }
    uint8_t *out = dst + row * dst_row_bytes; int isum;
    if (VDT == T_Q8_2_X4) {                      // iqk_quantize.cpp:1072-1166 (x86 branch)
        const float amax = quad_max(amax8(v0, v1));
        const uint32_t db = float_to_bf16_bits(amax / 127.f);
        const float d = bf16_bits_to_float(db), id = d > 0 ? 1.f / d : 0.f;
        const uint2 q = quant8(v0, v1, id, isum);
        isum = quad_sum(isum);
        if (!active) return;
        const int b = j >> 2, nb = (int)(K >> 5), nb4 = 4 * (nb / 4);
        uint8_t *blk; int doff, soff, qoff;
        if (b < nb4) { blk = out + (long)(b >> 2) * 144; const int ir = b & 3; doff = 2 * ir; soff = 8 + 2 * ir; qoff = 16 + 32 * ir; }
        else         { blk = out + (long)b * 36; doff = 0; soff = 2; qoff = 4; }
        if ((j & 3) == 0) { *reinterpret_cast<uint16_t *>(blk + doff) = (uint16_t)db; *reinterpret_cast<int16_t *>(blk + soff) = (int16_t)isum; }
        uint32_t *qp = reinterpret_cast<uint32_t *>(blk + qoff + 8 * (j & 3)); qp[0] = q.x; qp[1] = q.y;
    } else {                                     // iqk_quantize.cpp:3809-3875 (AVX2 branch); block_q8_K = 296 bytes
        const float amax = group_max<32>(amax8(v0, v1));
        const float d = amax / 127.f, id = amax != 0.0f ? 127.f / amax : 0.0f;
        const uint2 q = quant8(v0, v1, id, isum);
        const int s16 = isum + __shfl_xor(isum, 1, 64);          // 16-element sums (pairs of lanes)
        const int s32 = s16 + __shfl_xor(s16, 2, 64);            // 32-element sums
        const int lane32 = j & 31;
        uint8_t *blk = out + (long)(j >> 5) * 296;
        if (VDT == T_Q8_K32) {                    // bsums storage = 8 floats d*sum_32 ; sum = their in-order float sum
            const float bs = d * (float)s32;
            float tot = 0.f;
#pragma unroll
            for (int ib = 0; ib < 8; ++ib) tot += __shfl(bs, (threadIdx.x & 32) + 4 * ib, 64);
            if (!active) return;
            if ((lane32 & 3) == 0) *reinterpret_cast<float *>(blk + 264 + 4 * (lane32 >> 2)) = bs;
            if (lane32 == 0) { *reinterpret_cast<float *>(blk) = d; *reinterpret_cast<float *>(blk + 4) = tot; }
        } else {
            const int s16s = (int)(short)s16;                       // int16 storage
            int tot = (lane32 & 1) ? 0 : s16s; tot = group_sum<32>(tot);
            if (!active) return;
            if ((lane32 & 1) == 0) *reinterpret_cast<int16_t *>(blk + 264 + 2 * (lane32 >> 1)) = (int16_t)s16;
            if (lane32 == 0) { *reinterpret_cast<float *>(blk) = d; *reinterpret_cast<float *>(blk + 4) = d * (float)tot; }
        }
        uint32_t *qp = reinterpret_cast<uint32_t *>(blk + 8 + 8 * lane32); qp[0] = q.x; qp[1] = q.y;
    }
}

// f32 rows -> f16 activations for the MFMA path, in the SLAB layout the GEMM streams:
//     X16[slab = k / 64][row < xrows][k % 64]      (128 bytes per row per slab; rows >= nrows are written as zeros)
// with the middle two of every 4 consecutive k swapped (order 0,2,1,3): the weight fragments of the GEMM come out in that order.
// so the activation tile of a workgroup (32*NT consecutive rows x 64 k) is ONE contiguous run of 4*NT KiB.  With plain row-major
// f16 rows the tile is 32*NT pieces of 128 B at a stride of 2*K bytes: for K = 4096 every piece maps to the same L2 channel.
static __device__ __forceinline__ long x16_slab_index(long row, long k, long xrows) { return ((k >> 6) * xrows + row) * 64 + (k & 63); }
// One workgroup per destination row (grid.x = xrows: no 65535 limit).  The source row is either row `r` of a dense f32 matrix or -- the
// MUL_MAT_ID gather -- the activation row of the (token, slot) pair sorted to position r.
// f16 RANGE GUARD: |x| > 65504 would become inf and poison the whole output column (the reference quantizes activations to int8 with a
// per-block scale and cannot overflow).  A row whose amax exceeds 2^14 is scaled down by a power of two s (exact in f32, and the f16
// rounding of x / s is the rounding of x at the same relative precision); s goes to xscale[r] and the GEMM epilogue multiplies the
// accumulators of that token by it (exact).  Rows within range get s = 1: bit-identical to an unguarded conversion.
__global__ void __launch_bounds__(256) rows_to_f16_slab_kernel(const uint8_t *B, long strideB, int n_b, long nb11, long nb12, int n_used, const int *pairs_sorted, long npairs,
                                                               long K, long nrows, __half *dst, long xrows, float *xscale) {
    __shared__ float red[4];
    const long r = blockIdx.x;
    const float *src = nullptr;
    if (pairs_sorted) {
        if (r < npairs) { int pr = pairs_sorted[r]; if (pr < 0 || pr >= npairs) pr = 0;        // rows of invalid ids leave the tail of pairs_sorted unwritten
            const int t = pr / n_used, sl = pr - t * n_used; src = reinterpret_cast<const float *>(B + (long)t * nb12 + (n_b == 1 ? 0 : (long)sl * nb11)); }
    } else if (r < nrows) src = reinterpret_cast<const float *>(B + r * strideB);
    float amax = 0.f;
    // rows of up to 16384 values: the thread's (up to 16) float4 pieces are requested TOGETHER and stay in registers for the second pass -- one memory round trip per row
    // instead of two dependent ones with four loads in flight each (the launch is latency-bound: 512 x 4096 took 6.3 us, 512 x 14336 18 us)
    constexpr int RP = 16; float4 keep[RP]; const bool in_regs = K <= 1024L * RP;
    if (src && in_regs) {
#pragma unroll
        for (int i = 0; i < RP; ++i) { const long k = 4L * threadIdx.x + 1024L * i; keep[i] = k < K ? *reinterpret_cast<const float4 *>(src + k) : make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
        for (int i = 0; i < RP; ++i) amax = fmaxf(amax, fmaxf(fmaxf(fabsf(keep[i].x), fabsf(keep[i].y)), fmaxf(fabsf(keep[i].z), fabsf(keep[i].w))));
    } else
    if (src) for (long k = 4L * threadIdx.x; k < K; k += 1024) { const float4 v = *reinterpret_cast<const float4 *>(src + k); amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)))); }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float s = 1.f, inv = 1.f;
    if (amax > 16384.f && amax < 3.0e38f) { int e; (void)frexpf(amax, &e); s = ldexpf(1.f, e - 14); inv = ldexpf(1.f, 14 - e); }     // amax / s in [2^13, 2^14)
    if (threadIdx.x == 0) xscale[r] = s;
    if (src && in_regs) {
#pragma unroll
        for (int i = 0; i < RP; ++i) {
            const long k = 4L * threadIdx.x + 1024L * i;
            if (k < K) {
                const float4 v = make_float4(keep[i].x * inv, keep[i].y * inv, keep[i].z * inv, keep[i].w * inv);
                __half2 *o = reinterpret_cast<__half2 *>(dst + x16_slab_index(r, k, xrows));
                o[0] = __floats2half2_rn(v.x, v.z); o[1] = __floats2half2_rn(v.y, v.w);
            }
        }
        return;
    }
    for (long k = 4L * threadIdx.x; k < K; k += 1024) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (src) { v = *reinterpret_cast<const float4 *>(src + k); v.x *= inv; v.y *= inv; v.z *= inv; v.w *= inv; }
        __half2 *o = reinterpret_cast<__half2 *>(dst + x16_slab_index(r, k, xrows));
        o[0] = __floats2half2_rn(v.x, v.z); o[1] = __floats2half2_rn(v.y, v.w);      // k order (0,2,1,3) inside every group of 4: gemm_mfma.cuh pack8
    }
}

// ------------------------------------------------------------------------------------------------
// run-time repack base <-> row-interleaved (_R4), the device form of iqk_repack_tensor (iqk_quantize.cpp:8535-8582) and its
// inverse.  One thread per (row, block); in both layouts a row's bytes / nibbles / bit-fields are owned by that row alone
// (shared bytes such as scales_h only mix entries of the SAME row), so threads never write the same byte.
// The integer fields of a block are pulled out element by element with the layouts' position maps and re-inserted -- a
// pure permutation of bits, hence lossless in both directions.  Upload-time tool, not a hot kernel.
template <int BASE, bool TO_R4>
__global__ void repack_r4_kernel(const uint8_t *src, uint8_t *dst, long nrows, long K, long stride) {
    constexpr int BS = type_block_elems(BASE), TS = type_block_bytes(BASE);
    const long nblk = K / BS, idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nrows * nblk) return;
    const long row = idx / nblk, ibl = idx - row * nblk; const int r = (int)(row & 3);
    const uint8_t *bsrc = src + row * stride + ibl * TS;                         // base block of this row
    const uint8_t *rsrc = src + (row >> 2) * 4 * stride + ibl * (4 * TS);        // interleaved block of this row's group
    uint8_t *bdst = dst + row * stride + ibl * TS, *rdst = dst + (row >> 2) * 4 * stride + ibl * (4 * TS);
    const uint8_t *b = TO_R4 ? bsrc : rsrc; uint8_t *o = TO_R4 ? rdst : bdst;

    if (BASE == T_IQ4_NL) {                        // block_iq4_nl_r4 {half d[4]; u8 qs[64]}
        uint8_t out[16] = {0}; uint8_t acc[64];    // acc only used as scratch when writing R4 nibbles (own nibbles OR-ed into zeroed bytes)
        if (TO_R4) {
            *reinterpret_cast<uint16_t *>(o + 2 * r) = (uint16_t)ld16(b);
            for (int i = 0; i < 4; ++i) {          // own bytes 4r+i + {0,16,32,48}
                uint8_t v0 = 0, v16 = 0, v32 = 0, v48 = 0;
                auto nib = [&](int e) -> int { return e < 16 ? (b[2 + e] & 15) : (b[2 + e - 16] >> 4); };
                v0 = (uint8_t)(nib(i) | (nib(i + 8) << 4)); v16 = (uint8_t)(nib(i + 16) | (nib(i + 24) << 4));
                v32 = (uint8_t)(nib(i + 4) | (nib(i + 12) << 4)); v48 = (uint8_t)(nib(i + 20) | (nib(i + 28) << 4));
                o[8 + 4 * r + i] = v0; o[8 + 4 * r + i + 16] = v16; o[8 + 4 * r + i + 32] = v32; o[8 + 4 * r + i + 48] = v48;
            }
        } else {
            *reinterpret_cast<uint16_t *>(o) = (uint16_t)ld16(b + 2 * r);
            for (int e = 0; e < 32; ++e) { const int v = (b[8 + r4_nib_byte(0, r, e)] >> r4_nib_shift(e)) & 15; if (e < 16) out[e] |= (uint8_t)v; else out[e - 16] |= (uint8_t)(v << 4); }
            for (int i = 0; i < 16; ++i) o[2 + i] = out[i];
        }
        (void)acc; return;
    }

    // ---- 256-element super-blocks: gather the row's integer fields
    uint8_t q[256];           // element payload: Q4_K 4 bit, Q5_K 5 bit, Q6_K 6 bit (unsigned), IQ2_S/IQ3_S unused
    if (BASE == T_Q4_K || BASE == T_Q5_K) {
        int sc[8], mn[8]; uint16_t d, dm;
        if (TO_R4) {
            d = (uint16_t)ld16(b); dm = (uint16_t)ld16(b + 2);
            const uint8_t *s = b + 4, *qh = b + 16, *qs = b + (BASE == T_Q5_K ? 48 : 16);
            for (int j = 0; j < 8; ++j) { if (j < 4) { sc[j] = s[j] & 63; mn[j] = s[j + 4] & 63; } else { sc[j] = (s[j + 4] & 15) | ((s[j - 4] >> 6) << 4); mn[j] = (s[j + 4] >> 4) | ((s[j] >> 6) << 4); } }
            for (int e = 0; e < 256; ++e) { const int g = e >> 6, l = e & 31; int v = (e & 32) ? (qs[32 * g + l] >> 4) : (qs[32 * g + l] & 15);
                if (BASE == T_Q5_K) v |= ((qh[l] >> (2 * g + ((e >> 5) & 1))) & 1) << 4; q[e] = (uint8_t)v; }
            uint8_t *sh = o + 16, *sl = o + 32, *oqh = o + 64, *oqs = o + (BASE == T_Q5_K ? 192 : 64);
            *reinterpret_cast<uint16_t *>(o + 2 * r) = d; *reinterpret_cast<uint16_t *>(o + 2 * (r + 4)) = dm;
            for (int ib = 0; ib < 8; ++ib) {
                const int is = 4 * ib + r;
                sl[is] = (uint8_t)((sc[ib] & 15) | ((mn[ib] & 15) << 4));
            }
            // scales_h byte (4 ib + r) holds the high bits of entries is = 4 ib + r (low nibble) and is + 16 (high nibble): both of THIS row
            for (int ib = 0; ib < 4; ++ib) { const int is = 4 * ib + r; const int h0 = (sc[ib] >> 4) | ((mn[ib] >> 4) << 2), h1 = (sc[ib + 4] >> 4) | ((mn[ib + 4] >> 4) << 2); sh[is] = (uint8_t)(h0 | (h1 << 4)); }
            for (int ib = 0; ib < 8; ++ib)
                for (int i = 0; i < 4; ++i) {
                    const uint8_t *L = q + 32 * ib;
                    oqs[64 * ib + 4 * r + i]      = (uint8_t)((L[i] & 15) | ((L[i + 8] & 15) << 4));
                    oqs[64 * ib + 4 * r + i + 16] = (uint8_t)((L[i + 16] & 15) | ((L[i + 24] & 15) << 4));
                    oqs[64 * ib + 4 * r + i + 32] = (uint8_t)((L[i + 4] & 15) | ((L[i + 12] & 15) << 4));
                    oqs[64 * ib + 4 * r + i + 48] = (uint8_t)((L[i + 20] & 15) | ((L[i + 28] & 15) << 4));
                    if (BASE == T_Q5_K)
                        oqh[16 * ib + 4 * r + i] = (uint8_t)(((L[i] >> 4) << 0) | ((L[i + 8] >> 4) << 1) | ((L[i + 4] >> 4) << 2) | ((L[i + 12] >> 4) << 3) |
                                                             ((L[i + 16] >> 4) << 4) | ((L[i + 24] >> 4) << 5) | ((L[i + 20] >> 4) << 6) | ((L[i + 28] >> 4) << 7));
                }
        } else {
            const uint8_t *sh = b + 16, *sl = b + 32, *rqh = b + 64, *rqs = b + (BASE == T_Q5_K ? 192 : 64);
            d = (uint16_t)ld16(b + 2 * r); dm = (uint16_t)ld16(b + 2 * (r + 4));
            for (int ib = 0; ib < 8; ++ib) {
                const int is = 4 * ib + r, h = (sh[is & 15] >> (4 * (is >> 4))) & 15;
                sc[ib] = (sl[is] & 15) | ((h & 3) << 4); mn[ib] = (sl[is] >> 4) | ((h & 12) << 2);
                for (int e = 0; e < 32; ++e) {
                    int v = (rqs[r4_nib_byte(ib, r, e)] >> r4_nib_shift(e)) & 15;
                    if (BASE == T_Q5_K) { const int i = e & 3, g = e >> 2; const int bit = (g & 4) | ((g & 1) << 1) | ((g >> 1) & 1); v |= ((rqh[16 * ib + 4 * r + i] >> bit) & 1) << 4; }
                    q[32 * ib + e] = (uint8_t)v;
                }
            }
            *reinterpret_cast<uint16_t *>(o) = d; *reinterpret_cast<uint16_t *>(o + 2) = dm;
            uint8_t *s = o + 4, *oqh = o + 16, *oqs = o + (BASE == T_Q5_K ? 48 : 16);
            for (int j = 0; j < 4; ++j) {          // 6-bit packing of ggml-quants.c:2036-2043, inverted
                s[j]     = (uint8_t)((sc[j] & 63) | ((sc[j + 4] >> 4) << 6));
                s[j + 4] = (uint8_t)((mn[j] & 63) | ((mn[j + 4] >> 4) << 6));
                s[j + 8] = (uint8_t)((sc[j + 4] & 15) | ((mn[j + 4] & 15) << 4));
            }
            for (int g = 0; g < 4; ++g) for (int l = 0; l < 32; ++l) oqs[32 * g + l] = (uint8_t)((q[64 * g + l] & 15) | ((q[64 * g + 32 + l] & 15) << 4));
            if (BASE == T_Q5_K) for (int l = 0; l < 32; ++l) { int v = 0; for (int j = 0; j < 8; ++j) v |= ((q[32 * j + l] >> 4) & 1) << j; oqh[l] = (uint8_t)v; }
        }
        return;
    }
    if (BASE == T_Q6_K) {     // block_q6_k_r4 {half d[4]; i8 scales[64]; u8 qh[256]; u8 ql[512]}
        uint8_t scv[16]; uint16_t d;
        if (TO_R4) {
            d = (uint16_t)ld16(b + 208);
            for (int i = 0; i < 16; ++i) scv[i] = b[192 + i];
            for (int e = 0; e < 256; ++e) { const int n = e >> 7, rr = e & 127, l = rr & 31, k = rr >> 5; const uint8_t *ql = b + 64 * n, *qh = b + 128 + 32 * n;
                const int lo = (k & 2) ? (ql[l + 32 * (k & 1)] >> 4) : (ql[l + 32 * (k & 1)] & 15); q[e] = (uint8_t)(lo | (((qh[l] >> (2 * k)) & 3) << 4)); }
            uint8_t *scales = o + 8, *oqh = o + 72, *oql = o + 328;
            *reinterpret_cast<uint16_t *>(o + 2 * r) = d;
            for (int ib = 0; ib < 8; ++ib) {
                scales[8 * ib + r] = scv[2 * ib]; scales[8 * ib + r + 4] = scv[2 * ib + 1];
                const uint8_t *L = q + 32 * ib;
                for (int i = 0; i < 4; ++i) {
                    oql[64 * ib + 4 * r + i]      = (uint8_t)((L[i] & 15) | ((L[i + 8] & 15) << 4));
                    oql[64 * ib + 4 * r + i + 16] = (uint8_t)((L[i + 16] & 15) | ((L[i + 24] & 15) << 4));
                    oql[64 * ib + 4 * r + i + 32] = (uint8_t)((L[i + 4] & 15) | ((L[i + 12] & 15) << 4));
                    oql[64 * ib + 4 * r + i + 48] = (uint8_t)((L[i + 20] & 15) | ((L[i + 28] & 15) << 4));
                    oqh[32 * ib + 4 * r + i]      = (uint8_t)((L[i] >> 4) | ((L[i + 8] >> 4) << 2) | ((L[i + 4] >> 4) << 4) | ((L[i + 12] >> 4) << 6));
                    oqh[32 * ib + 4 * r + i + 16] = (uint8_t)((L[i + 16] >> 4) | ((L[i + 24] >> 4) << 2) | ((L[i + 20] >> 4) << 4) | ((L[i + 28] >> 4) << 6));
                }
            }
        } else {
            const uint8_t *scales = b + 8, *rqh = b + 72, *rql = b + 328;
            d = (uint16_t)ld16(b + 2 * r);
            for (int ib = 0; ib < 8; ++ib) {
                scv[2 * ib] = scales[8 * ib + r]; scv[2 * ib + 1] = scales[8 * ib + r + 4];
                for (int e = 0; e < 32; ++e) { const int i = e & 3, g = e >> 2; const int shq = ((g & 1) << 2) | (g & 2);
                    q[32 * ib + e] = (uint8_t)(((rql[r4_nib_byte(ib, r, e)] >> r4_nib_shift(e)) & 15) | (((rqh[32 * ib + 4 * r + i + 16 * (g >> 2)] >> shq) & 3) << 4)); }
            }
            *reinterpret_cast<uint16_t *>(o + 208) = d;
            for (int i = 0; i < 16; ++i) o[192 + i] = scv[i];
            for (int n = 0; n < 2; ++n) for (int l = 0; l < 32; ++l) {
                const uint8_t *L = q + 128 * n;
                o[64 * n + l]      = (uint8_t)((L[l] & 15) | ((L[l + 64] & 15) << 4));
                o[64 * n + l + 32] = (uint8_t)((L[l + 32] & 15) | ((L[l + 96] & 15) << 4));
                o[128 + 32 * n + l] = (uint8_t)((L[l] >> 4) | ((L[l + 32] >> 4) << 2) | ((L[l + 64] >> 4) << 4) | ((L[l + 96] >> 4) << 6));
            }
        }
        return;
    }
    if (BASE == T_IQ2_S) {    // block_iq2_s_r4 {half d[4]; u8 qs[128]; u8 qh[32]; u8 signs[128]; u8 scales[32]} : whole bytes move
        if (TO_R4) {
            *reinterpret_cast<uint16_t *>(o + 2 * r) = (uint16_t)ld16(b);
            for (int ib = 0; ib < 8; ++ib) { o[296 + 4 * ib + r] = b[74 + ib]; o[136 + 4 * ib + r] = b[66 + ib];
                for (int i = 0; i < 4; ++i) { o[8 + 16 * ib + 4 * r + i] = b[2 + 4 * ib + i]; o[168 + 16 * ib + 4 * r + i] = b[34 + 4 * ib + i]; } }
        } else {
            *reinterpret_cast<uint16_t *>(o) = (uint16_t)ld16(b + 2 * r);
            for (int ib = 0; ib < 8; ++ib) { o[74 + ib] = b[296 + 4 * ib + r]; o[66 + ib] = b[136 + 4 * ib + r];
                for (int i = 0; i < 4; ++i) { o[2 + 4 * ib + i] = b[8 + 16 * ib + 4 * r + i]; o[34 + 4 * ib + i] = b[168 + 16 * ib + 4 * r + i]; } }
        }
        return;
    }
    if (BASE == T_IQ3_S) {    // block_iq3_s_r4 {half d[4]; u8 qs[256]; u8 qh[32]; u8 signs[128]; u8 scales[16]}
        if (TO_R4) {
            const uint8_t *bqs = b + 2, *bqh = b + 66, *bsg = b + 74, *bsc = b + 106;
            uint8_t *qs = o + 8, *qh = o + 264, *sg = o + 296, *scl = o + 424;
            *reinterpret_cast<uint16_t *>(o + 2 * r) = (uint16_t)ld16(b);
            for (int ib = 0; ib < 4; ++ib) {       // scales byte l%16 holds entries l and l+16 (same row): ib and ib+4
                const int lo = (bsc[ib >> 1] >> (4 * (ib & 1))) & 15, hi = (bsc[(ib + 4) >> 1] >> (4 * (ib & 1))) & 15;
                scl[4 * ib + r] = (uint8_t)(lo | (hi << 4));
            }
            for (int ib = 0; ib < 8; ++ib) {
                qh[4 * ib + r] = bqh[ib];
                for (int i = 0; i < 8; ++i) qs[32 * ib + r + 8 * (i & 3) + 4 * (i >> 2)] = bqs[8 * ib + i];
                for (int j = 0; j < 4; ++j) { int v = 0;
                    for (int bit = 0; bit < 8; ++bit) { const int e = 16 * (bit >> 2) + 4 * (bit & 3) + j; v |= ((bsg[4 * ib + (e >> 3)] >> (e & 7)) & 1) << bit; }
                    sg[16 * ib + 4 * r + j] = (uint8_t)v; }
            }
        } else {
            const uint8_t *qs = b + 8, *qh = b + 264, *sg = b + 296, *scl = b + 424;
            uint8_t *oqs = o + 2, *oqh = o + 66, *osg = o + 74, *osc = o + 106;
            *reinterpret_cast<uint16_t *>(o) = (uint16_t)ld16(b + 2 * r);
            for (int k2 = 0; k2 < 4; ++k2) {       // base scales[k2]: low nibble = ib 2k2, high = ib 2k2+1
                const int l0 = 4 * (2 * k2) + r, l1 = 4 * (2 * k2 + 1) + r;
                osc[k2] = (uint8_t)(((scl[l0 & 15] >> (4 * (l0 >> 4))) & 15) | (((scl[l1 & 15] >> (4 * (l1 >> 4))) & 15) << 4));
            }
            for (int ib = 0; ib < 8; ++ib) {
                oqh[ib] = qh[4 * ib + r];
                for (int i = 0; i < 8; ++i) oqs[8 * ib + i] = qs[32 * ib + r + 8 * (i & 3) + 4 * (i >> 2)];
                for (int l = 0; l < 4; ++l) { int v = 0;
                    for (int bit = 0; bit < 8; ++bit) { const int e = 8 * l + bit; v |= ((sg[16 * ib + 4 * r + (e & 3)] >> (((e >> 2) & 3) + 4 * (e >> 4))) & 1) << bit; }
                    osg[4 * ib + l] = (uint8_t)v; }
            }
        }
        return;
    }
}

// ------------------------------------------------------------------------------------------------
// In-process GGML_OP_REDUCE (ADD): the single-process form of the tensor-parallel exchange (ggml.c:6166-6189, reduce.cu:125-598).
// One launch on the executing device reads every partial directly from its owner's HBM (peer access over xGMI), adds them in f32 in
// ascending device order (deterministic) and writes the sum back into EVERY listed buffer (partials and copy-only targets alike).
// Sized for the decode message (16-32 KB: one small launch instead of an N-step ring); prompt-size messages go the same way.
#define REDUCE_MAX_PEERS 16
// Sliced form (cdna4_reduce_peers_slice; the reference's variant for prompt-size messages where every device reduces its own 1/N of the vector, reduce.cu:448-533): the launch
// covers the 16-byte vectors [v_begin, v_end) only (+ the ragged tail when `tail` is set), and the host runs one such launch per device on that device's stream -- every GPU
// then reads (N-1)/N and writes (N-1)/N of the message over its own links instead of one GPU moving 2 (N-1) x the message.
struct ReducePeersArgs { void *buf[REDUCE_MAX_PEERS]; int n; unsigned partial_mask; long count; long v_begin, v_end; int tail; };
template <typename T>
__global__ void reduce_peers_kernel(const ReducePeersArgs a) {
    constexpr int V = 16 / sizeof(T);                      // elements per 16-byte access
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const long nvec = a.count / V;
    for (long i = a.v_begin + (long)blockIdx.x * blockDim.x + threadIdx.x; i < a.v_end; i += (long)gridDim.x * blockDim.x) {
        float acc[V];
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] = 0.f;
        for (int j = 0; j < a.n; ++j) {
            if (!a.buf[j] || !((a.partial_mask >> j) & 1u)) continue;
            union { u32x4 v; T t[V]; } u; u.v = reinterpret_cast<const u32x4 *>(a.buf[j])[i];
#pragma unroll
            for (int e = 0; e < V; ++e) acc[e] += (float)u.t[e];
        }
        union { u32x4 v; T t[V]; } o;
#pragma unroll
        for (int e = 0; e < V; ++e) o.t[e] = (T)acc[e];
        for (int j = 0; j < a.n; ++j) if (a.buf[j]) reinterpret_cast<u32x4 *>(a.buf[j])[i] = o.v;
    }
    // tail elements (count not a multiple of V)
    if (a.tail)
    for (long i = nvec * V + (long)blockIdx.x * blockDim.x + threadIdx.x; i < a.count; i += (long)gridDim.x * blockDim.x) {
        float acc = 0.f;
        for (int j = 0; j < a.n; ++j) if (a.buf[j] && ((a.partial_mask >> j) & 1u)) acc += (float)reinterpret_cast<const T *>(a.buf[j])[i];
        for (int j = 0; j < a.n; ++j) if (a.buf[j]) reinterpret_cast<T *>(a.buf[j])[i] = (T)acc;
    }
}

// Q8_0 partial sums (the reference's cparams.reduce_type = q8_0: prompt-size partials travel as block_q8_0, reduce.cu:20-43 k_add<block_q8_0>): one thread per 32-block
// sums the de-quantized partials in f32 in ascending device order and re-quantizes ONCE -- d = amax / 127 kept in f32 for the division, stored as f16, q = roundf(x / d) --
// into every listed buffer.  (The reference adds pairwise along its ring and re-quantizes after every hop: one rounding here instead of N - 1, so the results agree with
// it to the Q8_0 step, not bit for bit; with two partials the arithmetic is identical.)
__global__ void reduce_peers_q8_0_kernel(const ReducePeersArgs a) {
    const long nb = a.count / 32;                           // blocks of {f16 d; int8 qs[32]} = 34 bytes
    for (long ib = a.v_begin + (long)blockIdx.x * blockDim.x + threadIdx.x; ib < a.v_end; ib += (long)gridDim.x * blockDim.x) {
        if (ib >= nb) break;
        float x[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) x[i] = 0.f;
        for (int j = 0; j < a.n; ++j) {
            if (!a.buf[j] || !((a.partial_mask >> j) & 1u)) continue;
            const uint8_t *b = reinterpret_cast<const uint8_t *>(a.buf[j]) + ib * 34;
            const float d = half_bits_to_float(ld16(b));
#pragma unroll
            for (int i = 0; i < 32; ++i) x[i] += d * (float)(int)(int8_t)b[2 + i];
        }
        float amax = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) amax = fmaxf(amax, fabsf(x[i]));
        const float d = amax / 127, id = d > 0 ? 1 / d : 0.f;
        uint8_t o[34]; const __half dh = __float2half_rn(d); memcpy(o, &dh, 2);
#pragma unroll
        for (int i = 0; i < 32; ++i) o[2 + i] = (uint8_t)(int8_t)(int)roundf(x[i] * id);
        for (int j = 0; j < a.n; ++j) if (a.buf[j]) { uint8_t *w = reinterpret_cast<uint8_t *>(a.buf[j]) + ib * 34;
#pragma unroll
            for (int i = 0; i < 17; ++i) reinterpret_cast<uint16_t *>(w)[i] = reinterpret_cast<const uint16_t *>(o)[i]; }
    }
}

// ---- MUL_MAT_ID grouping on the device (replaces the host-side mmid_row_mapping + D2H sync of ggml-cuda.cu:2786-2834 and the
// CPU's matrix_rows construction ggml.c:18146-18205).  One workgroup: count pairs per expert (LDS atomics), scan, emit
//   pairs_sorted[pos] = token * n_used + slot   (grouped by expert)
//   tiles[i] = {expert, first sorted row, valid rows}  for every BN-row token tile (unused tiles: expert = -1)
// and zero the output rows of invalid ids (ggml.c:18178-18187).
__global__ void __launch_bounds__(1024) moe_sort_kernel(const int32_t *ids, long ids_nb1, int n_tokens, int n_used, int n_expert, int BN, int max_tiles,
                                                        int *pairs_sorted, int *tiles, float *C, long nb1, long nb2, int M) {
    extern __shared__ int sm[];            // counts[n_expert], offsets[n_expert + 1], cursor[n_expert]
    int *counts = sm, *offsets = sm + n_expert, *cursor = offsets + n_expert + 1;
    const int npairs = n_tokens * n_used;
    for (int e = threadIdx.x; e < n_expert; e += blockDim.x) counts[e] = 0;
    __syncthreads();
    auto id_of = [&](int p) { const int t = p / n_used; return reinterpret_cast<const int32_t *>(reinterpret_cast<const uint8_t *>(ids) + (long)t * ids_nb1)[p - t * n_used]; };
    for (int p = threadIdx.x; p < npairs; p += blockDim.x) { const int e = id_of(p); if (e >= 0 && e < n_expert) atomicAdd(&counts[e], 1); }
    __syncthreads();
    // exclusive scans over the experts of (pair count, tile count): thread e owns expert e (n_expert <= blockDim.x, checked by the host);
    // the packed 64-bit sum (tiles << 32 | pairs) is scanned once (Hillis-Steele in LDS, log2(1024) = 10 rounds)
    unsigned long long *scan = reinterpret_cast<unsigned long long *>(cursor + n_expert + (n_expert & 1 ? 0 : 1));   // 8-byte aligned scratch behind cursor[]
    {
        const int e = threadIdx.x;
        const int c = e < n_expert ? counts[e] : 0;
        unsigned long long v = ((unsigned long long)((c + BN - 1) / BN) << 32) | (unsigned)c;
        scan[e] = v;
        __syncthreads();
        for (int d = 1; d < (int)blockDim.x; d <<= 1) {
            const unsigned long long add = e >= d ? scan[e - d] : 0ull;
            __syncthreads();
            v += add; scan[e] = v;
            __syncthreads();
        }
        const unsigned long long total = scan[blockDim.x - 1];
        const unsigned long long excl = v - (((unsigned long long)((c + BN - 1) / BN) << 32) | (unsigned)c);
        const int off = (int)(excl & 0xffffffffu); int nt = (int)(excl >> 32);
        if (e < n_expert) {
            offsets[e] = off; cursor[e] = off;
            for (int r = 0; r < c; r += BN) { tiles[3 * nt] = e; tiles[3 * nt + 1] = off + r; tiles[3 * nt + 2] = min(BN, c - r); ++nt; }
        }
        if (e == 0) offsets[n_expert] = (int)(total & 0xffffffffu);
        for (int i = (int)(total >> 32) + e; i < max_tiles; i += blockDim.x) { tiles[3 * i] = -1; tiles[3 * i + 1] = 0; tiles[3 * i + 2] = 0; }
    }
    __syncthreads();
    for (int p0 = 0; p0 < npairs; p0 += blockDim.x) {
        const int p = p0 + threadIdx.x; const bool live = p < npairs;
        const int e = live ? id_of(p) : 0; const bool valid = e >= 0 && e < n_expert;
        if (live && valid) pairs_sorted[atomicAdd(&cursor[e], 1)] = p;
        // rows of invalid ids are zeroed (ggml.c:18178-18187) by the whole wave, one row at a time (they are rare; one thread per row
        // made a batch with many of them crawl)
        unsigned long long bad = __ballot(live && !valid);
        while (bad) {
            const int l = __builtin_ctzll(bad); bad &= bad - 1;
            const int pb = p0 + (threadIdx.x & ~63) + l, t = pb / n_used;
            float *row = C + (long)t * nb2 + (long)(pb - t * n_used) * nb1;
            for (int i = threadIdx.x & 63; i < M; i += 64) row[i] = 0.f;
        }
    }
    // within an expert the order of pairs depends on atomics; every output row is computed independently, so results do not
}

// one-time (per context) expansion of the codebooks into global memory (layout: gemv.cuh "LDS tables", iq_tables_offset)
__global__ void iq_tables_init_kernel(const uint16_t *packed, uint8_t *out) {
    expand_iq2s_grid(packed + GRID_IQ2S, out); expand_iq3s_grid(packed + GRID_IQ3S, out + iq_tables_offset(T_IQ3_S));
    expand_iq2_grid(packed + GRID_IQ2XXS, 256, out + iq_tables_offset(T_IQ2_XXS)); expand_iq2_grid(packed + GRID_IQ2XS, 512, out + iq_tables_offset(T_IQ2_XS));
    expand_iq3xxs_grid(packed + GRID_IQ3XXS, out + iq_tables_offset(T_IQ3_XXS));
    expand_iq1_grid(packed + GRID_IQ1S, out + iq_tables_offset(T_IQ1_S));
}

// ------------------------------------------------------------------------------------------------
// GET_ROWS (ggml.c:19808, ggml-cuda/getrows.cu): dst[i0, i10, i11, i12] = to_float(src[:, ids[i10, i11, i12], i11, i12])[i0].  TYPE = F32, F16 or a
// base quant type (de-quantized with the L0 element decoder: token_embd rows of a quantized model).
template <int TYPE>
__global__ void get_rows_kernel(TD src, TD ids, TD dst, const uint16_t *grid, long total) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long i0 = i % dst.ne[0]; long r = i / dst.ne[0]; const long i10 = r % dst.ne[1]; r /= dst.ne[1]; const long i11 = r % dst.ne[2], i12 = r / dst.ne[2];
        const long row = *reinterpret_cast<const int32_t *>(ids.data + i10 * ids.nb[0] + i11 * ids.nb[1] + i12 * ids.nb[2]);
        const char *sr = src.data + row * src.nb[1] + (src.ne[2] == 1 ? 0 : i11) * src.nb[2] + (src.ne[3] == 1 ? 0 : i12) * src.nb[3];
        float v;
        if (TYPE == T_F32) v = reinterpret_cast<const float *>(sr)[i0];
        else if (TYPE == T_F16) v = __half2float(reinterpret_cast<const __half *>(sr)[i0]);
        else { constexpr int BS = type_block_elems(TYPE), TS = type_block_bytes(TYPE); v = dequant_base_elem<TYPE>(reinterpret_cast<const uint8_t *>(sr) + type_row_meta(TYPE) + (i0 / BS) * TS, (int)(i0 % BS), grid, reinterpret_cast<const uint8_t *>(sr)); }
        *reinterpret_cast<float *>(dst.data + i0 * dst.nb[0] + i10 * dst.nb[1] + i11 * dst.nb[2] + i12 * dst.nb[3]) = v;
    }
}
