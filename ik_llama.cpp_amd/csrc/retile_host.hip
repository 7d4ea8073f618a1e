// Host-side re-tiling of the row-interleaved (`_R4` / `_R8`) weight formats that have no device re-tiling kernel <-> their base types: IQ2_K_R4 IQ3_K_R4 IQ4_K_R4 IQ5_K_R4
// IQ4_KS_R4 IQ5_KS_R4 (the forms the reference CUDA backend lists for MUL_MAT, ggml-cuda.cu:4893-4898), Q4_0_R8 Q5_0_R4 Q6_0_R4 Q8_0_R8 MXFP4_R8 Q2_K_R4 Q3_K_R4 IQ4_XS_R8
// IQ2_XXS_R4 IQ2_XS_R4 IQ3_XXS_R4 IQ2_BN_R4 (CPU-only in the reference).
//
// The interleave exists so that one AVX load of activations feeds four (eight) rows; a 64-lane wavefront amortises the activations anyway, so on MI355X these tensors are
// stored in the BASE tiling and served by the base types' kernels.  The conversion runs once per tensor at upload (and its inverse at download), on the host,
// between the file bytes and the H2D copy.  Layouts restated from the reference's repack functions (iqk_quantize.cpp:5304-8088, cited per format below) and the block structs
// of ggml-common.h.  Both directions go through ONE description per format (a block is decoded into its logical fields, the fields are encoded into the other format), so the
// two directions cannot disagree; tests/test_retile_host.py pins the bytes against the reference's own iqk_repack_tensor (live and as committed fixtures) and checks the round trips.
//
// No device code in this translation unit.  Built with -fno-vectorize -fno-slp-vectorize (build.py): this toolchain's clang -O3 turns the 4-byte-per-row
// accesses of the interleaved formats into 16-byte loads / read-modify-writes that reach past a row's bytes and past the end of the buffer (found by
// the guard bytes of tests/test_retile_host.py and by AddressSanitizer); the conversion runs once per tensor, its speed is that of the H2D copy beside it.
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/ggml_hip_cdna4.h"

int cdna4_set_err(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));     // (api_internal.h: thread-local cdna4_last_error(); returns `code`)
#define set_err cdna4_set_err

namespace {

// logical content of one 256-weight super-block of one row
struct Fields {
    uint16_t d, dmin;      // f16 bits of the block scale (types with a per-row scale: unused) and, Q2_K, of the block minimum
    uint8_t  ex[16];       // the "shifted table" flag of each 16-weight sub-block (types with 32-weight sub-blocks: unused)
    uint8_t  sl[16];       // scale, low part (4 bits per 16-weight sub-block; the *_KS types: the whole scale byte of a 32-weight sub-block in sl[0..7])
    uint8_t  sh[16];       // scale, high part (0, 1 or 2 bits)
    uint8_t  L[256];       // quant index of every weight (2 ... 5 bits)
};

// ---- the base formats ------------------------------------------------------------------------------------------------------------------
// header of IQ2_K / IQ3_K / IQ4_K / IQ5_K blocks: {f16 d; u16 extra; ...}: bit 2 ib + h of `extra` belongs to half h of 32-block ib
inline void hdr_get(const uint8_t *b, Fields &f) { uint16_t ex; memcpy(&f.d, b, 2); memcpy(&ex, b + 2, 2); for (int j = 0; j < 16; ++j) f.ex[j] = (ex >> j) & 1; }
inline void hdr_put(uint8_t *b, const Fields &f) { uint16_t ex = 0; for (int j = 0; j < 16; ++j) ex |= (uint16_t)(f.ex[j] & 1) << j; memcpy(b, &f.d, 2); memcpy(b + 2, &ex, 2); }
// scales_l[8]: low nibble = sub-block 2 ib, high nibble = sub-block 2 ib + 1
inline void sl_get(const uint8_t *s, Fields &f) { for (int ib = 0; ib < 8; ++ib) { f.sl[2 * ib] = s[ib] & 0xf; f.sl[2 * ib + 1] = s[ib] >> 4; } }
inline void sl_put(uint8_t *s, const Fields &f) { for (int ib = 0; ib < 8; ++ib) s[ib] = (f.sl[2 * ib] & 0xf) | (f.sl[2 * ib + 1] << 4); }
// 2-bit planes qs[64]: weights 128 g + 32 p + j (p = 0..3, j = 0..31) sit in byte 32 g + j at bit 2 p
inline void q2_get(const uint8_t *qs, Fields &f) { for (int e = 0; e < 256; ++e) f.L[e] = (qs[32 * (e >> 7) + (e & 31)] >> (2 * ((e >> 5) & 3))) & 3; }
inline void q2_put(uint8_t *qs, const Fields &f) { for (int g = 0; g < 2; ++g) for (int j = 0; j < 32; ++j) { const uint8_t *l = f.L + 128 * g + j; qs[32 * g + j] = (uint8_t)((l[0] & 3) | ((l[32] & 3) << 2) | ((l[64] & 3) << 4) | ((l[96] & 3) << 6)); } }
// IQ3_K third bit qh[32]: weight 32 ib + j in byte j at bit ib
inline void q3h_get(const uint8_t *qh, Fields &f) { for (int e = 0; e < 256; ++e) f.L[e] |= ((qh[e & 31] >> (e >> 5)) & 1) << 2; }
inline void q3h_put(uint8_t *qh, const Fields &f) { for (int j = 0; j < 32; ++j) { unsigned v = 0; for (int ib = 0; ib < 8; ++ib) v |= (unsigned)((f.L[32 * ib + j] >> 2) & 1) << ib; qh[j] = (uint8_t)v; } }
// IQ4_K / IQ4_KS nibbles qs[128]: weights 32 ib + j (low nibble) and 32 ib + 16 + j (high nibble) in byte 16 ib + j
inline void q4_get(const uint8_t *qs, Fields &f) { for (int ib = 0; ib < 8; ++ib) for (int j = 0; j < 16; ++j) { f.L[32 * ib + j] = qs[16 * ib + j] & 0xf; f.L[32 * ib + 16 + j] = qs[16 * ib + j] >> 4; } }
inline void q4_put(uint8_t *qs, const Fields &f) { for (int ib = 0; ib < 8; ++ib) for (int j = 0; j < 16; ++j) qs[16 * ib + j] = (f.L[32 * ib + j] & 0xf) | ((f.L[32 * ib + 16 + j] & 0xf) << 4); }
// IQ5_K / IQ5_KS: nibbles qs[128]: weights 64 g + j (low) and 64 g + 32 + j (high) in byte 32 g + j; fifth bit qh[32]: byte j, bit 2 g + (high ? 1 : 0)
inline void q5_get(const uint8_t *qs, const uint8_t *qh, Fields &f) {
    for (int g = 0; g < 4; ++g) for (int j = 0; j < 32; ++j) {
        f.L[64 * g + j]      = (qs[32 * g + j] & 0xf) | (((qh[j] >> (2 * g)) & 1) << 4);
        f.L[64 * g + 32 + j] = (qs[32 * g + j] >> 4)  | (((qh[j] >> (2 * g + 1)) & 1) << 4);
    }
}
inline void q5_put(uint8_t *qs, uint8_t *qh, const Fields &f) {
    for (int j = 0; j < 32; ++j) {
        unsigned h = 0;
        for (int g = 0; g < 4; ++g) {
            const uint8_t a = f.L[64 * g + j], b = f.L[64 * g + 32 + j];
            qs[32 * g + j] = (uint8_t)((a & 0xf) | ((b & 0xf) << 4));
            h |= (unsigned)((a >> 4) & 1) << (2 * g) | (unsigned)((b >> 4) & 1) << (2 * g + 1);
        }
        qh[j] = (uint8_t)h;
    }
}

// ---- the interleaved formats: one block holds super-block ibl of FOUR rows; row k of the group ------------------------------------------
// extra[8]: byte k = first halves, byte k + 4 = second halves of row k's 32-blocks (bit ib)
inline void ex4_get(const uint8_t *ex, int k, Fields &f) { for (int ib = 0; ib < 8; ++ib) { f.ex[2 * ib] = (ex[k] >> ib) & 1; f.ex[2 * ib + 1] = (ex[k + 4] >> ib) & 1; } }
inline void ex4_put(uint8_t *ex, int k, const Fields &f) { uint8_t a = 0, b = 0; for (int ib = 0; ib < 8; ++ib) { a |= (f.ex[2 * ib] & 1) << ib; b |= (f.ex[2 * ib + 1] & 1) << ib; } ex[k] = a; ex[k + 4] = b; }
// the 64 sub-block scales of the four rows are numbered i = 8 ib + 4 h + k (h = half of the 32-block); scales_l[32]: nibble i / 32 of byte i % 32;
// scales_h: SHB bits per scale, byte i % NH at bit SHB * (i / NH) with NH = 64 * SHB / 8 bytes
inline int  sc_index(int ib, int h, int k) { return 8 * ib + 4 * h + k; }
inline void sl4_get(const uint8_t *s, int k, Fields &f) { for (int j = 0; j < 16; ++j) { const int i = sc_index(j >> 1, j & 1, k); f.sl[j] = (s[i & 31] >> (4 * (i >> 5))) & 0xf; } }
inline void sl4_put(uint8_t *s, int k, const Fields &f) { for (int j = 0; j < 8; ++j) { const int i = sc_index(j >> 1, j & 1, k); s[i] = (uint8_t)((f.sl[j] & 0xf) | ((f.sl[j + 8] & 0xf) << 4)); } }     // (i < 32: sub-block j of 32-blocks 0..3 = low nibble, sub-block j + 8 = the same byte's high nibble)
template <int SHB> inline void sh4_get(const uint8_t *s, int k, Fields &f) { constexpr int NH = 8 * SHB; for (int j = 0; j < 16; ++j) { const int i = sc_index(j >> 1, j & 1, k); f.sh[j] = (s[i % NH] >> (SHB * (i / NH))) & ((1 << SHB) - 1); } }
template <int SHB> inline void sh4_put(uint8_t *s, int k, const Fields &f) {         // every byte of scales_h belongs to one row (byte % 4 == k): assembled, then stored once
    constexpr int NH = 8 * SHB, PER = 8 / SHB;           // PER scales per byte: scale numbers bb, bb + NH, bb + 2 NH, ...
    for (int bb = k; bb < NH; bb += 4) { unsigned v = 0; for (int t = 0; t < PER; ++t) { const int i = bb + NH * t; v |= (unsigned)(f.sh[2 * (i >> 3) + ((i >> 2) & 1)] & ((1 << SHB) - 1)) << (SHB * t); } s[bb] = (uint8_t)v; }
}
// 2-bit quants qs[256]: weight e = 16 h + 4 s + i of 32-block ib in byte 32 ib + 16 h + 4 k + i at bit 2 s
inline void q2r_get(const uint8_t *qs, int k, Fields &f) { for (int ib = 0; ib < 8; ++ib) for (int e = 0; e < 32; ++e) f.L[32 * ib + e] = (qs[32 * ib + 16 * (e >> 4) + 4 * k + (e & 3)] >> (2 * ((e >> 2) & 3))) & 3; }
inline void q2r_put(uint8_t *qs, int k, const Fields &f) { for (int ib = 0; ib < 8; ++ib) for (int h = 0; h < 2; ++h) for (int i = 0; i < 4; ++i) { const uint8_t *l = f.L + 32 * ib + 16 * h + i; qs[32 * ib + 16 * h + 4 * k + i] = (uint8_t)((l[0] & 3) | ((l[4] & 3) << 2) | ((l[8] & 3) << 4) | ((l[12] & 3) << 6)); } }
// IQ3_K_R4 third bit qh[128]: weight e = 4 q + i of 32-block ib in byte 16 ib + 4 k + i at bit q
inline void q3hr_get(const uint8_t *qh, int k, Fields &f) { for (int ib = 0; ib < 8; ++ib) for (int e = 0; e < 32; ++e) f.L[32 * ib + e] |= ((qh[16 * ib + 4 * k + (e & 3)] >> (e >> 2)) & 1) << 2; }
inline void q3hr_put(uint8_t *qh, int k, const Fields &f) { for (int ib = 0; ib < 8; ++ib) for (int i = 0; i < 4; ++i) { unsigned v = 0; for (int q = 0; q < 8; ++q) v |= (unsigned)((f.L[32 * ib + 4 * q + i] >> 2) & 1) << q; qh[16 * ib + 4 * k + i] = (uint8_t)v; } }
// 4-bit quants qs[512]: weight e = 4 q + i of 32-block ib: byte 64 ib + 32 (q & 1) + 16 (q >> 2) + 4 k + i, high nibble if q & 2
// ({0-3 | 8-11}, {16-19 | 24-27}, {4-7 | 12-15}, {20-23 | 28-31}: so that an unpacked 16-byte vector holds 4 consecutive weights of each row)
inline int  q4r_byte(int ib, int e, int k) { const int q = e >> 2; return 64 * ib + 32 * (q & 1) + 16 * (q >> 2) + 4 * k + (e & 3); }
inline void q4r_get(const uint8_t *qs, int k, Fields &f) { for (int ib = 0; ib < 8; ++ib) for (int e = 0; e < 32; ++e) f.L[32 * ib + e] = (qs[q4r_byte(ib, e, k)] >> (4 * ((e >> 3) & 1))) & 0xf; }
inline void q4r_put(uint8_t *qs, int k, const Fields &f) { for (int ib = 0; ib < 8; ++ib) for (int e = 0; e < 32; ++e) if (!(e & 8)) qs[q4r_byte(ib, e, k)] = (uint8_t)((f.L[32 * ib + e] & 0xf) | ((f.L[32 * ib + e + 8] & 0xf) << 4)); }      // (e and e + 8 share a byte)
// fifth bit qh[128]: weight e = 4 q + i in byte 16 ib + 4 k + i at bit (q >> 1) + 4 (q & 1)
inline void q5hr_get(const uint8_t *qh, int k, Fields &f) { for (int ib = 0; ib < 8; ++ib) for (int e = 0; e < 32; ++e) { const int q = e >> 2; f.L[32 * ib + e] |= ((qh[16 * ib + 4 * k + (e & 3)] >> ((q >> 1) + 4 * (q & 1))) & 1) << 4; } }
inline void q5hr_put(uint8_t *qh, int k, const Fields &f) { for (int ib = 0; ib < 8; ++ib) for (int i = 0; i < 4; ++i) { unsigned v = 0; for (int q = 0; q < 8; ++q) v |= (unsigned)((f.L[32 * ib + 4 * q + i] >> 4) & 1) << ((q >> 1) + 4 * (q & 1)); qh[16 * ib + 4 * k + i] = (uint8_t)v; } }

// ---- per type: block size, row header, and the four codecs -------------------------------------------------------------------------------
// get_base / put_base: one base block <-> its fields;  get_r / put_r: row k of one interleaved block <-> the same fields.
// Every put assembles a destination byte completely and stores it ONCE: the read-modify-write form (`byte |= field << shift` over a zeroed block) 
// also wrote bytes that belong to other rows when vectorized
struct Iq2k {
    using F = Fields; static constexpr int R = 4, EL = 256, BS = 76, ROW_META = 0;           // {d, extra, scales[8], qs[64]}            | {d[4], extra[8], scales[32], qs[256]}
    static void get_base(const uint8_t *b, Fields &f) { hdr_get(b, f); sl_get(b + 4, f); q2_get(b + 12, f); }
    static void put_base(uint8_t *b, const Fields &f) { hdr_put(b, f); sl_put(b + 4, f); q2_put(b + 12, f); }
    static void get_r(const uint8_t *b, int k, Fields &f) { memcpy(&f.d, b + 2 * k, 2); ex4_get(b + 8, k, f); sl4_get(b + 16, k, f); q2r_get(b + 48, k, f); }
    static void put_r(uint8_t *b, int k, const Fields &f) { memcpy(b + 2 * k, &f.d, 2); ex4_put(b + 8, k, f); sl4_put(b + 16, k, f); q2r_put(b + 48, k, f); }
};
struct Iq3k {
    using F = Fields; static constexpr int R = 4, EL = 256, BS = 110, ROW_META = 0;          // {d, extra, scales_h (u16), scales_l[8], qs[64], qh[32]} | {d[4], extra[8], scales_h[8], scales_l[32], qs[256], qh[128]}
    static void get_base(const uint8_t *b, Fields &f) { hdr_get(b, f); uint16_t sh; memcpy(&sh, b + 4, 2); for (int j = 0; j < 16; ++j) f.sh[j] = (sh >> j) & 1; sl_get(b + 6, f); q2_get(b + 14, f); q3h_get(b + 78, f); }
    static void put_base(uint8_t *b, const Fields &f) { hdr_put(b, f); uint16_t sh = 0; for (int j = 0; j < 16; ++j) sh |= (uint16_t)(f.sh[j] & 1) << j; memcpy(b + 4, &sh, 2); sl_put(b + 6, f); q2_put(b + 14, f); q3h_put(b + 78, f); }
    static void get_r(const uint8_t *b, int k, Fields &f) { memcpy(&f.d, b + 2 * k, 2); ex4_get(b + 8, k, f); sh4_get<1>(b + 16, k, f); sl4_get(b + 24, k, f); q2r_get(b + 56, k, f); q3hr_get(b + 312, k, f); }
    static void put_r(uint8_t *b, int k, const Fields &f) { memcpy(b + 2 * k, &f.d, 2); ex4_put(b + 8, k, f); sh4_put<1>(b + 16, k, f); sl4_put(b + 24, k, f); q2r_put(b + 56, k, f); q3hr_put(b + 312, k, f); }
};
// scales_h[4] of IQ4_K / IQ5_K: 2 bits per sub-block, sub-block j in byte j / 4 at bit 2 (j % 4)
inline void sh2_get(const uint8_t *s, Fields &f) { for (int j = 0; j < 16; ++j) f.sh[j] = (s[j >> 2] >> (2 * (j & 3))) & 3; }
inline void sh2_put(uint8_t *s, const Fields &f) { for (int b = 0; b < 4; ++b) s[b] = (uint8_t)((f.sh[4 * b] & 3) | ((f.sh[4 * b + 1] & 3) << 2) | ((f.sh[4 * b + 2] & 3) << 4) | ((f.sh[4 * b + 3] & 3) << 6)); }
struct Iq4k {
    using F = Fields; static constexpr int R = 4, EL = 256, BS = 144, ROW_META = 0;          // {d, extra, scales_h[4], scales_l[8], qs[128]}  | {d[4], extra[8], scales_h[16], scales_l[32], qs[512]}
    static void get_base(const uint8_t *b, Fields &f) { hdr_get(b, f); sh2_get(b + 4, f); sl_get(b + 8, f); q4_get(b + 16, f); }
    static void put_base(uint8_t *b, const Fields &f) { hdr_put(b, f); sh2_put(b + 4, f); sl_put(b + 8, f); q4_put(b + 16, f); }
    static void get_r(const uint8_t *b, int k, Fields &f) { memcpy(&f.d, b + 2 * k, 2); ex4_get(b + 8, k, f); sh4_get<2>(b + 16, k, f); sl4_get(b + 32, k, f); q4r_get(b + 64, k, f); }
    static void put_r(uint8_t *b, int k, const Fields &f) { memcpy(b + 2 * k, &f.d, 2); ex4_put(b + 8, k, f); sh4_put<2>(b + 16, k, f); sl4_put(b + 32, k, f); q4r_put(b + 64, k, f); }
};
struct Iq5k {
    using F = Fields; static constexpr int R = 4, EL = 256, BS = 176, ROW_META = 0;          // {d, extra, scales_h[4], scales_l[8], qs[128], qh[32]} | {d[4], extra[8], scales_h[16], scales_l[32], qs[512], qh[128]}
    static void get_base(const uint8_t *b, Fields &f) { hdr_get(b, f); sh2_get(b + 4, f); sl_get(b + 8, f); q5_get(b + 16, b + 144, f); }
    static void put_base(uint8_t *b, const Fields &f) { hdr_put(b, f); sh2_put(b + 4, f); sl_put(b + 8, f); q5_put(b + 16, b + 144, f); }
    static void get_r(const uint8_t *b, int k, Fields &f) { memcpy(&f.d, b + 2 * k, 2); ex4_get(b + 8, k, f); sh4_get<2>(b + 16, k, f); sl4_get(b + 32, k, f); q4r_get(b + 64, k, f); q5hr_get(b + 576, k, f); }
    static void put_r(uint8_t *b, int k, const Fields &f) { memcpy(b + 2 * k, &f.d, 2); ex4_put(b + 8, k, f); sh4_put<2>(b + 16, k, f); sl4_put(b + 32, k, f); q4r_put(b + 64, k, f); q5hr_put(b + 576, k, f); }
};
// the *_KS types: an f32 scale in front of every row (four of them in front of a row group), one scale byte per 32-weight sub-block; interleaved: scales[4 ib + k]
struct Iq4ks {
    using F = Fields; static constexpr int R = 4, EL = 256, BS = 136, ROW_META = 4;          // {scales[8], qs[128]}                    | {scales[32], qs[512]}
    static void get_base(const uint8_t *b, Fields &f) { memcpy(f.sl, b, 8); q4_get(b + 8, f); }
    static void put_base(uint8_t *b, const Fields &f) { memcpy(b, f.sl, 8); q4_put(b + 8, f); }
    static void get_r(const uint8_t *b, int k, Fields &f) { for (int ib = 0; ib < 8; ++ib) f.sl[ib] = b[4 * ib + k]; q4r_get(b + 32, k, f); }
    static void put_r(uint8_t *b, int k, const Fields &f) { for (int ib = 0; ib < 8; ++ib) b[4 * ib + k] = f.sl[ib]; q4r_put(b + 32, k, f); }
};
struct Iq5ks {
    using F = Fields; static constexpr int R = 4, EL = 256, BS = 168, ROW_META = 4;          // {scales[8], qs[128], qh[32]}            | {scales[32], qs[512], qh[128]}
    static void get_base(const uint8_t *b, Fields &f) { memcpy(f.sl, b, 8); q5_get(b + 8, b + 136, f); }
    static void put_base(uint8_t *b, const Fields &f) { memcpy(b, f.sl, 8); q5_put(b + 8, b + 136, f); }
    static void get_r(const uint8_t *b, int k, Fields &f) { for (int ib = 0; ib < 8; ++ib) f.sl[ib] = b[4 * ib + k]; q4r_get(b + 32, k, f); q5hr_get(b + 544, k, f); }
    static void put_r(uint8_t *b, int k, const Fields &f) { for (int ib = 0; ib < 8; ++ib) b[4 * ib + k] = f.sl[ib]; q4r_put(b + 32, k, f); q5hr_put(b + 544, k, f); }
};

// ---- Q2_K / Q3_K: the 2-bit planes, the third-bit plane (hmask) and the interleaved quants are those of IQ2_K / IQ3_K (iqk_quantize.cpp:6428-6460, 6545-6573) ----
struct Q2k {
    using F = Fields; static constexpr int R = 4, EL = 256, BS = 84, ROW_META = 0;      // {scales[16], qs[64], d, dmin}      | {d[4], dmin[4], scales[64], qs[256]}
    static void get_base(const uint8_t *b, Fields &f) { memcpy(f.sl, b, 16); q2_get(b + 16, f); memcpy(&f.d, b + 80, 2); memcpy(&f.dmin, b + 82, 2); }
    static void put_base(uint8_t *b, const Fields &f) { memcpy(b, f.sl, 16); q2_put(b + 16, f); memcpy(b + 80, &f.d, 2); memcpy(b + 82, &f.dmin, 2); }
    static void get_r(const uint8_t *b, int k, Fields &f) { memcpy(&f.d, b + 2 * k, 2); memcpy(&f.dmin, b + 8 + 2 * k, 2); for (int j = 0; j < 16; ++j) f.sl[j] = b[16 + 4 * j + k]; q2r_get(b + 80, k, f); }
    static void put_r(uint8_t *b, int k, const Fields &f) { memcpy(b + 2 * k, &f.d, 2); memcpy(b + 8 + 2 * k, &f.dmin, 2); for (int j = 0; j < 16; ++j) b[16 + 4 * j + k] = f.sl[j]; q2r_put(b + 80, k, f); }
};
struct Q3k {
    using F = Fields; static constexpr int R = 4, EL = 256, BS = 110, ROW_META = 0;     // {hmask[32], qs[64], scales[12], d} | {d[4], scales_h[16], scales_l[32], qh[128], qs[256]}
    // the sixteen 6-bit scales: low nibble of scale j in byte j % 8 (nibble j / 8), its two high bits in byte 8 + j % 4 at bit 2 (j / 4)   (the kmask shuffle of ggml-quants.c)
    static void get_base(const uint8_t *b, Fields &f) {
        q2_get(b + 32, f); q3h_get(b, f); memcpy(&f.d, b + 108, 2);
        for (int j = 0; j < 16; ++j) { f.sl[j] = (b[96 + (j & 7)] >> (4 * (j >> 3))) & 0xf; f.sh[j] = (b[104 + (j & 3)] >> (2 * (j >> 2))) & 3; }
    }
    static void put_base(uint8_t *b, const Fields &f) {
        q3h_put(b, f); q2_put(b + 32, f); memcpy(b + 108, &f.d, 2);
        for (int j = 0; j < 8; ++j) b[96 + j] = (uint8_t)((f.sl[j] & 0xf) | ((f.sl[j + 8] & 0xf) << 4));
        for (int j = 0; j < 4; ++j) b[104 + j] = (uint8_t)((f.sh[j] & 3) | ((f.sh[j + 4] & 3) << 2) | ((f.sh[j + 8] & 3) << 4) | ((f.sh[j + 12] & 3) << 6));
    }
    static void get_r(const uint8_t *b, int k, Fields &f) { memcpy(&f.d, b + 2 * k, 2); sh4_get<2>(b + 8, k, f); sl4_get(b + 24, k, f); q2r_get(b + 184, k, f); q3hr_get(b + 56, k, f); }
    static void put_r(uint8_t *b, int k, const Fields &f) { memcpy(b + 2 * k, &f.d, 2); sh4_put<2>(b + 8, k, f); sl4_put(b + 24, k, f); q3hr_put(b + 56, k, f); q2r_put(b + 184, k, f); }
};

// ---- IQ4_XS_R8: eight rows; one 6-bit scale per 32-block, scale number i = 8 ib + k (iqk_quantize.cpp:5742-5772) ----
struct Iq4xs {
    using F = Fields; static constexpr int R = 8, EL = 256, BS = 136, ROW_META = 0;     // {d, scales_h (u16), scales_l[4], qs[128]} | {d[8], scales_h[16], scales_l[32], qs[1024]}
    static void get_base(const uint8_t *b, Fields &f) { memcpy(&f.d, b, 2); uint16_t sh; memcpy(&sh, b + 2, 2); for (int ib = 0; ib < 8; ++ib) { f.sh[ib] = (sh >> (2 * ib)) & 3; f.sl[ib] = (b[4 + (ib >> 1)] >> (4 * (ib & 1))) & 0xf; } q4_get(b + 8, f); }
    static void put_base(uint8_t *b, const Fields &f) { memcpy(b, &f.d, 2); uint16_t sh = 0; for (int ib = 0; ib < 8; ++ib) sh |= (uint16_t)(f.sh[ib] & 3) << (2 * ib); memcpy(b + 2, &sh, 2);
                                                        for (int j = 0; j < 4; ++j) b[4 + j] = (uint8_t)((f.sl[2 * j] & 0xf) | ((f.sl[2 * j + 1] & 0xf) << 4)); q4_put(b + 8, f); }
    // scales_l[32]: byte 8 (ib % 4) + k, nibble ib / 4;  scales_h[16]: byte 8 (ib % 2) + k, bit 2 (ib / 2);  quants: weight e = 4 q + i of 32-block ib in byte 128 ib + 32 (q / 2) + 4 k + i, nibble q % 2
    static void get_r(const uint8_t *b, int k, Fields &f) {
        memcpy(&f.d, b + 2 * k, 2);
        for (int ib = 0; ib < 8; ++ib) { f.sh[ib] = (b[16 + 8 * (ib & 1) + k] >> (2 * (ib >> 1))) & 3; f.sl[ib] = (b[32 + 8 * (ib & 3) + k] >> (4 * (ib >> 2))) & 0xf; }
        for (int ib = 0; ib < 8; ++ib) for (int e = 0; e < 32; ++e) { const int q = e >> 2; f.L[32 * ib + e] = (b[64 + 128 * ib + 32 * (q >> 1) + 4 * k + (e & 3)] >> (4 * (q & 1))) & 0xf; }
    }
    static void put_r(uint8_t *b, int k, const Fields &f) {
        memcpy(b + 2 * k, &f.d, 2);
        for (int h = 0; h < 2; ++h) b[16 + 8 * h + k] = (uint8_t)((f.sh[h] & 3) | ((f.sh[h + 2] & 3) << 2) | ((f.sh[h + 4] & 3) << 4) | ((f.sh[h + 6] & 3) << 6));
        for (int j = 0; j < 4; ++j) b[32 + 8 * j + k] = (uint8_t)((f.sl[j] & 0xf) | ((f.sl[j + 4] & 0xf) << 4));
        for (int ib = 0; ib < 8; ++ib) for (int c = 0; c < 4; ++c) for (int i = 0; i < 4; ++i) b[64 + 128 * ib + 32 * c + 4 * k + i] = (uint8_t)((f.L[32 * ib + 8 * c + i] & 0xf) | ((f.L[32 * ib + 8 * c + 4 + i] & 0xf) << 4));
    }
};

// ---- the 32-weight block types (iqk_quantize.cpp:5304-5322 q4_0, :5350-5368 mxfp4, :5438-5452 q8_0, :5551-5576 q5_0, :5649-5674 q6_0) ----
struct Blk32 { uint8_t hdr[2]; uint8_t q[32]; };     // scale bytes; payload bytes (Q4_0 / MXFP4 / Q8_0) or quant indices (Q5_0: 5 bits, Q6_0: 6 bits)
template <int HB> struct Nib8 {                        // Q4_0_R8 (HB = 2: f16 d) / MXFP4_R8 (HB = 1: E8M0 byte): the 16 nibble bytes stay whole; {hdr, qs[16]} | {hdr[8], qs[128]}: byte 4 l + i -> 32 l + 4 k + i
    using F = Blk32; static constexpr int R = 8, EL = 32, BS = HB + 16, ROW_META = 0;
    static void get_base(const uint8_t *b, Blk32 &f) { memcpy(f.hdr, b, HB); memcpy(f.q, b + HB, 16); }
    static void put_base(uint8_t *b, const Blk32 &f) { memcpy(b, f.hdr, HB); memcpy(b + HB, f.q, 16); }
    static void get_r(const uint8_t *b, int k, Blk32 &f) { memcpy(f.hdr, b + HB * k, HB); for (int l = 0; l < 4; ++l) memcpy(f.q + 4 * l, b + 8 * HB + 32 * l + 4 * k, 4); }
    static void put_r(uint8_t *b, int k, const Blk32 &f) { memcpy(b + HB * k, f.hdr, HB); for (int l = 0; l < 4; ++l) memcpy(b + 8 * HB + 32 * l + 4 * k, f.q + 4 * l, 4); }
};
struct Q80 {                                           // {d, qs[32]} | {d[8], qs[256]}: byte 4 l + i -> 32 l + 4 k + i, byte 16 + 4 l + i -> 128 + 32 l + 4 k + i
    using F = Blk32; static constexpr int R = 8, EL = 32, BS = 34, ROW_META = 0;
    static void get_base(const uint8_t *b, Blk32 &f) { memcpy(f.hdr, b, 2); memcpy(f.q, b + 2, 32); }
    static void put_base(uint8_t *b, const Blk32 &f) { memcpy(b, f.hdr, 2); memcpy(b + 2, f.q, 32); }
    static void get_r(const uint8_t *b, int k, Blk32 &f) { memcpy(f.hdr, b + 2 * k, 2); for (int l = 0; l < 8; ++l) memcpy(f.q + 4 * l, b + 16 + 128 * (l >> 2) + 32 * (l & 3) + 4 * k, 4); }
    static void put_r(uint8_t *b, int k, const Blk32 &f) { memcpy(b + 2 * k, f.hdr, 2); for (int l = 0; l < 8; ++l) memcpy(b + 16 + 128 * (l >> 2) + 32 * (l & 3) + 4 * k, f.q + 4 * l, 4); }
};
// Q5_0_R4 / Q6_0_R4 quants: byte 16 l + 4 k + i holds weights i + l1 (low nibble) and i + l1 + 8 (high nibble), l1 = 4 (l / 2) + 16 (l % 2)
inline int  q56_first(int l) { return 4 * (l >> 1) + 16 * (l & 1); }
inline void q56r_get(const uint8_t *qs, int k, Blk32 &f) { for (int l = 0; l < 4; ++l) for (int i = 0; i < 4; ++i) { const uint8_t v = qs[16 * l + 4 * k + i]; f.q[i + q56_first(l)] = v & 0xf; f.q[i + q56_first(l) + 8] = v >> 4; } }
inline void q56r_put(uint8_t *qs, int k, const Blk32 &f) { for (int l = 0; l < 4; ++l) for (int i = 0; i < 4; ++i) qs[16 * l + 4 * k + i] = (uint8_t)((f.q[i + q56_first(l)] & 0xf) | ((f.q[i + q56_first(l) + 8] & 0xf) << 4)); }
struct Q50 {                                           // {d, qh (u32), qs[16]} | {d[4], qh[16], qs[64]}
    using F = Blk32; static constexpr int R = 4, EL = 32, BS = 22, ROW_META = 0;
    static void get_base(const uint8_t *b, Blk32 &f) { memcpy(f.hdr, b, 2); uint32_t qh; memcpy(&qh, b + 2, 4); for (int j = 0; j < 16; ++j) { f.q[j] = (b[6 + j] & 0xf) | (((qh >> j) & 1) << 4); f.q[j + 16] = (b[6 + j] >> 4) | (((qh >> (j + 16)) & 1) << 4); } }
    static void put_base(uint8_t *b, const Blk32 &f) { memcpy(b, f.hdr, 2); uint32_t qh = 0; for (int j = 0; j < 32; ++j) qh |= (uint32_t)((f.q[j] >> 4) & 1) << j; memcpy(b + 2, &qh, 4); for (int j = 0; j < 16; ++j) b[6 + j] = (uint8_t)((f.q[j] & 0xf) | ((f.q[j + 16] & 0xf) << 4)); }
    // qh[4 k + i]: bit l = fifth bit of weight i + l1(l), bit 4 + l = fifth bit of weight i + l1(l) + 8
    static void get_r(const uint8_t *b, int k, Blk32 &f) { memcpy(f.hdr, b + 2 * k, 2); q56r_get(b + 24, k, f); for (int l = 0; l < 4; ++l) for (int i = 0; i < 4; ++i) { const uint8_t h = b[8 + 4 * k + i]; f.q[i + q56_first(l)] |= ((h >> l) & 1) << 4; f.q[i + q56_first(l) + 8] |= ((h >> (4 + l)) & 1) << 4; } }
    static void put_r(uint8_t *b, int k, const Blk32 &f) { memcpy(b + 2 * k, f.hdr, 2); q56r_put(b + 24, k, f); for (int i = 0; i < 4; ++i) { unsigned h = 0; for (int l = 0; l < 4; ++l) h |= (unsigned)((f.q[i + q56_first(l)] >> 4) & 1) << l | (unsigned)((f.q[i + q56_first(l) + 8] >> 4) & 1) << (4 + l); b[8 + 4 * k + i] = (uint8_t)h; } }
};
struct Q60 {                                           // {d, qh[8], qs[16]} | {d[4], qh[32], qs[64]}
    using F = Blk32; static constexpr int R = 4, EL = 32, BS = 26, ROW_META = 0;
    // qh[j % 8], nibble j / 8 (j < 16): its low two bits are bits 4-5 of weight j, its high two bits those of weight j + 16
    static void get_base(const uint8_t *b, Blk32 &f) { memcpy(f.hdr, b, 2); for (int j = 0; j < 16; ++j) { const uint8_t h = (b[2 + (j & 7)] >> (4 * (j >> 3))) & 0xf; f.q[j] = (b[10 + j] & 0xf) | ((h & 3) << 4); f.q[j + 16] = (b[10 + j] >> 4) | ((h >> 2) << 4); } }
    static void put_base(uint8_t *b, const Blk32 &f) { memcpy(b, f.hdr, 2); for (int j = 0; j < 8; ++j) { const unsigned lo = ((f.q[j] >> 4) & 3) | (((f.q[j + 16] >> 4) & 3) << 2), hi = ((f.q[j + 8] >> 4) & 3) | (((f.q[j + 24] >> 4) & 3) << 2); b[2 + j] = (uint8_t)(lo | (hi << 4)); }
                                                       for (int j = 0; j < 16; ++j) b[10 + j] = (uint8_t)((f.q[j] & 0xf) | ((f.q[j + 16] & 0xf) << 4)); }
    // qh[16 (l % 2) + 4 k + i]: bits 2 (l / 2) .. +1 = bits 4-5 of weight i + l1(l), bits 4 + 2 (l / 2) .. +1 = those of weight i + l1(l) + 8
    static void get_r(const uint8_t *b, int k, Blk32 &f) { memcpy(f.hdr, b + 2 * k, 2); q56r_get(b + 40, k, f); for (int l = 0; l < 4; ++l) for (int i = 0; i < 4; ++i) { const uint8_t h = b[8 + 16 * (l & 1) + 4 * k + i]; f.q[i + q56_first(l)] |= ((h >> (2 * (l >> 1))) & 3) << 4; f.q[i + q56_first(l) + 8] |= ((h >> (4 + 2 * (l >> 1))) & 3) << 4; } }
    static void put_r(uint8_t *b, int k, const Blk32 &f) { memcpy(b + 2 * k, f.hdr, 2); q56r_put(b + 40, k, f);
        for (int p = 0; p < 2; ++p) for (int i = 0; i < 4; ++i) { unsigned h = 0; for (int t = 0; t < 2; ++t) { const int l = 2 * t + p; h |= (unsigned)((f.q[i + q56_first(l)] >> 4) & 3) << (2 * t) | (unsigned)((f.q[i + q56_first(l) + 8] >> 4) & 3) << (4 + 2 * t); } b[8 + 16 * p + 4 * k + i] = (uint8_t)h; } }
};

// ---- IQ2_XXS / IQ3_XXS / IQ2_XS: the interleaved forms re-code the 7-bit sign index (iqk_quantize.cpp:7631-7643 `scrambled_sign`, a 128-entry table there).  The code
// is computable: with y = s >> 1, P(y) = y ^ (y >> 1) ^ (y >> 2) ^ ... (the running parity of the sign bits from the top), the table holds P(y) for an even number of set
// bits in s and its 7-bit complement for an odd number.  Built once, together with its inverse.
struct SignCode {
    uint8_t fwd[128], inv[128];
    SignCode() {
        for (int s = 0; s < 128; ++s) { int y = s >> 1, p = 0; for (int t = 0; t < 6; ++t) p ^= y >> t; const int odd = __builtin_popcount(s) & 1; fwd[s] = (uint8_t)((odd ? ~p : p) & 0x7f); }
        for (int s = 0; s < 128; ++s) inv[fwd[s]] = (uint8_t)s;
    }
};
const SignCode &sign_code() { static const SignCode c; return c; }
struct Xxs { uint16_t d; uint8_t g[8][8]; uint8_t sg[8][4]; uint8_t sc[8]; };      // per 32-block: grid index bytes (4: IQ2_XXS, 8: IQ3_XXS), four 7-bit sign indices, the 4-bit scale
// the u32 of a 32-block: sign indices at bits 0, 7, 14, 21, scale at bit 28;  interleaved: byte j = (code(sign j) << 1) | bit j of the scale
inline void xxs_word_get(const uint8_t *w, int ib, Xxs &f) { uint32_t v; memcpy(&v, w, 4); for (int j = 0; j < 4; ++j) f.sg[ib][j] = (v >> (7 * j)) & 127; f.sc[ib] = v >> 28; }
inline void xxs_word_put(uint8_t *w, int ib, const Xxs &f) { uint32_t v = (uint32_t)(f.sc[ib] & 0xf) << 28; for (int j = 0; j < 4; ++j) v |= (uint32_t)(f.sg[ib][j] & 127) << (7 * j); memcpy(w, &v, 4); }
inline void xxs_sas_get(const uint8_t *w, int ib, Xxs &f) { const SignCode &c = sign_code(); unsigned sc = 0; for (int j = 0; j < 4; ++j) { f.sg[ib][j] = c.inv[w[j] >> 1]; sc |= (unsigned)(w[j] & 1) << j; } f.sc[ib] = (uint8_t)sc; }
inline void xxs_sas_put(uint8_t *w, int ib, const Xxs &f) { const SignCode &c = sign_code(); for (int j = 0; j < 4; ++j) w[j] = (uint8_t)((c.fwd[f.sg[ib][j] & 127] << 1) | ((f.sc[ib] >> j) & 1)); }
struct Iq2xxs {                                        // {d, 8 x {grid[4], u32}} | {d[4], sas[128], qs[128]}: sas word 4 ib + k, grid bytes at 16 ib + 4 k
    using F = Xxs; static constexpr int R = 4, EL = 256, BS = 66, ROW_META = 0;
    static void get_base(const uint8_t *b, Xxs &f) { memcpy(&f.d, b, 2); for (int ib = 0; ib < 8; ++ib) { memcpy(f.g[ib], b + 2 + 8 * ib, 4); xxs_word_get(b + 2 + 8 * ib + 4, ib, f); } }
    static void put_base(uint8_t *b, const Xxs &f) { memcpy(b, &f.d, 2); for (int ib = 0; ib < 8; ++ib) { memcpy(b + 2 + 8 * ib, f.g[ib], 4); xxs_word_put(b + 2 + 8 * ib + 4, ib, f); } }
    static void get_r(const uint8_t *b, int k, Xxs &f) { memcpy(&f.d, b + 2 * k, 2); for (int ib = 0; ib < 8; ++ib) { xxs_sas_get(b + 8 + 4 * (4 * ib + k), ib, f); memcpy(f.g[ib], b + 136 + 16 * ib + 4 * k, 4); } }
    static void put_r(uint8_t *b, int k, const Xxs &f) { memcpy(b + 2 * k, &f.d, 2); for (int ib = 0; ib < 8; ++ib) { xxs_sas_put(b + 8 + 4 * (4 * ib + k), ib, f); memcpy(b + 136 + 16 * ib + 4 * k, f.g[ib], 4); } }
};
struct Iq3xxs {                                        // {d, grid[64], 8 x u32} | {d[4], sas[128], qs[256]}: grid bytes at 32 ib + 8 k
    using F = Xxs; static constexpr int R = 4, EL = 256, BS = 98, ROW_META = 0;
    static void get_base(const uint8_t *b, Xxs &f) { memcpy(&f.d, b, 2); for (int ib = 0; ib < 8; ++ib) { memcpy(f.g[ib], b + 2 + 8 * ib, 8); xxs_word_get(b + 66 + 4 * ib, ib, f); } }
    static void put_base(uint8_t *b, const Xxs &f) { memcpy(b, &f.d, 2); for (int ib = 0; ib < 8; ++ib) { memcpy(b + 2 + 8 * ib, f.g[ib], 8); xxs_word_put(b + 66 + 4 * ib, ib, f); } }
    static void get_r(const uint8_t *b, int k, Xxs &f) { memcpy(&f.d, b + 2 * k, 2); for (int ib = 0; ib < 8; ++ib) { xxs_sas_get(b + 8 + 4 * (4 * ib + k), ib, f); memcpy(f.g[ib], b + 136 + 32 * ib + 8 * k, 8); } }
    static void put_r(uint8_t *b, int k, const Xxs &f) { memcpy(b + 2 * k, &f.d, 2); for (int ib = 0; ib < 8; ++ib) { xxs_sas_put(b + 8 + 4 * (4 * ib + k), ib, f); memcpy(b + 136 + 32 * ib + 8 * k, f.g[ib], 8); } }
};
struct Xs { uint16_t d; uint16_t v[32]; uint8_t sc[8]; };      // {9-bit grid index | 7-bit sign index << 9} per 8 weights; a scale byte per 32-block
struct Iq2xs {                                         // {d, u16 qs[32], scales[8]} | {d[4], u16 qs[128], scales[32]}: word 16 ib + 4 k + i with the re-coded sign index, scales[4 ib + k]
    using F = Xs; static constexpr int R = 4, EL = 256, BS = 74, ROW_META = 0;
    static void get_base(const uint8_t *b, Xs &f) { memcpy(&f.d, b, 2); memcpy(f.v, b + 2, 64); memcpy(f.sc, b + 66, 8); }
    static void put_base(uint8_t *b, const Xs &f) { memcpy(b, &f.d, 2); memcpy(b + 2, f.v, 64); memcpy(b + 66, f.sc, 8); }
    static void get_r(const uint8_t *b, int k, Xs &f) { const SignCode &c = sign_code(); memcpy(&f.d, b + 2 * k, 2);
        for (int ib = 0; ib < 8; ++ib) { for (int i = 0; i < 4; ++i) { uint16_t v; memcpy(&v, b + 8 + 2 * (16 * ib + 4 * k + i), 2); f.v[4 * ib + i] = (uint16_t)((v & 511) | (c.inv[v >> 9] << 9)); } f.sc[ib] = b[264 + 4 * ib + k]; } }
    static void put_r(uint8_t *b, int k, const Xs &f) { const SignCode &c = sign_code(); memcpy(b + 2 * k, &f.d, 2);
        for (int ib = 0; ib < 8; ++ib) { for (int i = 0; i < 4; ++i) { const uint16_t v = (uint16_t)((f.v[4 * ib + i] & 511) | (c.fwd[f.v[4 * ib + i] >> 9] << 9)); memcpy(b + 8 + 2 * (16 * ib + 4 * k + i), &v, 2); } b[264 + 4 * ib + k] = f.sc[ib]; } }
};

// ---- IQ2_BN_R4: f32 row scale, 64 ternary weights in 16 bytes (iqk_quantize.cpp:5931-5968): byte 16 l + 4 k + i of the interleaved 64-byte block collects the l-th
// 2-bit fields of base bytes i, i + 4, i + 8, i + 12 ----
struct Bn { uint8_t q[16]; };
struct Iq2bn {
    using F = Bn; static constexpr int R = 4, EL = 64, BS = 16, ROW_META = 4;
    static void get_base(const uint8_t *b, Bn &f) { memcpy(f.q, b, 16); }
    static void put_base(uint8_t *b, const Bn &f) { memcpy(b, f.q, 16); }
    static void get_r(const uint8_t *b, int k, Bn &f) { for (int c = 0; c < 4; ++c) for (int i = 0; i < 4; ++i) { unsigned v = 0; for (int l = 0; l < 4; ++l) v |= (unsigned)((b[16 * l + 4 * k + i] >> (2 * c)) & 3) << (2 * l); f.q[4 * c + i] = (uint8_t)v; } }
    static void put_r(uint8_t *b, int k, const Bn &f) { for (int l = 0; l < 4; ++l) for (int i = 0; i < 4; ++i) { unsigned v = 0; for (int c = 0; c < 4; ++c) v |= (unsigned)((f.q[4 * c + i] >> (2 * l)) & 3) << (2 * c); b[16 * l + 4 * k + i] = (uint8_t)v; } }
};


// one group of R rows: base rows src + k * row_size <-> the interleaved group (R * row_size bytes: the R row scales, then nblock interleaved blocks)
template <class T> void group_to_r(const uint8_t *src, uint8_t *dst, int64_t nblock, size_t row_size) {
    for (int k = 0; k < T::R; ++k) memcpy(dst + T::ROW_META * k, src + k * row_size, T::ROW_META);
    uint8_t *y = dst + T::R * T::ROW_META;
    typename T::F f;
    for (int64_t ibl = 0; ibl < nblock; ++ibl) for (int k = 0; k < T::R; ++k) {
        memset(&f, 0, sizeof(f));
        T::get_base(src + k * row_size + T::ROW_META + ibl * T::BS, f);
        T::put_r(y + ibl * T::R * T::BS, k, f);
    }
}
template <class T> void group_to_base(const uint8_t *src, uint8_t *dst, int64_t nblock, size_t row_size) {
    for (int k = 0; k < T::R; ++k) memcpy(dst + k * row_size, src + T::ROW_META * k, T::ROW_META);
    const uint8_t *y = src + T::R * T::ROW_META;
    typename T::F f;
    for (int64_t ibl = 0; ibl < nblock; ++ibl) for (int k = 0; k < T::R; ++k) {
        memset(&f, 0, sizeof(f));
        T::get_r(y + ibl * T::R * T::BS, k, f);
        T::put_base(dst + k * row_size + T::ROW_META + ibl * T::BS, f);
    }
}
template <class T> void retile_groups(const uint8_t *src, uint8_t *dst, int64_t g0, int64_t g1, int64_t nblock, bool to_base) {
    const size_t row_size = T::ROW_META + (size_t)nblock * T::BS;
    for (int64_t g = g0; g < g1; ++g) {
        if (to_base) group_to_base<T>(src + g * T::R * row_size, dst + g * T::R * row_size, nblock, row_size);
        else         group_to_r<T>(src + g * T::R * row_size, dst + g * T::R * row_size, nblock, row_size);
    }
}

using groups_fn = void (*)(const uint8_t *, uint8_t *, int64_t, int64_t, int64_t, bool);
struct TypeInfo { int r_type, base_type, rows, elems; groups_fn fn; };
template <class T> constexpr TypeInfo info(int r_type, int base_type) { return {r_type, base_type, T::R, T::EL, retile_groups<T>}; }
const TypeInfo *type_info(int r_type) {
    static const TypeInfo table[] = {       // enum ggml_type ids (ggml.h:391-490)
        info<Iq2k>(CDNA4_TYPE_IQ2_K_R4, 137), info<Iq3k>(CDNA4_TYPE_IQ3_K_R4, 138), info<Iq4k>(CDNA4_TYPE_IQ4_K_R4, 139), info<Iq5k>(CDNA4_TYPE_IQ5_K_R4, 140),
        info<Iq4ks>(CDNA4_TYPE_IQ4_KS_R4, 144), info<Iq5ks>(CDNA4_TYPE_IQ5_KS_R4, 152),
        info<Nib8<2>>(CDNA4_TYPE_Q4_0_R8, 2), info<Q50>(CDNA4_TYPE_Q5_0_R4, 6), info<Q60>(CDNA4_TYPE_Q6_0_R4, 133), info<Q80>(CDNA4_TYPE_Q8_0_R8, 8), info<Nib8<1>>(CDNA4_TYPE_MXFP4_R8, 39),
        info<Q2k>(CDNA4_TYPE_Q2_K_R4, 10), info<Q3k>(CDNA4_TYPE_Q3_K_R4, 11), info<Iq4xs>(CDNA4_TYPE_IQ4_XS_R8, 23),
        info<Iq2xxs>(CDNA4_TYPE_IQ2_XXS_R4, 16), info<Iq2xs>(CDNA4_TYPE_IQ2_XS_R4, 17), info<Iq3xxs>(CDNA4_TYPE_IQ3_XXS_R4, 18), info<Iq2bn>(CDNA4_TYPE_IQ2_BN_R4, 135),
    };
    for (const TypeInfo &t : table) if (t.r_type == r_type) return &t;
    return nullptr;
}

}  // namespace

int cdna4_retile_r4_host_base_type(int r_type) { const TypeInfo *t = type_info(r_type); return t ? t->base_type : -1; }
int cdna4_retile_r4_host_rows(int r_type) { const TypeInfo *t = type_info(r_type); return t ? t->rows : 0; }

int cdna4_retile_r4_host(int r_type, const void *src, void *dst, int64_t nrows, int64_t ne00, int to_base, int n_threads) {
    const TypeInfo *ti = type_info(r_type);
    if (!ti) return set_err(CDNA4_E_UNSUPPORTED, "host re-tiling: type %d is not a host re-tiled row-interleaved type", r_type);
    if (!src || !dst || src == dst) return set_err(CDNA4_E_INVALID, "host re-tiling runs out of place");
    if (nrows < 0 || nrows % ti->rows || ne00 <= 0 || ne00 % ti->elems) return set_err(CDNA4_E_INVALID, "host re-tiling of type %d: nrows %% %d == 0 and ne00 %% %d == 0 required (nrows %lld, ne00 %lld)", r_type, ti->rows, ti->elems, (long long)nrows, (long long)ne00);
    const groups_fn fn = ti->fn;
    const int64_t groups = nrows / ti->rows, nblock = ne00 / ti->elems;
    int nt = n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency();
    if (nt < 1) nt = 1;
    if (nt > 64) nt = 64;
    if ((int64_t)nt > groups) nt = (int)(groups > 0 ? groups : 1);
    if (groups * ti->rows * ne00 < (int64_t)4 << 20) nt = 1;      // (small tensors: a thread costs more than the work)
    const uint8_t *s = (const uint8_t *)src; uint8_t *d = (uint8_t *)dst;
    if (nt == 1) { fn(s, d, 0, groups, nblock, to_base != 0); return CDNA4_OK; }
    std::vector<std::thread> pool;
    for (int t = 0; t < nt; ++t) {
        const int64_t g0 = groups * t / nt, g1 = groups * (t + 1) / nt;
        pool.emplace_back([=] { fn(s, d, g0, g1, nblock, to_base != 0); });
    }
    for (auto &th : pool) th.join();
    return CDNA4_OK;
}
