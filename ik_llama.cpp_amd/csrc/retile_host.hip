// Host-side re-tiling of the row-interleaved (`_R4`) forms of ik's non-linear types: IQ2_K_R4, IQ3_K_R4, IQ4_K_R4, IQ5_K_R4, IQ4_KS_R4,
// IQ5_KS_R4 (the `_R4` weight types the reference CUDA backend lists for MUL_MAT, ggml-cuda.cu:4893-4898) <-> their base types.
//
// The interleave exists so that one AVX load of activations feeds four rows; a 64-lane wavefront amortises the activations anyway, so on
// MI355X these tensors are stored in the BASE tiling and served by the base types' kernels.  The conversion runs once per
// tensor at upload (and its inverse at download), on the host, between the file bytes and the H2D copy.  Layouts restated from the
// reference's repack functions -- iqk_quantize.cpp:7533-7572 (iq2_k), :7398-7446 (iq3_k), :6639-6683 (iq4_k), :6775-6822 (iq5_k),
// :5829-5862 (iq4_ks), :6892-6932 (iq5_ks) -- and the block structs ggml-common.h:610-778.  Both directions go through ONE description
// per format (a block is decoded into its logical fields, the fields are encoded into the other format), so the two directions cannot
// disagree; tests/test_retile_host.py pins the bytes against the reference's own iqk_repack_tensor and checks the round trip.
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
    uint16_t d;            // f16 bits of the block scale (types with a per-row scale: unused)
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
// get_base / put_base: one base block <-> Fields;  get_r4 / put_r4: row k of one interleaved block <-> Fields.
// Every put assembles a destination byte completely and stores it ONCE: the read-modify-write form (`byte |= field << shift` over a zeroed block) 
// also wrote bytes that belong to other rows when vectorized
struct Iq2k {
    static constexpr int BS = 76, ROW_META = 0;           // {d, extra, scales[8], qs[64]}            | {d[4], extra[8], scales[32], qs[256]}
    static void get_base(const uint8_t *b, Fields &f) { hdr_get(b, f); sl_get(b + 4, f); q2_get(b + 12, f); }
    static void put_base(uint8_t *b, const Fields &f) { hdr_put(b, f); sl_put(b + 4, f); q2_put(b + 12, f); }
    static void get_r4(const uint8_t *b, int k, Fields &f) { memcpy(&f.d, b + 2 * k, 2); ex4_get(b + 8, k, f); sl4_get(b + 16, k, f); q2r_get(b + 48, k, f); }
    static void put_r4(uint8_t *b, int k, const Fields &f) { memcpy(b + 2 * k, &f.d, 2); ex4_put(b + 8, k, f); sl4_put(b + 16, k, f); q2r_put(b + 48, k, f); }
};
struct Iq3k {
    static constexpr int BS = 110, ROW_META = 0;          // {d, extra, scales_h (u16), scales_l[8], qs[64], qh[32]} | {d[4], extra[8], scales_h[8], scales_l[32], qs[256], qh[128]}
    static void get_base(const uint8_t *b, Fields &f) { hdr_get(b, f); uint16_t sh; memcpy(&sh, b + 4, 2); for (int j = 0; j < 16; ++j) f.sh[j] = (sh >> j) & 1; sl_get(b + 6, f); q2_get(b + 14, f); q3h_get(b + 78, f); }
    static void put_base(uint8_t *b, const Fields &f) { hdr_put(b, f); uint16_t sh = 0; for (int j = 0; j < 16; ++j) sh |= (uint16_t)(f.sh[j] & 1) << j; memcpy(b + 4, &sh, 2); sl_put(b + 6, f); q2_put(b + 14, f); q3h_put(b + 78, f); }
    static void get_r4(const uint8_t *b, int k, Fields &f) { memcpy(&f.d, b + 2 * k, 2); ex4_get(b + 8, k, f); sh4_get<1>(b + 16, k, f); sl4_get(b + 24, k, f); q2r_get(b + 56, k, f); q3hr_get(b + 312, k, f); }
    static void put_r4(uint8_t *b, int k, const Fields &f) { memcpy(b + 2 * k, &f.d, 2); ex4_put(b + 8, k, f); sh4_put<1>(b + 16, k, f); sl4_put(b + 24, k, f); q2r_put(b + 56, k, f); q3hr_put(b + 312, k, f); }
};
// scales_h[4] of IQ4_K / IQ5_K: 2 bits per sub-block, sub-block j in byte j / 4 at bit 2 (j % 4)
inline void sh2_get(const uint8_t *s, Fields &f) { for (int j = 0; j < 16; ++j) f.sh[j] = (s[j >> 2] >> (2 * (j & 3))) & 3; }
inline void sh2_put(uint8_t *s, const Fields &f) { for (int b = 0; b < 4; ++b) s[b] = (uint8_t)((f.sh[4 * b] & 3) | ((f.sh[4 * b + 1] & 3) << 2) | ((f.sh[4 * b + 2] & 3) << 4) | ((f.sh[4 * b + 3] & 3) << 6)); }
struct Iq4k {
    static constexpr int BS = 144, ROW_META = 0;          // {d, extra, scales_h[4], scales_l[8], qs[128]}  | {d[4], extra[8], scales_h[16], scales_l[32], qs[512]}
    static void get_base(const uint8_t *b, Fields &f) { hdr_get(b, f); sh2_get(b + 4, f); sl_get(b + 8, f); q4_get(b + 16, f); }
    static void put_base(uint8_t *b, const Fields &f) { hdr_put(b, f); sh2_put(b + 4, f); sl_put(b + 8, f); q4_put(b + 16, f); }
    static void get_r4(const uint8_t *b, int k, Fields &f) { memcpy(&f.d, b + 2 * k, 2); ex4_get(b + 8, k, f); sh4_get<2>(b + 16, k, f); sl4_get(b + 32, k, f); q4r_get(b + 64, k, f); }
    static void put_r4(uint8_t *b, int k, const Fields &f) { memcpy(b + 2 * k, &f.d, 2); ex4_put(b + 8, k, f); sh4_put<2>(b + 16, k, f); sl4_put(b + 32, k, f); q4r_put(b + 64, k, f); }
};
struct Iq5k {
    static constexpr int BS = 176, ROW_META = 0;          // {d, extra, scales_h[4], scales_l[8], qs[128], qh[32]} | {d[4], extra[8], scales_h[16], scales_l[32], qs[512], qh[128]}
    static void get_base(const uint8_t *b, Fields &f) { hdr_get(b, f); sh2_get(b + 4, f); sl_get(b + 8, f); q5_get(b + 16, b + 144, f); }
    static void put_base(uint8_t *b, const Fields &f) { hdr_put(b, f); sh2_put(b + 4, f); sl_put(b + 8, f); q5_put(b + 16, b + 144, f); }
    static void get_r4(const uint8_t *b, int k, Fields &f) { memcpy(&f.d, b + 2 * k, 2); ex4_get(b + 8, k, f); sh4_get<2>(b + 16, k, f); sl4_get(b + 32, k, f); q4r_get(b + 64, k, f); q5hr_get(b + 576, k, f); }
    static void put_r4(uint8_t *b, int k, const Fields &f) { memcpy(b + 2 * k, &f.d, 2); ex4_put(b + 8, k, f); sh4_put<2>(b + 16, k, f); sl4_put(b + 32, k, f); q4r_put(b + 64, k, f); q5hr_put(b + 576, k, f); }
};
// the *_KS types: an f32 scale in front of every row (four of them in front of a row group), one scale byte per 32-weight sub-block; interleaved: scales[4 ib + k]
struct Iq4ks {
    static constexpr int BS = 136, ROW_META = 4;          // {scales[8], qs[128]}                    | {scales[32], qs[512]}
    static void get_base(const uint8_t *b, Fields &f) { memcpy(f.sl, b, 8); q4_get(b + 8, f); }
    static void put_base(uint8_t *b, const Fields &f) { memcpy(b, f.sl, 8); q4_put(b + 8, f); }
    static void get_r4(const uint8_t *b, int k, Fields &f) { for (int ib = 0; ib < 8; ++ib) f.sl[ib] = b[4 * ib + k]; q4r_get(b + 32, k, f); }
    static void put_r4(uint8_t *b, int k, const Fields &f) { for (int ib = 0; ib < 8; ++ib) b[4 * ib + k] = f.sl[ib]; q4r_put(b + 32, k, f); }
};
struct Iq5ks {
    static constexpr int BS = 168, ROW_META = 4;          // {scales[8], qs[128], qh[32]}            | {scales[32], qs[512], qh[128]}
    static void get_base(const uint8_t *b, Fields &f) { memcpy(f.sl, b, 8); q5_get(b + 8, b + 136, f); }
    static void put_base(uint8_t *b, const Fields &f) { memcpy(b, f.sl, 8); q5_put(b + 8, b + 136, f); }
    static void get_r4(const uint8_t *b, int k, Fields &f) { for (int ib = 0; ib < 8; ++ib) f.sl[ib] = b[4 * ib + k]; q4r_get(b + 32, k, f); q5hr_get(b + 544, k, f); }
    static void put_r4(uint8_t *b, int k, const Fields &f) { for (int ib = 0; ib < 8; ++ib) b[4 * ib + k] = f.sl[ib]; q4r_put(b + 32, k, f); q5hr_put(b + 544, k, f); }
};

// one group of four rows: base rows src + k * row_size <-> the interleaved group (4 * row_size bytes: the four row scales, then nblock interleaved blocks)
template <class T> void group_to_r4(const uint8_t *src, uint8_t *dst, int64_t nblock, size_t row_size) {
    for (int k = 0; k < 4; ++k) memcpy(dst + T::ROW_META * k, src + k * row_size, T::ROW_META);
    uint8_t *y = dst + 4 * T::ROW_META;
    Fields f;
    for (int64_t ibl = 0; ibl < nblock; ++ibl) for (int k = 0; k < 4; ++k) {
        T::get_base(src + k * row_size + T::ROW_META + ibl * T::BS, f);
        T::put_r4(y + ibl * 4 * T::BS, k, f);
    }
}
template <class T> void group_to_base(const uint8_t *src, uint8_t *dst, int64_t nblock, size_t row_size) {
    for (int k = 0; k < 4; ++k) memcpy(dst + k * row_size, src + T::ROW_META * k, T::ROW_META);
    const uint8_t *y = src + 4 * T::ROW_META;
    Fields f;
    for (int64_t ibl = 0; ibl < nblock; ++ibl) for (int k = 0; k < 4; ++k) {
        memset(&f, 0, sizeof(f));
        T::get_r4(y + ibl * 4 * T::BS, k, f);
        T::put_base(dst + k * row_size + T::ROW_META + ibl * T::BS, f);
    }
}
template <class T> void retile_groups(const uint8_t *src, uint8_t *dst, int64_t g0, int64_t g1, int64_t nblock, bool to_base) {
    const size_t row_size = T::ROW_META + (size_t)nblock * T::BS;
    for (int64_t g = g0; g < g1; ++g) {
        if (to_base) group_to_base<T>(src + g * 4 * row_size, dst + g * 4 * row_size, nblock, row_size);
        else         group_to_r4<T>(src + g * 4 * row_size, dst + g * 4 * row_size, nblock, row_size);
    }
}

using groups_fn = void (*)(const uint8_t *, uint8_t *, int64_t, int64_t, int64_t, bool);
groups_fn groups_of(int r4_type) {
    switch (r4_type) {
        case CDNA4_TYPE_IQ2_K_R4:  return retile_groups<Iq2k>;
        case CDNA4_TYPE_IQ3_K_R4:  return retile_groups<Iq3k>;
        case CDNA4_TYPE_IQ4_K_R4:  return retile_groups<Iq4k>;
        case CDNA4_TYPE_IQ5_K_R4:  return retile_groups<Iq5k>;
        case CDNA4_TYPE_IQ4_KS_R4: return retile_groups<Iq4ks>;
        case CDNA4_TYPE_IQ5_KS_R4: return retile_groups<Iq5ks>;
        default: return nullptr;
    }
}

}  // namespace

int cdna4_retile_r4_host_base_type(int r4_type) { return groups_of(r4_type) ? r4_type - 200 : -1; }      // enum ggml_type: the _R4 ids are base + 200 (ggml.h:461-490)

int cdna4_retile_r4_host(int r4_type, const void *src, void *dst, int64_t nrows, int64_t ne00, int to_base, int n_threads) {
    groups_fn fn = groups_of(r4_type);
    if (!fn) return set_err(CDNA4_E_UNSUPPORTED, "host re-tiling: type %d is not one of IQ2_K_R4 IQ3_K_R4 IQ4_K_R4 IQ5_K_R4 IQ4_KS_R4 IQ5_KS_R4", r4_type);
    if (!src || !dst || src == dst) return set_err(CDNA4_E_INVALID, "host re-tiling runs out of place");
    if (nrows < 0 || nrows % 4 || ne00 <= 0 || ne00 % 256) return set_err(CDNA4_E_INVALID, "host re-tiling: nrows %% 4 == 0 and ne00 %% 256 == 0 required (nrows %lld, ne00 %lld)", (long long)nrows, (long long)ne00);
    const int64_t groups = nrows / 4, nblock = ne00 / 256;
    int nt = n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency();
    if (nt < 1) nt = 1;
    if (nt > 64) nt = 64;
    if ((int64_t)nt > groups) nt = (int)(groups > 0 ? groups : 1);
    if (groups * nblock < 4096) nt = 1;                      // (small tensors: a thread costs more than the work)
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
