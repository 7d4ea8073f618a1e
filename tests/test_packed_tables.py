"""CPU: the compact device encodings of the codebooks (csrc/iq_grids_packed.inc, expanded on the device by gemv.cuh expand_*) and the packed value tables of
cdna4_common.cuh decode to exactly the tables the oracle uses (oracle/iq_grids.h, iqk_oracle.c) -- which are pinned against the reference through the
dequantization tests.  Catches a stale / mistyped table without a GPU."""
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def c_array(text, name):
    m = re.search(r"\b%s\s*(\[[^\]]*\])+\s*=\s*\{" % re.escape(name), text)
    assert m, name
    i = m.end(); depth = 1
    while depth:                     # balanced braces (rows of a 2-D table are flattened)
        depth += {"{": 1, "}": -1}.get(text[i], 0); i += 1
    body = re.sub(r"/\*.*?\*/|//[^\n]*", "", text[m.end():i - 1], flags=re.S)
    return [int(x.rstrip("uUlL"), 0) for x in re.findall(r"-?0x[0-9a-fA-F]+[uUlL]*|-?\d+[uUlL]*", body)]


PACKED = open(os.path.join(ROOT, "ik_llama.cpp_amd/csrc/iq_grids_packed.inc")).read()
GRIDS = open(os.path.join(ROOT, "oracle/iq_grids.h")).read()
COMMON = open(os.path.join(ROOT, "ik_llama.cpp_amd/csrc/cdna4_common.cuh")).read()
ORACLE = open(os.path.join(ROOT, "oracle/iqk_oracle.c")).read()


def entry_bytes(v, n):
    return [(v >> (8 * j)) & 0xff for j in range(n)]


def test_iq2_family_grids():         # 2 bits per magnitude -> {8, 25, 43} (expand_iq2_grid: t * 17 + 8 + (t >> 1))
    for packed, full, n in (("k_iq2s_grid_packed", "oracle_iq2s_grid", 1024), ("k_iq2xxs_grid_packed", "oracle_iq2xxs_grid", 256), ("k_iq2xs_grid_packed", "oracle_iq2xs_grid", 512)):
        p, g = c_array(PACKED, packed), c_array(GRIDS, full)
        assert len(p) == len(g) == n
        for w, v in zip(p, g):
            assert [t * 17 + 8 + (t >> 1) for t in [(w >> (2 * j)) & 3 for j in range(8)]] == entry_bytes(v, 8)


def test_iq3_family_grids():         # 3 bits per magnitude: IQ3_S 2 t + 1 ; IQ3_XXS 8 t + 4 (+ 2 for t = 7)
    p, g = c_array(PACKED, "k_iq3s_grid_packed"), c_array(GRIDS, "oracle_iq3s_grid")
    assert len(p) == len(g) == 512
    for w, v in zip(p, g):
        assert [2 * ((w >> (3 * j)) & 7) + 1 for j in range(4)] == entry_bytes(v, 4)
    p, g = c_array(PACKED, "k_iq3xxs_grid_packed"), c_array(GRIDS, "oracle_iq3xxs_grid")
    assert len(p) == len(g) == 256
    for w, v in zip(p, g):
        assert [8 * t + 4 + (2 if t == 7 else 0) for t in [(w >> (3 * j)) & 7 for j in range(4)]] == entry_bytes(v, 4)


def test_iq1_grid_and_its_two_images():      # 2 bits per value (g + 1); the kernels read signed bytes 8 g + 1 / 8 g - 1 (expand_iq1_grid, iq_fill_lds)
    p, g = c_array(PACKED, "k_iq1s_grid_packed"), c_array(GRIDS, "oracle_iq1s_grid")
    assert len(p) == len(g) == 2048
    for w, v in zip(p, g):
        vals = [((w >> (2 * j)) & 3) - 1 for j in range(8)]
        assert [x & 0xff for x in vals] == entry_bytes(v, 8) and set(vals) <= {-1, 0, 1}
        plus = sum(((8 * x + 1) & 0xff) << (8 * j) for j, x in enumerate(vals))
        for half in (plus & 0xffffffff, plus >> 32):          # the 8 g - 1 image is derived per dword without a borrow between bytes
            minus = (((half ^ 0x80808080) - 0x02020202) ^ 0x80808080) & 0xffffffff
            assert [(b - 256 if b > 127 else b) for b in entry_bytes(minus, 4)] == [(b - 256 if b > 127 else b) - 2 for b in entry_bytes(half, 4)]


def packed_bytes(words):
    return [((b - 256) if b > 127 else b) for w in words for b in entry_bytes(w, 4)]


def test_value_tables():
    assert packed_bytes(c_array(COMMON, "k_iq4nl_packed")) == c_array(ORACLE, "k_iq4nl")
    assert packed_bytes(c_array(COMMON, "k_mxfp4_packed")) == c_array(ORACLE, "k_mxfp4")
    assert packed_bytes(c_array(COMMON, "k_iq5nl_packed")) == c_array(ORACLE, "k_iq5nl")
    assert packed_bytes(c_array(COMMON, "k_iq6nl_packed")) == c_array(ORACLE, "k_iq6nl")
    n2, n3 = c_array(ORACLE, "k_iq2nl"), c_array(ORACLE, "k_iq3nl")
    assert packed_bytes(c_array(COMMON, "k_iq2nl_packed")) == n2 + [v + 5 for v in n2]                  # second half = first + 5
    assert packed_bytes(c_array(COMMON, "k_iq3nl_packed")) == n3 + [v + 4 for v in n3]                  # + 4
    kl = c_array(ORACLE, "k_iq2kl")                                                                      # 32 pairs, flattened
    assert packed_bytes(c_array(COMMON, "k_iq2kl_v0")) == kl[0::2] and packed_bytes(c_array(COMMON, "k_iq2kl_v1")) == kl[1::2]


def test_e8m0_half():           # 2^(e - 128), the two denormal cases spelled out (ggml-impl.h:40-45)
    def dev(x):
        return np.array([(x - 1) << 23 if x >= 2 else (0x00400000 if x else 0x00200000)], np.uint32).view(np.float32)[0]
    for e in range(0, 255):
        assert dev(e) == np.float32(2.0) ** np.float32(e - 128), e


def test_type_lists_agree():
    """a weight type has to be registered in five places (build.py translation units, api_internal.h dispatch macro, weight_type_ok, the Python size tables, the
    oracle's lists): they must name the same set"""
    from oracle import bindings as ob
    csrc = os.path.join(ROOT, "ik_llama.cpp_amd/csrc")
    build = open(os.path.join(ROOT, "ik_llama.cpp_amd/build.py")).read()
    tus = set(int(x) for x in re.search(r"^BASE_TYPES = \[(.*?)\]", build, re.M).group(1).split(",")) | \
        set(int(x) for x in re.findall(r"\d+", re.search(r"^GEMV_ONLY_TYPES = \[(.*?)\]", build, re.M).group(1)))
    hdr = open(os.path.join(csrc, "api_internal.h")).read()
    macro = set(int(x) for x in re.findall(r"X\((\d+)\)", re.search(r"#define CDNA4_FOR_BASE_TYPES\(X\)(.*)", hdr).group(1))) | \
        set(int(x) for x in re.findall(r"X\((\d+)\)", re.search(r"#define CDNA4_FOR_GEMV_ONLY_TYPES\(X\)(.*)", hdr).group(1)))
    enum = dict((n, int(v)) for n, v in re.findall(r"\b(T_[A-Z0-9_]+) = (\d+)", COMMON))
    api = open(os.path.join(csrc, "cdna4_api.hip")).read()
    body = api[api.index("static bool weight_type_ok"):api.index("int    cdna4_type_supported")]
    ok = set(enum[n] for n in re.findall(r"case (T_[A-Z0-9_]+):", body))
    base_ok = set(t for t in ok if t < 200)
    served = set(ob.BASE_TYPES) | set(ob.LEGACY_TYPES)
    bitnet = set(ob.BITNET_TYPES)        # IQ1_BN / IQ2_BN: a translation unit of their own (gemv_bitnet.hip: plain MUL_MAT + de-quantization), outside the per-type kernel families
    kt = set(ob.KT_TYPES)                # trellis types: decode units only (GEMV_ONLY_TYPES); prompts go through the f16 instance of the GEMM
    gemv_only = set(int(x) for x in re.findall(r"\d+", re.search(r"^GEMV_ONLY_TYPES = \[(.*?)\]", build, re.M).group(1)))
    assert gemv_only == kt == set(int(x) for x in re.findall(r"X\((\d+)\)", re.search(r"#define CDNA4_FOR_GEMV_ONLY_TYPES\(X\)(.*)", hdr).group(1)))
    served |= kt
    assert tus == macro == served and base_ok == served | bitnet, (sorted(tus ^ macro), sorted(base_ok ^ (served | bitnet)), sorted(tus ^ served))
    assert set(t for t in ok if t >= 200) == set(ob.R4_TYPES)
    conv = open(os.path.join(csrc, "convert.hip")).read()                                 # de-quantization and get_rows switches
    dq = set(enum[n] for n in re.findall(r"DQ\((T_[A-Z0-9_]+)\)", conv)); gr = set(enum[n] for n in re.findall(r"GR\((T_[A-Z0-9_]+)\)", conv))
    assert set(t for t in dq if t < 200) == served | bitnet and set(t for t in dq if t >= 200) == set(ob.R4_TYPES), sorted(set(t for t in dq if t < 200) ^ (served | bitnet))
    assert gr - {enum["T_F32"], enum["T_F16"]} == served, sorted((gr - {0, 1}) ^ served)
    pkg_src = open(os.path.join(ROOT, "ik_llama.cpp_amd/cdna4.py")).read()
    sizes = dict((int(k), int(v)) for k, v in re.findall(r"(\d+): (\d+)", re.search(r"^TYPE_SIZE = \{(.*?)\}", pkg_src, re.M).group(1)))
    for t in served | bitnet:
        assert sizes[t] == ob.TYPE_SIZE[t], t


def test_grouped_gemm_band_order_covers_every_tile_once():
    """The grouped (MUL_MAT_ID) prompt GEMM maps a workgroup index to (weight-row tile, token tile) through RB bands of row tiles x 8 / RB token phases over the 8 XCDs
    (gemm_mfma.cuh: kernel prologue `if (a.moe_tiles)`, host `launch_gemm_ks`).  Restated here: every (row tile, token tile) pair is visited exactly once, workgroups
    outside the pairs exit, and every XCD gets the same number of token tiles of the USED prefix of the table (+- 1) -- the property the order exists for."""
    for MT in (1, 4, 6, 13, 16, 32, 36, 112):
        for ntl in (1, 5, 17, 64, 257):
            rb = 8
            while rb > 1 and MT % rb:
                rb >>= 1                                          # launch_gemm_ks: the most bands that divide the row tiles evenly
            tp = 8 // rb; band = (MT + rb - 1) // rb; grid = 8 * band * ((ntl + tp - 1) // tp)
            seen = {}
            per_xcd = [set() for _ in range(8)]
            for b in range(grid):
                xcd = b & 7; li = b >> 3; r = xcd % rb; t = xcd // rb
                nl = li // band; n_tile = nl * tp + t; m_tile = r * band + (li - nl * band)
                if m_tile >= MT or n_tile >= ntl:
                    continue
                assert (m_tile, n_tile) not in seen, (MT, ntl, b, seen[(m_tile, n_tile)])
                seen[(m_tile, n_tile)] = b; per_xcd[xcd].add(n_tile)
            assert len(seen) == MT * ntl, (MT, ntl, len(seen))
            used = max(1, (2 * ntl) // 3)                         # the used entries come first in the tile table
            counts = [sum(1 for n in s_ if n < used) for s_ in per_xcd if s_]
            assert max(counts) - min(counts) <= 1, (MT, ntl, counts)
