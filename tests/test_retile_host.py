"""Host-side re-tiling of the `_R4` forms of ik's non-linear types (csrc/retile_host.hip, cdna4_retile_r4_host): bytes pinned against the
reference's own repacker (iqk_repack_tensor, iqk_quantize.cpp:8535-8583), exact round trip, and the reference's `_R4` de-quantizer applied to
OUR interleaved bytes.  Pure host code: runs without a GPU."""
import ctypes as C

import numpy as np
import pytest

from common import gaussian_weights_f32, random_block_bytes
from conftest import load_package
from oracle import bindings as ob

# base -> row-interleaved type (ggml.h:461-490); ROWS = rows per interleaved group
R4_HOST = {ob.IQ2_K: 337, ob.IQ3_K: 338, ob.IQ4_K: 339, ob.IQ5_K: 340, ob.IQ4_KS: 344, ob.IQ5_KS: 352,
           ob.Q4_0: 202, ob.Q5_0: 206, ob.Q6_0: 233, ob.Q8_0: 208, ob.MXFP4: 353, ob.Q2_K: 210, ob.Q3_K: 211, ob.IQ4_XS: 223,
           ob.IQ2_XXS: 216, ob.IQ2_XS: 217, ob.IQ3_XXS: 218, ob.IQ2_BN: 335}
R4_NAMES = {337: "iq2_k_r4", 338: "iq3_k_r4", 339: "iq4_k_r4", 340: "iq5_k_r4", 344: "iq4_ks_r4", 352: "iq5_ks_r4", 202: "q4_0_r8", 206: "q5_0_r4", 233: "q6_0_r4", 208: "q8_0_r8", 353: "mxfp4_r8",
            210: "q2_k_r4", 211: "q3_k_r4", 223: "iq4_xs_r8", 216: "iq2_xxs_r4", 217: "iq2_xs_r4", 218: "iq3_xxs_r4", 335: "iq2_bn_r4"}
ROWS = {r: (8 if R4_NAMES[r].endswith("_r8") else 4) for r in R4_NAMES}
BASE_NAME = dict(ob.NAMES); BASE_NAME[ob.IQ2_BN] = "iq2_bn"


def retile(lib, r4_t, w, k, to_base, threads=1):
    w = np.ascontiguousarray(w, dtype=np.uint8); out = np.empty_like(w)
    rc = lib.cdna4_retile_r4_host(r4_t, w.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), w.shape[0], k, 1 if to_base else 0, threads)
    assert rc == 0, lib.cdna4_last_error()
    return out


@pytest.fixture(scope="module")
def lib():
    return load_package().load_library()


def test_served_set(lib):
    for base, r4 in R4_HOST.items():
        assert lib.cdna4_retile_r4_host_base_type(r4) == base and lib.cdna4_retile_r4_host_rows(r4) == ROWS[r4]
    for t in (212, 220, 12, 139, 341, 219, 229, 397, 398, 399, 230, 0, -1):          # device-retiled _R4 types, base types, IQ6_K's id + 200 (no such type), IQ1_S_R4 / IQ1_M_R4 / Q8_K_R16 / Q8_KV_R8 / Q8_K_R8 / BF16_R16: not host-retiled
        assert lib.cdna4_retile_r4_host_base_type(t) == -1 and lib.cdna4_retile_r4_host_rows(t) == 0


@pytest.mark.parametrize("base", list(R4_HOST), ids=lambda t: BASE_NAME[t])
@pytest.mark.parametrize("m,k", [(8, 256), (8, 1024), (24, 512), (64, 4096)])
def test_bytes_equal_iqk_repack_tensor_and_round_trip(base, m, k, lib, ref):
    r4 = R4_HOST[base]
    for w in (ref.quantize(base, gaussian_weights_f32(m, k, 3)), random_block_bytes(base, m, k, 4 + m)):
        new_t, want = ref.repack_tensor(base, w, k)
        assert new_t == r4, (new_t, r4)
        got = retile(lib, r4, w, k, to_base=False)
        assert np.array_equal(got, want.reshape(got.shape))
        assert np.array_equal(retile(lib, r4, got, k, to_base=True), w)


@pytest.mark.parametrize("base", list(R4_HOST), ids=lambda t: BASE_NAME[t])
def test_bytes_equal_the_committed_iqk_repack_tensor_output(base, lib):
    """the same pin without the reference library: tests/golden/r4_host_golden.npz holds iqk_repack_tensor's output (tests/golden/make_golden_r4_host.py)"""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "r4_host_golden.npz"))
    r4 = R4_HOST[base]; k = int(g["meta"][1])
    for tag in ("q", "b"):
        w, want = g["base_%s_%d" % (tag, r4)], g["r_%s_%d" % (tag, r4)]
        assert np.array_equal(retile(lib, r4, w, k, to_base=False), want)
        assert np.array_equal(retile(lib, r4, want, k, to_base=True), w)


@pytest.mark.parametrize("base", list(R4_HOST), ids=lambda t: BASE_NAME[t])
def test_every_bit_pattern_survives_the_round_trip(base, lib):
    """both directions are bijections on the raw bytes (also for bit patterns no quantizer emits): base -> _R4 -> base and _R4 -> base -> _R4"""
    r4 = R4_HOST[base]; k = 768
    rng = np.random.default_rng(11)
    w = rng.integers(0, 256, size=(16, ob.row_size(base, k)), dtype=np.uint8)
    # (IQ2_XXS / IQ3_XXS / IQ2_XS: the sign index is re-coded through a bijection of the 7-bit codes -- still a bijection of the bytes)
    assert np.array_equal(retile(lib, r4, retile(lib, r4, w, k, False), k, True), w)
    assert np.array_equal(retile(lib, r4, retile(lib, r4, w, k, True), k, False), w)
    if base in (ob.IQ2_K, ob.IQ3_K, ob.IQ4_K, ob.IQ5_K, ob.IQ4_KS, ob.IQ5_KS, ob.Q4_0, ob.Q8_0, ob.MXFP4, ob.Q2_K, ob.IQ4_XS):       # pure bit / byte permutations with uniform fields
        for fill in (0x00, 0xff):
            w[:] = fill
            assert np.array_equal(retile(lib, r4, w, k, False), w) and np.array_equal(retile(lib, r4, w, k, True), w)


@pytest.mark.parametrize("base", list(R4_HOST), ids=lambda t: BASE_NAME[t])
def test_reference_r4_dequantizer_reads_our_interleave(base, lib, ref):
    """dequantize_row_*_r4 of the reference applied to OUR interleaved bytes gives the base type's values, row by row"""
    r4 = R4_HOST[base]; m, k = 8, 1024; R = ROWS[r4]
    if base == ob.IQ2_BN:
        pytest.skip("the reference has no stand-alone row de-quantizer for the BitNet types (oracle restatement only)")
    if base == ob.MXFP4:
        pytest.skip("dequantize_row_mxfp4_r8 walks k / 32 blocks instead of (k / 8) / 32 (iqk_quantize.cpp:4333): it runs 8x past its input and output")
    f = getattr(ref.lib, "dequantize_row_" + R4_NAMES[r4]); f.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]; f.restype = None
    fb = getattr(ref.lib, "dequantize_row_" + BASE_NAME[base]); fb.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]; fb.restype = None
    for w in (ref.quantize(base, gaussian_weights_f32(m, k, 5)), random_block_bytes(base, m, k, 6)):
        wr = retile(lib, r4, w, k, to_base=False)
        out = np.empty((m, k), np.float32); want = np.empty((m, k), np.float32)
        for r in range(0, m, R):
            f(wr[r:].ctypes.data_as(C.c_void_p), out[r:].ctypes.data_as(C.c_void_p), R * k)
        for r in range(m):
            fb(w[r:].ctypes.data_as(C.c_void_p), want[r:].ctypes.data_as(C.c_void_p), k)
        if base in (ob.IQ2_K, ob.IQ3_K, ob.IQ4_K, ob.IQ5_K, ob.IQ4_KS, ob.IQ5_KS):
            assert np.array_equal(out.view(np.uint32), want.view(np.uint32))
        else:       # the reference's interleaved de-quantizers of the older types round differently from their base twins (d * (q - 16) against d * q - 16 d ...): same weights to f32 rounding
            assert np.allclose(out, want, rtol=2e-6, atol=1e-9), float(np.max(np.abs(out - want)))


@pytest.mark.parametrize("base", list(R4_HOST), ids=lambda t: BASE_NAME[t])
def test_guard_bytes_around_the_destination_stay_untouched(base, lib):
    """a row group owns 4-byte pieces of the interleaved blocks: no access may be widened past them (the last group's pieces end at the buffer's end)"""
    r4 = R4_HOST[base]; rng = np.random.default_rng(7)
    for m, k in ((8, 256), (8, 1024), (64, 4096)):
        n = m * ob.row_size(base, k)
        for to_base in (0, 1):
            src = np.full(n + 512, 0xAB, np.uint8); dst = np.full(n + 512, 0xCD, np.uint8)
            src[256:256 + n] = rng.integers(0, 256, n, dtype=np.uint8)
            assert lib.cdna4_retile_r4_host(r4, C.c_void_p(src.ctypes.data + 256), C.c_void_p(dst.ctypes.data + 256), m, k, to_base, 1) == 0
            assert (dst[:256] == 0xCD).all() and (dst[256 + n:] == 0xCD).all(), (m, k, to_base)
            assert (src[:256] == 0xAB).all() and (src[256 + n:] == 0xAB).all()


def test_threads_and_argument_checks(lib):
    base, r4, k = ob.IQ4_K, 339, 4096
    w = np.random.default_rng(2).integers(0, 256, size=(4096, ob.row_size(base, k)), dtype=np.uint8)        # 16384 row-group blocks: takes the threaded path
    a = retile(lib, r4, w, k, False, threads=1)
    for nt in (0, 3, 8):
        assert np.array_equal(retile(lib, r4, w, k, False, threads=nt), a)
    p = w.ctypes.data_as(C.c_void_p); o = np.empty_like(w).ctypes.data_as(C.c_void_p)
    assert lib.cdna4_retile_r4_host(r4, p, o, 6, k, 1, 1) == -2            # nrows % 4
    assert lib.cdna4_retile_r4_host(202, p, o, 12, k, 1, 1) == -2          # Q4_0_R8: nrows % 8
    assert lib.cdna4_retile_r4_host(r4, p, o, 8, 128, 1, 1) == -2          # ne00 % 256
    assert lib.cdna4_retile_r4_host(r4, p, p, 8, k, 1, 1) == -2            # in place
    assert lib.cdna4_retile_r4_host(212, p, o, 8, k, 1, 1) == -1           # Q4_K_R4 is re-tiled on the device (cdna4_unrepack_r4)
