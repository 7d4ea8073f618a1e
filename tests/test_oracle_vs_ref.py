"""Pins the oracle (oracle/iqk_oracle.c) against the REAL reference library built from /root/reference
(oracle/_ref, oracle/Makefile).  Skipped where that library cannot run."""
import numpy as np
import pytest

from common import activations, gaussian_weights_f32, random_block_bytes
from oracle import bindings as ob


def bits(a):
    return a.view(np.uint32)


@pytest.mark.parametrize("t", ob.BASE_TYPES, ids=lambda t: ob.NAMES[t])
def test_dequant_bit_exact(t, oracle, ref):
    k = 2048
    for w in (ref.quantize(t, gaussian_weights_f32(16, k, 1)), random_block_bytes(t, 16, k, 2)):
        assert np.array_equal(bits(ref.dequantize(t, w, k)), bits(oracle.dequantize(t, w, k)))


@pytest.mark.parametrize("t", ob.BASE_TYPES, ids=lambda t: ob.NAMES[t])
def test_r4_repack_and_dequant_bit_exact(t, oracle, ref):
    k = 1024
    for w in (ref.quantize(t, gaussian_weights_f32(8, k, 3)), random_block_bytes(t, 8, k, 4)):
        wr = oracle.repack_r4(t, w, k)
        a = ref.dequantize(ob.R4_OF[t], wr, k)
        # the reference's own R4 dequantizer applied to OUR repack reproduces the base dequant => layout is the reference's
        assert np.array_equal(bits(a), bits(ref.dequantize(t, w, k)))
        assert np.array_equal(bits(a), bits(oracle.dequantize(ob.R4_OF[t], wr, k)))


@pytest.mark.parametrize("t", ob.BASE_TYPES, ids=lambda t: ob.NAMES[t])
@pytest.mark.parametrize("m,k", [(4, 256), (8, 1024), (64, 4096), (12, 512)])
def test_r4_bytes_equal_iqk_repack_tensor(t, m, k, oracle, ref):
    """the row-interleaved layout pinned against the reference's OWN repacker (iqk_repack_tensor, iqk_quantize.cpp:8535-8583: the function
    `llama-quantize --repack` and -rtr call), byte for byte, all six types (IQ2_S included)"""
    for w in (ref.quantize(t, gaussian_weights_f32(m, k, 3)), random_block_bytes(t, m, k, 4)):
        new_t, want = ref.repack_tensor(t, w, k)
        assert new_t == ob.R4_OF[t], (new_t, ob.R4_OF[t])
        assert np.array_equal(want, oracle.repack_r4(t, w, k).reshape(want.shape))


@pytest.mark.parametrize("vdt", [ob.Q8_2_X4, ob.Q8_K, ob.Q8_K32])
def test_activation_quantizers_bit_exact(vdt, oracle, ref):
    x = activations(12, 2048, 5)
    x[1, ::256] = 1e3; x[2] = 0; x[3, :300] *= 1e-30; x[4] *= 1e20
    x[5, :32] = np.array([127.49, -127.5] * 16, dtype=np.float32)
    assert np.array_equal(ref.quantize_activations(vdt, x), oracle.quantize_activations(vdt, x))


def test_q8_2_x4_ragged_tail(oracle, ref):
    x = activations(3, 160, 6)          # 5 blocks of 32: one x4 group + a block_q8_2 tail
    assert np.array_equal(ref.quantize_activations(ob.Q8_2_X4, x), oracle.quantize_activations(ob.Q8_2_X4, x))


@pytest.mark.parametrize("t", ob.BASE_TYPES + ob.R4_TYPES, ids=lambda t: ob.NAMES[t])
@pytest.mark.parametrize("n", [1, 2, 8])
def test_mul_mat_matches_reference_direct_kernels(t, n, oracle, ref):
    """N <= 8: the reference runs its direct int8 kernels; the oracle states the same arithmetic (exact int32 block
    sums, f32 scale accumulate) so results agree to f32 summation order."""
    m, k = 64, 1024
    base = ob.BASE_OF.get(t, t)
    w = ref.quantize(base, gaussian_weights_f32(m, k, 7))
    if t in ob.BASE_OF:
        w = oracle.repack_r4(base, w, k)
    x = activations(n, k, 8 + n, outliers=(n == 2))
    vdt = ob.vec_dot_type(t)
    xq = oracle.dequantize_activations(vdt, oracle.quantize_activations(vdt, x), k)
    _, sum_abs = oracle.mul_mat_f64(t, w, xq)
    err = np.max(np.abs(ref.mul_mat(t, w, x).astype(np.float64) - oracle.mul_mat(t, w, x)) / sum_abs)
    assert err < 2e-6, err      # f32 summation-order noise only (IQ4_NL: the reference adds a -128*sum(y) correction term)


@pytest.mark.parametrize("op", [6, 10, 14, 15], ids=["relu", "silu", "swiglu_oai", "gelu"])
@pytest.mark.parametrize("bias,limit", [(False, 0.0), (True, 0.0), (False, 0.75), (True, 2.0)])
def test_fused_up_gate_epilogue_matches_reference(op, bias, limit, oracle, ref):
    """the fused up*gate epilogue incl. biases, `limit` clamp and the SWIGLU_OAI form against the reference's own
    iqk_moe_fused_up_gate (iqk_mul_mat.cpp:783-857 -> mul_mat_up_gate_NxM :136-236)."""
    t, m, k, n = ob.Q4_K, 64, 1024, 3
    wu = ref.quantize(t, gaussian_weights_f32(m, k, 21) * 40); wg = ref.quantize(t, gaussian_weights_f32(m, k, 22) * 40)
    x = activations(n, k, 23)
    rng = np.random.default_rng(24)
    ub = rng.normal(0, 1, m).astype(np.float32) if bias else None
    gb = rng.normal(0, 1, m).astype(np.float32) if bias else None
    want = ref.fused_up_gate(t, op, wu, wg, x, ub, gb, limit)
    got = oracle.fused_up_gate(t, op, wu, wg, x, ub, gb, limit)
    # scale of the products (|up| * |act(gate)| up to a few units here); both sides differ by f32 summation order in the dots
    # and by the reference's vectorised expf/tanhf in SILU/GELU (~1e-6 relative)
    tol = 2e-5 * max(1.0, float(np.max(np.abs(want))))
    assert np.max(np.abs(want - got)) < tol, (np.max(np.abs(want - got)), tol)
    if limit > 0:                        # the clamps must actually bite in this data
        plain = oracle.fused_up_gate(t, op, wu, wg, x, ub, gb, 0.0)
        assert np.max(np.abs(plain - got)) > 1e-3


# types whose reference AVX-512 kernel (mul_mat_iqX_k_q8_K_AVX512, values + 128 through _mm512_maddubs_epi16) saturates int16 pair sums on full-range int8 activations
SATURATING = (ob.IQ4_XS, ob.IQ4_K, ob.IQ5_K, ob.IQ4_KS, ob.IQ5_KS, ob.IQ4_KSS, ob.IQ6_K)


# ---- more weight types: legacy 32-blocks, the remaining K / IQ types, ik's non-linear types
@pytest.mark.parametrize("t", ob.LEGACY_TYPES, ids=lambda t: ob.NAMES[t])
def test_legacy_dequant_bit_exact(t, oracle, ref):
    k = 2048
    for w in (ref.quantize(t, gaussian_weights_f32(16, k, 1)), random_block_bytes(t, 16, k, 2)):
        assert np.array_equal(bits(ref.dequantize(t, w, k)), bits(oracle.dequantize(t, w, k)))


@pytest.mark.parametrize("t", ob.LEGACY_TYPES, ids=lambda t: ob.NAMES[t])
@pytest.mark.parametrize("n", [1, 2, 8])
def test_legacy_mul_mat_matches_reference_direct_kernels(t, n, oracle, ref):
    """mul_mat_qX_1_q8_2_T<Q4_0_1_Unpacker / Q8_0_1_Unpacker> (iqk_gemm_legacy_quants.cpp:753-770,2338-2358) and mul_mat_iqX_k_q8_K_AVX512<DequantizerIQ4XS>
    (iqk_gemm_kquants.cpp:292-332,423-465): unsigned quants + a sum(y) correction in the reference, signed dot in the oracle -- equal to f32 rounding.
    IQ4_XS (and IQ4_K, IQ5_K, IQ4_KS, IQ5_KS, same kernel): the reference's AVX-512 kernel adds PAIRS of (codebook + 128) x int8 products into int16 with saturation (_mm512_maddubs_epi16: up to
    2 x 241 x 127 > 32767), so with activations that fill the int8 range it deviates from its own exact arithmetic by ~1e-2 of sum|w x| on some rows;
    the oracle states the exact sum, so the tight bar is checked on activations whose int8 values stay small (one outlier per 256 sets the scale); on
    N(0, 1) activations the reference's kernel is 6e-4 ... 5e-3 (NMSE) away from its own exact form -- a loose sanity bar only."""
    m, k = 64, 1024
    w = ref.quantize(t, gaussian_weights_f32(m, k, 7))
    vdt = ob.vec_dot_type(t)
    for outliers in ((True,) if t in SATURATING else (n == 2,)):
        x = activations(n, k, 8 + n, outliers=outliers)
        xq = oracle.dequantize_activations(vdt, oracle.quantize_activations(vdt, x), k)
        _, sum_abs = oracle.mul_mat_f64(t, w, xq)
        err = np.max(np.abs(ref.mul_mat(t, w, x).astype(np.float64) - oracle.mul_mat(t, w, x)) / sum_abs)
        assert err < 2e-6, err
    if t in SATURATING:
        from common import nmse
        x = activations(n, k, 8 + n)
        assert nmse(oracle.mul_mat(t, w, x), ref.mul_mat(t, w, x)) < 2e-2


# ---- BitNet types (oracle ahead of the device path): Q8_K64 activations byte-exact, the two mat-mul kernels, the value a weight has
@pytest.mark.parametrize("k", [512, 1024, 576 + 64, 4096])
def test_q8_k64_activations_byte_exact(k, oracle, ref):
    for seed, outliers in ((1, False), (2, True)):
        x = activations(3, k, seed, outliers=outliers)
        assert np.array_equal(ref.quantize_activations(ob.Q8_K64, x), oracle.quantize_activations(ob.Q8_K64, x))


@pytest.mark.parametrize("t", ob.BITNET_TYPES, ids=["iq1_bn", "iq2_bn"])
@pytest.mark.parametrize("n", [1, 2, 8])
def test_bitnet_mul_mat_matches_reference_kernels(t, n, oracle, ref):
    """mul_mat_iq1bn_q8_K64 / mul_mat_iq2bn_q8_K64 (iqk_gemm_1bit.cpp:1247-1447): exact per-class integer sums over the row, one fma per class, a fixed 4-way sum,
    the row scale -- restated in that order, so the results agree to the last bit; K with an odd number of 64-blocks takes the kernels' tail path"""
    for m, k in ((64, 1024), (16, 576)):
        for w in (ref.quantize(t, gaussian_weights_f32(m, k, 7)), random_block_bytes(t, m, k, 8)):
            x = activations(n, k, 9 + n)
            assert np.array_equal(bits(ref.mul_mat(t, w, x)), bits(oracle.mul_mat(t, w, x)))


@pytest.mark.parametrize("t", ob.BITNET_TYPES, ids=["iq1_bn", "iq2_bn"])
def test_bitnet_weight_values(t, oracle, ref):
    """the real quantizer maps a row to max|x| x {-1, 0, +1} (threshold max / 2): the oracle's de-quantized row is that"""
    m, k = 8, 1024
    wf = gaussian_weights_f32(m, k, 3); w = ref.quantize(t, wf)
    mx = np.abs(wf).max(axis=1, keepdims=True)
    if t == ob.IQ1_BN:
        mx = mx.astype(np.float16).astype(np.float32)
    want = np.where(np.abs(wf) < 0.5 * np.abs(wf).max(axis=1, keepdims=True), 0.0, np.sign(wf)).astype(np.float32) * mx
    assert np.array_equal(oracle.dequantize(t, w, k), want)


# ---- trellis types (IQ1_KT / IQ2_KT / IQ3_KT / IQ4_KT): the generator, the four block layouts, the 8-lane f32 accumulation of the AVX2 kernels
@pytest.mark.parametrize("t", ob.KT_TYPES, ids=lambda t: ob.NAMES[t])
def test_kt_dequant_bit_exact(t, oracle, ref):
    """dequantize_row_iqX_kt (the scalar to_float: row scale x block scale x generator value, WITHOUT the 1.05 / 1.01 of the mat-mul kernels)"""
    k = 1024
    for w in (ref.quantize(t, gaussian_weights_f32(4, k, 1)), random_block_bytes(t, 16, k, 2)):
        assert np.array_equal(bits(ref.dequantize(t, w, k)), bits(oracle.dequantize(t, w, k)))


@pytest.mark.parametrize("t", ob.KT_TYPES, ids=lambda t: ob.NAMES[t])
@pytest.mark.parametrize("n", [1, 2, 8])
def test_kt_mul_mat_matches_reference_kernels_to_the_bit(t, n, oracle, ref):
    """mul_mat_iqX_kt_q8_2_x4_T (iqk_gemm_ktquants.cpp): exact int32 sums per 16 weights, eight f32 fma accumulators (32-block b of a 128, half h), hsum_float_8 -- restated
    in that order, so the f32 results are IDENTICAL, on real quantizer output and on random-bit blocks"""
    k = 1024
    for w in (ref.quantize(t, gaussian_weights_f32(4, k, 7)), random_block_bytes(t, 24, k, 8)):
        x = activations(n, k, 9 + n, outliers=(n == 2))
        assert np.array_equal(bits(ref.mul_mat(t, w, x)), bits(oracle.mul_mat(t, w, x)))


def test_kt_matmul_scale_factor(oracle, ref):
    """the reference's mat-mul kernels scale IQ2_KT rows by 1.05 and IQ3_KT rows by 1.01 on top of what its to_float returns (iqk_gemm_ktquants.cpp:424,705,763,849; CUDA alike):
    mul_mat(w, x) == factor * (x_q . to_float(w)) to rounding"""
    k = 1024
    for t, f in ((ob.IQ2_KT, 1.05), (ob.IQ3_KT, 1.01), (ob.IQ4_KT, 1.0), (ob.IQ1_KT, 1.0)):
        w = random_block_bytes(t, 8, k, 3); x = activations(2, k, 4)
        xq = oracle.dequantize_activations(ob.Q8_2_X4, oracle.quantize_activations(ob.Q8_2_X4, x), k)
        want = f * (xq.astype(np.float64) @ ref.dequantize(t, w, k).astype(np.float64).T)
        got = ref.mul_mat(t, w, x)
        assert np.max(np.abs(got - want)) <= 2e-5 * np.max(np.abs(want)), (ob.NAMES[t], np.max(np.abs(got - want)), np.max(np.abs(want)))
