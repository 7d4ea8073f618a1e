"""Oracle vs the committed golden vectors produced by the real reference (tests/golden/make_golden.py).
These run everywhere (no reference library needed), so the oracle stays pinned on the GPU box too."""
import os

import numpy as np
import pytest

from oracle import bindings as ob

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "iqk_golden.npz"))
M, K = [int(v) for v in G["meta"]]
ALL = ob.BASE_TYPES + ob.R4_TYPES


@pytest.mark.parametrize("t", ALL, ids=lambda t: ob.NAMES[t])
def test_oracle_dequant_matches_golden(t, oracle):
    for wk, dk in (("w_%d", "deq_%d"), ("wb_%d", "deqb_%d")):
        got = oracle.dequantize(t, G[wk % t], K)
        assert np.array_equal(got.view(np.uint32), G[dk % t].view(np.uint32))


@pytest.mark.parametrize("t", ob.BASE_TYPES, ids=lambda t: ob.NAMES[t])
def test_oracle_repack_matches_golden(t, oracle):
    assert np.array_equal(oracle.repack_r4(t, G["w_%d" % t], K), G["w_%d" % ob.R4_OF[t]])


@pytest.mark.parametrize("vdt", [ob.Q8_2_X4, ob.Q8_K, ob.Q8_K32])
def test_oracle_activation_quant_matches_golden(vdt, oracle):
    assert np.array_equal(oracle.quantize_activations(vdt, G["x"]), G["xq_%d" % vdt])


@pytest.mark.parametrize("t", ALL, ids=lambda t: ob.NAMES[t])
@pytest.mark.parametrize("n", [1, 2, 8])
def test_oracle_mul_mat_matches_golden(t, n, oracle):
    x = G["x"][:n]; w = G["w_%d" % t]; vdt = ob.vec_dot_type(t)
    xq = oracle.dequantize_activations(vdt, oracle.quantize_activations(vdt, x), K)
    _, sum_abs = oracle.mul_mat_f64(t, w, xq)
    err = np.max(np.abs(oracle.mul_mat(t, w, x).astype(np.float64) - G["mm_%d_n%d" % (t, n)]) / sum_abs)
    assert err < 2e-6, err
