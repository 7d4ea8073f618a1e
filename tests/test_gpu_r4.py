"""-m gpu: the row-interleaved (_R4) variants: device-side iqk_repack_tensor and its inverse are bit-exact;
mat-muls on _R4 tensors reproduce the reference's _R4 CPU kernels' arithmetic (Q8_K32 / Q8_K / Q8_2_X4 activations)."""
import os

import numpy as np
import pytest
import torch

from common import TOL_INT8_PATH, activations, make_weights, random_block_bytes
from oracle import bindings as ob
from test_gpu_parity import check_mul_mat, dev

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "iqk_golden.npz"))
GM, GK = [int(v) for v in G["meta"]]


@pytest.mark.parametrize("t", ob.BASE_TYPES, ids=lambda t: ob.NAMES[t])
def test_repack_and_inverse_bit_exact(t, backend, oracle):
    assert np.array_equal(backend.repack_r4(t, dev(G["w_%d" % t]), GK).cpu().numpy(), G["w_%d" % ob.R4_OF[t]])     # golden (reference repack layout)
    m, k = 64, 2048
    w = random_block_bytes(t, m, k, 5)
    r4 = backend.repack_r4(t, dev(w), k)
    assert np.array_equal(r4.cpu().numpy(), oracle.repack_r4(t, w, k))
    assert np.array_equal(backend.unrepack_r4(t, r4, k).cpu().numpy(), w)


@pytest.mark.parametrize("t", ob.R4_TYPES, ids=lambda t: ob.NAMES[t])
@pytest.mark.parametrize("n", [1, 2, 8])
def test_r4_gemv_golden(t, n, backend, oracle):
    x = G["x"][:n]; w = G["w_%d" % t]; vdt = ob.vec_dot_type(t)
    got = backend.mul_mat(t, dev(w), dev(x)).cpu().numpy()
    xq = oracle.dequantize_activations(vdt, oracle.quantize_activations(vdt, x), GK)
    _, sum_abs = oracle.mul_mat_f64(t, w, xq)
    err = np.max(np.abs(got.astype(np.float64) - G["mm_%d_n%d" % (t, n)]) / sum_abs)      # vs the REAL reference _R4 kernels
    assert err < TOL_INT8_PATH, err


@pytest.mark.parametrize("t", ob.R4_TYPES, ids=lambda t: ob.NAMES[t])
@pytest.mark.parametrize("m,k", [(256, 4096), (64, 14336), (132, 1024)])
def test_r4_gemv_shapes(t, m, k, backend, oracle):
    w = make_weights(t, m, k, 40 + t, oracle)
    for n, seed in ((1, 1), (3, 2)):
        check_mul_mat(backend, oracle, t, w, activations(n, k, seed, outliers=(n == 3)), int8_path=True)


@pytest.mark.parametrize("t", [ob.Q4_K_R4, ob.Q6_K_R4, ob.IQ4_NL_R4, ob.IQ3_S_R4], ids=lambda t: ob.NAMES[t])
def test_r4_prefill(t, backend, oracle):
    w = make_weights(t, 256, 1024, 60 + t, oracle)
    check_mul_mat(backend, oracle, t, w, activations(48, 1024, 3), int8_path=False)


def test_weight_cache_invalidation(backend, oracle):
    t, m, k = ob.Q4_K_R4, 64, 1024
    w1 = make_weights(t, m, k, 1, oracle); w2 = make_weights(t, m, k, 2, oracle); x = dev(activations(1, k, 3))
    wd = dev(w1)
    a = backend.mul_mat(t, wd, x).clone()
    wd.copy_(dev(w2))                       # tensor bytes change in place (set_tensor): the owner must invalidate
    backend.invalidate_weight_cache(wd)
    b = backend.mul_mat(t, wd, x)
    assert not torch.equal(a, b)
    assert torch.equal(b, backend.mul_mat(t, dev(w2), x))
    backend.invalidate_weight_cache()


@pytest.mark.parametrize("t", [ob.R4_OF[ob.Q4_K], ob.R4_OF[ob.Q6_K], ob.R4_OF[ob.IQ4_NL]], ids=lambda t: ob.NAMES[t])
def test_r4_experts_mul_mat_id(t, backend, oracle):
    """_R4 expert tensors through MUL_MAT_ID / MOE_FUSED_UP_GATE (a8 x a11): decode path reproduces the _R4 CPU arithmetic."""
    base = ob.BASE_OF[t]; m, k, n_expert, n_used, n_tok = 96, 512, 4, 2, 3
    ws = np.stack([oracle.repack_r4(base, random_block_bytes(base, m, k, 700 + e), k) for e in range(n_expert)])
    wg = np.stack([oracle.repack_r4(base, random_block_bytes(base, m, k, 750 + e), k) for e in range(n_expert)])
    x = activations(n_tok, k, 71).reshape(n_tok, 1, k)
    ids = np.random.default_rng(7).integers(0, n_expert, size=(n_tok, n_used)).astype(np.int32); ids[1, 1] = -1
    got = backend.mul_mat_id(t, dev(ws), dev(x), dev(ids)).cpu().numpy()
    want = oracle.mul_mat_id(t, ws, x, ids)
    assert np.allclose(got, want, rtol=2e-5, atol=2e-6 * np.abs(want).max())
    assert np.all(got[ids < 0] == 0)
    got = backend.moe_fused_up_gate(t, dev(ws), dev(wg), dev(x), dev(ids), op=10).cpu().numpy()
    for tk in range(n_tok):
        for s in range(n_used):
            e = ids[tk, s]
            if e < 0:
                assert np.all(got[tk, s] == 0); continue
            w1 = oracle.fused_up_gate(t, 10, ws[e], wg[e], x[tk])[0]
            assert np.allclose(got[tk, s], w1, rtol=2e-5, atol=2e-6 * max(1.0, np.abs(w1).max()))
    backend.invalidate_weight_cache()
