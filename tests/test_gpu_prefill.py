"""-m gpu: prompt-processing path (N > 8): dequant -> f16 MFMA GEMM (north_star) and its int8-dot alternative.

Parity bars: L1 |out - fp64 accumulate(L0 weights x f16-rounded activations)| <= 1e-3 * sum|w*x| ; L2 NMSE vs the
CPU-arithmetic result <= 5e-4 (the reference's own MUL_MAT tolerance, tests/test-backend-ops.cpp:979-981)."""
import numpy as np
import pytest
import torch

from common import NMSE_VS_CPU, TOL_FP_ACCUM, activations, make_weights, nmse
from oracle import bindings as ob
from test_gpu_parity import check_mul_mat, dev

pytestmark = pytest.mark.gpu
MFMA_TYPES = [ob.Q4_K, ob.Q5_K, ob.Q6_K, ob.IQ4_NL, ob.IQ2_S, ob.IQ3_S]


@pytest.mark.parametrize("t", MFMA_TYPES, ids=lambda t: ob.NAMES[t])
@pytest.mark.parametrize("m,k,n", [(256, 1024, 32), (130, 2048, 9), (384, 4096, 100), (128, 512, 512), (96, 14336, 40), (700, 1024, 300)])
def test_mfma_gemm_shapes(t, m, k, n, backend, oracle):
    w = make_weights(t, m, k, 500 + t, oracle)
    check_mul_mat(backend, oracle, t, w, activations(n, k, n, outliers=(n == 40)), int8_path=False)


@pytest.mark.parametrize("t", [ob.Q4_K, ob.Q6_K], ids=lambda t: ob.NAMES[t])
def test_mfma_gemm_real_quantizer_weights_vs_reference(t, backend, oracle, ref):
    """weights from the reference quantizer, result compared with the REAL reference CPU backend (iqk_mul_mat, which for
    N >= 32 takes its repack path -- lossy for Q6_K): NMSE bar of the reference's own backend test."""
    m, k, n = 512, 4096, 64
    w = make_weights(t, m, k, 9, oracle, ref=ref); x = activations(n, k, 10)
    got = backend.mul_mat(t, dev(w), dev(x)).cpu().numpy()
    assert nmse(got, ref.mul_mat(t, w, x)) < NMSE_VS_CPU


@pytest.mark.parametrize("t", [ob.Q4_K, ob.Q6_K], ids=lambda t: ob.NAMES[t])
def test_fused_up_gate_prefill(t, backend, oracle):
    m, k, n = 200, 1024, 48
    wu = make_weights(t, m, k, 21, oracle); wg = make_weights(t, m, k, 22, oracle); x = activations(n, k, 23)
    got = backend.fused_up_gate(t, dev(wu), dev(wg), dev(x), op=10).cpu().numpy()
    xh = x.astype(np.float16).astype(np.float32)
    u, _ = oracle.mul_mat_f64(t, wu, xh); g, _ = oracle.mul_mat_f64(t, wg, xh)
    want = (g * 0.5 * (1 + np.tanh(0.5 * g))) * u        # silu(g) = g*sigmoid(g), overflow-free form
    assert nmse(got, want) < 1e-6


def test_int8_prefill_mode_matches_cpu_arithmetic(backend, oracle):
    """CDNA4_PREFILL_INT8_DOT: the CPU path's arithmetic also for N > 8 (parity mode)."""
    t = ob.Q4_K; w = make_weights(t, 256, 2048, 5, oracle); x = activations(20, 2048, 6)
    backend.set_prefill_mode(1)
    try:
        check_mul_mat(backend, oracle, t, w, x, int8_path=True)
    finally:
        backend.set_prefill_mode(0)


def test_prefill_linearity_full_size(backend):
    """Size-independent check at the BASELINE shape (14336 x 4096 Q4_K, N=512): columns are independent."""
    from common import random_block_bytes
    t, m, k, n = ob.Q4_K, 14336, 4096, 512
    w = dev(random_block_bytes(t, m, k, 1)); x = torch.randn(n, k, device="cuda")
    full = backend.mul_mat(t, w, x)
    part = backend.mul_mat(t, w, x[100:164].contiguous())
    # (different N may choose a different token tile / K split, i.e. another f32 summation order)
    assert torch.allclose(full[100:164], part, rtol=1e-4, atol=1e-4 * float(full.abs().max()))
    assert torch.isfinite(full).all()
