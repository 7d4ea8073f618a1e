"""-m gpu: cases added in round 2 -- the hazards the round-1 review found (ADVICE.md / VERDICT.md "What's weak"):
f16 range of the prefill activations, MUL_MAT_ID batches beyond the 65535-workgroup grid limits, many invalid expert ids,
and sentinel (overrun) checks around every output buffer (tests/test-backend-ops.cpp:440-476 style)."""
import numpy as np
import pytest
import torch

from common import NMSE_VS_CPU, TOL_FP_ACCUM, activations, make_weights, nmse
from oracle import bindings as ob
from test_gpu_parity import dev

pytestmark = pytest.mark.gpu
SENTINEL = np.float32(-7.5e37)


@pytest.mark.parametrize("t", [ob.Q4_K, ob.Q6_K, ob.IQ2_S], ids=lambda t: ob.NAMES[t])
def test_prefill_activation_outliers_beyond_f16_range(t, backend, oracle):
    """|x| > 65504 must not become inf in the f16 MFMA path (the reference quantizes activations to int8 per block and cannot
    overflow): rows are scaled by a power of two before the f16 rounding and the epilogue undoes it (convert.cuh)."""
    m, k, n = 256, 1024, 48
    w = make_weights(t, m, k, 77, oracle); x = activations(n, k, 78)
    x[3, 17] = 3.0e5; x[5] *= 1.0e6; x[7, ::64] = -9.0e4; x[9, 100] = 7.0e4        # single outliers, a whole large row, just above 65504
    got = backend.mul_mat(t, dev(w), dev(x)).cpu().numpy()
    assert np.all(np.isfinite(got))
    c64, sum_abs = oracle.mul_mat_f64(t, w, x)
    # f16 rounding of the (scaled) activations: <= 2^-11 relative per element => <= 4.9e-4 of sum|w*x|, plus the weight rounding
    # Q4_K: the packed-f16 de-quantization rounds d*sc and dmin*m to f16 before combining them -- exactly what the reference's own prompt
    # path does (iqk_convert_q4_k_q8_1_r8 stores both as fp16: iqk_gemm_kquants.cpp:2241-2249) -- so a weight that is the small difference
    # of two large terms carries their rounding; when ONE activation dominates a row the error shows relative to sum|w*x| of that row.
    assert np.max(np.abs(got - c64) / sum_abs) < (1e-2 if t == ob.Q4_K else TOL_FP_ACCUM)
    assert nmse(got, c64) < 1e-6
    # rows without large values are converted exactly as before (scale 1): same rows run alone agree to summation order
    alone = backend.mul_mat(t, dev(w), dev(x[10:26])).cpu().numpy()
    assert np.allclose(got[10:26], alone, rtol=0, atol=2e-6 * np.abs(alone).max())


@pytest.mark.parametrize("t", [ob.Q4_K, ob.Q6_K, ob.Q5_K, ob.IQ4_NL], ids=lambda t: ob.NAMES[t])
def test_exact_f16_prefill_mode_keeps_the_north_star_bar_on_outlier_rows(t, backend, oracle):
    """CDNA4_PREFILL_MFMA_F16_EXACT, the run-time parity switch of the prompt path: every weight enters the tile as its L0 value rounded ONCE to f16 (the default Q4_K /
    Q6_K tiles round the block scale to f16 before the product, like the reference's own prompt repack) -- the north-star bar of 1e-3 of sum|w*x| then holds for Q4_K
    too on rows that a single huge activation dominates, where the default tile is only held to 1e-2 (test above)."""
    m, k, n = 256, 1024, 48
    w = make_weights(t, m, k, 77, oracle); x = activations(n, k, 78)
    x[3, 17] = 3.0e5; x[5] *= 1.0e6; x[7, ::64] = -9.0e4; x[9, 100] = 7.0e4
    backend.set_prefill_mode(2)
    try:
        got = backend.mul_mat(t, dev(w), dev(x)).cpu().numpy()
        wu = make_weights(t, 128, k, 81, oracle); wg = make_weights(t, 128, k, 82, oracle)
        fused = backend.fused_up_gate(t, dev(wu), dev(wg), dev(x), op=6).cpu().numpy()
    finally:
        backend.set_prefill_mode(0)
    c64, sum_abs = oracle.mul_mat_f64(t, w, x)
    assert np.all(np.isfinite(got)) and np.max(np.abs(got - c64) / sum_abs) < TOL_FP_ACCUM
    u, _ = oracle.mul_mat_f64(t, wu, x); g, _ = oracle.mul_mat_f64(t, wg, x)
    assert np.all(np.isfinite(fused)) and nmse(fused, np.maximum(g, 0) * u) < 1e-6


def test_fused_up_gate_prefill_outliers(backend, oracle):
    t, m, k, n = ob.Q4_K, 128, 1024, 40
    wu = make_weights(t, m, k, 81, oracle); wg = make_weights(t, m, k, 82, oracle); x = activations(n, k, 83)
    x[2] *= 3.0e5
    got = backend.fused_up_gate(t, dev(wu), dev(wg), dev(x), op=6).cpu().numpy()           # RELU: no exp of a huge number in the check
    u, _ = oracle.mul_mat_f64(t, wu, x); g, _ = oracle.mul_mat_f64(t, wg, x)
    want = np.maximum(g, 0) * u
    assert np.all(np.isfinite(got)) and nmse(got, want) < 1e-5


def test_mul_mat_id_more_than_65535_pairs_decode_path(backend, oracle):
    """id-indexed GEMV path with more (token, slot) pairs than grid.y allows: launches are chunked with a pair offset."""
    t, m, k, n_expert, n_used, n_tok = ob.Q4_K, 32, 256, 4, 2, 33000                    # 66000 pairs
    ws = np.stack([make_weights(t, m, k, 400 + e, oracle) for e in range(n_expert)])
    x = activations(n_tok, k, 41).reshape(n_tok, 1, k)
    ids = np.random.default_rng(6).integers(0, n_expert, size=(n_tok, n_used)).astype(np.int32)
    backend.set_prefill_mode(1)                      # CDNA4_PREFILL_INT8_DOT: forces the id-GEMV path for any batch
    try:
        got = backend.mul_mat_id(t, dev(ws), dev(x), dev(ids)).cpu().numpy()
    finally:
        backend.set_prefill_mode(0)
    sel = np.r_[0:50, 32760:32790, n_tok - 40:n_tok]                                   # both sides of the 65535-pair boundary
    want = oracle.mul_mat_id(t, ws, x[sel], ids[sel])
    assert np.allclose(got[sel], want, rtol=2e-5, atol=2e-6 * np.abs(want).max())


def test_mul_mat_id_large_ubatch_grouped(backend, oracle):
    """-ub 8192-style batch with 8 used experts: 65536+ sorted rows through the grouped MFMA path (rows on grid.x now)."""
    t, m, k, n_expert, n_used, n_tok = ob.Q4_K, 128, 256, 8, 8, 8300
    ws = np.stack([make_weights(t, m, k, 500 + e, oracle) for e in range(n_expert)])
    x = activations(n_tok, k, 43).reshape(n_tok, 1, k)
    ids = np.stack([np.random.default_rng(100 + i).permutation(n_expert) for i in range(n_tok)]).astype(np.int32)
    got = backend.mul_mat_id(t, dev(ws), dev(x), dev(ids)).cpu().numpy()
    sel = np.r_[0:16, 4000:4016, n_tok - 16:n_tok]
    want = oracle.mul_mat_id(t, ws, x[sel], ids[sel])
    assert nmse(got[sel], want) < NMSE_VS_CPU


def test_mul_mat_id_many_invalid_ids_grouped(backend, oracle):
    """a third of the ids invalid in a prompt-size batch: their rows are zero, the rest unaffected (ggml.c:18178-18187)."""
    t, m, k, n_expert, n_used, n_tok = ob.Q4_K, 160, 512, 8, 2, 300
    ws = np.stack([make_weights(t, m, k, 600 + e, oracle) for e in range(n_expert)])
    x = activations(n_tok * n_used, k, 45).reshape(n_tok, n_used, k)
    rng = np.random.default_rng(8); ids = rng.integers(0, n_expert, size=(n_tok, n_used)).astype(np.int32)
    bad = rng.random((n_tok, n_used)) < 0.33; ids[bad] = np.where(rng.random(bad.sum()) < 0.5, -1, n_expert + 3)
    out = torch.full((n_tok, n_used, m), float(SENTINEL), device="cuda")
    got = backend.mul_mat_id(t, dev(ws), dev(x), dev(ids), out=out).cpu().numpy()
    want = oracle.mul_mat_id(t, ws, x, np.where(bad, -1, ids).astype(np.int32))
    assert np.all(got[bad] == 0)
    assert nmse(got[~bad], want[~bad]) < NMSE_VS_CPU


@pytest.mark.parametrize("n", [1, 3, 8, 9, 40, 130])
@pytest.mark.parametrize("t", [ob.Q4_K, ob.Q6_K, ob.IQ3_S], ids=lambda t: ob.NAMES[t])
def test_no_overrun_around_output(t, n, backend, oracle):
    """sentinel rows / columns around the result (test-backend-ops.cpp:440-476): a ragged M inside a wider-strided C, guard rows before
    and after -- kernels that rely on clamped re-reads and padded rows must never write outside [n, m]."""
    m, k, stride = 203, 1024, 256
    w = make_weights(t, m, k, 900 + t, oracle); x = activations(n, k, 91)
    buf = torch.full((n + 2, stride), float(SENTINEL), device="cuda")
    out = buf[1:n + 1, :m]
    backend.mul_mat(t, dev(w), dev(x), out=out)
    res = buf.cpu().numpy()
    assert np.all(res[0] == SENTINEL) and np.all(res[n + 1] == SENTINEL) and np.all(res[1:n + 1, m:] == SENTINEL)
    assert nmse(res[1:n + 1, :m], oracle.mul_mat(t, w, x)) < NMSE_VS_CPU


@pytest.mark.parametrize("n", [1, 4, 64])
def test_no_overrun_fused_and_moe(n, backend, oracle):
    t, m, k = ob.Q4_K, 203, 512
    wu = make_weights(t, m, k, 31, oracle); wg = make_weights(t, m, k, 32, oracle); x = activations(n, k, 33)
    buf = torch.full((n + 2, 256), float(SENTINEL), device="cuda")
    backend.fused_up_gate(t, dev(wu), dev(wg), dev(x), out=buf[1:n + 1, :m])
    res = buf.cpu().numpy()
    assert np.all(res[0] == SENTINEL) and np.all(res[n + 1] == SENTINEL) and np.all(res[1:n + 1, m:] == SENTINEL)
    n_expert, n_used = 4, 2
    ws = np.stack([make_weights(t, m, k, 700 + e, oracle) for e in range(n_expert)])
    ids = np.random.default_rng(9).integers(0, n_expert, size=(n, n_used)).astype(np.int32)
    buf3 = torch.full((n + 2, n_used, 256), float(SENTINEL), device="cuda")
    backend.mul_mat_id(t, dev(ws), dev(x.reshape(n, 1, k)), dev(ids), out=buf3[1:n + 1, :, :m])
    r3 = buf3.cpu().numpy()
    assert np.all(r3[0] == SENTINEL) and np.all(r3[n + 1] == SENTINEL) and np.all(r3[1:n + 1, :, m:] == SENTINEL)
    assert nmse(r3[1:n + 1, :, :m], oracle.mul_mat_id(t, ws, x.reshape(n, 1, k), ids)) < NMSE_VS_CPU
