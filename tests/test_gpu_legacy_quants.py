"""-m gpu: the legacy 32-block types Q4_0 and Q8_0 (iqk_gemm_legacy_quants.cpp) on every kernel family -- L0 dequant bit-exact,
decode GEMV (1 / 3 / 8 columns) in the CPU path's int8 arithmetic, the MFMA prompt GEMM, fused up*gate, MUL_MAT_ID, and through the shim
against the reference CPU backend.  Same bars as tests/test_gpu_parity.py / test_gpu_prefill.py."""
import os

import numpy as np
import pytest
import torch

from common import NMSE_VS_CPU, activations, gaussian_weights_f32, make_weights, nmse
from oracle import bindings as ob
from test_gpu_parity import check_mul_mat, dev
from test_oracle_vs_ref import SATURATING

pytestmark = pytest.mark.gpu
T = ob.LEGACY_TYPES


@pytest.mark.parametrize("t", T, ids=lambda t: ob.NAMES[t])
def test_dequantize_bit_exact(t, backend, oracle):
    m, k = 64, 4096
    w = make_weights(t, m, k, 11 + t, oracle)
    got = backend.dequantize(t, dev(w), k).cpu().numpy()
    assert np.array_equal(got.view(np.uint32), oracle.dequantize(t, w, k).view(np.uint32))
    h = backend.dequantize(t, dev(w), k, dtype=torch.float16).cpu().numpy()
    assert np.array_equal(h, oracle.dequantize(t, w, k).astype(np.float16))


@pytest.mark.parametrize("t", T, ids=lambda t: ob.NAMES[t])
@pytest.mark.parametrize("m,k", [(512, 4096), (96, 14336), (257, 1024), (130, 3072), (64, 256), (33, 2048), (128, 8192)])
def test_gemv_shapes(t, m, k, backend, oracle):
    w = make_weights(t, m, k, 100 + t, oracle)
    for n, seed in ((1, 1), (3, 2), (8, 3)):
        check_mul_mat(backend, oracle, t, w, activations(n, k, seed, outliers=(n == 3)), int8_path=True)


@pytest.mark.parametrize("t", T, ids=lambda t: ob.NAMES[t])
def test_gemv_real_quantizer_weights_vs_reference(t, backend, oracle, ref):
    """(IQ4_XS: activations with one large value per 256 -- the reference's AVX-512 kernel saturates int16 pair sums on full-range int8 activations,
    tests/test_oracle_vs_ref.py; the device computes the exact sums)"""
    m, k = 256, 4096
    w = ref.quantize(t, gaussian_weights_f32(m, k, 5)); x = activations(2, k, 6, outliers=(t in SATURATING))
    got = backend.mul_mat(t, dev(w), dev(x)).cpu().numpy()
    assert nmse(got, ref.mul_mat(t, w, x)) < 1e-10


@pytest.mark.parametrize("t", T, ids=lambda t: ob.NAMES[t])
@pytest.mark.parametrize("m,k,n", [(256, 1024, 32), (130, 2048, 9), (384, 4096, 100), (128, 512, 512), (96, 14336, 40), (700, 1024, 300)])
def test_mfma_gemm_shapes(t, m, k, n, backend, oracle):
    w = make_weights(t, m, k, 500 + t, oracle)
    check_mul_mat(backend, oracle, t, w, activations(n, k, n, outliers=(n == 40)), int8_path=False)


@pytest.mark.parametrize("t", [ob.Q5_0, ob.IQ4_KS, ob.IQ3_XXS, ob.Q4_K], ids=lambda t: ob.NAMES[t])
def test_f16_route_of_types_without_a_tile(t, backend, oracle, monkeypatch):
    """the generic prompt route (chunked de-quantization to f16 + the f16 instance of the MFMA GEMM), forced: every type has a tile of its own at present.
    Small chunks so that the row-chunk loop and the fused up*gate form are covered."""
    monkeypatch.setenv("CDNA4_FORCE_F16_ROUTE", "1"); monkeypatch.setenv("CDNA4_F16_CHUNK_MB", "1")
    m, k, n = 700, 1024, 40
    w = make_weights(t, m, k, 600 + t, oracle)
    check_mul_mat(backend, oracle, t, w, activations(n, k, 9), int8_path=False)
    wu = make_weights(t, 192, 2048, 21, oracle); wg = make_weights(t, 192, 2048, 22, oracle); x = activations(48, 2048, 23)
    got = backend.fused_up_gate(t, dev(wu), dev(wg), dev(x), op=10).cpu().numpy()
    xh = x.astype(np.float16).astype(np.float32)
    u, _ = oracle.mul_mat_f64(t, wu, xh); g, _ = oracle.mul_mat_f64(t, wg, xh)
    assert nmse(got, (g * 0.5 * (1 + np.tanh(0.5 * g))) * u) < 1e-6


@pytest.mark.parametrize("t", T, ids=lambda t: ob.NAMES[t])
@pytest.mark.parametrize("n", [64, 512])
def test_mfma_gemm_model_shape_vs_reference(t, n, backend, ref):
    from test_gpu_prefill import tiled_real_weights
    m, k = 4096, 4096
    w = tiled_real_weights(ref, t, m, k, 40 + t); x = activations(n, k, 41)
    got = backend.mul_mat(t, dev(w), dev(x)).cpu().numpy()
    assert nmse(got, ref.mul_mat(t, w, x, nth=16)) < NMSE_VS_CPU


@pytest.mark.parametrize("t", T, ids=lambda t: ob.NAMES[t])
@pytest.mark.parametrize("n", [2, 48])
def test_fused_up_gate(t, n, backend, oracle):
    m, k = 192, 2048
    wu = make_weights(t, m, k, 21, oracle); wg = make_weights(t, m, k, 22, oracle); x = activations(n, k, 23)
    got = backend.fused_up_gate(t, dev(wu), dev(wg), dev(x), op=10).cpu().numpy()
    if n <= 8:
        want = oracle.fused_up_gate(t, 10, wu, wg, x)
        assert np.allclose(got, want, rtol=2e-5, atol=2e-6 * np.abs(want).max())
    else:
        xh = x.astype(np.float16).astype(np.float32)
        u, _ = oracle.mul_mat_f64(t, wu, xh); g, _ = oracle.mul_mat_f64(t, wg, xh)
        assert nmse(got, (g * 0.5 * (1 + np.tanh(0.5 * g))) * u) < 1e-6


@pytest.mark.parametrize("t", T, ids=lambda t: ob.NAMES[t])
@pytest.mark.parametrize("n_tok", [1, 5, 96])
def test_mul_mat_id(t, n_tok, backend, oracle):
    m, k, n_expert, n_used = 200, 512, 8, 2
    ws = np.stack([make_weights(t, m, k, 700 + e, oracle) for e in range(n_expert)])
    x = activations(n_tok, k, 31).reshape(n_tok, 1, k)
    ids = np.random.default_rng(5).integers(0, n_expert, size=(n_tok, n_used)).astype(np.int32)
    got = backend.mul_mat_id(t, dev(ws), dev(x), dev(ids)).cpu().numpy()
    cpu = oracle.mul_mat_id(t, ws, x, ids)
    assert nmse(got, cpu) < NMSE_VS_CPU
    if n_tok <= 8:
        assert nmse(got, cpu) < 1e-10


@pytest.fixture(scope="module")
def host():
    from ggml_host import SHIM, GgmlHost
    if ob.ref_path() is None or not os.path.exists(SHIM):
        pytest.skip("needs oracle/_ref (reference libggml) and the prebuilt backend shim")
    h = GgmlHost()
    gpu = h.shim.ggml_backend_cuda_init(0, None, None); cpu = h.g.ggml_backend_cpu_init(); h.g.ggml_backend_cpu_set_n_threads(cpu, 8)
    yield h, gpu, cpu
    h.g.ggml_backend_free(gpu); h.g.ggml_backend_free(cpu)


@pytest.mark.parametrize("t", T, ids=lambda t: ob.NAMES[t])
@pytest.mark.parametrize("m,n,k", [(512, 1, 4096), (256, 64, 1024)])
def test_mul_mat_through_the_shim_vs_cpu_backend(t, m, n, k, host):
    h, gpu, cpu = host
    w = h.ref.quantize(t, gaussian_weights_f32(m, k, 1)); x = activations(n, k, 2)

    def build(ctx):
        a = h.g.ggml_new_tensor_2d(ctx, t, k, m); b = h.g.ggml_new_tensor_2d(ctx, 0, k, n)
        return {"a": a, "b": b}, h.g.ggml_mul_mat(ctx, a, b)
    got, sup = h.run(gpu, build, {"a": w, "b": x}); want, _ = h.run(cpu, build, {"a": w, "b": x})
    assert sup and nmse(got, want) < (NMSE_VS_CPU if t not in SATURATING else 2e-2)       # (the CPU kernel's int16 saturation, see above)
    if n <= 8 and t not in SATURATING:
        assert nmse(got, want) < 1e-10


@pytest.mark.parametrize("t", T, ids=lambda t: ob.NAMES[t])
def test_get_rows_through_the_shim(t, host):
    h, gpu, cpu = host
    n_embd, n_vocab, n_tok = 512, 96, 13
    w = h.ref.quantize(t, np.random.default_rng(10).standard_normal((n_vocab, n_embd)).astype(np.float32))
    ids = np.random.default_rng(11).integers(0, n_vocab, n_tok).astype(np.int32)

    def build(ctx):
        tw = h.g.ggml_new_tensor_2d(ctx, t, n_embd, n_vocab); ti = h.g.ggml_new_tensor_1d(ctx, 26, n_tok)
        return {"w": tw, "i": ti}, h.g.ggml_get_rows(ctx, tw, ti)
    got, sup = h.run(gpu, build, {"w": w, "i": ids}); want, _ = h.run(cpu, build, {"w": w, "i": ids})
    assert sup
    np.testing.assert_array_equal(got, want)
