"""-m gpu: the trellis weight types IQ1_KT / IQ2_KT / IQ3_KT / IQ4_KT -- decode units in csrc/gemv.cuh (Unit<T_IQx_KT>), element decoder in csrc/convert.cuh,
prompt batches through the f16 route.  The oracle reproduces the reference's kernels (mul_mat_iqX_kt_q8_2_x4_T) to the last bit (tests/test_oracle_vs_ref.py); the device
computes the same int32 block sums and the same scale products and differs only in the order of the f32 additions."""
import numpy as np
import pytest
import torch

from common import TOL_FP_ACCUM, activations, gaussian_weights_f32, random_block_bytes
from oracle import bindings as ob
from test_gpu_parity import dev

pytestmark = pytest.mark.gpu
IDS = [ob.NAMES[t] for t in ob.KT_TYPES]
TOL_DIRECT = 2e-6          # |device - oracle| / sum|w x| for the direct int8 kernels (the bar of the other weight types)


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def rel_err(got, t, w, x, oracle):
    k = x.shape[1]
    xq = oracle.dequantize_activations(ob.Q8_2_X4, oracle.quantize_activations(ob.Q8_2_X4, x), k)
    _, sum_abs = oracle.mul_mat_f64(t, w, xq)
    return np.max(np.abs(got.astype(np.float64) - oracle.mul_mat(t, w, x)) / np.maximum(sum_abs, 1e-30))


@pytest.mark.parametrize("t", ob.KT_TYPES, ids=IDS)
def test_kt_dequant_bit_exact(t, backend, oracle):
    """to_float (and get_rows): the scalar dequantize_row_iqX_kt -- without the mat-mul kernels' 1.05 / 1.01"""
    for m, k in ((16, 1024), (5, 256)):
        w = random_block_bytes(t, m, k, 3)
        got = backend.dequantize(t, dev(w), k).cpu().numpy()
        assert np.array_equal(bits(got), bits(oracle.dequantize(t, w, k)))
        h = backend.dequantize(t, dev(w), k, dtype=torch.float16).cpu().numpy()
        assert np.array_equal(h, oracle.dequantize(t, w, k).astype(np.float16))


@pytest.mark.parametrize("t", ob.KT_TYPES, ids=IDS)
@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 8])
@pytest.mark.parametrize("m,k", [(64, 1024), (33, 512), (256, 4096), (48, 14336)])
def test_kt_decode_vs_oracle(t, n, m, k, backend, oracle):
    """1 ... 8 columns on the decode units; 512 / 1024 = rows shorter than a wave's 64 units, 14336 = a row of several K-slices, 33 rows = a ragged last group"""
    w = random_block_bytes(t, m, k, 11 + t); x = activations(n, k, 12 + n, outliers=(n == 2))
    got = backend.mul_mat(t, dev(w), dev(x)).cpu().numpy()
    assert rel_err(got, t, w, x, oracle) < TOL_DIRECT


@pytest.mark.parametrize("t", ob.KT_TYPES, ids=IDS)
def test_kt_golden(t, backend):
    """the committed reference outputs (tests/golden/iqk_golden_kt.npz): de-quantization bit for bit, the decode mat-mul to the f32 summation order"""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "iqk_golden_kt.npz")); k = int(g["meta"][1])
    for wk, dk, mk in (("w_%d", "deq_%d", "mm_%d_n%d"), ("wb_%d", "deqb_%d", "mmb_%d_n%d")):
        w = g[wk % t]
        assert np.array_equal(bits(backend.dequantize(t, dev(w), k).cpu().numpy()), bits(g[dk % t]))
        for n in (1, 2, 8):
            got = backend.mul_mat(t, dev(w), dev(g["x"][:n])).cpu().numpy(); want = g[mk % (t, n)]
            assert np.max(np.abs(got - want)) <= 1e-5 * np.max(np.abs(want)) + 1e-30, (ob.NAMES[t], n)


@pytest.mark.parametrize("t", ob.KT_TYPES, ids=IDS)
def test_kt_fused_up_gate_and_moe_decode(t, backend, oracle):
    """the fused up*gate launch and the id-indexed MoE launch instantiate the same units"""
    m, k = 96, 1024
    up = random_block_bytes(t, m, k, 31); gate = random_block_bytes(t, m, k, 32); x = activations(1, k, 33)
    got = backend.fused_up_gate(t, dev(up), dev(gate), dev(x)).cpu().numpy()
    u = oracle.mul_mat(t, up, x).astype(np.float64); g = oracle.mul_mat(t, gate, x).astype(np.float64)
    want = u * (g / (1 + np.exp(-g)))
    assert np.max(np.abs(got - want)) <= 2e-5 * np.max(np.abs(want))
    n_expert, n_used = 4, 2
    ws = np.stack([random_block_bytes(t, m, k, 40 + e) for e in range(n_expert)])
    ids = np.array([[2, 0]], np.int32)
    out = backend.mul_mat_id(t, dev(ws), dev(x.reshape(1, 1, k)), dev(ids)).cpu().numpy()
    for s in range(n_used):
        want = oracle.mul_mat(t, ws[ids[0, s]], x)[0]
        assert np.max(np.abs(out[0, s] - want)) <= 2e-5 * np.max(np.abs(want))


@pytest.mark.parametrize("t", ob.KT_TYPES, ids=IDS)
@pytest.mark.parametrize("m,k,n", [(256, 1024, 48), (130, 4096, 512), (64, 2304, 40)])
def test_kt_prompt_batches(t, m, k, n, backend, oracle):
    """N > 8: rows de-quantized to f16 WITH the mat-mul factor (1.05 / 1.01 on the row scale, as every mat-mul kernel of the reference), then the f16 MFMA GEMM;
    K = 2304 (a multiple of 256 but not of ... 128 x odd) exercises the plain tiling on an odd tile count"""
    w = random_block_bytes(t, m, k, 21 + t); x = activations(n, k, 22)
    got = backend.mul_mat(t, dev(w), dev(x)).cpu().numpy()
    c64, sum_abs = oracle.mul_mat_f64(t, w, x.astype(np.float16).astype(np.float32))
    assert np.max(np.abs(got - c64) / np.maximum(sum_abs, 1e-30)) < TOL_FP_ACCUM


def test_kt_real_quantizer_weights_vs_reference(backend, oracle, ref):
    """weights from the reference's trellis quantizer (a few rows: the search is slow), decode results against the REAL reference kernels"""
    for t in ob.KT_TYPES:
        w = ref.quantize(t, gaussian_weights_f32(8, 1024, 5)); x = activations(4, 1024, 6)
        got = backend.mul_mat(t, dev(w), dev(x)).cpu().numpy(); want = ref.mul_mat(t, w, x)
        assert np.max(np.abs(got - want)) <= 1e-5 * np.max(np.abs(want)), ob.NAMES[t]


@pytest.mark.parametrize("t", [ob.IQ2_KT, ob.IQ4_KT], ids=["iq2_kt", "iq4_kt"])
@pytest.mark.parametrize("chunk_mb", [None, "0"], ids=["one-chunk", "expert-by-expert"])
def test_kt_moe_prompt_grouped_through_f16(t, chunk_mb, backend, oracle, monkeypatch):
    """MUL_MAT_ID / MOE_FUSED_UP_GATE prompt batches of a decode-only type: pairs grouped by expert on the device, experts de-quantized to f16 a chunk at a time, the f16
    instance of the grouped GEMM (CDNA4_F16_MOE_CHUNK_MB=0: one expert per chunk -- every chunk's launch skips the other experts' tiles)"""
    if chunk_mb is not None:
        monkeypatch.setenv("CDNA4_F16_MOE_CHUNK_MB", chunk_mb)
    m, k, n_expert, n_used, n_tok = 160, 1024, 4, 2, 72
    ws = np.stack([random_block_bytes(t, m, k, 50 + e) for e in range(n_expert)]); wg = np.stack([random_block_bytes(t, m, k, 60 + e) for e in range(n_expert)])
    x = activations(n_tok, k, 51); rng = np.random.default_rng(52)
    ids = np.stack([rng.permutation(n_expert)[:n_used] for _ in range(n_tok)]).astype(np.int32); ids[5, 1] = -1       # (an invalid id: a zero row)
    out = backend.mul_mat_id(t, dev(ws), dev(x.reshape(n_tok, 1, k)), dev(ids)).cpu().numpy()
    fused = backend.moe_fused_up_gate(t, dev(ws), dev(wg), dev(x.reshape(n_tok, 1, k)), dev(ids)).cpu().numpy()
    xh = x.astype(np.float16).astype(np.float32)
    up = [oracle.mul_mat_f64(t, ws[e], xh) for e in range(n_expert)]; gt = [oracle.mul_mat_f64(t, wg[e], xh) for e in range(n_expert)]
    for tok in range(n_tok):
        for s in range(n_used):
            e = ids[tok, s]
            if e < 0:
                assert not out[tok, s].any() and not fused[tok, s].any(); continue
            c64, sa = up[e][0][tok], up[e][1][tok]
            assert np.max(np.abs(out[tok, s] - c64) / np.maximum(sa, 1e-30)) < TOL_FP_ACCUM
            g64 = gt[e][0][tok]; want = c64 * (g64 / (1 + np.exp(-g64)))
            assert np.max(np.abs(fused[tok, s] - want)) <= 2e-3 * np.max(np.abs(want)), (tok, s)
