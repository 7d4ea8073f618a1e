"""-m gpu: BitNet weights (IQ1_BN / IQ2_BN) -- csrc/gemv_bitnet.hip.  The oracle is pinned BIT-exactly against the reference kernels
(mul_mat_iq1bn_q8_K64 / mul_mat_iq2bn_q8_K64, quantize_row_q8_K64: tests/test_oracle_vs_ref.py); the device restates the same integers and the same five float
operations, so decode results are compared bit for bit.  Prompt batches go through the f16 route (de-quantize + f16 MFMA GEMM): the usual bar against the fp64 accumulate."""
import numpy as np
import pytest
import torch

from common import TOL_FP_ACCUM, activations, gaussian_weights_f32, random_block_bytes
from oracle import bindings as ob
from test_gpu_parity import dev

pytestmark = pytest.mark.gpu
IDS = ["iq1_bn", "iq2_bn"]


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def bn_weights(t, m, k, seed):
    """random ternary digits behind a row scale of realistic size (the quantizer stores max|w| of the row: 1e-2 .. 1e-1 for real layers; random_block_bytes' 2e-4 is an f16
    SUBNORMAL and would make the f16 prompt route look lossy)"""
    w = random_block_bytes(t, m, k, seed); rng = np.random.default_rng(seed + 1000)
    if t == ob.IQ2_BN:
        w[:, :4] = rng.uniform(0.01, 0.1, size=(m, 1)).astype(np.float32).view(np.uint8)
    else:
        w[:, :2] = rng.uniform(0.01, 0.1, size=(m, 1)).astype(np.float16).view(np.uint8)
    return w


@pytest.mark.parametrize("k", [512, 1024, 640, 4096, 14336])
def test_q8_k64_quantizer_byte_exact(k, backend, oracle):
    for seed, outliers in ((1, False), (2, True)):
        x = activations(5, k, seed, outliers=outliers); x[3] = 0
        got = backend.quantize_activations(ob.Q8_K64, dev(x)).cpu().numpy()
        assert np.array_equal(got, oracle.quantize_activations(ob.Q8_K64, x))


@pytest.mark.parametrize("t", ob.BITNET_TYPES, ids=IDS)
def test_bitnet_dequant_bit_exact(t, backend, oracle):
    for m, k in ((16, 1024), (5, 576)):
        w = random_block_bytes(t, m, k, 3)
        got = backend.dequantize(t, dev(w), k).cpu().numpy()
        assert np.array_equal(bits(got), bits(oracle.dequantize(t, w, k)))
        h = backend.dequantize(t, dev(w), k, dtype=torch.float16).cpu().numpy()
        assert np.array_equal(h, oracle.dequantize(t, w, k).astype(np.float16))


@pytest.mark.parametrize("t", ob.BITNET_TYPES, ids=IDS)
@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 8])
@pytest.mark.parametrize("m,k", [(64, 1024), (33, 576), (256, 4096), (40, 8192)])
def test_bitnet_decode_bit_exact(t, n, m, k, backend, oracle):
    """exact per-class int32 sums over the row, one fma per class, the fixed 4-way sum, the row scale: identical to the oracle (= the reference kernels) to the last bit;
    576 = an odd number of 64-blocks, 33 rows = a ragged last workgroup"""
    w = random_block_bytes(t, m, k, 11 + t); x = activations(n, k, 12 + n, outliers=(n == 2))
    got = backend.mul_mat(t, dev(w), dev(x)).cpu().numpy()
    assert np.array_equal(bits(got), bits(oracle.mul_mat(t, w, x)))
    # pre-quantized activations (what a caller that already holds the Q8_K64 rows hands over) give the same bits
    xq = oracle.quantize_activations(ob.Q8_K64, x)
    got_q = backend.mul_mat(t, dev(w), dev(xq), x_type=ob.Q8_K64).cpu().numpy()
    assert np.array_equal(bits(got_q), bits(got))


@pytest.mark.parametrize("t", ob.BITNET_TYPES, ids=IDS)
@pytest.mark.parametrize("m,k,n", [(256, 1024, 48), (130, 4096, 512), (64, 576, 40)])
def test_bitnet_prompt_batches(t, m, k, n, backend, oracle):
    """N > 8: de-quantized to f16 (row scale x {-1, 0, 1}: exact in f16 up to the scale's rounding) and the f16 MFMA GEMM; K = 576 (not a multiple of 128) runs
    zero-padded to 640"""
    w = bn_weights(t, m, k, 21 + t); x = activations(n, k, 22)
    got = backend.mul_mat(t, dev(w), dev(x)).cpu().numpy()
    c64, sum_abs = oracle.mul_mat_f64(t, w, x.astype(np.float16).astype(np.float32))
    assert np.max(np.abs(got - c64) / np.maximum(sum_abs, 1e-30)) < TOL_FP_ACCUM


def test_bitnet_real_quantizer_weights_vs_reference(backend, oracle, ref):
    """weights from the reference quantizer, decode results against the REAL reference kernels: bit for bit"""
    for t in ob.BITNET_TYPES:
        w = ref.quantize(t, gaussian_weights_f32(96, 2048, 5)); x = activations(4, 2048, 6)
        got = backend.mul_mat(t, dev(w), dev(x)).cpu().numpy()
        assert np.array_equal(bits(got), bits(ref.mul_mat(t, w, x)))


def test_bitnet_unsupported_forms_say_so(backend, oracle):
    from ik_llama_cpp_amd import Cdna4Error
    t = ob.IQ2_BN; w = dev(random_block_bytes(t, 64, 1024, 1)); x = dev(activations(1, 1024, 2))
    with pytest.raises(Cdna4Error) as ei:
        backend.fused_up_gate(t, w, w, x)
    assert ei.value.code == -1
