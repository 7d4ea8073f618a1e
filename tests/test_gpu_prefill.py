"""-m gpu: prompt-processing path (N > 8): dequant -> f16 MFMA GEMM (north_star) and its int8-dot alternative.

Parity bars: L1 |out - fp64 accumulate(L0 weights x f16-rounded activations)| <= 1e-3 * sum|w*x| ; L2 NMSE vs the
CPU-arithmetic result <= 5e-4 (the reference's own MUL_MAT tolerance, tests/test-backend-ops.cpp:979-981)."""
import numpy as np
import pytest
import torch

from common import np_up_gate_combine, NMSE_VS_CPU, TOL_FP_ACCUM, activations, make_weights, nmse  # noqa: F401
from oracle import bindings as ob
from test_gpu_parity import check_mul_mat, dev

pytestmark = pytest.mark.gpu
MFMA_TYPES = [ob.Q4_K, ob.Q5_K, ob.Q6_K, ob.IQ4_NL, ob.IQ2_S, ob.IQ3_S]


@pytest.mark.parametrize("t", MFMA_TYPES, ids=lambda t: ob.NAMES[t])
@pytest.mark.parametrize("m,k,n", [(256, 1024, 32), (130, 2048, 9), (384, 4096, 100), (128, 512, 512), (96, 14336, 40), (700, 1024, 300),
                                   (192, 8192, 64), (160, 3584, 48)])          # last two: Llama-3-70B TP=8 slices (K = 8192, K-slice 3584)
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


def tiled_real_weights(ref, t, m, k, seed, distinct=256):
    """[m, row_size] of REAL quantizer output: `distinct` independently quantized rows repeated down the matrix (the sub-4-bit quantizers take
    minutes for 58 M weights; every row is still genuine quantizer output and the kernels see the full-size shape)"""
    from common import gaussian_weights_f32
    base = ref.quantize(t, gaussian_weights_f32(distinct, k, seed))
    return np.ascontiguousarray(np.tile(base, ((m + distinct - 1) // distinct, 1))[:m])


@pytest.mark.parametrize("t", MFMA_TYPES, ids=lambda t: ob.NAMES[t])
@pytest.mark.parametrize("m,k", [(14336, 4096), (4096, 14336)], ids=["up", "down"])
@pytest.mark.parametrize("n", [64, 512])
def test_mfma_gemm_model_shapes_vs_reference(t, m, k, n, backend, ref):
    """all six types on the Llama-3-8B FFN shapes at prompt sizes against the REAL reference CPU kernels (iqk_mul_mat; for N >= 32 that is its
    repack-to-Q8 path, SURVEY F2 -- itself lossy, hence the reference's own NMSE bar), 16 host threads"""
    w = tiled_real_weights(ref, t, m, k, 40 + t); x = activations(n, k, 41)
    got = backend.mul_mat(t, dev(w), dev(x)).cpu().numpy()
    want = ref.mul_mat(t, w, x, nth=16)
    assert nmse(got, want) < NMSE_VS_CPU, nmse(got, want)


@pytest.mark.parametrize("t", [ob.Q4_K, ob.Q6_K], ids=lambda t: ob.NAMES[t])
def test_fused_up_gate_prefill(t, backend, oracle):
    m, k, n = 200, 1024, 48
    wu = make_weights(t, m, k, 21, oracle); wg = make_weights(t, m, k, 22, oracle); x = activations(n, k, 23)
    got = backend.fused_up_gate(t, dev(wu), dev(wg), dev(x), op=10).cpu().numpy()
    xh = x.astype(np.float16).astype(np.float32)
    u, _ = oracle.mul_mat_f64(t, wu, xh); g, _ = oracle.mul_mat_f64(t, wg, xh)
    want = (g * 0.5 * (1 + np.tanh(0.5 * g))) * u        # silu(g) = g*sigmoid(g), overflow-free form
    assert nmse(got, want) < 1e-6


@pytest.mark.parametrize("op", [10, 14, 15], ids=["silu", "swiglu_oai", "gelu"])
def test_fused_up_gate_epilogue_prefill(op, backend, oracle):
    """biases / limit / SWIGLU_OAI in the MFMA kernel's epilogue vs fp64 on f16-rounded activations."""
    t, m, k, n = ob.Q4_K, 200, 1024, 48
    wu = make_weights(t, m, k, 51, oracle); wg = make_weights(t, m, k, 52, oracle); x = activations(n, k, 53)
    u, _ = oracle.mul_mat_f64(t, wu, x)
    x *= np.float32(2.5 / np.std(u))          # dots ~ N(0, 2.5^2): the clamps (limit, +-7) bite on a minority of the entries, not on all of them
    xh = x.astype(np.float16).astype(np.float32)
    u, _ = oracle.mul_mat_f64(t, wu, xh); g, _ = oracle.mul_mat_f64(t, wg, xh)
    rng = np.random.default_rng(54)
    ub = rng.normal(0, 1, m).astype(np.float32); gb = rng.normal(0, 1, m).astype(np.float32)
    for up_b, gate_b, limit in ((ub, gb, 0.0), (None, None, 2.0), (ub, gb, 3.0)):
        got = backend.fused_up_gate(t, dev(wu), dev(wg), dev(x), op=op, up_b=None if up_b is None else dev(up_b),
                                    gate_b=None if gate_b is None else dev(gate_b), limit=limit).cpu().numpy()
        want = np_up_gate_combine(op, u, g, up_b, gate_b, limit)
        assert nmse(got, want) < 1e-6, (op, limit)
        assert nmse(np_up_gate_combine(op, u, g), want) > 1e-3            # biases / clamps change the result materially


def test_moe_fused_up_gate_biases_grouped_prefill(backend, oracle):
    t, m, k, n_expert, n_used, n_tok = ob.Q4_K, 160, 512, 8, 2, 96
    wu = np.stack([make_weights(t, m, k, 800 + e, oracle) for e in range(n_expert)])
    wg = np.stack([make_weights(t, m, k, 900 + e, oracle) for e in range(n_expert)])
    x = activations(n_tok, k, 33).reshape(n_tok, 1, k)
    x *= np.float32(2.5 / np.std(oracle.mul_mat_f64(t, wu[0], x[:8, 0])[0]))
    rng = np.random.default_rng(6); ids = rng.integers(0, n_expert, size=(n_tok, n_used)).astype(np.int32)
    ub = rng.normal(0, 1, (n_expert, m)).astype(np.float32); gb = rng.normal(0, 1, (n_expert, m)).astype(np.float32)
    xh = x.astype(np.float16).astype(np.float32)
    for op, limit in ((10, 0.0), (14, 0.0), (10, 2.0)):
        got = backend.moe_fused_up_gate(t, dev(wu), dev(wg), dev(x), dev(ids), op=op, up_b=dev(ub), gate_b=dev(gb), limit=limit).cpu().numpy()
        want = np.zeros_like(got, dtype=np.float64)
        for tk in range(n_tok):
            for s in range(n_used):
                e = ids[tk, s]
                u, _ = oracle.mul_mat_f64(t, wu[e], xh[tk]); g, _ = oracle.mul_mat_f64(t, wg[e], xh[tk])
                want[tk, s] = np_up_gate_combine(op, u[0], g[0], ub[e], gb[e], limit)
        assert nmse(got, want) < 1e-6, (op, limit)


def test_int8_prefill_mode_matches_cpu_arithmetic(backend, oracle):
    """CDNA4_PREFILL_INT8_DOT: the CPU path's arithmetic also for N > 8 (parity mode)."""
    t = ob.Q4_K; w = make_weights(t, 256, 2048, 5, oracle); x = activations(20, 2048, 6)
    backend.set_prefill_mode(1)
    try:
        check_mul_mat(backend, oracle, t, w, x, int8_path=True)
    finally:
        backend.set_prefill_mode(0)


def test_prefill_full_size_against_oracle_rows(backend, oracle):
    """BASELINE shape (14336 x 4096 Q4_K, N = 512): a row subset spread over the matrix (first / last rows of workgroup tiles, the final partial
    tile) against the oracle's fp64 accumulate on f16-rounded activations, every token column; plus column independence across launch geometries."""
    from common import random_block_bytes
    t, m, k, n = ob.Q4_K, 14336, 4096, 512
    w = random_block_bytes(t, m, k, 1); x = activations(n, k, 2)
    full = backend.mul_mat(t, dev(w), dev(x))
    rows = np.unique(np.concatenate([np.arange(0, 4), np.arange(126, 130), np.arange(7167, 7171), np.arange(14332, 14336), np.random.default_rng(3).integers(0, m, 48)]))
    want, sum_abs = oracle.mul_mat_f64(t, w[rows], x.astype(np.float16).astype(np.float32))
    got = full[:, torch.from_numpy(rows).cuda()].cpu().numpy()
    assert np.max(np.abs(got - want) / sum_abs) < TOL_FP_ACCUM
    part = backend.mul_mat(t, dev(w), dev(x[100:164]))           # (another token tile / K split: a different f32 summation order)
    assert torch.allclose(full[100:164], part, rtol=1e-4, atol=1e-4 * float(full.abs().max()))
    assert torch.isfinite(full).all()


def _edge_rows(m, seed, extra):
    """first / last rows of the 128-, 224- and 256-row workgroup tiles, the matrix ends, and `extra` random rows"""
    fixed = [0, 1, 127, 128, 223, 224, 255, 256, m // 2 - 1, m // 2, m - 257, m - 256, m - 129, m - 128, m - 2, m - 1]
    return np.unique(np.concatenate([np.array(fixed), np.random.default_rng(seed).integers(0, m, extra)]))


@pytest.fixture
def gemm_form(backend):
    """sets the process-wide prompt-GEMM form for one test (cdna4_set_gemm_form), back to the default afterwards"""
    def set_form(f):
        backend.set_gemm_form(f)
    yield set_form
    backend.set_gemm_form(1)


def _wlds_serves(t, m, n, fused):
    """the default dispatch rule of launch_gemm_wlds on a 256-CU GPU (gemm_wlds.cuh): fused launches whose 256-token tiles give >= 1.7 workgroups per CU and fill >= 85 %
    (Q4_K / Q5_K: 95 %) of whole rounds of workgroups"""
    wgs = -(-m // (128 if fused else 256)) * -(-n // 256)
    need = 0.95 if t in (ob.Q4_K, ob.Q5_K) else 0.85
    return fused and wgs >= 1.7 * 256 and wgs / (-(-wgs // 256) * 256) * (n / (-(-n // 256) * 256)) >= need


def _ppf_serves(t, m, k, n, fused):
    """the default dispatch rule of mul_mat_ppf on a 256-CU GPU (cdna4_api.hip): the large-batch route -- weights de-quantized once into an f16 image + gemm_ppf -- from 2048 tokens
    on (Q4_K / Q5_K fused launches from 3072, and never their rows < K shapes) when the 256 x 256 tiles fill >= 0.9 of the CUs and >= 80 % of whole rounds"""
    packed = t in (ob.Q4_K, ob.Q5_K)
    wgs = -(-m // (128 if fused else 256)) * -(-n // 256)
    fill = wgs / (-(-wgs // 256) * 256) * (n / (-(-n // 256) * 256))
    return n >= ((3072 if fused else 2048) if packed else 2048) and wgs >= 0.9 * 256 and fill >= 0.8 and not (packed and not fused and m < k)


@pytest.mark.parametrize("form", [1, 2], ids=["default", "shared-tile"])
@pytest.mark.parametrize("t", [ob.Q4_K, ob.Q6_K], ids=lambda t: ob.NAMES[t])
@pytest.mark.parametrize("m,k", [(14336, 4096), (4096, 14336)], ids=["up", "down"])
@pytest.mark.parametrize("n", [2048, 4096])
def test_prefill_4k_tokens_against_oracle_rows(form, t, m, k, n, backend, oracle, gemm_form):
    """The sizes the north-star prefill target is quoted on (2048 / 4096 tokens, the Llama-3-8B FFN shapes): a row subset over every workgroup-tile edge against the oracle's
    fp64 accumulate on f16-rounded activations, EVERY token column (the reference runs the same path at any N: iqk_mul_mat.cpp:537-571); the kernel and launch geometry that
    served the shape are asserted -- form 1 (default; plain mat-muls keep the per-wave de-quantizing kernel: 8-wave 256-row workgroups = MW 2 where that grid fills whole
    rounds of the 256 CUs, K-halved workgroups at 4096 x 2048), form 2: the workgroup-shared weight tile kernel forced onto the same shapes -- and a token block in the middle of the
    batch is recomputed by a launch of a different geometry (column independence across tile / super-column order)."""
    from common import random_block_bytes
    gemm_form(form)
    w = random_block_bytes(t, m, k, 70 + t); x = activations(n, k, 71)
    xd = dev(x)
    full = backend.mul_mat(t, dev(w), xd)
    info = backend.last_launch_info()
    assert info["upgate"] == 0 and info["nt"] == 8 and info["ksplit"] == 1, info
    if form == 2:
        assert info["kernel"] == "gemm_wlds" and info["type"] == t and info["grid"] == "%dx1x1" % ((m // 256) * (n // 256)), info
    elif _ppf_serves(t, m, k, n, False):        # round 6: the large-batch route (type 1 = the f16 weight image)
        assert info["kernel"] == "gemm_ppf" and info["type"] == 1 and info["grid"] == "%dx1x1" % ((m // 256) * (n // 256)), info
    else:
        assert info["type"] == t, info
        # 14336 rows: 112 x n / 256 four-wave workgroups.  4096 rows x 4096 tokens: 16 x 16 = 256 eight-wave workgroups of 256 rows (MW 2: one full round of the CUs);
        # 4096 rows x 2048 tokens: 32 x 8 = 256 workgroups of two K-halves (KS 2)
        want_mw, want_ks = (1, 1) if m == 14336 else ((2, 1) if n == 4096 else (1, 2))
        assert info["kernel"] == "gemm_mfma" and (info["mw"], info["ks"]) == (want_mw, want_ks), info
    assert info["g"] >= 1 and (n // 256) % info["g"] == 0, info
    rows = _edge_rows(m, 72, 24 if k == 4096 else 8)
    want, sum_abs = oracle.mul_mat_f64(t, w[rows], x.astype(np.float16).astype(np.float32))
    got = full[:, torch.from_numpy(rows).cuda()].cpu().numpy()
    assert np.max(np.abs(got - want) / sum_abs) < TOL_FP_ACCUM
    lo = n // 2 - 96; part = backend.mul_mat(t, dev(w), xd[lo:lo + 192].contiguous())           # 192 tokens: 128-token tiles, another grid
    assert backend.last_launch_info() != info
    assert torch.allclose(full[lo:lo + 192], part, rtol=1e-4, atol=1e-4 * float(full.abs().max()))
    assert torch.isfinite(full).all()


@pytest.mark.parametrize("form", [0, 1], ids=["per-wave", "default"])
@pytest.mark.parametrize("t", [ob.Q4_K, ob.Q6_K], ids=lambda t: ob.NAMES[t])
@pytest.mark.parametrize("n", [2048, 4096])
def test_fused_up_gate_4k_tokens_against_oracle_rows(form, t, n, backend, oracle, gemm_form):
    """the launch bench.py times as `n4096` -- fused up*gate 2 x 14336 x 4096 -- against fp64 on a row subset, every token; SILU(gate) * up like ggml.c:18653-18722.
    form 1: gemm_wlds (112 x n / 256 workgroups of 128 rows x {up, gate}); form 0: gemm_mfma on 256-row workgroups (MW 2) at 4096 tokens, 128-row ones at 2048"""
    from common import random_block_bytes
    gemm_form(form)
    m, k = 14336, 4096
    wu = random_block_bytes(t, m, k, 80 + t); wg = random_block_bytes(t, m, k, 81 + t); x = activations(n, k, 82)
    xd = dev(x)
    full = backend.fused_up_gate(t, dev(wu), dev(wg), xd, op=10)
    info = backend.last_launch_info()
    if form == 1 and _ppf_serves(t, m, k, n, True):                     # round 6: Q6_K from 2048 tokens, Q4_K at 4096: f16 weight image + gemm_ppf
        assert info["kernel"] == "gemm_ppf" and info["type"] == 1 and info["upgate"] == 1 and info["grid"] == "%dx1x1" % (112 * (n // 256)), info
    elif form == 1 and _wlds_serves(t, m, n, True):                     # (what is left for the shared-tile kernel by default: nothing at these two sizes, kept for the rule)
        assert info["kernel"] == "gemm_wlds" and info["upgate"] == 1 and info["grid"] == "%dx1x1" % (112 * (n // 256)), info
    else:
        # 4096 tokens: 56 x 32 = 1792 eight-wave workgroups = 7 whole rounds of the 256 CUs (MW 2); 2048 tokens: 896 would be 3.5 rounds -> 128-row workgroups
        mw = 2 if n == 4096 else 1
        assert info["kernel"] == "gemm_mfma" and info["upgate"] == 1 and info["nt"] == 4 and info["mw"] == mw and info["grid"] == "%dx1x1" % ((112 // mw) * (n // 128)), info
    rows = _edge_rows(m, 83, 16)
    xh = x.astype(np.float16).astype(np.float32)
    u, _ = oracle.mul_mat_f64(t, wu[rows], xh); g, _ = oracle.mul_mat_f64(t, wg[rows], xh)
    want = (g * 0.5 * (1 + np.tanh(0.5 * g))) * u
    got = full[:, torch.from_numpy(rows).cuda()].cpu().numpy()
    assert nmse(got, want) < 1e-6
    assert np.max(np.abs(got - want)) < 2e-3 * np.max(np.abs(want))
    lo = n // 2 - 32; part = backend.fused_up_gate(t, dev(wu), dev(wg), xd[lo:lo + 64].contiguous(), op=10)
    assert backend.last_launch_info()["kernel"] == "gemm_mfma" and backend.last_launch_info()["mw"] == 1
    assert torch.allclose(full[lo:lo + 64], part, rtol=1e-4, atol=1e-4 * float(full.abs().max()))
    assert torch.isfinite(full).all()


@pytest.mark.parametrize("t", MFMA_TYPES, ids=lambda t: ob.NAMES[t])
@pytest.mark.parametrize("m,k,n,fused", [(1024, 1024, 512, False), (700, 2048, 300, False), (384, 4096, 256, True), (130, 512, 1000, True), (512, 14336, 256, False)])
def test_shared_weight_tile_kernel_is_bit_identical_to_the_per_wave_kernel(t, m, k, n, fused, backend, oracle, gemm_form):
    """gemm_wlds (the weight tile de-quantized once per workgroup into LDS) against gemm_mfma (every wave its own B fragments) on the same inputs: same products, same k-step
    pairing, same accumulation order => the SAME BITS, for all six scope types, ragged row / token counts (partial tiles on both edges), plain and fused; and both against the
    oracle.  Form 2 forces the shared-tile kernel onto grids it would not be chosen for."""
    wu = make_weights(t, m, k, 300 + t, oracle); wg = make_weights(t, m, k, 301 + t, oracle) if fused else None
    x = activations(n, k, 302, outliers=True); xd = dev(x)

    def run():
        return backend.fused_up_gate(t, dev(wu), dev(wg), xd, op=10) if fused else backend.mul_mat(t, dev(wu), xd)
    gemm_form(0); ref = run(); info0 = backend.last_launch_info()
    gemm_form(2); got = run(); info2 = backend.last_launch_info()
    assert info0["kernel"] == "gemm_mfma" and info2["kernel"] == "gemm_wlds", (info0, info2)
    if info0["ksplit"] == 1 and info0["ks"] == 1:
        assert torch.equal(ref, got)                          # (a K-split / K-halved launch of the per-wave kernel adds its slices in another order)
    else:
        assert torch.allclose(ref, got, rtol=1e-4, atol=1e-4 * float(ref.abs().max()))
    if fused:
        xh = x.astype(np.float16).astype(np.float32)
        u, _ = oracle.mul_mat_f64(t, wu, xh); g, _ = oracle.mul_mat_f64(t, wg, xh)
        assert nmse(got.cpu().numpy(), (g * 0.5 * (1 + np.tanh(0.5 * g))) * u) < 1e-6
    else:
        want, sum_abs = oracle.mul_mat_f64(t, wu, x.astype(np.float16).astype(np.float32))
        assert np.max(np.abs(got.cpu().numpy() - want) / sum_abs) < (1e-2 if t == ob.Q4_K else TOL_FP_ACCUM)      # (Q4_K: f16-rounded scales; outlier activations)


@pytest.mark.parametrize("t", MFMA_TYPES, ids=lambda t: ob.NAMES[t])
@pytest.mark.parametrize("m,k,n,fused", [(1024, 1024, 512, False), (700, 2048, 300, False), (384, 4096, 256, True), (130, 512, 1000, True), (512, 14336, 256, False), (256, 256, 256, False)])
def test_ping_pong_kernel_is_bit_identical_to_the_per_wave_kernel(t, m, k, n, fused, backend, oracle, gemm_form):
    """gemm_pp (round 6: the two waves of a SIMD alternate between matrix and load / de-quantize intervals, loads in flight across the workgroup barriers) against gemm_mfma on
    the same inputs: same products, same k-step pairing, same accumulation order => the SAME BITS for all six scope types, ragged row / token counts, plain and fused, two
    128-wide K tiles (the shortest row of a 256-block type) up to 112 of them; and against the oracle.  Form 3 forces the kernel onto grids it would not be chosen for.
    Outlier activations (1e3) on every 256th value make a stale or early LDS read visible: a fragment read before its DMA has landed changes bits."""
    wu = make_weights(t, m, k, 400 + t, oracle); wg = make_weights(t, m, k, 401 + t, oracle) if fused else None
    x = activations(n, k, 402, outliers=True); xd = dev(x)

    def run():
        return backend.fused_up_gate(t, dev(wu), dev(wg), xd, op=10) if fused else backend.mul_mat(t, dev(wu), xd)
    gemm_form(0); ref = run(); info0 = backend.last_launch_info()
    gemm_form(3); got = run(); info3 = backend.last_launch_info()
    assert info0["kernel"] == "gemm_mfma" and info3["kernel"] == "gemm_pp", (info0, info3)
    for _ in range(3):                                        # (a race would not repeat: the same launch three more times)
        assert torch.equal(got, run())
    if info0["ksplit"] == 1 and info0["ks"] == 1:
        assert torch.equal(ref, got)
    else:
        assert torch.allclose(ref, got, rtol=1e-4, atol=1e-4 * float(ref.abs().max()))
    if fused:
        xh = x.astype(np.float16).astype(np.float32)
        u, _ = oracle.mul_mat_f64(t, wu, xh); g, _ = oracle.mul_mat_f64(t, wg, xh)
        assert nmse(got.cpu().numpy(), (g * 0.5 * (1 + np.tanh(0.5 * g))) * u) < 1e-6
    else:
        want, sum_abs = oracle.mul_mat_f64(t, wu, x.astype(np.float16).astype(np.float32))
        assert np.max(np.abs(got.cpu().numpy() - want) / sum_abs) < (1e-2 if t == ob.Q4_K else TOL_FP_ACCUM)


@pytest.mark.parametrize("t", MFMA_TYPES, ids=lambda t: ob.NAMES[t])
@pytest.mark.parametrize("m,k,n,fused", [(1024, 1024, 512, False), (700, 2048, 300, False), (384, 4096, 256, True), (130, 512, 1000, True), (512, 14336, 256, False), (256, 256, 256, False)])
def test_f16_image_route_is_bit_identical_to_the_per_wave_kernel(t, m, k, n, fused, backend, oracle, gemm_form):
    """the large-batch route of round 6 (form 4 forces it onto any shape): the weights de-quantized ONCE into an f16 image (dequant_slab_kernel: WTile::frag values, i.e. the bits
    the fused kernels multiply), then gemm_ppf -- both operands by LDS-DMA, the k pairing of the weight type's fused kernels -- against gemm_mfma on the same inputs: the SAME BITS
    for all six scope types, ragged row / token counts (partial tiles on both edges: image rows past M, token rows past N), plain and fused; three more launches must repeat the
    bits (a DMA piece read before it has landed would not); and against the oracle."""
    wu = make_weights(t, m, k, 500 + t, oracle); wg = make_weights(t, m, k, 501 + t, oracle) if fused else None
    x = activations(n, k, 502, outliers=True); xd = dev(x)

    def run():
        return backend.fused_up_gate(t, dev(wu), dev(wg), xd, op=10) if fused else backend.mul_mat(t, dev(wu), xd)
    gemm_form(0); ref = run(); info0 = backend.last_launch_info()
    gemm_form(4); got = run(); info4 = backend.last_launch_info()
    assert info0["kernel"] == "gemm_mfma" and info4["kernel"] == "gemm_ppf" and info4["upgate"] == int(fused), (info0, info4)
    for _ in range(3):
        assert torch.equal(got, run())
    if info0["ksplit"] == 1 and info0["ks"] == 1:
        assert torch.equal(ref, got)
    else:
        assert torch.allclose(ref, got, rtol=1e-4, atol=1e-4 * float(ref.abs().max()))
    if fused:
        xh = x.astype(np.float16).astype(np.float32)
        u, _ = oracle.mul_mat_f64(t, wu, xh); g, _ = oracle.mul_mat_f64(t, wg, xh)
        assert nmse(got.cpu().numpy(), (g * 0.5 * (1 + np.tanh(0.5 * g))) * u) < 1e-6
    else:
        want, sum_abs = oracle.mul_mat_f64(t, wu, x.astype(np.float16).astype(np.float32))
        assert np.max(np.abs(got.cpu().numpy() - want) / sum_abs) < (1e-2 if t == ob.Q4_K else TOL_FP_ACCUM)


@pytest.mark.parametrize("t", [ob.Q4_0, ob.Q8_0, ob.IQ4_XS, ob.Q2_K, ob.Q3_K, ob.IQ2_XXS, ob.IQ3_XXS, ob.IQ4_KS, ob.IQ1_S, ob.MXFP4], ids=lambda t: ob.NAMES[t])
def test_f16_image_route_other_weight_types(t, backend, oracle, gemm_form):
    """every weight type with a prompt tile takes the large-batch route through the same image kernel (codebook types expand their tables in its prologue, the row-scaled types read
    their row scale): against the per-wave kernel and the oracle"""
    m, k, n = 300, 1024, 260
    w = make_weights(t, m, k, 600 + t, oracle); x = activations(n, k, 601); xd = dev(x)
    gemm_form(0); ref = backend.mul_mat(t, dev(w), xd)
    gemm_form(4); got = backend.mul_mat(t, dev(w), xd); info = backend.last_launch_info()
    assert info["kernel"] == "gemm_ppf", info
    assert torch.allclose(ref, got, rtol=1e-4, atol=1e-4 * float(ref.abs().max()))
    want, sum_abs = oracle.mul_mat_f64(t, w, x.astype(np.float16).astype(np.float32))
    assert np.max(np.abs(got.cpu().numpy() - want) / np.maximum(sum_abs, 1e-30)) < TOL_FP_ACCUM


@pytest.mark.parametrize("t", [ob.Q4_K, ob.IQ2_S], ids=lambda t: ob.NAMES[t])
@pytest.mark.parametrize("n", [1, 5, 8, 33, 200])
def test_no_writes_outside_the_result(t, n, backend, oracle):
    """sentinel check in the style of tests/test-backend-ops.cpp:440-476: the result lives in the middle of a larger buffer filled with a canary, rows
    are strided (stride_C > M); nothing but the [n, m] result may change -- the kernels clamp out-of-range re-reads and pad token tiles internally"""
    m, k = 200, 1024                                   # not a multiple of the 128-row / 64-row tiles
    w = make_weights(t, m, k, 77, oracle); x = activations(n, k, 78)
    canary = 1.2345e30; stride = m + 56; lead = 512
    buf = torch.full((lead + n * stride + 512,), canary, device="cuda", dtype=torch.float32)
    out = buf[lead:lead + n * stride].view(n, stride)[:, :m]
    backend.mul_mat(t, dev(w), dev(x), out=out)
    torch.cuda.synchronize()
    mask = torch.ones_like(buf, dtype=torch.bool); mask[lead:lead + n * stride].view(n, stride)[:, :m] = False
    assert bool((buf[mask] == canary).all()), "a kernel wrote outside its result"
    want = backend.mul_mat(t, dev(w), dev(x))              # (a contiguous result may take a K-split launch: another f32 summation order)
    assert torch.allclose(out, want, rtol=1e-4, atol=1e-4 * float(want.abs().max()))


def _moe_reference(oracle, t, ws, x, ids, ws_gate=None):
    """fp64 accumulate per (token, slot) on f16-rounded activations; fused up*gate (SILU) when ws_gate is given."""
    n_tok, n_b, k = x.shape; n_used = ids.shape[1]; m = ws.shape[1]
    out = np.zeros((n_tok, n_used, m), np.float64)
    xh = x.astype(np.float16).astype(np.float32)
    for tk in range(n_tok):
        for s in range(n_used):
            e = ids[tk, s]
            if e < 0 or e >= ws.shape[0]:
                continue
            xv = xh[tk, 0 if n_b == 1 else s][None, :]
            u, _ = oracle.mul_mat_f64(t, ws[e], xv)
            if ws_gate is None:
                out[tk, s] = u[0]
            else:
                g, _ = oracle.mul_mat_f64(t, ws_gate[e], xv)
                out[tk, s] = (g[0] * 0.5 * (1 + np.tanh(0.5 * g[0]))) * u[0]
    return out


@pytest.mark.parametrize("t", [ob.Q4_K, ob.Q6_K, ob.IQ2_S], ids=lambda t: ob.NAMES[t])
@pytest.mark.parametrize("n_expert,n_used,n_tok,n_b", [(8, 2, 64, 2), (4, 1, 200, 1), (8, 4, 40, 1), (2, 1, 600, 1)])
def test_mul_mat_id_grouped_prefill(t, n_expert, n_used, n_tok, n_b, backend, oracle):
    """MUL_MAT_ID at prompt sizes: pairs are grouped by expert on the device (no host row mapping), one grouped MFMA GEMM."""
    m, k = 200, 512
    ws = np.stack([make_weights(t, m, k, 700 + e, oracle) for e in range(n_expert)])
    x = activations(n_tok * n_b, k, 31).reshape(n_tok, n_b, k)
    ids = np.random.default_rng(5).integers(0, n_expert, size=(n_tok, n_used)).astype(np.int32)
    ids[1, 0] = -1; ids[n_tok - 1, n_used - 1] = n_expert + 3            # invalid ids -> zero rows
    got = backend.mul_mat_id(t, dev(ws), dev(x), dev(ids)).cpu().numpy()
    want = _moe_reference(oracle, t, ws, x, ids)
    assert nmse(got, want) < 1e-6
    assert np.all(got[1, 0] == 0) and np.all(got[n_tok - 1, n_used - 1] == 0)
    cpu = oracle.mul_mat_id(t, ws, x, np.where((ids < 0) | (ids >= n_expert), -1, ids).astype(np.int32))
    assert nmse(got, cpu) < NMSE_VS_CPU


@pytest.mark.parametrize("n_expert,n_used,n_tok", [(128, 8, 80), (67, 4, 96), (129, 2, 300)])
def test_mul_mat_id_many_experts(n_expert, n_used, n_tok, backend, oracle):
    """Many experts, few pairs each (>= 4 per expert on average so that the grouped MFMA path is taken): the device-side grouping scans
    pair and tile counts over all experts in parallel (even / odd expert counts, experts without any pair, partially filled tiles)."""
    t, m, k = ob.Q4_K, 96, 256
    base = [make_weights(t, m, k, 1700 + e, oracle) for e in range(8)]
    ws = np.stack([np.roll(base[e % 8], e, axis=0) for e in range(n_expert)])      # distinct per expert (row rotation), cheap to build
    x = activations(n_tok, k, 33).reshape(n_tok, 1, k)
    rng = np.random.default_rng(9)
    ids = np.stack([rng.permutation(n_expert - 3)[:n_used] for _ in range(n_tok)]).astype(np.int32)     # the last 3 experts never used
    ids[2, 0] = -1
    assert n_tok * n_used >= 4 * n_expert
    got = backend.mul_mat_id(t, dev(ws), dev(x), dev(ids)).cpu().numpy()
    want = _moe_reference(oracle, t, ws, x, ids)
    assert nmse(got, want) < 1e-6
    assert np.all(got[2, 0] == 0)


def test_mul_mat_id_small_batch_takes_decode_path(backend, oracle):
    """Fewer than 4 pairs per expert: the id-indexed GEMV serves the batch (CPU int8 arithmetic, so it matches the iqk oracle to f32
    summation order -- tighter than the f16 MFMA path could)."""
    t, m, k, n_expert, n_used, n_tok = ob.Q4_K, 96, 256, 64, 4, 12
    base = [make_weights(t, m, k, 1800 + e, oracle) for e in range(8)]
    ws = np.stack([np.roll(base[e % 8], e, axis=0) for e in range(n_expert)])
    x = activations(n_tok, k, 34).reshape(n_tok, 1, k)
    rng = np.random.default_rng(10)
    ids = np.stack([rng.permutation(n_expert)[:n_used] for _ in range(n_tok)]).astype(np.int32)
    got = backend.mul_mat_id(t, dev(ws), dev(x), dev(ids)).cpu().numpy()
    cpu = oracle.mul_mat_id(t, ws, x, ids)
    assert np.allclose(got, cpu, rtol=2e-5, atol=2e-6 * np.abs(cpu).max())


@pytest.mark.parametrize("n_expert,n_used,n_tok", [(8, 2, 96), (4, 2, 120), (2, 1, 560)])      # 24 / 60 / 280 pairs per expert: 32- / 64- / 128-token tiles
def test_moe_fused_up_gate_grouped_prefill(n_expert, n_used, n_tok, backend, oracle):
    t, m, k = ob.Q4_K, 160, 512
    wu = np.stack([make_weights(t, m, k, 800 + e, oracle) for e in range(n_expert)])
    wg = np.stack([make_weights(t, m, k, 900 + e, oracle) for e in range(n_expert)])
    x = activations(n_tok, k, 33).reshape(n_tok, 1, k)
    ids = np.random.default_rng(6).integers(0, n_expert, size=(n_tok, n_used)).astype(np.int32)
    got = backend.moe_fused_up_gate(t, dev(wu), dev(wg), dev(x), dev(ids), op=10).cpu().numpy()
    assert nmse(got, _moe_reference(oracle, t, wu, x, ids, ws_gate=wg)) < 1e-6


def test_mul_mat_multi_prefill_qkv(backend, oracle):
    """q/k/v at prompt sizes: activations converted once, same-type matrices in one MFMA launch; identical to separate calls."""
    k, n = 1024, 96
    wq = make_weights(ob.Q4_K, 512, k, 1, oracle); wk = make_weights(ob.Q4_K, 128, k, 2, oracle); wv6 = make_weights(ob.Q6_K, 128, k, 3, oracle)
    wv4 = make_weights(ob.Q4_K, 200, k, 4, oracle)
    x = dev(activations(n, k, 9))
    for types, ws in (([ob.Q4_K, ob.Q4_K, ob.Q4_K], [wq, wk, wv4]), ([ob.Q4_K, ob.Q4_K, ob.Q6_K], [wq, wk, wv6])):
        outs = backend.mul_mat_multi(types, [dev(w) for w in ws], x)
        for t, w, o in zip(types, ws, outs):
            assert torch.allclose(o, backend.mul_mat(t, dev(w), x), rtol=1e-5, atol=1e-5 * float(o.abs().max()))


@pytest.mark.parametrize("t", [ob.Q4_K, ob.Q6_K], ids=lambda t: ob.NAMES[t])
@pytest.mark.parametrize("m,k,n", [(4096, 14336, 512), (4096, 4096, 512), (1024, 8192, 300), (384, 16384, 100)])
def test_split_k_prompt_gemm_is_deterministic_and_writes_strided_results(t, m, k, n, backend, oracle):
    """Grids smaller than the chip are split over K (gemm_mfma.cuh).  The slices' partial tiles meet in the workspace (write-through stores, one agent-scope ticket per
    workgroup) and the last workgroup to arrive adds them in slice order: bit-identical results run after run BY DEFAULT since round 4 (rounds 1-3 accumulated with f32 atomics
    in arrival order unless cdna4_set_deterministic was set; the reference's mmq.cuh stream-k fix-up is deterministic too), no zero-fill of the result, any result stride, and
    the counters re-arm themselves (a third call after two).  Other launches in between reuse the same workspace slabs: a stale slab would show."""
    w = dev(make_weights(t, m, k, 77 + t, oracle)); x = dev(activations(n, k, 78)); x2 = dev(activations(n, k, 79))
    a = backend.mul_mat(t, w, x)
    other = backend.mul_mat(t, w, x2)                           # (same slabs, other data)
    b = backend.mul_mat(t, w, x)
    big = torch.full((n, m + 64), 7.0, dtype=torch.float32, device=x.device)
    c = backend.mul_mat(t, w, x, out=big[:, :m])                 # rows m + 64 floats apart, canary behind every row
    torch.cuda.synchronize()
    assert not torch.equal(a, other)
    assert torch.equal(a, b) and torch.equal(a, c)
    assert bool((big[:, m:] == 7.0).all())
    # a row subset against the fp64 accumulate of the L0 weights x f16-rounded activations (the bar of every prompt test)
    rows = np.r_[0:8, m // 2:m // 2 + 8, m - 8:m]
    c64, sum_abs = oracle.mul_mat_f64(t, np.ascontiguousarray(w.cpu().numpy()[rows]), x.cpu().numpy().astype(np.float16).astype(np.float32))
    assert np.max(np.abs(a.cpu().numpy()[:, rows] - c64) / np.maximum(sum_abs, 1e-30)) < TOL_FP_ACCUM


@pytest.mark.parametrize("t", [ob.Q4_0, ob.Q8_0, ob.IQ4_NL, ob.MXFP4, ob.Q5_1], ids=lambda t: ob.NAMES[t])
@pytest.mark.parametrize("m,k,n", [(96, 2880, 40), (256, 2880, 512), (64, 192, 33)])
def test_prompt_batches_with_row_lengths_off_the_128_grid(t, m, k, n, backend, oracle):
    """32-block types on rows that are a multiple of 64 but not of 128 (gpt-oss: n_embd 2880): the prompt GEMM walks 128-wide K tiles, so these shapes run through the
    f16 route with both operands zero-padded to the next multiple of 128 (rounds 1-2 sent them down the decode kernels column group by column group); with a fused
    up*gate launch as well"""
    w = make_weights(t, m, k, 600 + t, oracle); x = activations(n, k, 601)
    got = backend.mul_mat(t, dev(w), dev(x)).cpu().numpy()
    c64, sum_abs = oracle.mul_mat_f64(t, w, x.astype(np.float16).astype(np.float32))
    assert np.max(np.abs(got - c64) / np.maximum(sum_abs, 1e-30)) < TOL_FP_ACCUM
    wg = make_weights(t, m, k, 602 + t, oracle)
    fused = backend.fused_up_gate(t, dev(w), dev(wg), dev(x), op=6).cpu().numpy()
    g64, _ = oracle.mul_mat_f64(t, wg, x.astype(np.float16).astype(np.float32))
    assert nmse(fused, np.maximum(g64, 0) * c64) < 1e-6
