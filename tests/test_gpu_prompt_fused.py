"""-m gpu: prompt batches of cdna4_mul_mat_multi_fused / cdna4_fused_up_gate_fused -- [ADD +] FUSED_RMS_NORM + the f16 activation image as ONE launch in front of the matrix-core
GEMM(s) (csrc/ops.hip norm_to_f16_slab_kernel) -- against the launches they replace through the C ABI (cdna4_op_add_rms_norm / cdna4_op_rms_norm, then cdna4_mul_mat_multi /
cdna4_fused_up_gate): the sum must be bit-identical, the mat-mul results bit-identical too (same sums in the same order, same conversion), and within the prompt bar of float64
arithmetic on the CPU restatement's weights; what the fused form does not serve is declined with CDNA4_E_UNSUPPORTED."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from common import random_block_bytes  # noqa: E402
from oracle import bindings as ob  # noqa: E402

pytestmark = pytest.mark.gpu
P, I, L64, F = C.c_void_p, C.c_int, C.c_long, C.c_float


class Fusion(C.Structure):      # cdna4_fusion
    _fields_ = [("norm_w", P), ("norm_eps", F), ("residual", P), ("qkv", P), ("add_b", P), ("add_dst", P)]


@pytest.fixture(scope="module")
def env():
    import nt_bench as nb
    hip = nb.Hip(); lib = nb.load_lib(os.path.join(ROOT, "ik_llama.cpp_amd", "libggml-hip-cdna4.so"))
    TP = C.POINTER(nb.Tensor)
    lib.cdna4_op_add_rms_norm.argtypes = [P, TP, TP, TP, TP, F, TP, P]
    lib.cdna4_op_rms_norm.argtypes = [P, TP, TP, F, TP, P]
    lib.cdna4_mul_mat_multi.argtypes = [P, I, C.POINTER(L64), L64, L64, C.POINTER(I), C.POINTER(P), C.POINTER(L64), I, P, L64, C.POINTER(P), C.POINTER(L64), P]
    lib.cdna4_mul_mat_multi_fused.argtypes = lib.cdna4_mul_mat_multi.argtypes[:-1] + [C.POINTER(Fusion), P]
    lib.cdna4_fused_up_gate.argtypes = [P, L64, L64, L64, I, I, P, P, L64, I, P, L64, P, L64, P]
    lib.cdna4_fused_up_gate_fused.argtypes = [P, L64, L64, L64, I, I, P, P, L64, I, P, L64, P, P, F, P, L64, C.POINTER(Fusion), P]
    ctx = lib.cdna4_init(0)
    assert ctx, lib.cdna4_last_error()
    yield nb, hip, lib, ctx
    lib.cdna4_free(ctx)


def _norm64(x, w, eps):
    x = x.astype(np.float64)
    return x / np.sqrt((x * x).mean(-1, keepdims=True) + eps) * w.astype(np.float64)


@pytest.mark.parametrize("with_add", [False, True])
@pytest.mark.parametrize("types,rows,K,n_tok", [((ob.Q4_K, ob.Q4_K, ob.Q6_K), (512, 128, 128), 4096, 512), ((ob.Q4_K,), (384,), 2048, 77), ((ob.Q5_K, ob.Q5_K), (256, 256), 1024, 9),
                                                ((ob.IQ4_NL,), (256,), 14336, 130)])
def test_prompt_norm_rides_in_the_image_launch(types, rows, K, n_tok, with_add, env):
    nb, hip, lib, ctx = env
    rng = np.random.default_rng(8); eps = 1e-5
    x = rng.standard_normal((n_tok, K)).astype(np.float32) * 3
    b = rng.standard_normal((n_tok, K)).astype(np.float32); wn = (1 + 0.1 * rng.standard_normal(K)).astype(np.float32)
    wn[7] = 4.0e4                                                                         # (normed values beyond 2^14: the image's per-row range guard scales these rows)
    ws = [random_block_bytes(t, m, K, 40 + i) for i, (t, m) in enumerate(zip(types, rows))]
    xd, bd, wnd = hip.upload(x), hip.upload(b), hip.upload(wn); wd = [hip.upload(w) for w in ws]
    s1, s2, y = hip.malloc(4 * n_tok * K), hip.malloc(4 * n_tok * K), hip.malloc(4 * n_tok * K)
    c1 = [hip.malloc(4 * n_tok * m) for m in rows]; c2 = [hip.malloc(4 * n_tok * m) for m in rows]
    n = len(types); nx = (L64 * n)(*rows); ty = (I * n)(*types); ap = (P * n)(*wd); sa = (L64 * n)(*[w.shape[1] for w in ws]); sc = (L64 * n)(*rows)
    tx = nb.tensor(xd, 0, [K, n_tok, 1, 1], 4); tb = nb.tensor(bd, 0, [K, n_tok, 1, 1], 4); ts = nb.tensor(s1, 0, [K, n_tok, 1, 1], 4); ty_ = nb.tensor(y, 0, [K, n_tok, 1, 1], 4)
    tw = nb.tensor(wnd, 0, [K, 1, 1, 1], 4)
    # the launches it replaces
    if with_add:
        assert lib.cdna4_op_add_rms_norm(ctx, C.byref(tx), C.byref(tb), C.byref(ts), C.byref(tw), eps, C.byref(ty_), None) == 0, lib.cdna4_last_error()
    else:
        assert lib.cdna4_op_rms_norm(ctx, C.byref(tx), C.byref(tw), eps, C.byref(ty_), None) == 0, lib.cdna4_last_error()
    assert lib.cdna4_mul_mat_multi(ctx, n, nx, n_tok, K, ty, ap, sa, 0, y, 4 * K, (P * n)(*c1), sc, None) == 0, lib.cdna4_last_error()
    # the fused call
    fx = Fusion(wnd, eps, None, None, bd if with_add else None, s2 if with_add else None)
    assert lib.cdna4_mul_mat_multi_fused(ctx, n, nx, n_tok, K, ty, ap, sa, 0, xd, 4 * K, (P * n)(*c2), sc, C.byref(fx), None) == 0, lib.cdna4_last_error()
    hip.check(hip.h.hipDeviceSynchronize(), "sync")
    if with_add:
        np.testing.assert_array_equal(hip.download(s1, (n_tok, K), np.float32).view(np.uint32), hip.download(s2, (n_tok, K), np.float32).view(np.uint32))
    xs = x + b if with_add else x
    yn = _norm64(xs, wn, eps)
    for i, (t, m) in enumerate(zip(types, rows)):
        r1, r2 = hip.download(c1[i], (n_tok, m), np.float32), hip.download(c2[i], (n_tok, m), np.float32)
        np.testing.assert_array_equal(r1.view(np.uint32), r2.view(np.uint32), err_msg="matrix %d" % i)
        want = yn @ ob.Oracle().dequantize(t, ws[i], K).astype(np.float64).T
        assert np.sum((r2 - want) ** 2) / np.sum(want ** 2) < 2e-6
    for d in [xd, bd, wnd, s1, s2, y] + wd + c1 + c2:
        hip.h.hipFree(d)


@pytest.mark.parametrize("with_add", [False, True])
def test_prompt_norm_rides_in_the_up_gate_image_launch(with_add, env):
    nb, hip, lib, ctx = env
    rng = np.random.default_rng(9); eps = 1e-6; K, m, n_tok, t = 4096, 1024, 200, ob.Q4_K
    x = rng.standard_normal((n_tok, K)).astype(np.float32); b = rng.standard_normal((n_tok, K)).astype(np.float32); wn = (1 + 0.1 * rng.standard_normal(K)).astype(np.float32)
    wu, wg = random_block_bytes(t, m, K, 50), random_block_bytes(t, m, K, 51)
    xd, bd, wnd, wud, wgd = hip.upload(x), hip.upload(b), hip.upload(wn), hip.upload(wu), hip.upload(wg)
    s1, s2, y, c1, c2 = hip.malloc(4 * n_tok * K), hip.malloc(4 * n_tok * K), hip.malloc(4 * n_tok * K), hip.malloc(4 * n_tok * m), hip.malloc(4 * n_tok * m)
    tx = nb.tensor(xd, 0, [K, n_tok, 1, 1], 4); tb = nb.tensor(bd, 0, [K, n_tok, 1, 1], 4); ts = nb.tensor(s1, 0, [K, n_tok, 1, 1], 4); ty_ = nb.tensor(y, 0, [K, n_tok, 1, 1], 4)
    tw = nb.tensor(wnd, 0, [K, 1, 1, 1], 4)
    if with_add:
        assert lib.cdna4_op_add_rms_norm(ctx, C.byref(tx), C.byref(tb), C.byref(ts), C.byref(tw), eps, C.byref(ty_), None) == 0, lib.cdna4_last_error()
    else:
        assert lib.cdna4_op_rms_norm(ctx, C.byref(tx), C.byref(tw), eps, C.byref(ty_), None) == 0, lib.cdna4_last_error()
    SILU = 10       # GGML_UNARY_OP_SILU
    assert lib.cdna4_fused_up_gate(ctx, m, n_tok, K, SILU, t, wud, wgd, wu.shape[1], 0, y, 4 * K, c1, m, None) == 0, lib.cdna4_last_error()
    fx = Fusion(wnd, eps, None, None, bd if with_add else None, s2 if with_add else None)
    assert lib.cdna4_fused_up_gate_fused(ctx, m, n_tok, K, SILU, t, wud, wgd, wu.shape[1], 0, xd, 4 * K, None, None, 0.0, c2, m, C.byref(fx), None) == 0, lib.cdna4_last_error()
    hip.check(hip.h.hipDeviceSynchronize(), "sync")
    if with_add:
        np.testing.assert_array_equal(hip.download(s1, (n_tok, K), np.float32).view(np.uint32), hip.download(s2, (n_tok, K), np.float32).view(np.uint32))
    r1, r2 = hip.download(c1, (n_tok, m), np.float32), hip.download(c2, (n_tok, m), np.float32)
    np.testing.assert_array_equal(r1.view(np.uint32), r2.view(np.uint32))
    yn = _norm64(x + b if with_add else x, wn, eps); o = ob.Oracle()
    u = yn @ o.dequantize(t, wu, K).astype(np.float64).T; g = yn @ o.dequantize(t, wg, K).astype(np.float64).T
    want = u * g / (1 + np.exp(-g))
    assert np.sum((r2 - want) ** 2) / np.sum(want ** 2) < 2e-6
    for d in (xd, bd, wnd, wud, wgd, s1, s2, y, c1, c2):
        hip.h.hipFree(d)


def test_prompt_fusion_declines_what_it_does_not_serve(env):
    nb, hip, lib, ctx = env
    K, m = 4096, 128; w = random_block_bytes(ob.Q4_K, m, K, 3); wd = hip.upload(w); buf = hip.malloc(4 * 64 * K); cbuf = hip.malloc(4 * 64 * m)
    nx = (L64 * 1)(m); ty = (I * 1)(ob.Q4_K); ap = (P * 1)(wd); sa = (L64 * 1)(w.shape[1]); sc = (L64 * 1)(m); cp = (P * 1)(cbuf)
    def call(n_tok, fx, stride=4 * K):
        return lib.cdna4_mul_mat_multi_fused(ctx, 1, nx, n_tok, K, ty, ap, sa, 0, buf, stride, cp, sc, C.byref(fx), None)
    assert call(64, Fusion(buf, 1e-5, buf, None, None, None)) == -1          # residual epilogue on a prompt batch
    assert call(4, Fusion(buf, 1e-5, None, None, None, None)) == -1           # 2..8 rows: neither the decode form nor the prompt form
    assert call(1, Fusion(buf, 1e-5, None, None, buf, buf)) == -1             # the ADD in front of the norm: prompt batches only
    assert call(64, Fusion(buf, 1e-5, None, None, buf, None)) != 0            # add_b without add_dst
    assert call(64, Fusion(buf, 1e-5, None, None, None, None), stride=4 * K + 4) == -1      # rows off the 16-byte grid
    for d in (wd, buf, cbuf):
        hip.h.hipFree(d)
