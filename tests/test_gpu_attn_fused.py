"""-m gpu: cdna4_attn_out_fused -- FLASH_ATTN_EXT + attn_output MUL_MAT + residual ADD of one decoded token as ONE launch (csrc/gemv_attn.hip) -- against the two launches
it replaces (cdna4_op_flash_attn, then cdna4_mul_mat_multi_fused with the residual), through the C ABI: the attention row and the result must be BIT-IDENTICAL (same attention
body, same mat-vec body, same order of every sum), launch after launch with fresh q / visible keys (the hand-off reuses the same row and the same ticket words every launch:
a stale row or a lost ticket shows as a wrong result), and both must sit within the mat-mul bar of float64 arithmetic on the CPU restatement's weights."""
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


@pytest.fixture(scope="module")
def env():
    import nt_bench as nb
    hip = nb.Hip(); lib = nb.load_lib(os.path.join(ROOT, "ik_llama.cpp_amd", "libggml-hip-cdna4.so"))
    TP = C.POINTER(nb.Tensor)
    lib.cdna4_attn_out_fused.argtypes = [P, TP, TP, TP, TP, TP, F, F, F, L64, L64, I, P, L64, P, P, P]
    lib.cdna4_op_flash_attn.argtypes = [P, TP, TP, TP, TP, TP, F, F, F, P]
    lib.cdna4_mul_mat_multi_fused.argtypes = [P, I, C.POINTER(L64), L64, L64, C.POINTER(I), C.POINTER(P), C.POINTER(L64), I, P, L64, C.POINTER(P), C.POINTER(L64), P, P]
    ctx = lib.cdna4_init(0)
    assert ctx, lib.cdna4_last_error()
    yield nb, hip, lib, ctx
    lib.cdna4_free(ctx)


class Fusion(C.Structure):      # cdna4_fusion {norm_w, norm_eps, residual, qkv, add_b, add_dst}
    _fields_ = [("norm_w", P), ("norm_eps", F), ("residual", P), ("qkv", P), ("add_b", P), ("add_dst", P)]


@pytest.mark.parametrize("t,n_head,n_head_kv,n_kv,m", [(ob.Q4_K, 32, 8, 256, 4096), (ob.Q6_K, 32, 8, 256, 4096), (ob.Q5_K, 32, 8, 128, 4096), (ob.IQ4_NL, 32, 8, 320, 4096),
                                                       (ob.Q4_K, 32, 32, 192, 1024), (ob.Q4_K, 24, 8, 256, 3072), (ob.Q6_K, 32, 4, 64, 512)])
def test_attn_out_fused_matches_the_two_launches_bit_for_bit(t, n_head, n_head_kv, n_kv, m, env):
    nb, hip, lib, ctx = env
    D = 128; K = n_head * D; rng = np.random.default_rng(5)
    w = random_block_bytes(t, m, K, 21); kk = rng.standard_normal((n_head_kv, n_kv, D)).astype(np.float16); vv = rng.standard_normal((n_head_kv, n_kv, D)).astype(np.float16)
    wd, kd, vd = hip.upload(w), hip.upload(kk), hip.upload(vv)
    qd, md, rd = hip.malloc(4 * K), hip.malloc(2 * 32 * n_kv), hip.malloc(4 * m)
    a1, a2, c1, c2 = hip.malloc(4 * K), hip.malloc(4 * K), hip.malloc(4 * m), hip.malloc(4 * m)
    tq = nb.tensor(qd, 0, [D, 1, n_head, 1], 4); tk = nb.tensor(kd, 1, [D, n_kv, n_head_kv, 1], 2); tv = nb.tensor(vd, 1, [D, n_kv, n_head_kv, 1], 2); tm = nb.tensor(md, 1, [n_kv, 32, 1, 1], 2)
    ta1 = nb.tensor(a1, 0, [D, n_head, 1, 1], 4); ta2 = nb.tensor(a2, 0, [D, n_head, 1, 1], 4)
    scale = 1.0 / np.sqrt(D); g = n_head // n_head_kv
    wf = ob.Oracle().dequantize(t, w, K).astype(np.float64)
    for it in range(12):
        q = rng.standard_normal((n_head, 1, D)).astype(np.float32); res = rng.standard_normal(m).astype(np.float32); nvis = n_kv if it == 0 else int(rng.integers(1, n_kv + 1))
        mask = np.zeros((32, n_kv), np.float16); mask[:, nvis:] = -np.inf
        for dst, src in ((qd, q), (md, mask), (rd, res)):
            hip.check(hip.h.hipMemcpy(dst, src.ctypes.data_as(P), src.nbytes, 1), "H2D")
        for buf, n in ((a1, 4 * K), (a2, 4 * K), (c1, 4 * m), (c2, 4 * m)):
            hip.check(hip.h.hipMemset(buf, 0xff, n), "memset")
        # the two launches
        assert lib.cdna4_op_flash_attn(ctx, C.byref(tq), C.byref(tk), C.byref(tv), C.byref(tm), C.byref(ta1), scale, 0.0, 0.0, None) == 0, lib.cdna4_last_error()
        fx = Fusion(None, 0.0, rd, None, None, None); nx = (L64 * 1)(m); ty = (I * 1)(t); ap = (P * 1)(wd); sa = (L64 * 1)(w.shape[1]); cp = (P * 1)(c1); sc = (L64 * 1)(m)
        assert lib.cdna4_mul_mat_multi_fused(ctx, 1, nx, 1, K, ty, ap, sa, 0, a1, 4 * K, cp, sc, C.byref(fx), None) == 0, lib.cdna4_last_error()
        # the one launch (three times back to back: the tickets re-arm themselves)
        for _ in range(3):
            rc = lib.cdna4_attn_out_fused(ctx, C.byref(tq), C.byref(tk), C.byref(tv), C.byref(tm), C.byref(ta2), scale, 0.0, 0.0, m, K, t, wd, w.shape[1], rd, c2, None)
            assert rc == 0, lib.cdna4_last_error()
        hip.check(hip.h.hipDeviceSynchronize(), "sync")
        at1, at2 = hip.download(a1, (K,), np.float32), hip.download(a2, (K,), np.float32); r1, r2 = hip.download(c1, (m,), np.float32), hip.download(c2, (m,), np.float32)
        np.testing.assert_array_equal(at1.view(np.uint32), at2.view(np.uint32), err_msg="attention row, launch %d" % it)
        np.testing.assert_array_equal(r1.view(np.uint32), r2.view(np.uint32), err_msg="result, launch %d" % it)
        # and the arithmetic itself: float64 attention, exact weights
        want_a = np.empty((n_head, D))
        for h in range(n_head):
            s_ = q[h, 0].astype(np.float64) @ kk[h // g, :nvis].astype(np.float64).T * scale; s_ -= s_.max(); pr = np.exp(s_); pr /= pr.sum()
            want_a[h] = pr @ vv[h // g, :nvis].astype(np.float64)
        assert np.sum((at2 - want_a.reshape(-1)) ** 2) / np.sum(want_a ** 2) < 1e-10
        want = wf @ want_a.reshape(-1) + res
        assert np.sum((r2 - want) ** 2) / np.sum(want ** 2) < 5e-4
    for d in (wd, kd, vd, qd, md, rd, a1, a2, c1, c2):
        hip.h.hipFree(d)


def test_attn_out_fused_declines_what_it_does_not_serve(env):
    """contexts at or above the split-KV threshold, batches, unsupported weight types: CDNA4_E_UNSUPPORTED (the shim then issues the three nodes)"""
    nb, hip, lib, ctx = env
    D, n_head, n_head_kv = 128, 32, 8; K = n_head * D; m = 512
    w = random_block_bytes(ob.Q4_K, m, K, 3); wd = hip.upload(w); buf = hip.malloc(4 << 20)
    def call(n_kv, n_tok=1, t=ob.Q4_K):
        tq = nb.tensor(buf, 0, [D, n_tok, n_head, 1], 4); tk = nb.tensor(buf, 1, [D, n_kv, n_head_kv, 1], 2); ta = nb.tensor(buf, 0, [D, n_head, n_tok, 1], 4)
        return lib.cdna4_attn_out_fused(ctx, C.byref(tq), C.byref(tk), C.byref(tk), None, C.byref(ta), 0.1, 0.0, 0.0, m, K, t, wd, w.shape[1], buf, buf, None)
    assert call(1024) == -1 and call(256, n_tok=2) == -1          # CDNA4_E_UNSUPPORTED
    n_head = 64; K = n_head * D                                   # rows of more than 4096 weights (two K-slices per row): not served
    w2 = random_block_bytes(ob.Q4_K, 64, K, 4); wd2 = hip.upload(w2)
    tq = nb.tensor(buf, 0, [D, 1, n_head, 1], 4); tk = nb.tensor(buf, 1, [D, 256, 8, 1], 2); ta = nb.tensor(buf, 0, [D, n_head, 1, 1], 4)
    assert lib.cdna4_attn_out_fused(ctx, C.byref(tq), C.byref(tk), C.byref(tk), None, C.byref(ta), 0.1, 0.0, 0.0, 64, K, ob.Q4_K, wd2, w2.shape[1], buf, buf, None) == -1
    hip.h.hipFree(wd2)
    hip.h.hipFree(wd); hip.h.hipFree(buf)


@pytest.mark.parametrize("t,n_head,n_head_kv,n_kv,m", [(ob.Q4_K, 32, 8, 256, 4096), (ob.Q6_K, 32, 8, 256, 4096), (ob.Q5_K, 32, 8, 128, 4096), (ob.IQ4_NL, 32, 8, 320, 4096),
                                                       (ob.Q4_K, 24, 8, 256, 3072), (ob.Q6_K, 16, 4, 64, 512)])
def test_attention_q8_hand_off_matches_the_f32_row_bit_for_bit(t, n_head, n_head_kv, n_kv, m, env):
    """cdna4_op_flash_attn_q8: the attention of one decoded token that ALSO emits its row as block_q8_2_x4, then the attn_output mat-vec + residual called with typeB = Q8_2_X4
    (quantize once, consume everywhere: ggml.c:17955-17964).  The f32 row equals cdna4_op_flash_attn's bit for bit, the q8 bytes equal the CPU restatement of
    quantize_row_q8_2_x4 (iqk_quantize.cpp:1072-1175) applied to that row byte for byte, and the mat-vec result equals the one computed from the f32 row bit for bit (its
    prologue would have produced the same bytes)."""
    nb, hip, lib, ctx = env
    TP = C.POINTER(nb.Tensor)
    lib.cdna4_op_flash_attn_q8.argtypes = [P, TP, TP, TP, TP, TP, F, F, F, P, P]
    D = 128; K = n_head * D; rng = np.random.default_rng(7); orc = ob.Oracle()
    w = random_block_bytes(t, m, K, 22); kk = rng.standard_normal((n_head_kv, n_kv, D)).astype(np.float16); vv = rng.standard_normal((n_head_kv, n_kv, D)).astype(np.float16)
    wd, kd, vd = hip.upload(w), hip.upload(kk), hip.upload(vv)
    qrow = K // 128 * 144
    qd, md, rd, q8d = hip.malloc(4 * K), hip.malloc(2 * 32 * n_kv), hip.malloc(4 * m), hip.malloc(qrow)
    a1, a2, c1, c2 = hip.malloc(4 * K), hip.malloc(4 * K), hip.malloc(4 * m), hip.malloc(4 * m)
    tq = nb.tensor(qd, 0, [D, 1, n_head, 1], 4); tk = nb.tensor(kd, 1, [D, n_kv, n_head_kv, 1], 2); tv = nb.tensor(vd, 1, [D, n_kv, n_head_kv, 1], 2); tm = nb.tensor(md, 1, [n_kv, 32, 1, 1], 2)
    ta1 = nb.tensor(a1, 0, [D, n_head, 1, 1], 4); ta2 = nb.tensor(a2, 0, [D, n_head, 1, 1], 4)
    scale = 1.0 / np.sqrt(D)
    for it in range(6):
        q = (rng.standard_normal((n_head, 1, D)) * (1.0 + 3.0 * (it % 3))).astype(np.float32); res = rng.standard_normal(m).astype(np.float32); nvis = n_kv if it == 0 else int(rng.integers(1, n_kv + 1))
        mask = np.zeros((32, n_kv), np.float16); mask[:, nvis:] = -np.inf
        for dst, src in ((qd, q), (md, mask), (rd, res)):
            hip.check(hip.h.hipMemcpy(dst, src.ctypes.data_as(P), src.nbytes, 1), "H2D")
        for buf, n in ((a1, 4 * K), (a2, 4 * K), (c1, 4 * m), (c2, 4 * m), (q8d, qrow)):
            hip.check(hip.h.hipMemset(buf, 0xff, n), "memset")
        nx = (L64 * 1)(m); ty = (I * 1)(t); ap = (P * 1)(wd); sa = (L64 * 1)(w.shape[1]); sc = (L64 * 1)(m)
        # f32 hand-off
        assert lib.cdna4_op_flash_attn(ctx, C.byref(tq), C.byref(tk), C.byref(tv), C.byref(tm), C.byref(ta1), scale, 0.0, 0.0, None) == 0, lib.cdna4_last_error()
        fx = Fusion(None, 0.0, rd, None, None, None); cp = (P * 1)(c1)
        assert lib.cdna4_mul_mat_multi_fused(ctx, 1, nx, 1, K, ty, ap, sa, 0, a1, 4 * K, cp, sc, C.byref(fx), None) == 0, lib.cdna4_last_error()
        # q8 hand-off
        assert lib.cdna4_op_flash_attn_q8(ctx, C.byref(tq), C.byref(tk), C.byref(tv), C.byref(tm), C.byref(ta2), scale, 0.0, 0.0, q8d, None) == 0, lib.cdna4_last_error()
        cp2 = (P * 1)(c2)
        assert lib.cdna4_mul_mat_multi_fused(ctx, 1, nx, 1, K, ty, ap, sa, 99, q8d, qrow, cp2, sc, C.byref(fx), None) == 0, lib.cdna4_last_error()
        hip.check(hip.h.hipDeviceSynchronize(), "sync")
        at1, at2 = hip.download(a1, (K,), np.float32), hip.download(a2, (K,), np.float32); r1, r2 = hip.download(c1, (m,), np.float32), hip.download(c2, (m,), np.float32)
        q8 = hip.download(q8d, (qrow,), np.uint8)
        np.testing.assert_array_equal(at1.view(np.uint32), at2.view(np.uint32), err_msg="attention row, launch %d" % it)
        want_q8 = orc.quantize_activations(ob.Q8_2_X4, at1.reshape(1, K)).reshape(-1)
        np.testing.assert_array_equal(q8, want_q8, err_msg="q8 image, launch %d" % it)
        np.testing.assert_array_equal(r1.view(np.uint32), r2.view(np.uint32), err_msg="mat-vec result, launch %d" % it)
    # more keys than the per-head kernel serves (the split-KV form from 384 keys on): declined, the caller takes the f32 path
    big = hip.malloc(2 * n_head_kv * 512 * D); tkb = nb.tensor(big, 1, [D, 512, n_head_kv, 1], 2); mb = hip.malloc(2 * 32 * 512); tmb = nb.tensor(mb, 1, [512, 32, 1, 1], 2)
    assert lib.cdna4_op_flash_attn_q8(ctx, C.byref(tq), C.byref(tkb), C.byref(tkb), C.byref(tmb), C.byref(ta2), scale, 0.0, 0.0, q8d, None) == -1      # CDNA4_E_UNSUPPORTED
    for d in (wd, kd, vd, qd, md, rd, q8d, a1, a2, c1, c2, big, mb):
        hip.h.hipFree(d)
