"""-m gpu: the drop-in boundary end to end -- the REAL reference libggml (oracle/_ref) as the host, our shim
(ggml_backend_cuda_* symbols + vtables) as the backend, one-op graphs built with the reference's own API, compared with the
reference CPU backend at its own tolerance (NMSE <= 5e-4, tests/test-backend-ops.cpp:979-981) -- the new-repo form of
test_mul_mat / test_mul_mat_id of test-backend-ops.cpp:966-1076,2265-2350."""
import os

import numpy as np
import pytest

from common import NMSE_VS_CPU, activations, gaussian_weights_f32, nmse
from oracle import bindings as ob

pytestmark = pytest.mark.gpu
F32, I32 = 0, 26


@pytest.fixture(scope="module")
def host():
    from ggml_host import SHIM, GgmlHost
    if ob.ref_path() is None or not os.path.exists(SHIM):
        pytest.skip("needs oracle/_ref (reference libggml) and the prebuilt backend shim")
    h = GgmlHost()
    assert h.shim.ggml_backend_cuda_get_device_count() >= 1
    gpu = h.shim.ggml_backend_cuda_init(0, None, None); cpu = h.g.ggml_backend_cpu_init(); h.g.ggml_backend_cpu_set_n_threads(cpu, 8)
    assert gpu and h.shim.ggml_backend_is_cuda(gpu) and not h.shim.ggml_backend_is_cuda(cpu)
    yield h, gpu, cpu
    h.g.ggml_backend_free(gpu); h.g.ggml_backend_free(cpu)


@pytest.mark.parametrize("t", ob.BASE_TYPES, ids=lambda t: ob.NAMES[t])
@pytest.mark.parametrize("m,n,k", [(16, 1, 256), (16, 16, 256), (512, 1, 4096), (256, 64, 1024)])   # first two: test-backend-ops.cpp:2265-2310
def test_mul_mat_vs_cpu_backend(t, m, n, k, host):
    h, gpu, cpu = host
    w = h.ref.quantize(t, gaussian_weights_f32(m, k, 1)); x = activations(n, k, 2)

    def build(ctx):
        a = h.g.ggml_new_tensor_2d(ctx, t, k, m); b = h.g.ggml_new_tensor_2d(ctx, F32, k, n)
        return {"a": a, "b": b}, h.g.ggml_mul_mat(ctx, a, b)
    got, sup = h.run(gpu, build, {"a": w, "b": x}); want, _ = h.run(cpu, build, {"a": w, "b": x})
    assert sup
    assert nmse(got, want) < NMSE_VS_CPU
    if n <= 8:       # decode: same int8 arithmetic as the CPU backend -> orders of magnitude tighter
        assert nmse(got, want) < 1e-10


@pytest.mark.parametrize("t,m,n,k,batch", [(ob.Q4_K, 256, 1, 1024, 1), (ob.Q6_K, 128, 5, 512, 1), (ob.IQ4_NL, 256, 48, 1024, 1), (ob.Q4_K, 64, 3, 512, 2)])
def test_mul_mat_f16_src1_with_quantised_weights(t, m, n, k, batch, host):
    """f16 activations with quantised weights: the CUDA backend's supports_op tolerates them (ggml-cuda.cu:4844-4847), the CPU path asserts f32 (ggml.c:18162) -- so the check
    is against THIS backend's f32 path on the same f16-representable values: bit-identical (the f16 row is converted to f32 in a scratch buffer, then the same launches)."""
    h, gpu, cpu = host
    F16 = 1
    w = h.ref.quantize(t, gaussian_weights_f32(m, k, 7)); x16 = activations(n * batch, k, 8).astype(np.float16)

    def build_t(xt):
        def build(ctx):
            a = h.g.ggml_new_tensor_2d(ctx, t, k, m); b = h.g.ggml_new_tensor_3d(ctx, xt, k, n, batch)
            return {"a": a, "b": b}, h.g.ggml_mul_mat(ctx, a, b)
        return build
    got, sup = h.run(gpu, build_t(F16), {"a": w, "b": x16}); want, sup32 = h.run(gpu, build_t(F32), {"a": w, "b": x16.astype(np.float32)})
    assert sup and sup32
    np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))


def test_fused_up_gate_vs_cpu_backend(host):
    h, gpu, cpu = host
    t, m, n, k = ob.Q4_K, 256, 2, 1024
    wu = h.ref.quantize(t, gaussian_weights_f32(m, k, 3)); wg = h.ref.quantize(t, gaussian_weights_f32(m, k, 4)); x = activations(n, k, 5)

    def build(ctx):
        u = h.g.ggml_new_tensor_2d(ctx, t, k, m); g = h.g.ggml_new_tensor_2d(ctx, t, k, m); b = h.g.ggml_new_tensor_2d(ctx, F32, k, n)
        return {"u": u, "g": g, "b": b}, h.g.ggml_fused_up_gate(ctx, u, g, b, 10)      # GGML_UNARY_OP_SILU
    got, sup = h.run(gpu, build, {"u": wu, "g": wg, "b": x}); want, _ = h.run(cpu, build, {"u": wu, "g": wg, "b": x})
    assert sup and nmse(got, want) < 1e-9


def test_mul_mat_id_vs_cpu_backend(host):
    h, gpu, cpu = host
    t, m, k, n_expert, n_used, n_tok = ob.Q4_K, 512, 256, 8, 2, 5          # test-backend-ops.cpp:2319-2350 shape family
    ws = np.stack([h.ref.quantize(t, gaussian_weights_f32(m, k, 10 + e)) for e in range(n_expert)])
    x = activations(n_tok * n_used, k, 6).reshape(n_tok, n_used, k)
    ids = np.random.default_rng(0).integers(0, n_expert, size=(n_tok, n_used)).astype(np.int32)

    def build(ctx):
        a = h.g.ggml_new_tensor_3d(ctx, t, k, m, n_expert); b = h.g.ggml_new_tensor_3d(ctx, F32, k, n_used, n_tok)
        i = h.g.ggml_new_tensor_2d(ctx, I32, n_used, n_tok)
        return {"a": a, "b": b, "i": i}, h.g.ggml_mul_mat_id(ctx, a, b, i)
    got, sup = h.run(gpu, build, {"a": ws, "b": x, "i": ids}); want, _ = h.run(cpu, build, {"a": ws, "b": x, "i": ids})
    assert sup and nmse(got, want) < 1e-9


@pytest.mark.parametrize("op", [10, 14], ids=["silu", "swiglu_oai"])
def test_moe_up_gate_with_biases_vs_cpu_backend(op, host):
    """GGML_OP_MOE_FUSED_UP_GATE with per-expert biases (src[4], src[5]) built by the reference's own ggml_moe_up_gate_ext."""
    h, gpu, cpu = host
    t, m, k, n_expert, n_used, n_tok = ob.Q4_K, 256, 256, 4, 2, 3
    wu = np.stack([h.ref.quantize(t, gaussian_weights_f32(m, k, 30 + e) * 30) for e in range(n_expert)])
    wg = np.stack([h.ref.quantize(t, gaussian_weights_f32(m, k, 40 + e) * 30) for e in range(n_expert)])
    x = activations(n_tok, k, 7).reshape(n_tok, 1, k)
    rng = np.random.default_rng(1); ids = rng.integers(0, n_expert, size=(n_tok, n_used)).astype(np.int32)
    ub = rng.normal(0, 1, (n_expert, m)).astype(np.float32); gb = rng.normal(0, 1, (n_expert, m)).astype(np.float32)

    def build(ctx):
        u = h.g.ggml_new_tensor_3d(ctx, t, k, m, n_expert); g = h.g.ggml_new_tensor_3d(ctx, t, k, m, n_expert)
        b = h.g.ggml_new_tensor_3d(ctx, F32, k, 1, n_tok); i = h.g.ggml_new_tensor_2d(ctx, I32, n_used, n_tok)
        bu = h.g.ggml_new_tensor_2d(ctx, F32, m, n_expert); bg = h.g.ggml_new_tensor_2d(ctx, F32, m, n_expert)
        return {"u": u, "g": g, "b": b, "i": i, "bu": bu, "bg": bg}, h.g.ggml_moe_up_gate_ext(ctx, u, g, b, i, bu, bg, op)
    inp = {"u": wu, "g": wg, "b": x, "i": ids, "bu": ub, "bg": gb}
    got, sup = h.run(gpu, build, inp); want, _ = h.run(cpu, build, inp)
    assert sup and nmse(got, want) < 1e-9


@pytest.mark.parametrize("n_tok,bias", [(3, False), (3, True), (96, True)], ids=["decode", "decode_bias", "prefill_bias"])
def test_moe_merged_up_gate_tensor_vs_cpu_backend(n_tok, bias, host):
    """GGML_OP_MOE_FUSED_UP_GATE with up and gate MERGED in one tensor per expert (src[1] == NULL; ggml.c:18470-18600: rows [0, ne01 / 2) gate, the rest up, the
    biases of both in src[4]): built by the reference's own ggml_moe_up_gate_ext(as_up, NULL, ...) and compared with its CPU backend; the shim serves it from the
    ordinary kernels with two pointers into the one tensor"""
    h, gpu, cpu = host
    t, m, k, n_expert, n_used, op = ob.Q4_K, 256, 256, 4, 2, 10
    w = np.stack([h.ref.quantize(t, gaussian_weights_f32(2 * m, k, 60 + e) * 30) for e in range(n_expert)])
    x = activations(n_tok, k, 9).reshape(n_tok, 1, k)
    rng = np.random.default_rng(2); ids = np.stack([rng.permutation(n_expert)[:n_used] for _ in range(n_tok)]).astype(np.int32)
    ub = rng.normal(0, 1, (n_expert, 2 * m)).astype(np.float32)

    def build(ctx):
        u = h.g.ggml_new_tensor_3d(ctx, t, k, 2 * m, n_expert)
        b = h.g.ggml_new_tensor_3d(ctx, F32, k, 1, n_tok); i = h.g.ggml_new_tensor_2d(ctx, I32, n_used, n_tok)
        tens = {"u": u, "b": b, "i": i}
        bu = None
        if bias:
            bu = h.g.ggml_new_tensor_2d(ctx, F32, 2 * m, n_expert); tens["bu"] = bu
        return tens, h.g.ggml_moe_up_gate_ext(ctx, u, None, b, i, bu, None, op)
    inp = {"u": w, "b": x, "i": ids}
    if bias:
        inp["bu"] = ub
    got, sup = h.run(gpu, build, inp); want, _ = h.run(cpu, build, inp)
    assert sup, "the merged form must run on the device"
    assert got.size == n_tok * n_used * m and nmse(got, want) < (1e-9 if n_tok <= 8 else 5e-4)


@pytest.mark.parametrize("n", [1, 40], ids=["decode", "prefill"])
def test_qkv_sharing_src1_vs_cpu_backend(n, host):
    """three MUL_MATs on the same activations (q,k Q4_K + v Q6_K, the Q4_K_M attention block): the shim hands them to
    cdna4_mul_mat_multi as one call (one decode launch); every output must match the CPU backend."""
    h, gpu, cpu = host
    k = 1024
    ws = {"q": (ob.Q4_K, 512), "k": (ob.Q4_K, 128), "v": (ob.Q6_K, 128)}
    wq = {name: h.ref.quantize(t, gaussian_weights_f32(m, k, 50 + i)) for i, (name, (t, m)) in enumerate(ws.items())}
    x = activations(n, k, 8)

    def build(ctx):
        b = h.g.ggml_new_tensor_2d(ctx, F32, k, n)
        tens = {"b": b}; outs = []
        for name, (t, m) in ws.items():
            tens[name] = h.g.ggml_new_tensor_2d(ctx, t, k, m); outs.append(h.g.ggml_mul_mat(ctx, tens[name], b))
        return tens, outs
    inp = dict(wq, b=x)
    got, sup = h.run(gpu, build, inp); want, _ = h.run(cpu, build, inp)
    assert sup
    for a, b_ in zip(got, want):
        assert nmse(a, b_) < (1e-9 if n == 1 else NMSE_VS_CPU)


def test_reduce_node(host):
    """GGML_OP_REDUCE built by the reference's own ggml_reduce (ggml.c:6166-6189): after graph_compute every src holds the sum.  Both
    partials live on device 0 here (one GPU per box); the code path -- peer buffers, stream ordering, one launch -- is the multi-device one."""
    import ctypes as C
    h, gpu, _ = host
    n = 4096
    a0 = np.random.default_rng(0).standard_normal(n).astype(np.float32); a1 = np.random.default_rng(1).standard_normal(n).astype(np.float32)
    keep = {}

    def build(ctx):
        t0 = h.g.ggml_new_tensor_1d(ctx, F32, n); t1 = h.g.ggml_new_tensor_1d(ctx, F32, n)
        arr = (C.c_void_p * 2)(t0, t1)
        r = h.g.ggml_reduce(ctx, arr, 2, 2)                      # GGML_OP_ADD == 2
        keep["t0"], keep["t1"] = t0, t1
        return {"t0": t0, "t1": t1}, [r, t0]                      # read back the result view (== t1) and the other partial
    got, sup = h.run(gpu, build, {"t0": a0, "t1": a1})
    assert sup
    assert np.array_equal(got[0], a0 + a1) and np.array_equal(got[1], a0 + a1)


def test_reduce_node_q8_0_partials(host):
    """the reference's cparams.reduce_type = q8_0 (llama-build-context.cpp builds the partial sums' REDUCE on Q8_0 copies; reduce.cu:20-43): two Q8_0 partials through
    ggml_reduce -> every src holds the re-quantized sum; with two partials the arithmetic is the reference kernel's own (x = d0 q0 + d1 q1, d = amax / 127, roundf)."""
    import ctypes as C
    h, gpu, _ = host
    Q8_0 = 8; n = 4096
    a = [np.random.default_rng(5 + j).standard_normal(n).astype(np.float32) for j in range(2)]
    q = [np.frombuffer(h.ref.quantize(Q8_0, x.reshape(1, n)), np.uint8).copy() for x in a]

    def vals(img):
        b = img.reshape(-1, 34); return b[:, :2].copy().view(np.float16).astype(np.float32) * b[:, 2:].view(np.int8).astype(np.float32)
    x = vals(q[0]) + vals(q[1]); d = np.abs(x).max(1) / np.float32(127); inv = (np.float32(1) / d).astype(np.float32)
    qq = np.sign(x * inv[:, None]) * np.floor(np.abs(x * inv[:, None]) + np.float32(0.5))
    want = np.zeros((n // 32, 34), np.uint8); want[:, :2] = d.astype(np.float16).view(np.uint8).reshape(-1, 2); want[:, 2:] = qq.astype(np.int8).view(np.uint8)

    def build(ctx):
        t0 = h.g.ggml_new_tensor_1d(ctx, Q8_0, n); t1 = h.g.ggml_new_tensor_1d(ctx, Q8_0, n)
        arr = (C.c_void_p * 2)(t0, t1)
        return {"t0": t0, "t1": t1}, [h.g.ggml_reduce(ctx, arr, 2, 2), t0]
    got, sup = h.run(gpu, build, {"t0": q[0], "t1": q[1]})
    assert sup
    for g_ in got:
        assert np.array_equal(np.asarray(g_).view(np.uint8).reshape(-1), want.reshape(-1))


def test_unsupported_ops_are_declined(host):
    """supports_op must be false for anything off the hot path so the scheduler keeps it on its own backend."""
    h, gpu, _ = host

    def build(ctx):     # a big dense f32 x f32 mat-mul: not a quantized weight and not a router-sized one
        a = h.g.ggml_new_tensor_2d(ctx, F32, 64, 2048); b = h.g.ggml_new_tensor_2d(ctx, F32, 64, 2)
        return {"a": a, "b": b}, h.g.ggml_mul_mat(ctx, a, b)
    g = h.g
    ctx = g.ggml_init(h.ref.InitParams(g.ggml_tensor_overhead() * 8 + (1 << 16), None, True))
    _, out = build(ctx)
    assert not g.ggml_backend_supports_op(gpu, out)
    g.ggml_free(ctx)


@pytest.mark.parametrize("t", ob.R4_TYPES, ids=lambda t: ob.NAMES[t])
@pytest.mark.parametrize("n", [1, 40], ids=["decode", "prefill"])
def test_r4_weights_retiled_at_upload(t, n, host, oracle):
    """_R4 tensors (offline-repacked GGUFs): set_tensor un-interleaves them ONCE into the base tiling (no shadow copy, no
    pointer-keyed cache), the mat-mul keeps the _R4 kernels' activation arithmetic, get_tensor hands the interleaved bytes back unchanged."""
    import ctypes as C
    h, gpu, cpu = host
    m, k = 64, 1024
    base = ob.BASE_OF[t]
    w = oracle.repack_r4(base, h.ref.quantize(base, gaussian_weights_f32(m, k, 60 + t)), k); x = activations(n, k, 61)
    back = {}

    def build(ctx):
        a = h.g.ggml_new_tensor_2d(ctx, t, k, m); b = h.g.ggml_new_tensor_2d(ctx, F32, k, n)
        back["a"] = a
        return {"a": a, "b": b}, h.g.ggml_mul_mat(ctx, a, b)

    class Probe:            # read the weight tensor back after the compute, before the buffer is freed
        pass
    orig_free = h.g.ggml_backend_buffer_free

    def grab(buf):
        out = np.empty_like(w); h.g.ggml_backend_tensor_get(back["a"], out.ctypes.data_as(C.c_void_p), 0, out.nbytes); back["bytes"] = out
        orig_free(buf)
    h.g.ggml_backend_buffer_free = grab
    try:
        got, sup = h.run(gpu, build, {"a": w, "b": x})
    finally:
        h.g.ggml_backend_buffer_free = orig_free
    want, _ = h.run(cpu, build, {"a": w, "b": x})
    assert sup
    assert np.array_equal(back["bytes"], w)                    # exact round trip of the interleaved file bytes
    assert nmse(got, want) < (1e-9 if n == 1 else NMSE_VS_CPU)


def test_r4_weights_uploaded_in_pieces(host, oracle):
    """the model loader uploads big tensors in chunks (llama-model-loader.cpp:1204-1240): a piecewise-written _R4 tensor is re-tiled at its
    first use; a later partial overwrite goes back through the file layout"""
    import ctypes as C
    h, gpu, cpu = host
    t, m, k = ob.R4_OF[ob.Q4_K], 64, 1024
    w1 = oracle.repack_r4(ob.Q4_K, h.ref.quantize(ob.Q4_K, gaussian_weights_f32(m, k, 71)), k)
    w2 = oracle.repack_r4(ob.Q4_K, h.ref.quantize(ob.Q4_K, gaussian_weights_f32(m, k, 72)), k)
    x = activations(1, k, 73)
    g = h.g
    ctx = g.ggml_init(h.ref.InitParams(g.ggml_tensor_overhead() * 16 + g.ggml_graph_overhead() + (1 << 16), None, True))
    a = g.ggml_new_tensor_2d(ctx, t, k, m); b = g.ggml_new_tensor_2d(ctx, F32, k, 1); o = g.ggml_mul_mat(ctx, a, b)
    gf = g.ggml_new_graph(ctx); g.ggml_build_forward_expand(gf, o)
    buf = g.ggml_backend_alloc_ctx_tensors(ctx, gpu)
    g.ggml_backend_tensor_set(b, x.ctypes.data_as(C.c_void_p), 0, x.nbytes)

    def upload(w, pieces):
        flat = np.ascontiguousarray(w).reshape(-1); step = (flat.size // pieces + 143) // 144 * 144
        for o0 in range(0, flat.size, step):
            part = np.ascontiguousarray(flat[o0:o0 + step]); g.ggml_backend_tensor_set(a, part.ctypes.data_as(C.c_void_p), o0, part.nbytes)

    def compute():
        assert g.ggml_backend_graph_compute(gpu, gf) == 0
        r = np.empty(m, np.float32); g.ggml_backend_tensor_get(o, r.ctypes.data_as(C.c_void_p), 0, r.nbytes); return r
    def cpu_ref(w):
        def build(c2):
            a2 = g.ggml_new_tensor_2d(c2, t, k, m); b2 = g.ggml_new_tensor_2d(c2, F32, k, 1)
            return {"a": a2, "b": b2}, g.ggml_mul_mat(c2, a2, b2)
        return h.run(cpu, build, {"a": w, "b": x})[0]
    upload(w1, 3); r1 = compute()
    assert nmse(r1, cpu_ref(w1)) < 1e-9
    upload(w2, 2); r2 = compute()                              # overwrite a tensor that is already re-tiled
    assert nmse(r2, cpu_ref(w2)) < 1e-9
    g.ggml_backend_buffer_free(buf); g.ggml_free(ctx)


@pytest.mark.parametrize("n_tok", [1, 4, 40], ids=["decode1", "decode4", "prefill"])
def test_moe_block_two_nodes_vs_cpu_backend(n_tok, host):
    """MOE_FUSED_UP_GATE followed by the MUL_MAT_ID that consumes it with the same ids (the expert FFN block): for decode-size batches the shim
    runs both graph nodes in ONE C-ABI call like ggml_cuda_moe_up_gate_unary does (ggml-cuda.cu:3062-3185); both outputs must match the CPU."""
    h, gpu, cpu = host
    t, ff, k, n_expert, n_used = ob.Q4_K, 512, 256, 4, 2
    wu = np.stack([h.ref.quantize(t, gaussian_weights_f32(ff, k, 80 + e) * 20) for e in range(n_expert)])
    wg = np.stack([h.ref.quantize(t, gaussian_weights_f32(ff, k, 90 + e) * 20) for e in range(n_expert)])
    wd = np.stack([h.ref.quantize(ob.Q6_K, gaussian_weights_f32(k, ff, 100 + e)) for e in range(n_expert)])
    x = activations(n_tok, k, 9).reshape(n_tok, 1, k)
    ids = np.stack([np.random.default_rng(20 + i).permutation(n_expert)[:n_used] for i in range(n_tok)]).astype(np.int32)

    def build(ctx):
        u = h.g.ggml_new_tensor_3d(ctx, t, k, ff, n_expert); g = h.g.ggml_new_tensor_3d(ctx, t, k, ff, n_expert); d = h.g.ggml_new_tensor_3d(ctx, ob.Q6_K, ff, k, n_expert)
        b = h.g.ggml_new_tensor_3d(ctx, F32, k, 1, n_tok); i = h.g.ggml_new_tensor_2d(ctx, I32, n_used, n_tok)
        f = h.g.ggml_moe_up_gate_ext(ctx, u, g, b, i, None, None, 10)
        return {"u": u, "g": g, "d": d, "b": b, "i": i}, [h.g.ggml_mul_mat_id(ctx, d, f, i), f]
    inp = {"u": wu, "g": wg, "d": wd, "b": x, "i": ids}
    got, sup = h.run(gpu, build, inp); want, _ = h.run(cpu, build, inp)
    assert sup
    for a, b_ in zip(got, want):
        assert nmse(a, b_) < (1e-9 if n_tok <= 8 else NMSE_VS_CPU)
