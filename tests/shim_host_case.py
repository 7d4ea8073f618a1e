"""Stand-alone driver (child process of tests/test_shim_host_logic.py, started with LD_PRELOAD=<tests/fake_hip build>): the HOST logic of the backend shim on a stand-in HIP
runtime ("device" memory is host memory, kernel launches do nothing), with the real reference libggml as the host.  Checked here, without a GPU:

* host re-tiled interleaved weight types (18): a complete upload leaves the BASE-type bytes in the device buffer and get_tensor hands back the FILE bytes (whole and by
  pieces); a piecewise upload stays in the file layout until the tensor's first use in a graph, then becomes base bytes; a later partial overwrite patches the covered row groups in place (the tensor stays re-tiled); formerly it went back through the
  file layout; tensor copies between device buffers keep the tiling state;
* supports_op decisions for the interleaved types (row multiples, buffer ownership, unsupported forms, GET_ROWS);
* the split buffer type of `-sm graph` (two logical devices): a K-split (every split gets a column range of every row group, with a copy of the group's row scales for the
  `_KS` types) and a row split of interleaved tensors -- every split holds the BASE-type bytes of its slice after the upload, get_tensor gathers the FILE bytes back.
Mat-mul RESULTS are not looked at (no kernels run): parity of the same paths on an MI355X is tests/test_gpu_r4_host.py.  Exit code 0 = every case passed."""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
from common import gaussian_weights_f32  # noqa: E402
from oracle import bindings as ob  # noqa: E402
from r4_host_case import R4_HOST, interleave  # noqa: E402

F32, I32 = 0, 26
ROWS = {name: (8 if name.endswith("_r8") else 4) for name in R4_HOST}


def main():
    from ggml_host import GgmlHost
    h = GgmlHost(); g = h.g
    g.ggml_get_data.restype = C.c_void_p; g.ggml_get_data.argtypes = [C.c_void_p]
    g.ggml_backend_tensor_copy.restype = None; g.ggml_backend_tensor_copy.argtypes = [C.c_void_p, C.c_void_p]
    lib = C.CDLL(os.path.join(os.path.dirname(HERE), "ik_llama.cpp_amd", "libggml-hip-cdna4.so"))
    lib.cdna4_retile_r4_host.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int]
    assert h.shim.ggml_backend_cuda_get_device_count() in (1, 2), "not running on the stand-in runtime (LD_PRELOAD missing?)"
    gpu = h.shim.ggml_backend_cuda_init(0, None, None); cpu = g.ggml_backend_cpu_init()
    assert gpu
    failures = 0

    def report(case, ok, **kw):
        nonlocal failures
        failures += 0 if ok else 1
        print(json.dumps(dict(case=case, ok=bool(ok), **kw)), flush=True)

    def raw(t, n):          # the bytes in the "device" buffer as they are
        return np.ctypeslib.as_array((C.c_uint8 * n).from_address(g.ggml_get_data(t))).copy()

    def get(t, n, off=0):
        out = np.empty(n, np.uint8); g.ggml_backend_tensor_get(t, out.ctypes.data_as(C.c_void_p), off, n); return out

    def put(t, arr, off=0):
        arr = np.ascontiguousarray(arr); g.ggml_backend_tensor_set(t, arr.ctypes.data_as(C.c_void_p), off, arr.nbytes)

    m, k = 64, 1024
    for name, (base, r) in R4_HOST.items():
        wb = h.ref.quantize(base, gaussian_weights_f32(m, k, 100 + base)); wf = interleave(lib, r, wb, k); n = wf.nbytes
        wb2 = h.ref.quantize(base, gaussian_weights_f32(m, k, 200 + base)); wf2 = interleave(lib, r, wb2, k)
        ctx = g.ggml_init(h.ref.InitParams(g.ggml_tensor_overhead() * 16 + g.ggml_graph_overhead() + (1 << 16), None, True))
        a = g.ggml_new_tensor_2d(ctx, r, k, m); a2 = g.ggml_new_tensor_2d(ctx, r, k, m); b = g.ggml_new_tensor_2d(ctx, F32, k, 1); o = g.ggml_mul_mat(ctx, a, b)
        gf = g.ggml_new_graph(ctx); g.ggml_build_forward_expand(gf, o)
        buf = g.ggml_backend_alloc_ctx_tensors(ctx, gpu)
        # complete upload: base bytes on the device, file bytes back to the host (whole, and a piece from the middle)
        put(a, wf)
        ok = np.array_equal(raw(a, n), wb.reshape(-1)) and np.array_equal(get(a, n), wf.reshape(-1)) and np.array_equal(get(a, 1000, 4096), wf.reshape(-1)[4096:5096])
        # piecewise upload into a FRESH tensor: file layout until the first use in a graph ...
        pieces = 3; step = (n // pieces + 255) // 256 * 256
        for o0 in range(0, n, step):
            put(a2, wf2.reshape(-1)[o0:o0 + step], o0)
        ok = ok and np.array_equal(raw(a2, n), wf2.reshape(-1)) and np.array_equal(get(a2, n), wf2.reshape(-1))
        # ... and over an already re-tiled one: every piece patches the row groups it covers, the tensor stays re-tiled
        for o0 in range(0, n, step):
            put(a, wf2.reshape(-1)[o0:o0 + step], o0)
        ok = ok and np.array_equal(raw(a, n), wb2.reshape(-1))
        sup = bool(g.ggml_backend_supports_op(gpu, o))
        ok = ok and sup and g.ggml_backend_graph_compute(gpu, gf) == 0
        ok = ok and np.array_equal(raw(a, n), wb2.reshape(-1)) and np.array_equal(get(a, n), wf2.reshape(-1))
        # a partial overwrite of a re-tiled tensor patches the interleaved row groups it touches and leaves the tensor re-tiled: the first piece of the other file (it ends
        # in the middle of a group) over the second file, then a few bytes from the middle of a group
        put(a, wf.reshape(-1)[:step], 0)
        mixed = wf2.reshape(-1).copy(); mixed[:step] = wf.reshape(-1)[:step]
        put(a, wf.reshape(-1)[step + 777:step + 777 + 999], step + 777); mixed[step + 777:step + 777 + 999] = wf.reshape(-1)[step + 777:step + 777 + 999]
        mixed_base = np.empty_like(wf)
        assert lib.cdna4_retile_r4_host(r, mixed.ctypes.data_as(C.c_void_p), mixed_base.ctypes.data_as(C.c_void_p), m, k, 1, 1) == 0
        patch = dict(raw=bool(np.array_equal(raw(a, n), mixed_base.reshape(-1))), get=bool(np.array_equal(get(a, n), mixed)),
                     piece=bool(np.array_equal(get(a, 3001, step - 1500), mixed[step - 1500:step + 1501])))
        ok = ok and all(patch.values())
        # copies between device tensors keep the state: a re-tiled source makes a re-tiled destination
        put(a, wf); g.ggml_backend_tensor_copy(a, a2)
        ok = ok and np.array_equal(raw(a2, n), wb.reshape(-1)) and np.array_equal(get(a2, n), wf.reshape(-1))
        report("%s set / get / piecewise / overwrite / copy" % name, ok, supported=sup, **({} if ok else {"patch": patch}))
        g.ggml_backend_buffer_free(buf); g.ggml_free(ctx)

    # supports_op decisions
    def supported(build, backend_for_alloc=gpu):
        ctx = g.ggml_init(h.ref.InitParams(g.ggml_tensor_overhead() * 16 + g.ggml_graph_overhead() + (1 << 16), None, True))
        out = build(ctx); buf = g.ggml_backend_alloc_ctx_tensors(ctx, backend_for_alloc)
        s = bool(g.ggml_backend_supports_op(gpu, out)); g.ggml_backend_buffer_free(buf); g.ggml_free(ctx); return s

    def mm(t, rows, kk=1024, n=1):
        return lambda ctx: g.ggml_mul_mat(ctx, g.ggml_new_tensor_2d(ctx, t, kk, rows), g.ggml_new_tensor_2d(ctx, F32, kk, n))
    for name, (base, r) in R4_HOST.items():
        R = ROWS[name]
        report("supports_op %s" % name, supported(mm(r, 64)) and supported(mm(r, 64, n=40)) and not supported(mm(r, 64 + R // 2))          # whole row groups only
               and not supported(mm(r, 64), backend_for_alloc=cpu))                                                                        # re-tiled bytes exist only in our buffers
    report("supports_op base types / unsupported interleaved forms", supported(mm(ob.IQ4_K, 66)) and supported(mm(ob.Q4_K, 64), backend_for_alloc=cpu)
           and not supported(mm(219, 64)) and not supported(mm(229, 64)))                      # IQ1_S_R4, IQ1_M_R4: formats of their own
    report("supports_op MUL_MAT_ID over interleaved experts / GET_ROWS of an interleaved table",
           supported(lambda ctx: g.ggml_mul_mat_id(ctx, g.ggml_new_tensor_3d(ctx, 339, 512, 128, 8), g.ggml_new_tensor_3d(ctx, F32, 512, 2, 5), g.ggml_new_tensor_2d(ctx, I32, 2, 5)))
           and not supported(lambda ctx: g.ggml_get_rows(ctx, g.ggml_new_tensor_2d(ctx, 339, 512, 64), g.ggml_new_tensor_1d(ctx, I32, 7)))
           and supported(lambda ctx: g.ggml_get_rows(ctx, g.ggml_new_tensor_2d(ctx, ob.IQ4_K, 512, 64), g.ggml_new_tensor_1d(ctx, I32, 7))))
    # ---- split buffers (ggml-cuda.cu:852-1402 behaviour): parent tensor in the split buffer type, t->extra = {n_device, split_dim, tensor, splits[]}
    offs = os.environ.get("SHIM_CASE_TENSOR_OFFSETS")          # "offsetof(extra) sizeof(ggml_tensor)" from the reference's ggml.h (tests/test_shim_host_logic.py compiles the probe)
    if offs and h.shim.ggml_backend_cuda_get_device_count() >= 2:
        off_extra = int(offs.split()[0])

        class SplitExtra(C.Structure):
            _fields_ = [("n_device", C.c_int), ("split_dim", C.c_int), ("tensor", C.c_void_p), ("splits", C.POINTER(C.c_void_p))]
        h.shim.ggml_backend_cuda_split_buffer_type.restype = C.c_void_p; h.shim.ggml_backend_cuda_split_buffer_type.argtypes = [C.c_void_p]
        g.ggml_backend_alloc_ctx_tensors_from_buft.restype = C.c_void_p; g.ggml_backend_alloc_ctx_tensors_from_buft.argtypes = [C.c_void_p, C.c_void_p]
        ts = (C.c_float * 16)(*([0.5, 0.5] + [0.0] * 14)); buft = h.shim.ggml_backend_cuda_split_buffer_type(ts)
        m, k = 64, 2048
        for name, (base, r) in R4_HOST.items():
            meta = ob.ROW_META.get(base, 0); bs, blck = ob.TYPE_SIZE[base], ob.BLCK[base]
            wb = h.ref.quantize(base, gaussian_weights_f32(m, k, 300 + base)); wf = interleave(lib, r, wb, k)
            for dim, parts in ((0, (768, 1280)), (1, (24, 40))):          # K split (columns) / row split: two uneven parts, whole blocks / whole row groups
                # like llama-load-tensors.cpp:4630-4699: the parent first, its per-device tensors after it in the SAME context (the allocator sizes the buffer from them, the
                # parent's init_tensor gives them their device memory before the allocator reaches them)
                ctx = g.ggml_init(h.ref.InitParams(g.ggml_tensor_overhead() * 8 + (1 << 12), None, True))
                t = g.ggml_new_tensor_2d(ctx, r, k, m)
                sp = [g.ggml_new_tensor_2d(ctx, r, p_, m) if dim == 0 else g.ggml_new_tensor_2d(ctx, r, k, p_) for p_ in parts]
                arr = (C.c_void_p * 2)(*sp); ex = SplitExtra(2, dim, t, C.cast(arr, C.POINTER(C.c_void_p)))
                C.c_void_p.from_address(t + off_extra).value = C.addressof(ex)
                buf = g.ggml_backend_alloc_ctx_tensors_from_buft(ctx, buft)
                put(t, wf)
                ok = bool(buf); acc = 0
                for s_, p_ in zip(sp, parts):
                    if dim == 0:      # base rows of the slice: [row meta][blocks acc / blck ... (acc + p_) / blck)
                        rows = wb[:, :meta], wb[:, meta + (acc // blck) * bs: meta + ((acc + p_) // blck) * bs]
                        want = np.concatenate(rows, axis=1).reshape(-1)
                    else:
                        want = wb[acc:acc + p_].reshape(-1)
                    ok = ok and np.array_equal(raw(s_, want.size), want); acc += p_
                ok = ok and np.array_equal(get(t, wf.nbytes), wf.reshape(-1))
                report("%s split buffer, split_dim %d" % (name, dim), ok)
                g.ggml_backend_buffer_free(buf); g.ggml_free(ctx)
        # ---- merged ffn_gate_up_exps (-muge, llama-load-tensors.cpp:4403-4470): ONE row-split parent [K, 2 n_ff, n_expert] with explicit ranges per device, the file's
        # ffn_gate_exps / ffn_up_exps loaded through VIEWS of it (ggml-cuda.cu:890-968).  Every split must hold [its gate rows ; its up rows] of every expert afterwards.
        if len(offs.split()) >= 5:
            off_params, off_name = int(offs.split()[2]), int(offs.split()[4])
            K2, NF, NE, parts = 512, 64, 3, (24, 40)
            g.ggml_set_name.restype = C.c_void_p; g.ggml_set_name.argtypes = [C.c_void_p, C.c_char_p]
            for ty, tag in ((ob.Q4_K, "Q4_K"), (212, "Q4_K_R4")):
                base = ty if ty < 200 else ty - 200
                wg = h.ref.quantize(base, gaussian_weights_f32(NF * NE, K2, 910)).reshape(NE, NF, -1); wu = h.ref.quantize(base, gaussian_weights_f32(NF * NE, K2, 911)).reshape(NE, NF, -1)
                if ty >= 200:       # the file holds 4-row interleaved experts
                    orc = ob.Oracle(); wg_f = np.stack([orc.repack_r4(base, wg[e], K2) for e in range(NE)]); wu_f = np.stack([orc.repack_r4(base, wu[e], K2) for e in range(NE)])
                else:
                    wg_f, wu_f = wg, wu
                ctx = g.ggml_init(h.ref.InitParams(g.ggml_tensor_overhead() * 12 + (1 << 12), None, True))
                t = g.ggml_new_tensor_3d(ctx, ty, K2, 2 * NF, NE); g.ggml_set_name(t, b"blk.0.ffn_gate_up_exps.weight")
                sp = [g.ggml_new_tensor_3d(ctx, ty, K2, 2 * p_, NE) for p_ in parts]
                arr = (C.c_void_p * 2)(*sp); ex = SplitExtra(2, 1, t, C.cast(arr, C.POINTER(C.c_void_p)))
                C.c_void_p.from_address(t + off_extra).value = C.addressof(ex)
                # std::vector<std::vector<std::pair<int, int>>> as libstdc++ lays it out: {begin, end, end_of_storage}
                class Vec(C.Structure):
                    _fields_ = [("b", C.c_void_p), ("e", C.c_void_p), ("c", C.c_void_p)]
                pairs = [(C.c_int * 4)(0, parts[0], NF, parts[0]), (C.c_int * 4)(parts[0], parts[1], NF + parts[0], parts[1])]
                inner = (Vec * 2)(*[Vec(C.addressof(p_), C.addressof(p_) + 16, C.addressof(p_) + 16) for p_ in pairs])
                outer = Vec(C.addressof(inner), C.addressof(inner) + C.sizeof(inner), C.addressof(inner) + C.sizeof(inner))
                C.c_void_p.from_address(t + off_params).value = C.addressof(outer)
                g.ggml_view_3d.restype = C.c_void_p; g.ggml_view_3d.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_size_t, C.c_size_t, C.c_size_t]
                rowb = wg_f.shape[-1]
                vg = g.ggml_view_3d(ctx, t, K2, NF, NE, rowb, rowb * NF, 0); g.ggml_set_name(vg, b"blk.0.ffn_gate_exps.weight")
                vu = g.ggml_view_3d(ctx, t, K2, NF, NE, rowb, rowb * NF, rowb * NF * NE); g.ggml_set_name(vu, b"blk.0.ffn_up_exps.weight")
                buf = g.ggml_backend_alloc_ctx_tensors_from_buft(ctx, buft)
                put(vg, wg_f); put(vu, wu_f)
                ok = bool(buf); acc = 0
                for s_, p_ in zip(sp, parts):
                    want = np.concatenate([np.concatenate([wg_f[e, acc:acc + p_], wu_f[e, acc:acc + p_]]) for e in range(NE)]).reshape(-1)
                    ok = ok and np.array_equal(raw(s_, want.size), want); acc += p_
                report("merged ffn_gate_up_exps views into a row-split parent (%s)" % tag, ok)
                g.ggml_backend_buffer_free(buf); g.ggml_free(ctx)
    g.ggml_backend_free(gpu); g.ggml_backend_free(cpu)
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
