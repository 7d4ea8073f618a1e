"""debug: where does cdna4_op_norm_rope_store_kv differ from rms_norm + rope?  (position 0: cos = 1, sin = 0 -> the norm stage alone)"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import nt_bench as nb
P, I, L64, F = C.c_void_p, C.c_int, C.c_long, C.c_float
hip = nb.Hip(); lib = nb.load_lib(os.path.join(ROOT, "ik_llama.cpp_amd", "libggml-hip-cdna4.so")); TP = C.POINTER(nb.Tensor)
lib.cdna4_op_norm_rope_store_kv.argtypes = [P, TP, TP, F, TP, TP, TP, F, TP, TP, P, TP, TP, P, P, P, I, I, I, F, F, F, F, F, F, P]
lib.cdna4_op_rms_norm.argtypes = [P, TP, TP, F, TP, P]; lib.cdna4_op_rope.argtypes = [P, TP, P, P, TP, I, I, I, F, F, F, F, F, F, P]
lib.cdna4_op_rope_cache.argtypes = [P, P, L64, P, I, I, F, F, F, F, F, F, P]
ctx = lib.cdna4_init(0)
hd, n_head, n_kv, n_tok = 128, 16, 8, 4
nq, nk = hd * n_head, hd * n_kv
rng = np.random.default_rng(1)
xq = (rng.standard_normal((n_tok, nq)) * 2.5).astype(np.float32); xk = rng.standard_normal((n_tok, nk)).astype(np.float32); xv = rng.standard_normal((n_tok, nk)).astype(np.float32)
up = lambda a: hip.upload(a).value; ml = lambda n: hip.malloc(n).value
for mode in (0, 2):
    for p0 in (0, 17):
        for wscale in (0.0, 0.2):
            w = (1 + wscale * rng.standard_normal(hd)).astype(np.float32); pos = np.full(n_tok, p0, np.int32)
            qd, kd, vd, wd, pd = up(xq), up(xk), up(xv), up(w), up(pos); nqd, rq1, rq2, kc, vc = ml(4 * nq * n_tok), ml(4 * nq * n_tok), ml(4 * nq * n_tok), ml(2 * nk * n_tok), ml(2 * nk * n_tok)
            t3 = lambda p, h: nb.tensor(p, 0, [hd, h, n_tok, 1], 4)
            tw = nb.tensor(wd, 0, [hd, 1, 1, 1], 4); tv = nb.tensor(vd, 0, [nk, n_tok, 1, 1], 4); tkc, tvc = nb.tensor(kc, 1, [nk, n_tok, 1, 1], 2), nb.tensor(vc, 1, [nk, n_tok, 1, 1], 2)
            rp = (hd, mode, 40960, 1e6, 1.0, 0.0, 1.0, 32.0, 1.0)
            assert lib.cdna4_op_rope_cache(ctx, pd, n_tok, None, hd, 40960, 1e6, 1.0, 0.0, 1.0, 32.0, 1.0, None) == 0
            assert lib.cdna4_op_rms_norm(ctx, C.byref(t3(qd, n_head)), C.byref(tw), 1e-6, C.byref(t3(nqd, n_head)), None) == 0
            assert lib.cdna4_op_rope(ctx, C.byref(t3(nqd, n_head)), pd, None, C.byref(t3(rq1, n_head)), *rp, None) == 0
            assert lib.cdna4_op_norm_rope_store_kv(ctx, C.byref(t3(qd, n_head)), C.byref(tw), 1e-6, C.byref(t3(rq2, n_head)), C.byref(t3(kd, n_kv)), C.byref(tw), 1e-6, None, C.byref(tkc), None, C.byref(tv), C.byref(tvc),
                                                   None, pd, None, *rp, None) == 0, lib.cdna4_last_error()
            hip.check(hip.h.hipDeviceSynchronize(), "sync")
            n1 = hip.download(nqd, (n_tok * n_head, hd), np.float32); a = hip.download(rq1, (n_tok * n_head, hd), np.float32); b = hip.download(rq2, (n_tok * n_head, hd), np.float32)
            bad = a.view(np.uint32) != b.view(np.uint32)
            print("mode %d pos %2d wscale %.1f: %5d / %d differ; rows with a difference %d / %d; norm-only vs fused (pos 0 only meaningful) %d" % (mode, p0, wscale, bad.sum(), bad.size, bad.any(axis=1).sum(), bad.shape[0],
                  (n1.view(np.uint32) != b.view(np.uint32)).sum()), flush=True)
            if bad.any():
                r, c = np.argwhere(bad)[0]; print("   first: row %d col %d  six %r one %r   cols differing in that row: %s" % (r, c, a[r, c], b[r, c], np.flatnonzero(bad[r])[:16]))
