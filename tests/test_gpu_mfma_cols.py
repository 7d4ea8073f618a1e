"""-m gpu: the 2..8-column decode kernel on the int8 matrix cores (csrc/gemv_mfma.hip) against the oracle's CPU arithmetic -- every column count
it can serve (the dispatcher only picks it from 5 columns on; CDNA4_GEMV_MFMA_MIN_COLS=2 forces it, read once per process, hence the child
processes), both weight paths (per-lane loads / LDS-DMA staged tiles), plain and fused up*gate, K-split and chunked-K (K > 4096) shapes."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, numpy as np
sys.path.insert(0, "ROOT"); sys.path.insert(0, "ROOT/tests")
from __graft_entry__ import _load_package
from oracle import bindings as ob
from common import TOL_INT8_PATH, activations, make_weights
from test_gpu_parity import dev
be = _load_package().Cdna4Backend(0); orc = ob.Oracle()
for t in (ob.Q4_K, ob.Q5_K, ob.Q6_K):
    for (m, k) in ((200, 1024), (64, 14336), (130, 4096), (48, 256), (16, 8192)):
        w = make_weights(t, m, k, 300 + t, orc); w2 = make_weights(t, m, k, 400 + t, orc)
        for n in (2, 3, 5, 8):
            x = activations(n, k, 10 * n + t, outliers=(n == 3))
            vdt = ob.vec_dot_type(t)
            xq = orc.dequantize_activations(vdt, orc.quantize_activations(vdt, x), k)
            _, sum_abs = orc.mul_mat_f64(t, w, xq)
            got = be.mul_mat(t, dev(w), dev(x)).cpu().numpy(); cpu = orc.mul_mat(t, w, x)
            e = np.max(np.abs(got.astype(np.float64) - cpu) / sum_abs)
            assert e < TOL_INT8_PATH, ("plain", t, m, k, n, e)
            if k <= 4096 and n in (2, 8):
                g = be.fused_up_gate(t, dev(w), dev(w2), dev(x), op=10).cpu().numpy(); want = orc.fused_up_gate(t, 10, w, w2, x)
                assert np.allclose(g, want, rtol=2e-5, atol=2e-6 * np.abs(want).max()), ("fused", t, m, k, n, np.abs(g - want).max())
print("MFMA-COLS-OK")
'''.replace("ROOT", ROOT)


@pytest.mark.parametrize("lds", ["0", "1"], ids=["direct_loads", "lds_staged"])
def test_mfma_columns_all_counts(lds):
    env = dict(os.environ, CDNA4_GEMV_MFMA_MIN_COLS="2", CDNA4_GEMV_MFMA_LDS=lds, CDNA4_GEMV_MFMA_Q6_ALL="1")
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0 and b"MFMA-COLS-OK" in r.stdout, r.stderr.decode(errors="replace")[-3000:]
