"""-m gpu: the host re-tiled row-interleaved weight types -- IQ2_K_R4 IQ3_K_R4 IQ4_K_R4 IQ5_K_R4 IQ4_KS_R4 IQ5_KS_R4 (the interleaved forms the reference CUDA
backend lists for MUL_MAT, ggml-cuda.cu:4893-4898) and the CPU-only Q4_0_R8 Q5_0_R4 Q6_0_R4 Q8_0_R8 MXFP4_R8 Q2_K_R4 Q3_K_R4 IQ4_XS_R8 IQ2_XXS_R4 IQ2_XS_R4
IQ3_XXS_R4 IQ2_BN_R4 -- through the backend shim against the reference CPU backend (its base-type kernels on the un-interleaved tensor AND its own
interleaved kernels on the file bytes).  The cases live in tests/r4_host_case.py and run in a CHILD process (the shim aborts the process on an internal
error; a child keeps that away from the test session).  The host half (the re-tiling itself, cdna4_retile_r4_host) is pinned on the CPU in
tests/test_retile_host.py.  First run on an MI355X: profiles/r03_r4_host_cases.log (40 / 40 cases)."""
import os
import subprocess
import sys

import pytest

from oracle import bindings as ob

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
NAMES = ["iq2_k_r4", "iq3_k_r4", "iq4_k_r4", "iq5_k_r4", "iq4_ks_r4", "iq5_ks_r4", "q4_0_r8", "q5_0_r4", "q6_0_r4", "q8_0_r8", "mxfp4_r8", "q2_k_r4", "q3_k_r4", "iq4_xs_r8", "iq2_xxs_r4", "iq2_xs_r4",
         "iq3_xxs_r4", "iq2_bn_r4"]


@pytest.mark.parametrize("name", NAMES)
def test_host_retiled_r4_types_through_the_shim(name):
    from ggml_host import SHIM
    if ob.ref_path() is None or not os.path.exists(SHIM):
        pytest.skip("needs oracle/_ref (reference libggml) and the prebuilt backend shim")
    p = subprocess.run([sys.executable, os.path.join(HERE, "r4_host_case.py"), name], capture_output=True, text=True, timeout=600)
    print(p.stdout); print(p.stderr[-4000:], file=sys.stderr)
    assert p.returncode == 0, "child exit %d\n%s\n%s" % (p.returncode, p.stdout[-3000:], p.stderr[-3000:])


def test_gguf_of_host_retiled_types_through_libllama():
    """GGUFs whose weight tensors are host re-tiled interleaved types (the six the CUDA backend lists in one model, ten CPU-only forms in another) through the unmodified
    libllama: -ngl 99 (the loader's uploads are re-tiled by the shim's set_tensor) against -ngl 0 (the CPU backend's own interleaved kernels); tests/r4_host_llama_case.py.
    First run on an MI355X: profiles/r03_r4_host_llama.log."""
    from ggml_host import SHIM
    if ob.ref_path() is None or not os.path.exists(SHIM) or not os.path.exists(os.path.join(os.path.dirname(HERE), "oracle", "_ref", "llama", "bin", "llama_logits")):
        pytest.skip("needs oracle/_ref (reference libggml + llama_logits) and the prebuilt backend shim")
    p = subprocess.run([sys.executable, os.path.join(HERE, "r4_host_llama_case.py")], capture_output=True, text=True, timeout=900)
    print(p.stdout); print(p.stderr[-4000:], file=sys.stderr)
    assert p.returncode == 0, "child exit %d\n%s\n%s" % (p.returncode, p.stdout[-3000:], p.stderr[-3000:])
