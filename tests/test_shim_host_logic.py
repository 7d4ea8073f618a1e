"""CPU-side coverage of the backend shim's HOST logic (buffer set / get / re-tiling state machine of the host re-tiled interleaved weight types, supports_op decisions): the
shim and the C-ABI library run on a stand-in HIP runtime (tests/fake_hip/fake_hip.cpp: "device" memory is host memory, kernel launches do nothing) that a child process preloads;
the real reference libggml is the host, as in the -m gpu tests.  No result of a kernel is looked at here -- parity of the same paths on an MI355X is tests/test_gpu_r4_host.py."""
import os
import shutil
import subprocess
import sys

import pytest

from oracle import bindings as ob

HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(HERE)
SHIM = os.path.join(ROOT, "ik_llama.cpp_amd", "backend", "libggml-cuda-cdna4.so")


@pytest.fixture(scope="module")
def fake_hip(tmp_path_factory):
    if not shutil.which("g++") or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        pytest.skip("needs g++ and the HIP headers to build the stand-in runtime")
    if ob.ref_path() is None or not os.path.exists(SHIM):
        pytest.skip("needs oracle/_ref (reference libggml) and the prebuilt backend shim")
    out = str(tmp_path_factory.mktemp("fake_hip") / "libfakehip.so")
    subprocess.check_call(["g++", "-O1", "-shared", "-fPIC", "-w", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-o", out, os.path.join(HERE, "fake_hip", "fake_hip.cpp")])
    return out


def test_interleaved_weight_state_machine_and_supports_op_on_the_stand_in_runtime(fake_hip):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a real GPU is present: tests/test_gpu_r4_host.py covers these paths with their results")
    env = dict(os.environ); env["LD_PRELOAD"] = fake_hip
    p = subprocess.run([sys.executable, os.path.join(HERE, "shim_host_case.py")], capture_output=True, text=True, timeout=600, env=env)
    print(p.stdout); print(p.stderr[-3000:], file=sys.stderr)
    assert p.returncode == 0, "child exit %d\n%s\n%s" % (p.returncode, p.stdout[-3000:], p.stderr[-3000:])
    assert p.stdout.count('"ok": true') >= 38 and '"ok": false' not in p.stdout
