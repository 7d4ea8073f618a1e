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


def tensor_offsets(tmp):
    """offsetof(ggml_tensor, extra) and sizeof(ggml_tensor) from the reference's own header (the split-buffer cases set t->extra by hand); None where the header is absent"""
    hdr = "/root/reference/ggml/include"
    if not os.path.exists(os.path.join(hdr, "ggml.h")) or not shutil.which("gcc"):
        return None
    src = os.path.join(tmp, "offs.c"); exe = os.path.join(tmp, "offs")
    open(src, "w").write('#include <stdio.h>\n#include <stddef.h>\n#include "ggml.h"\nint main(void) { printf("%zu %zu %zu %zu %zu", offsetof(struct ggml_tensor, extra), sizeof(struct ggml_tensor), offsetof(struct ggml_tensor, op_params), offsetof(struct ggml_tensor, view_src), offsetof(struct ggml_tensor, name)); return 0; }\n')
    subprocess.check_call(["gcc", "-I" + hdr, src, "-o", exe])
    return subprocess.check_output([exe]).decode()


def no_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a real GPU is present: the -m gpu tests cover these paths with their results")


def test_interleaved_weight_state_machine_supports_op_and_split_buffers_on_the_stand_in_runtime(fake_hip, tmp_path):
    no_gpu()
    env = dict(os.environ); env["LD_PRELOAD"] = fake_hip; env["GGML_CDNA4_FAKE_DEVICES"] = "2"
    offs = tensor_offsets(str(tmp_path))
    if offs:
        env["SHIM_CASE_TENSOR_OFFSETS"] = offs
    p = subprocess.run([sys.executable, os.path.join(HERE, "shim_host_case.py")], capture_output=True, text=True, timeout=600, env=env)
    print(p.stdout); print(p.stderr[-3000:], file=sys.stderr)
    assert p.returncode == 0, "child exit %d\n%s\n%s" % (p.returncode, p.stdout[-3000:], p.stderr[-3000:])
    assert p.stdout.count('"ok": true') >= (76 if offs else 38) and '"ok": false' not in p.stdout       # 18 state machines + 20 supports_op (+ 36 split-buffer cases + 2 merged gate_up view cases)


@pytest.mark.parametrize("sm,n_dev", [("none", 1), ("layer", 2), ("graph", 2)])
def test_libllama_host_paths_run_on_the_stand_in_runtime(sm, n_dev, fake_hip, tmp_path):
    """the unmodified libllama loads GGUFs (plain K-quants, a MoE, device re-tiled _R4 tensors, the two host re-tiled mixes) with every layer offloaded and walks a 48-token
    prompt + 3 decode steps through the shim -- buffer types, split buffers of `-sm graph`, REDUCE nodes, the eager graph walk with every fusion decision -- without an abort.
    Kernels do nothing here, so the logits are not looked at (tests/test_gpu_llama.py, tests/test_gpu_r4_host.py do that on an MI355X)."""
    no_gpu()
    logits = os.path.join(ROOT, "oracle", "_ref", "llama", "bin", "llama_logits")
    if not os.path.exists(logits):
        pytest.skip("oracle/_ref/llama not built")
    import ctypes as C
    import numpy as np
    sys.path.insert(0, HERE)
    import gguf_synth as gs
    import r4_host_llama_case as rl
    from conftest import load_package
    gs.add_types(ob)
    for r, b in rl.BASE_OF.items():
        gs.TYPE_SIZE[r] = ob.TYPE_SIZE[b]; gs.BLCK[r] = ob.BLCK[b]
        if b in ob.ROW_META:
            gs.ROW_META[r] = ob.ROW_META[b]
    gs.TYPE_SIZE.update({gs.Q4_K + 200: 144, gs.Q6_K + 200: 210}); gs.BLCK.update({gs.Q4_K + 200: 256, gs.Q6_K + 200: 256})
    ref = ob.Ref(); orc = ob.Oracle(); lib = load_package().load_library()

    class Quant:
        def quantize(self, t, w):
            if t in rl.BASE_OF:
                q = ref.quantize(rl.BASE_OF[t], w); out = np.empty_like(q)
                assert lib.cdna4_retile_r4_host(t, q.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), q.shape[0], w.shape[1], 0, 1) == 0
                return out
            return orc.repack_r4(t - 200, ref.quantize(t - 200, w), w.shape[1]) if t in (gs.Q4_K + 200, gs.Q6_K + 200) else ref.quantize(t, w)

    def r4_mix(name, il, nl):
        return {"token_embd": gs.Q4_K, "output": gs.Q6_K}.get(name, gs.Q4_K + 200 if name != "attn_v" else gs.Q6_K + 200)
    models = {"dense": dict(), "moe": dict(n_expert=4, n_used=2, seed=2), "r4": dict(types=r4_mix, seed=3), "cuda_listed": dict(types=rl.cuda_listed_mix, seed=11), "cpu_only": dict(types=rl.cpu_only_mix, seed=12),
              "mha": dict(n_head=4, n_head_kv=4, seed=13), "mqa": dict(n_head=8, n_head_kv=1, n_embd=1024, n_ff=2048, seed=14), "wide_moe": dict(n_expert=8, n_used=3, n_ff=512, seed=15)}
    env = dict(os.environ); env["LD_PRELOAD"] = fake_hip; env["GGML_CDNA4_FAKE_DEVICES"] = str(n_dev); env["LLAMA_LOGITS_KV_OFFLOAD"] = "1"
    env["GGML_CDNA4_CHECK_USES"] = "1"       # the walk's use index (last reader of every tensor) is answered both ways -- index and scan -- and a difference aborts
    env["GGML_CDNA4_CHECK_OVERLAP"] = "1"    # every fused launch asserts its operand layout on the host (a result over an operand another workgroup still reads: round 4's
                                             # ROPE + ROPE + KV-store race -- the allocator puts the rotated K exactly over the un-rotated Q in these very graphs; the launch
                                             # no longer writes it)
    for tag, kw in models.items():
        if tag == "mqa" and sm == "graph":
            continue        # (one KV head cannot be split over two devices: the reference's own graph builder asserts, ggml.c:6179 GGML_ASSERT(nhave > 1))
        if tag in ("moe", "wide_moe") and sm == "graph":
            continue        # (under -sm graph part of the MoE block runs on the CPU backend, which follows the router's ids -- garbage here, where no kernel runs: not a host-logic check)
        path = gs.tiny_model(str(tmp_path / (tag + ".gguf")), Quant(), n_vocab=512, **kw)
        out = str(tmp_path / "logits.bin")
        p = subprocess.run([logits, path, "99", "48", "8", sm, out, "3"], capture_output=True, env=env, timeout=300)
        assert p.returncode == 0, (tag, sm, p.returncode, p.stderr.decode(errors="replace")[-2000:])
        assert os.path.getsize(out) == 4 * 512 * 4


def test_supports_op_agrees_with_the_entry_points_on_the_stand_in_runtime(fake_hip):
    """randomized one-op graphs (13 op families: types, shapes, broadcasts, strided / permuted / offset views, parameters at and beyond the edges): whatever supports_op accepts,
    the C-ABI entry point accepts too -- the shim aborts the process otherwise (tests/shim_fuzz_case.py; four more seeds x 150 rounds ran clean when this was written)"""
    no_gpu()
    env = dict(os.environ); env["LD_PRELOAD"] = fake_hip
    p = subprocess.run([sys.executable, os.path.join(HERE, "shim_fuzz_case.py"), "60", "1"], capture_output=True, text=True, timeout=900, env=env)
    tail = p.stdout[-1500:]
    assert p.returncode == 0, "child exit %d (the last `computing ...` line names the node)\n%s\n%s" % (p.returncode, tail, p.stderr[-3000:])
    assert "offered / accepted per op" in tail and p.stdout.count("computing ") > 300


@pytest.mark.parametrize("kw", [dict(), dict(n_expert=4, n_used=2, seed=2)], ids=["dense", "moe"])
def test_decode_graph_cache_captures_once_and_replays(kw, fake_hip, tmp_path):
    """the shim's HIP-graph cache on the stand-in runtime with capture emulation (FAKE_HIP_CAPTURE=1: captures are accepted, replays are no-ops): over 8 decode steps the token graph
    is seen once eagerly, captured once and replayed from then on -- graph key, KV slot table and replay bookkeeping are host logic (results: tests/test_gpu_llama.py
    test_decode_steps_replayed_from_a_hip_graph)"""
    no_gpu()
    import re
    logits = os.path.join(ROOT, "oracle", "_ref", "llama", "bin", "llama_logits")
    if not os.path.exists(logits):
        pytest.skip("oracle/_ref/llama not built")
    sys.path.insert(0, HERE)
    import gguf_synth as gs
    path = gs.tiny_model(str(tmp_path / "m.gguf"), ob.Ref(), n_vocab=512, **kw)
    env = dict(os.environ); env.update({"LD_PRELOAD": fake_hip, "FAKE_HIP_CAPTURE": "1", "GGML_CDNA4_STATS": "1", "LLAMA_LOGITS_KV_OFFLOAD": "1"})
    p = subprocess.run([logits, path, "99", "5", "8", "none", str(tmp_path / "o.bin"), "8"], capture_output=True, env=env, timeout=300)
    err = p.stderr.decode(errors="replace")
    assert p.returncode == 0, err[-2000:]
    m = re.search(r"graph_compute calls: (\d+) eager, (\d+) captured, (\d+) replayed, (\d+) capture failures", err)
    assert m, err[-1500:]
    eager, captured, replayed, failed = (int(v) for v in m.groups())
    assert captured >= 1 and replayed >= 5 and failed == 0 and eager + captured + replayed == 9, m.group(0)
    # the per-token input uploads (one backend on the device) ride the compute stream instead of blocking
    ms = re.search(r"small uploads queued on the compute stream instead of blocking copies: (\d+)", err)
    assert ms and int(ms.group(1)) >= 9, err[-1500:]


@pytest.mark.parametrize("tag,kw,stat,per_graph", [
    ("qwen3", dict(n_embd=256, n_ff=512, n_head=4, n_head_kv=2, n_layer=2, arch="qwen3", head_dim=64, qk_norm=True, tied=True, seed=21), r"q/k norms\+ROPE\+KV stores (\d+)", 2),
    ("moe1k", dict(n_embd=1024, n_ff=256, n_head=8, n_head_kv=4, n_layer=2, n_expert=4, n_used=2, seed=22), r"RMS_NORM in MoE router (\d+)", 2)], ids=["qwen3", "moe1k"])
def test_round5_fusion_matchers_fire_on_the_stand_in_runtime(tag, kw, stat, per_graph, fake_hip, tmp_path):
    """host logic of the two graph fusions of round 5 on the stand-in runtime (kernels do nothing; node order, the matchers, the operand-layout rules and the entry points' argument
    checks are real): a Qwen3-architecture graph takes the q / k norm + ROPE + KV-store launch in the prompt graph and in every decode graph, a MoE graph with 1024-wide rows folds
    ffn_norm into the router launch in every DECODE graph (one token: the allocator hands the dead un-normed row to the router's results, which only a single workgroup may ignore).
    Results of the same graphs: tests/test_gpu_llama.py on an MI355X."""
    no_gpu()
    import re
    logits = os.path.join(ROOT, "oracle", "_ref", "llama", "bin", "llama_logits")
    if not os.path.exists(logits):
        pytest.skip("oracle/_ref/llama not built")
    sys.path.insert(0, HERE)
    import gguf_synth as gs
    path = gs.tiny_model(str(tmp_path / (tag + ".gguf")), ob.Ref(), n_vocab=512, **kw)
    env = dict(os.environ); env.update({"LD_PRELOAD": fake_hip, "GGML_CDNA4_STATS": "1", "GGML_CDNA4_PARAMS": "graphs=0", "LLAMA_LOGITS_KV_OFFLOAD": "1", "GGML_CDNA4_CHECK_OVERLAP": "1"})
    p = subprocess.run([logits, path, "99", "48", "8", "none", str(tmp_path / "o.bin"), "3"], capture_output=True, env=env, timeout=300)
    err = p.stderr.decode(errors="replace")
    assert p.returncode == 0, err[-2000:]
    m = re.search(stat, err)
    assert m, err[-1500:]
    n_graphs = 4 if tag == "qwen3" else 3          # prompt + 3 decode steps | the decode steps only
    assert int(m.group(1)) == per_graph * n_graphs, (m.group(0), per_graph * n_graphs)
