import os, sys, subprocess, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import gguf_synth as gs
from oracle import bindings as ob
from common import nmse
ref = ob.Ref()
def iq_mix(name, il, nl):
    return {"attn_q": gs.IQ2_S, "attn_k": gs.IQ4_NL, "attn_v": gs.Q6_K, "attn_output": gs.IQ3_S, "ffn_gate": gs.IQ2_S, "ffn_up": gs.IQ2_S, "ffn_down": gs.Q5_K, "output": gs.Q6_K, "token_embd": gs.Q4_K}[name]
models = {"dense": gs.tiny_model("/tmp/d.gguf", ref, n_vocab=512), "iq": gs.tiny_model("/tmp/i.gguf", ref, n_vocab=512, types=iq_mix, seed=1)}
LOG = "/root/repo/oracle/_ref/llama/bin/llama_logits"
def logits(model, ngl, sm, env):
    e = dict(os.environ); e.update(env); out = "/tmp/l.bin"
    if ngl: e["LLAMA_LOGITS_KV_OFFLOAD"] = "1"
    r = subprocess.run([LOG, model, str(ngl), "48", "8", sm, out, "3"], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-500:]
    return np.fromfile(out, np.float32).reshape(4, 512)
for name, m in models.items():
    cpu = logits(m, 0, "none", {})
    for sm, env in (("none", {}), ("graph", {"GGML_CDNA4_FAKE_DEVICES": "2"}), ("graph", {"GGML_CDNA4_FAKE_DEVICES": "2", "GGML_CDNA4_NO_NORM_MM": "1"}), ("layer", {"GGML_CDNA4_FAKE_DEVICES": "2"})):
        g = logits(m, 99, sm, env)
        print(name, sm, sorted(env), ["%.1e" % nmse(g[i], cpu[i]) for i in range(4)], flush=True)
