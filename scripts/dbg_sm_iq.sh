#!/bin/bash
# developer helper: node trace of a -sm graph decode step on the sub-4-bit tiny model, two logical devices
cd /root/repo
python - <<'PY'
import sys; sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import gguf_synth as gs
from oracle import bindings as ob
def iq_mix(name, il, nl):
    return {"attn_q": gs.IQ2_S, "attn_k": gs.IQ4_NL, "attn_v": gs.Q6_K, "attn_output": gs.IQ3_S, "ffn_gate": gs.IQ2_S, "ffn_up": gs.IQ2_S, "ffn_down": gs.Q5_K, "output": gs.Q6_K, "token_embd": gs.Q4_K}[name]
gs.tiny_model('/tmp/iq.gguf', ob.Ref(), n_vocab=512, types=iq_mix, seed=1)
PY
mkdir -p gpurun_out
GGML_CDNA4_FAKE_DEVICES=2 GGML_CDNA4_TRACE=1 LLAMA_LOGITS_KV_OFFLOAD=1 timeout 120 oracle/_ref/llama/bin/llama_logits /tmp/iq.gguf 99 4 8 graph /tmp/o.bin 1 > gpurun_out/sm_iq.log 2>&1
echo rc=$?
grep "^cdna4\[" gpurun_out/sm_iq.log | tail -${1:-70} | cut -c1-200
