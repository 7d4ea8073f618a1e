#!/bin/bash
# developer helper: one -sm graph run of the logits harness on a synthetic MoE model with two logical devices, full log kept
cd /root/repo
python - <<'PY'
import sys; sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import gguf_synth as gs
from oracle import bindings as ob
gs.tiny_model('/tmp/moe.gguf', ob.Ref(), n_vocab=512, n_expert=4, n_used=2, seed=2)
PY
mkdir -p gpurun_out
GGML_CDNA4_FAKE_DEVICES=2 GGML_CDNA4_LOG_UNSUPPORTED=1 LLAMA_LOGITS_KV_OFFLOAD=1 timeout 120 oracle/_ref/llama/bin/llama_logits /tmp/moe.gguf 99 ${1:-48} 8 graph /tmp/o.bin 2 > gpurun_out/sm_graph.log 2>&1
echo rc=$?
grep -v "llama_model_loader: - kv\|print_info" gpurun_out/sm_graph.log | tail -${2:-60}
