#!/bin/bash
# developer helper: node trace (GGML_CDNA4_TRACE) of one decode step of the tiny dense model
cd /root/repo
python - <<'PY'
import sys; sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import gguf_synth as gs
from oracle import bindings as ob
gs.tiny_model('/tmp/dense.gguf', ob.Ref(), n_vocab=512)
PY
mkdir -p gpurun_out
GGML_CDNA4_TRACE=1 LLAMA_LOGITS_KV_OFFLOAD=1 timeout 120 oracle/_ref/llama/bin/llama_logits /tmp/dense.gguf 99 ${1:-4} 8 none /tmp/o.bin 1 > gpurun_out/trace.log 2>&1
echo rc=$?
grep "^cdna4\[" gpurun_out/trace.log | tail -${2:-40}
