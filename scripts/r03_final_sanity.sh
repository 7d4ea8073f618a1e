#!/bin/bash
# last GPU call of round 3 (torch-free, ~20 s of run): the host re-tiled GGUFs through libllama incl. the base-type twin check, then the relinked llama-bench on a tiny dense model
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
timeout 30 python tests/r4_host_llama_case.py > gpurun_out/r4h_llama2.log 2> gpurun_out/r4h_llama2.err; echo rc=$? >> gpurun_out/r4h_llama2.log
timeout 15 python - > gpurun_out/llama_bench_tiny.log 2>&1 <<'PY'
import os, subprocess, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import gguf_synth as gs
from oracle import bindings as ob
m = gs.tiny_model("/tmp/tiny_dense.gguf", ob.Ref(), n_vocab=512)
r = subprocess.run(["oracle/_ref/llama/bin/llama-bench", "-m", m, "-p", "64", "-n", "8", "-ngl", "99", "-fa", "1", "-t", "8", "-r", "2", "-o", "json"], capture_output=True, text=True, timeout=12)
print("rc", r.returncode); print(r.stdout[-1500:]); print(r.stderr[-500:])
PY
grep -c '"ok": true' gpurun_out/r4h_llama2.log; grep -v '"ok": true' gpurun_out/r4h_llama2.log | tail -5; grep -a "rc \|avg_ts\|gfx" gpurun_out/llama_bench_tiny.log | head -8
