#!/bin/bash
# round-3 GPU call 1: (a) the chain probe, (b) the driver's exact bench command with clocks sampled + A/B against the round-1 library,
# (c) a short bench like the builder's round-2 runs, (d) Qwen3-shaped GGUF through the shim with unsupported ops logged
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3a; mkdir -p $O
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O2 scripts/probes/chain_probe.hip -o /tmp/chain_probe > $O/chain_build.log 2>&1
timeout 150 /tmp/chain_probe 129 20 9 33 66 > $O/chain_probe.log 2>&1; echo "chain_probe rc=$?" >> $O/chain_probe.log
AB=$(ls ik_llama.cpp_amd/build/ab_*/libggml-hip-cdna4.so 2>/dev/null | head -1)
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 ${AB:+--ab-lib $AB} > $O/bench_driver.json 2> $O/bench_driver.err; echo "rc=$?" >> $O/bench_driver.err
timeout 300 python bench.py --steps 5 --warmup 1 --no-extra-configs --no-llama-bench --no-cpu-baseline --no-pmc > $O/bench_short.json 2> $O/bench_short.err; echo "rc=$?" >> $O/bench_short.err
python - > $O/qwen3_shim.log 2>&1 <<'PY'
import os, subprocess, sys
sys.path.insert(0, "tests")
import gguf_synth
p = "/tmp/qwen3.gguf"; gguf_synth.qwen3_06b_model(p)
env = dict(os.environ, GGML_CDNA4_LOG_UNSUPPORTED="1", GGML_CDNA4_STATS="1")
r = subprocess.run(["oracle/_ref/llama/bin/llama-bench", "-m", p, "-p", "128", "-n", "32", "-ngl", "99", "-fa", "1", "-t", "8", "-r", "3"], env=env, capture_output=True, timeout=300)
print(r.returncode); print(r.stdout.decode()[-2000:]); 
err = r.stderr.decode().splitlines()
seen = {}
for l in err:
    seen[l] = seen.get(l, 0) + 1
for l, n in list(seen.items())[:80]:
    print(n, l)
PY
rocm-smi --showclocks --showpower --showtemp > $O/smi_end.txt 2>&1
ls /sys/class/drm/ > $O/sysfs.txt 2>&1; for d in /sys/class/drm/card*/device; do echo $d; cat $d/pp_dpm_sclk $d/pp_dpm_mclk $d/pp_dpm_fclk 2>&1 | head -40; ls $d/hwmon/*/ 2>/dev/null | head -40; done >> $O/sysfs.txt 2>&1
