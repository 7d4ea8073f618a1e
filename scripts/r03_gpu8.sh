#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3h; mkdir -p $O
export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" timeout 300 python -m pytest tests/test_gpu_llama.py -q -k "more_weight_types or offloaded_vs_cpu or hip_graph" 2>&1 | grep -E "passed|failed|AssertionError: \(|^FAILED" > $O/$tag.log; }
python scripts/poison_hbm.py 0x7e > $O/poison1.log 2>&1; run p7e_default A=1
python scripts/poison_hbm.py 0xfb > $O/poison2.log 2>&1; run pfb_default A=1
python scripts/poison_hbm.py 0x7e >> $O/poison1.log 2>&1; run p7e_fusion0 GGML_CDNA4_PARAMS=fusion=0
python scripts/poison_hbm.py 0x7e >> $O/poison1.log 2>&1; run p7e_nomm GGML_CDNA4_NO_MM_FUSION=1
python scripts/poison_hbm.py 0x7e >> $O/poison1.log 2>&1; run p7e_nographs GGML_CDNA4_PARAMS=graphs=0
