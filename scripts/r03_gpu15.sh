#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3o; mkdir -p $O; rm -f $O/moe.log
export TMPDIR=/tmp
for nt in 1 2 4; do for rb in 8 1; do
  echo "== NT=$nt RB=$rb" >> $O/moe.log
  CDNA4_MOE_NT=$nt CDNA4_MOE_RB=$rb MB_MOE_T=128,512,2048 timeout 600 python scripts/microbench.py moe mixtral 2>&1 | grep "moe q4" >> $O/moe.log
  CDNA4_MOE_NT=$nt CDNA4_MOE_RB=$rb MB_MOE_T=128,512,2048 timeout 600 python scripts/microbench.py moe 2>&1 | grep "moe q4" >> $O/moe.log
done; done
python - <<'PY'
import re
for l in open('gpurun_out/r3o/moe.log'):
    if l.startswith('=='): print(l.strip()); continue
    m=re.search(r'E=(\d+).*T=\s*(\d+).*gate\s+([\d.]+) us.*down\s+([\d.]+) us', l)
    print("  E=%s T=%s fused %s down %s" % m.groups())
PY
