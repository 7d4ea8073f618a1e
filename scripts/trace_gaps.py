#!/usr/bin/env python3
"""Per-kernel duration AND the idle gap in front of each kernel from a rocprofv3 --kernel-trace CSV (decode tokens: kernels are grouped by short name; the gap of a kernel is
its start minus the previous kernel's end on the same queue).    python scripts/trace_gaps.py <run_kernel_trace.csv> [min_calls]"""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1]))); min_calls = int(sys.argv[2]) if len(sys.argv) > 2 else 50
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); m = re.match(r"(?:void )?([A-Za-z0-9_]+(?:<[^(]*>)?)", n); return (m.group(1) if m else n)[:70]
dur = collections.defaultdict(list); gap = collections.defaultdict(list); prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"]); k = short(r["Kernel_Name"])
    dur[k].append(e - s)
    if prev_end is not None and 0 <= s - prev_end < 200000: gap[k].append(s - prev_end)
    prev_end = max(e, prev_end or 0)
tot_d = sum(sum(v) for v in dur.values()); tot_g = sum(sum(v) for v in gap.values())
print("kernels %d  sum of durations %.2f ms  sum of gaps (< 200 us) %.2f ms  span %.2f ms" % (len(rows), tot_d / 1e6, tot_g / 1e6, (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e6))
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    if len(v) < min_calls: continue
    g = gap.get(k, [0]); v2 = sorted(v); g2 = sorted(g)
    print("%6d x  dur avg %7.2f med %7.2f us   gap-before avg %6.2f med %6.2f us   %s" % (len(v), sum(v) / len(v) / 1e3, v2[len(v2) // 2] / 1e3, sum(g) / max(len(g), 1) / 1e3, g2[len(g2) // 2] / 1e3, k))
