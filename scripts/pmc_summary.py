#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel: average counter value per dispatch.
FETCH_SIZE / WRITE_SIZE are reported in KB; on gfx950 a wide coalesced stream is under-reported by exactly 2x
(MI355X_MICROARCH.md, HBM / rocprofv3 section), so corrected bytes = 2 * 1024 * value.
usage: python scripts/pmc_summary.py <counter_collection.csv> "<command that produced it>" > profiles/<name>.json"""
import csv, json, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"command": sys.argv[2] if len(sys.argv) > 2 else "", "correction": "gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced stream "
       "(MI355X_MICROARCH.md HBM section): bytes = 2 * 1024 * FETCH_SIZE", "kernels": {}}
for k, cs in acc.items():
    if "at::native" in k or "rocclr" in k:
        continue
    e = {"dispatches": len(next(iter(cs.values())))}
    for c, v in cs.items():
        e[c + ("_KB_avg" if c.endswith("_SIZE") else "_avg")] = round(sum(v) / len(v), 2)
        if c == "FETCH_SIZE":
            e["hbm_read_bytes_per_launch_corrected"] = int(round(2 * 1024 * sum(v) / len(v)))
        if c == "WRITE_SIZE":
            e["hbm_write_bytes_per_launch_corrected"] = int(round(2 * 1024 * sum(v) / len(v)))
    out["kernels"][k] = e
json.dump(out, sys.stdout, indent=1)
