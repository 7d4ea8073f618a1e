#!/usr/bin/env python3
"""Fill most of the free HBM with a byte pattern and release it: the next process then finds non-zero garbage in freshly allocated device memory (a freshly leased box hands out
zero pages, which hides reads of uninitialised memory).    python scripts/poison_hbm.py [pattern-byte, default 0x7e] [fraction, default 0.9]"""
import sys
import torch
pat = int(sys.argv[1], 0) if len(sys.argv) > 1 else 0x7e
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.9
free, total = torch.cuda.mem_get_info()
bufs = []; left = int(free * frac); chunk = 8 << 30
while left > 0:
    n = min(chunk, left)
    try:
        b = torch.empty(n, dtype=torch.uint8, device="cuda"); b.fill_(pat); bufs.append(b)
    except RuntimeError:
        break
    left -= n
torch.cuda.synchronize()
print("poisoned %.1f GB with 0x%02x" % (sum(b.numel() for b in bufs) / 1e9, pat))
