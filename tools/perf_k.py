#!/usr/bin/env python3
"""K2 cost as a function of k (batch regime) -- development timing."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from mrag_amd.index import DenseIndex
d = 768
ix = DenseIndex(d)
g = torch.Generator(device="cuda").manual_seed(1)
for _ in range(4):
    ix.add(torch.randn(250000, d, device="cuda", generator=g))
for nq in (1000, 10000):
    q = torch.randn(nq, d, device="cuda", generator=g)
    for k in (1, 10, 16, 17, 20, 32, 64):
        ix.search(q, k)
        ts = []
        for _ in range(3):
            ix.search(q, k); ts.append(ix.last_timing_ms())
        ts.sort()
        print(f"nq={nq} k={k}: K2 {ts[1][0]:.3f} ms, search {ts[1][1]:.3f} ms", flush=True)
