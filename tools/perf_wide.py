#!/usr/bin/env python3
"""Development timing of the k > 64 path (the reference's 200-candidate pool), one MI355X."""
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
import os, time
for nq in (1, 8, 64, 1000, 10000):
    q = torch.randn(nq, d, device="cuda", generator=g)
    for k in (10, 64, 100, 200, 256):
        ix.search(q, k)
        ts = []
        for _ in range(7 if nq <= 1000 or k <= 64 else 3):
            ix.search(q, k); ts.append(ix.last_timing_ms())
        ts.sort(); ts = ts + ts[-1:] * 4
        print(f"nq={nq} k={k}: kernel {ts[3][0]:.3f} ms, search {ts[3][1]:.3f} ms"
              + (f", {ix.last_wide_redone()} queries redone by the streaming kernel (wide batch)" if k > 64 and nq > 32 else ""), flush=True)
