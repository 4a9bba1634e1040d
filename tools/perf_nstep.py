#!/usr/bin/env python3
"""One rank's share of the N-GPU step timed on ONE GPU: local search over 1M/N rows, then the
merge leg over a stand-in for the all-gather output (N copies of the local result with shifted ids).
The RCCL all-gather itself (1.2 MB per rank at C4) is not included."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from mrag_amd.index import DenseIndex
from mrag_amd.sharded import merge_gathered
d, nq, k = 768, 10000, 10
g = torch.Generator(device="cuda").manual_seed(1)
q = torch.randn(nq, d, device="cuda", generator=g)
for world in (1, 2, 4, 8):
    n = 1_000_000 // world
    ix = DenseIndex(d)
    for lo in range(0, n, 250000):
        ix.add(torch.randn(min(250000, n - lo), d, device="cuda", generator=g))
    sc, ids = ix.search(q, k); torch.cuda.synchronize()
    gs = torch.stack([sc - 0.001 * r for r in range(world)]).contiguous()
    gi = torch.stack([ids + n * r for r in range(world)]).contiguous()
    out = {}
    for mode in (["host", "device"] if world > 1 else ["local"]):
        bufs = {}
        ts = []
        for it in range(12):
            torch.cuda.synchronize(); t = time.perf_counter()
            sc, ids = ix.search(q, k)
            if world > 1:
                merge_gathered(gs, gi, bufs, merge=mode)
            else:
                sc.cpu(); ids.cpu()
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
        out[mode] = sorted(ts[2:])[len(ts[2:]) // 2] * 1e3
    print(f"world={world} rows/GPU={n}: K2 {ix.last_timing_ms()[0]:.3f} ms; step " + ", ".join(f"{m} {v:.3f} ms" for m, v in out.items()), flush=True)
    ix.close()
