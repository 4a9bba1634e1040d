#!/usr/bin/env python3
"""Development timing of the fused similarity + top-k kernel at the BASELINE shapes
(kernel time from hipEvents inside the library).  Not the graded benchmark: see bench.py."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
from mrag_amd.index import DenseIndex


def run(nq, n, d, k=10, iters=int(__import__('os').environ.get('ITERS', '5'))):
    g = torch.Generator(device="cuda").manual_seed(1234)
    ix = DenseIndex(d)
    step = 262144
    for lo in range(0, n, step):
        m = min(step, n - lo)
        ix.add(torch.randn(m, d, device="cuda", generator=g, dtype=torch.float32))
    q = torch.randn(nq, d, device="cuda", generator=g, dtype=torch.float32)
    torch.cuda.synchronize()
    gm, tm = [], []
    for _ in range(iters + 2):
        ix.search(q, k)
        a, b = ix.last_timing_ms()
        gm.append(a); tm.append(b)
    gm, tm = sorted(gm[2:]), sorted(tm[2:])
    fl = 2.0 * nq * n * d
    print(f"nq={nq} n={n} d={d} k={k}: gemm+topk {gm[len(gm)//2]:.3f} ms ({fl/gm[len(gm)//2]/1e9:.1f} TFLOP/s), "
          f"search {tm[len(tm)//2]:.3f} ms, {nq/tm[len(tm)//2]*1e3:.0f} q/s", flush=True)
    ix.close()


if __name__ == "__main__":
    shapes = [(1000, 100000, 768), (1000, 1000000, 768), (10000, 1000000, 768), (10000, 125000, 768), (1, 1000000, 768),
              (100, 5000, 384)]
    if len(sys.argv) > 1:
        shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
    for s in shapes:
        run(*s)
