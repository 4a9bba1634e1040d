#!/usr/bin/env python3
"""One rank's step of the N-GPU search on ONE GPU through a 1-rank RCCL group (force_collective): the one-shot exchange
against the two-half pipeline (all-gather of half A async under the local search of half B), per-phase device ms.
With one rank the collective itself is a local copy, so this measures what the split COSTS the local search (two
half-batch launches) -- the part a real 8-GPU run has to win back from the exchange it hides.
    python tools/perf_pipeline.py"""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch, torch.distributed as dist
from mrag_amd.sharded import ShardedDenseIndex
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29631")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
d, nq, k = 768, 10000, 10
g = torch.Generator(device="cuda").manual_seed(1)
q = torch.randn(nq, d, device="cuda", generator=g)
for world in (8, 4, 2):
    n = 1_000_000 // world
    sh = ShardedDenseIndex(d, n, rank=0, world=1, device=0)
    for lo in range(0, n, 250000):
        sh.add_local(torch.randn(min(250000, n - lo), d, device="cuda", generator=g))
    for pipe in (False, True):
        ph = []
        for it in range(12):
            sh.search(q, k, force_collective=True, pipeline=pipe)
            ph.append(dict(sh.last_phases))
        med = {kk: float(np.median([p.get(kk, 0.0) for p in ph[2:]])) for kk in ("local_ms", "pack_ms", "gather_ms", "merge_ms", "d2h_ms")}
        print(f"rows/GPU={n} (1/{world} of 1M) {'two halves' if pipe else 'one shot  '}: " + ", ".join(f"{a} {b:.3f}" for a, b in med.items())
              + f", sum {sum(med.values()):.3f} ms", flush=True)
    sh.index.close()
dist.destroy_process_group()
