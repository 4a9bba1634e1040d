#!/usr/bin/env python3
"""Host-side legs of the N-GPU step timed on this box: pinned D2H of the gathered partial
top-k and mrag_topk_merge with C++ threads."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from mrag_amd.index import topk_merge
rng = np.random.default_rng(0)
for nparts in (2, 4, 8):
    nq, k = 10000, 10
    sc = -np.sort(-rng.random((nparts, nq, k)).astype(np.float32), axis=2)
    ids = np.sort(rng.integers(0, 10**6, (nparts, nq, k)), axis=2)
    for nt in (1, 4, 16, 0):
        topk_merge(sc, ids, nt)
        t = time.perf_counter()
        for _ in range(10): topk_merge(sc, ids, nt)
        print(f"merge nparts={nparts} threads={nt}: {(time.perf_counter()-t)/10*1e3:.3f} ms", flush=True)
    if torch.cuda.is_available():
        gs = torch.from_numpy(sc).cuda(); gi = torch.from_numpy(ids).cuda()
        hs = torch.empty_like(gs, device="cpu").pin_memory(); hi = torch.empty_like(gi, device="cpu").pin_memory()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(10):
            hs.copy_(gs, non_blocking=True); hi.copy_(gi, non_blocking=True); torch.cuda.synchronize()
        print(f"  pinned D2H of {gs.numel()*12/1e6:.1f} MB: {(time.perf_counter()-t)/10*1e3:.3f} ms", flush=True)
        # the device leg that replaces both: mrag_topk_merge_device + D2H of the merged [nq, k] only
        from mrag_amd.index import topk_merge_device
        ms = torch.empty((nq, k), dtype=torch.float32, device="cuda"); mi = torch.empty((nq, k), dtype=torch.int64, device="cuda")
        hms = torch.empty((nq, k), dtype=torch.float32).pin_memory(); hmi = torch.empty((nq, k), dtype=torch.int64).pin_memory()
        topk_merge_device(gs, gi, ms, mi); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(10):
            topk_merge_device(gs, gi, ms, mi); hms.copy_(ms, non_blocking=True); hmi.copy_(mi, non_blocking=True); torch.cuda.synchronize()
        print(f"  device merge + D2H of the merged {nq*k*12/1e6:.1f} MB: {(time.perf_counter()-t)/10*1e3:.3f} ms", flush=True)
