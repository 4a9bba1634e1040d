#!/usr/bin/env python3
"""Development timings of the other BASELINE configs on one MI355X: IVF-flat (C5 per-GPU share) and the
encoder forward (C3).  Not the graded benchmark (bench.py)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch


def ivf(n=625_000, d=768, nlist=4096, nprobe=32, nq=10_000, k=10):
    from mrag_amd.index import IVFFlatIndex, DenseIndex
    g = torch.Generator(device="cuda").manual_seed(1)
    cent = torch.randn(4096, d, device="cuda", generator=g)
    ix = IVFFlatIndex(d, nlist)
    bf = DenseIndex(d)
    t0 = time.perf_counter()
    train = cent[torch.randint(0, 4096, (100_000,), device="cuda", generator=g)] + 0.3 * torch.randn(100_000, d, device="cuda", generator=g)
    ix.train(train, iters=5, seed=1)
    torch.cuda.synchronize(); t_train = time.perf_counter() - t0
    t0 = time.perf_counter()
    for lo in range(0, n, 125_000):
        m = min(125_000, n - lo)
        rows = cent[torch.randint(0, 4096, (m,), device="cuda", generator=g)] + 0.3 * torch.randn(m, d, device="cuda", generator=g)
        ix.add(rows); bf.add(rows)
    torch.cuda.synchronize(); t_add = time.perf_counter() - t0
    q = cent[torch.randint(0, 4096, (nq,), device="cuda", generator=g)] + 0.3 * torch.randn(nq, d, device="cuda", generator=g)
    ix.search(q, k, nprobe)
    ts = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); sc, ids = ix.search(q, k, nprobe); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)   # (device in / out: the call itself is asynchronous)
    bs, bi = bf.search(q, k); torch.cuda.synchronize()
    bi = bi.cpu().numpy()
    ids = ids.cpu().numpy() if torch.is_tensor(ids) else ids
    rec = np.mean([len(set(a.tolist()) & set(b.tolist())) / k for a, b in zip(ids, bi)])
    t = ix.last_timing()
    print(f"  list scan {t['scan_ms']:.3f} ms, whole search {t['total_ms']:.3f} ms, {t['n_wg']} workgroups, {t['scanned_rows']} rows")
    print(f"IVF n={n} nlist={nlist} nprobe={nprobe} nq={nq}: train {t_train:.2f}s add {t_add:.2f}s search {min(ts)*1e3:.1f} ms "
          f"({nq/min(ts):.0f} q/s) recall@{k} vs brute force {rec:.4f}", flush=True)


def encoder(arch="bge-base", B=2048, S=128, iters=3):
    from mrag_amd.encoder import HipSentenceEncoder, ARCHS
    enc = HipSentenceEncoder.from_seed(arch, seed=0)
    rng = np.random.default_rng(0)
    ids = rng.integers(1000, 30000, size=(B, S)).astype(np.int32)
    mask = np.ones((B, S), dtype=np.int32)
    enc.forward(ids, mask)
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter(); enc.forward(ids, mask); ts.append(time.perf_counter() - t0)
    a = ARCHS[arch]
    flop_tok = 2 * a["layers"] * (4 * a["hidden"] ** 2 + 2 * a["hidden"] * a["intermediate"]) + 4 * S * a["hidden"] * a["layers"]
    t = min(ts)
    print(f"encoder {arch} B={B} S={S}: {t*1e3:.1f} ms, {B/t:.0f} passages/s, {B*S*flop_tok/t/1e12:.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    what = sys.argv[1:] or ["ivf", "enc"]
    if "enc" in what:
        encoder("minilm-l6", 4096, 128)
    if "enc" in what or "enc-bge" in what:
        encoder("bge-base", 2048, 128)
    if "ivf" in what:
        ivf()
