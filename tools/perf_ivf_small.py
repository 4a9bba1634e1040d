import sys, time; sys.path.insert(0, ".")
import numpy as np, torch
from mrag_amd.index import IVFFlatIndex, DenseIndex
g = torch.Generator(device="cuda").manual_seed(1)
d, nlist, n = 768, 4096, 625_000
cent = torch.randn(4096, d, device="cuda", generator=g)
ix = IVFFlatIndex(d, nlist); bf = DenseIndex(d)
train = cent[torch.randint(0, 4096, (100_000,), device="cuda", generator=g)] + 0.3 * torch.randn(100_000, d, device="cuda", generator=g)
ix.train(train, iters=5, seed=1)
for lo in range(0, n, 125_000):
    rows = cent[torch.randint(0, 4096, (125_000,), device="cuda", generator=g)] + 0.3 * torch.randn(125_000, d, device="cuda", generator=g)
    ix.add(rows); bf.add(rows)
for nq in (1, 8, 32, 128, 1000, 10000):
    q = cent[torch.randint(0, 4096, (nq,), device="cuda", generator=g)] + 0.3 * torch.randn(nq, d, device="cuda", generator=g)
    ix.search(q, 10, 32); torch.cuda.synchronize()
    ts, dev = [], []
    for _ in range(10):
        torch.cuda.synchronize(); t0 = time.perf_counter(); sc, ids = ix.search(q, 10, 32); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        dev.append(ix.last_timing()["total_ms"])
    bs, bi = bf.search(q, 10); torch.cuda.synchronize()
    rec = np.mean([len(set(a.tolist()) & set(b.tolist())) / 10 for a, b in zip(ids.cpu().numpy(), bi.cpu().numpy())])
    print(f"IVF nq={nq}: host call {np.median(ts)*1e3:.3f} ms, device {np.median(dev):.3f} ms, recall@10 {rec:.3f}; brute force search {bf.last_timing_ms()[1]:.3f} ms", flush=True)
