#!/usr/bin/env python3
"""Development timing of the BM25 text channel on the device vs the CPU restatement (oracle) -- not the graded bench."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np


def main(n=int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000):
    from mrag_amd.text_index import HipBM25Index
    rng = np.random.default_rng(7)
    V = 30000
    p = 1.0 / np.arange(1, V + 1) ** 1.07
    p /= p.sum()
    lens = rng.integers(5, 40, size=n)
    words = rng.choice(V, size=int(lens.sum()), p=p)
    t0 = time.perf_counter()
    rows, at = [], 0
    for i, L in enumerate(lens):
        rows.append({"doc_id": f"T{i // 5}#{i % 5}", "title": f"T{i // 5}", "sent_id": i % 5, "text": " ".join(f"w{w}" for w in words[at:at + L])})
        at += L
    t_gen = time.perf_counter() - t0
    t0 = time.perf_counter(); ix = HipBM25Index(rows=rows); t_build = time.perf_counter() - t0
    queries = ["w0 w3 w77 w4000 w12", "w1 w2 w250 w9999", "w5 w17 w300 w29999 w4"]
    ix.search(queries, top_k=200)
    ts = []
    for _ in range(10):
        t0 = time.perf_counter(); got = ix.search(queries, top_k=200); ts.append(time.perf_counter() - t0)
    post = sum(ix.df.get(t, 0) for q in queries for t in q.split())
    print(f"BM25 n={n} docs, vocab {len(ix.df)}: build {t_build:.1f} s (host tokenise + CSR), search of {len(queries)} expanded queries "
          f"({post} postings) top-200: {np.median(ts)*1e3:.3f} ms", flush=True)
    if n <= 200_000:
        from oracle.bm25 import Bm25Oracle
        o = Bm25Oracle(rows)
        t0 = time.perf_counter(); want = o.search(queries, top_k=200); t_cpu = time.perf_counter() - t0
        print(f"  CPU restatement of the reference's search: {t_cpu*1e3:.1f} ms; identical: {got == want}")


if __name__ == "__main__":
    main()
