#!/usr/bin/env python3
"""End-to-end latency of ONE question through the drop-in (provider -> router -> DenseRetrievalBackend.run), the reference's own
online regime (app/system.py: one question per call).  Synthetic HotpotQA-shaped docs.jsonl, seeded MiniLM-L6-shaped encoder."""
import cProfile, io, pstats, sys, tempfile, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np

from mrag_amd import corpus
from mrag_amd.backend import DenseRetrievalBackend
from mrag_amd.dto import RetrievalIn
from mrag_amd.provider import HipEmbeddingProvider


def main(n_titles=20000):
    rng = np.random.default_rng(4)
    vocab = [f"w{i}" for i in range(5000)]
    rows = []
    for t in range(n_titles):
        for sid in range(int(rng.integers(2, 6))):
            rows.append({"doc_id": f"Title {t}#{sid}", "title": f"Title {t}", "sent_id": sid,
                         "text": " ".join(rng.choice(vocab, size=int(rng.integers(8, 24))))})
    td = tempfile.mkdtemp()
    docs = Path(td) / "docs.jsonl"
    corpus.write_docs_jsonl(docs, rows)
    prov = HipEmbeddingProvider(arch="minilm-l6", seed=5, embed_model="minilm-seed5")

    class Router:
        providers, policy = {"hip": prov}, {"embedding_provider": "hip"}
        def embed(self, *, model_hint, texts, require=None):
            return prov.embed(model=model_hint, texts=texts, require=require)

    for kw in ({}, {"text_channel": "bm25"}, {"text_channel": "bm25", "fuse_on_device": True}):
        be = DenseRetrievalBackend(Router(), index_path=str(docs), cache_dir=str(Path(td) / "cache"), **kw)
        q = " ".join(rng.choice(vocab, size=12))
        t0 = time.perf_counter()
        be.run(RetrievalIn(query=q, graph_id="", top_k=30, trace_id="t"))
        t_first = time.perf_counter() - t0
        ts = []
        for _ in range(30):
            q = " ".join(rng.choice(vocab, size=12))
            t0 = time.perf_counter(); out = be.run(RetrievalIn(query=q, graph_id="", top_k=30, trace_id="t")); ts.append(time.perf_counter() - t0)
        print(f"{len(rows)} sentences, {kw or 'dense only'}: first call {t_first:.2f} s (index build / cache), then "
              f"median {np.median(ts)*1e3:.2f} ms, min {min(ts)*1e3:.2f} ms per question; hits {len(out['hits'])}", flush=True)
        if kw.get("fuse_on_device") or not kw:
            pr = cProfile.Profile(); pr.enable()
            for _ in range(20):
                be.run(RetrievalIn(query=q, graph_id="", top_k=30, trace_id="t"))
            pr.disable()
            s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(14)
            print("\n".join(s.getvalue().splitlines()[:40]))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 20000)
