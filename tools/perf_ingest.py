#!/usr/bin/env python3
"""Cold build of the dense corpus index from a docs.jsonl: write an N-row HotpotQA-shaped file, then
``DenseRetrievalBackend._build_state`` through the bulk device path, with the phases timed.
    python tools/perf_ingest.py [rows=1000000] [arch=minilm-l6]"""
import sys, time, tempfile
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from bench import synthetic_sentences
from mrag_amd import corpus
from mrag_amd.backend import DenseRetrievalBackend
from mrag_amd.provider import HipEmbeddingProvider

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
arch = sys.argv[2] if len(sys.argv) > 2 else "minilm-l6"
t0 = time.perf_counter()
texts = synthetic_sentences(n, 1)
rows = [{"doc_id": f"Title {i // 4}#{i % 4}", "title": f"Title {i // 4}", "sent_id": i % 4, "text": t} for i, t in enumerate(texts)]
tmp = Path(tempfile.mkdtemp()) / "docs.jsonl"
corpus.write_docs_jsonl(tmp, rows)
print(f"wrote {n} rows, {tmp.stat().st_size / 1e6:.0f} MB in {time.perf_counter() - t0:.1f} s", flush=True)
del rows, texts
prov = HipEmbeddingProvider(arch=arch, seed=0, embed_model=f"{arch}-seed0")


class Router:
    providers, policy = {"hip": prov}, {"embedding_provider": "hip"}
    def embed(self, *, model_hint, texts, require=None):
        return prov.embed(model=model_hint, texts=texts, require=require)

prov.embed_device(["warm up"] * 4096)
t0 = time.perf_counter(); r = corpus.read_docs_jsonl(tmp); t_read = time.perf_counter() - t0
del r
be = DenseRetrievalBackend(Router(), index_path=str(tmp), bulk_batch=4096)
t0 = time.perf_counter()
st = be._build_state(f"{arch}-seed0", "t")
dt = time.perf_counter() - t0
print(f"{arch}: cold build of {len(st['index'])} rows in {dt:.1f} s ({n / dt:.0f} rows/s) via {st['ingest']}; of which reading docs.jsonl {t_read:.1f} s", flush=True)
