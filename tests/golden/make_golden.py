#!/usr/bin/env python3
"""Capture golden vectors F1-F5 from the REFERENCE's own functions.

Run in the build container only (the reference never travels):

    cd /root/repo && PYTHONHASHSEED=0 PYTHONDONTWRITEBYTECODE=1 \
        PYTHONPATH=/root/reference python tests/golden/make_golden.py

Outputs are DATA (inputs + the reference's outputs), written next to this file:
  f1_cosine.npz        DenseReranker._cosine / utils.similarity.cosine known answers
  f2_dense_score.json  DenseReranker.score through the real LLMRouter with a table provider
  f3_hybrid_run.json   HybridRetrievalBackend.run + RetrievalAdapter.retrieve, channel
                       inputs (BM25 / graph / dense raw) recorded beside the fused output
  f4_minmax.json       _minmax_norm / _normalize_id / _normalize_hit / mmr_diversify cases
  f5_bruteforce_c1.npz reference _cosine over the C1 shape (100 x 5000 x 384) -> top-10

Nothing from the reference is copied: the script only CALLS it.
"""
from __future__ import annotations

import hashlib
import json
import os
import sys
import tempfile
import zlib
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent.parent))

assert os.environ.get("PYTHONHASHSEED") == "0", "run with PYTHONHASHSEED=0 (tie order in the reference is hash dependent)"

from app.core.dto import RetrievalIn  # noqa: E402  (reference)
from app.core.llm_router import LLMRouter  # noqa: E402
from app.modules.retrieval.retrieval_adapter import RetrievalAdapter  # noqa: E402
from app.modules.retrieval.retrieval_backend import DenseReranker, HybridRetrievalBackend  # noqa: E402
from app.utils import similarity as ref_sim  # noqa: E402


def text_vec(text: str, dim: int = 8):
    """Deterministic table provider: vector seeded by crc32(text)."""
    rng = np.random.default_rng(zlib.crc32(text.encode("utf-8")))
    return [float(x) for x in rng.standard_normal(dim)]


class TableProvider:
    """Provider with the call shape the router uses (llm_router.py:115)."""

    def __init__(self, dim=8, fail_calls=(), bare_list=False, embed_model="fake-embed"):
        self.kwargs = {"embed_model": embed_model}
        self.dim, self.fail_calls, self.bare_list = dim, set(fail_calls), bare_list
        self.calls = []

    def embed(self, *, model, texts, require):
        idx = len(self.calls)
        self.calls.append({"model": model, "n": len(texts), "require": dict(require)})
        if idx in self.fail_calls:
            raise RuntimeError("injected embed failure")
        vecs = [text_vec(t, self.dim) for t in texts]
        return vecs if self.bare_list else {"vectors": vecs}

    def complete(self, *, model, prompt, require):
        return "alpha beta\n- gamma delta"


class RaisingRouter:
    """Router stand-in whose embed raises on chosen calls (exercises DenseReranker's own
    except branches, retrieval_backend.py:229-231,241-243)."""

    def __init__(self, fail_calls, dim=8):
        self.policy, self.providers = {}, {}
        self.fail_calls, self.dim, self.n = set(fail_calls), dim, 0

    def embed(self, *, model_hint, texts, require=None):
        i = self.n
        self.n += 1
        if i in self.fail_calls:
            raise RuntimeError("router-level failure")
        return {"vectors": [text_vec(t, self.dim) for t in texts]}


# ------------------------------------------------------------------------------ F1
def f1():
    dr = DenseReranker(router=None)
    rng = np.random.default_rng(20251004)
    A, B, out_dr, out_util, dims = [], [], [], [], []

    def add(a, b):
        A.append(np.asarray(a, dtype=np.float64)); B.append(np.asarray(b, dtype=np.float64))
        dims.append((len(a), len(b)))
        out_dr.append(dr._cosine(list(map(float, a)), list(map(float, b))))
        out_util.append(ref_sim.cosine(list(map(float, a)), list(map(float, b))))

    add([], [1.0]); add([1.0], []); add([0.0, 0.0, 0.0], [1.0, 2.0, 3.0]); add([1.0, 2.0, 3.0], [0.0, 0.0, 0.0])
    add([1.0, 2.0], [1.0, 2.0, 3.0]); add([1.0, 2.0, 3.0], [1.0, 2.0]); add([1.0, 0.0], [0.0, 1.0])
    add([1.0, 2.0, 3.0], [1.0, 2.0, 3.0]); add([1.0, 2.0, 3.0], [-1.0, -2.0, -3.0]); add([3.0], [-2.0])
    add([1e-200, 1e-200], [1e-200, 1e-200]); add([1e200, 1e200], [1e200, 1.0])
    for d in (3, 8, 8, 384, 384, 384, 768, 768, 768, 3072):
        add(rng.standard_normal(d), rng.standard_normal(d))
    for d in (384, 768):  # fp16-rounded unit vectors, the build's storage format
        a = rng.standard_normal(d); b = a + 0.05 * rng.standard_normal(d)
        a = (a / np.linalg.norm(a)).astype(np.float16).astype(np.float64)
        b = (b / np.linalg.norm(b)).astype(np.float16).astype(np.float64)
        add(a, b)
    dmax = max(max(x) for x in dims)
    pa = np.zeros((len(A), dmax)); pb = np.zeros((len(B), dmax))
    for i, (a, b) in enumerate(zip(A, B)):
        pa[i, :len(a)] = a; pb[i, :len(b)] = b
    np.savez_compressed(HERE / "f1_cosine.npz", a=pa, b=pb, dims=np.asarray(dims, dtype=np.int64),
                        out_dense_reranker=np.asarray(out_dr), out_utils_similarity=np.asarray(out_util))
    print("F1:", len(A), "pairs")


# ------------------------------------------------------------------------------ F2
def make_candidates(n, seed, missing=()):
    rng = np.random.default_rng(seed)
    words = ["alpha", "beta", "gamma", "delta", "epsilon", "zeta", "eta", "theta", "iota", "kappa"]
    out = []
    for i in range(n):
        txt = " ".join(rng.choice(words, size=5)) + f" #{i}"
        meta = {"kind": "sentence", "text": None if i in missing else txt, "doc": f"T{i % 7}", "sent_id": i % 5,
                "source": "bm25"}
        out.append({"id": f"sent::T{i % 7}#{i % 5}::{(i % 5) or ''}", "score": float(rng.random()), "meta": meta})
    return out


def f2():
    cases = []

    def run_case(name, router, prov, cands, max_pool, embed_batch, query="what is alpha gamma"):
        dr = DenseReranker(router=router, max_pool=max_pool, embed_batch=embed_batch)
        out = dr.score(query=query, candidates=cands, trace_id="t-f2")
        cases.append({"name": name, "query": query, "candidates": cands, "max_pool": max_pool,
                      "embed_batch": embed_batch, "resolved_model": dr._resolve_embed_model(),
                      "calls": (prov.calls if prov is not None else None),
                      "provider": ({"dim": prov.dim, "fail_calls": sorted(prov.fail_calls),
                                    "bare_list": prov.bare_list} if prov is not None else None),
                      "router_fail_calls": (sorted(router.fail_calls) if isinstance(router, RaisingRouter) else None),
                      "out": out})

    def real_router(prov, policy=None):
        return LLMRouter(providers={"hip": prov}, policy=policy or {"embedding_provider": "hip"})

    p = TableProvider(); run_case("basic_pool20_bs4", real_router(p), p, make_candidates(23, 1, missing={3, 11}), 20, 4)
    p = TableProvider(); run_case("bs50_single_chunk", real_router(p), p, make_candidates(30, 2), 200, 50)
    p = TableProvider(bare_list=True); run_case("bare_list_provider", real_router(p), p, make_candidates(12, 3), 200, 8)
    p = TableProvider(fail_calls={2}); run_case("provider_fails_chunk_router_zero3", real_router(p), p,
                                                make_candidates(20, 4), 200, 8)
    p = TableProvider(fail_calls={0}); run_case("provider_fails_query_router_zero3", real_router(p), p,
                                                make_candidates(10, 5), 200, 8)
    p = TableProvider(); run_case("policy_embedding_model_wins", real_router(
        p, {"embedding_provider": "hip", "embedding": [{"model": "policy-model", "provider": "hip"}]}), p,
        make_candidates(9, 6), 200, 8)
    p = TableProvider(); run_case("no_provider_named", LLMRouter(providers={"hip": p}, policy={}), p,
                                  make_candidates(9, 7), 200, 8)
    r = RaisingRouter({2}); run_case("router_raises_chunk_zero_fill", r, None, make_candidates(20, 8), 200, 8)
    r = RaisingRouter({0}); run_case("router_raises_query_empty", r, None, make_candidates(10, 9), 200, 8)
    p = TableProvider(); run_case("all_text_missing", real_router(p), p, make_candidates(5, 10, missing=set(range(5))), 200, 8)
    p = TableProvider(); run_case("empty_candidates", real_router(p), p, [], 200, 8)
    (HERE / "f2_dense_score.json").write_text(json.dumps({"cases": cases}, indent=1))
    print("F2:", len(cases), "cases")


# ------------------------------------------------------------------------------ F3
def synth_docs(n_titles=40, seed=7):
    rng = np.random.default_rng(seed)
    vocab = ["alpha", "beta", "gamma", "delta", "river", "city", "born", "film", "band", "album", "war",
             "king", "queen", "author", "novel", "team", "league", "science", "island", "bridge"]
    rows = []
    titles = [f"Title {i}" for i in range(n_titles)] + ["Title 3", "Title 8"]  # duplicated titles (ingest does not dedupe)
    for t in titles:
        for sid in range(int(rng.integers(3, 7))):
            text = " ".join(rng.choice(vocab, size=int(rng.integers(6, 12)))) + "."
            rows.append({"doc_id": f"{t}#{sid}", "title": t, "sent_id": sid, "text": text})
    return rows


def f3():
    rows = synth_docs()
    out_cases = []
    with tempfile.TemporaryDirectory() as td:
        docs = Path(td) / "docs.jsonl"
        docs.write_text("\n".join(json.dumps(r) for r in rows) + "\n")
        groot = Path(td) / "graph"
        (groot / "g1").mkdir(parents=True)
        gnodes = [{"id": "q1", "type": "question", "props": {"text": "alpha river"}}]
        gedges = []
        for i in range(6):
            gnodes.append({"id": f"Title 3::{i}", "type": "sentence",
                           "props": {"text": rows[i]["text"], "doc": "Title 3"}})
            if i:
                gedges.append({"type": "next_in_doc", "source": f"Title 3::{i - 1}", "target": f"Title 3::{i}"})
        gedges.append({"type": "q_match", "source": "q1", "target": "Title 3::2"})
        (groot / "g1" / "graph.json").write_text(json.dumps({"nodes": gnodes, "edges": gedges}))

        for name, graph_id, top_k, query in (("no_graph_top5", "", 5, "alpha river city"),
                                              ("graph_g1_top8", "g1", 8, "alpha river"),
                                              ("top_k_default", "", 0, "queen novel author")):
            prov = TableProvider()
            router = LLMRouter(providers={"hip": prov}, policy={"embedding_provider": "hip",
                                                                "default": [{"model": "m", "provider": "hip"}]})
            be = HybridRetrievalBackend(router=router, index_path=str(docs), graph_root=str(groot),
                                        bm25_pool_k=30, embed_batch=8, graph_window=1, default_top_k=7)
            rec = {}
            _ts, _ge, _ds = be.text.search, be.graph.expand, be.dense.score
            be.text.search = lambda **kw: rec.setdefault("t_hits_raw", _ts(**kw))
            be.graph.expand = lambda **kw: rec.setdefault("g_hits_raw", _ge(**kw))
            be.dense.score = lambda **kw: rec.setdefault("dense_scores_raw", _ds(**kw))
            req = RetrievalIn(query=query, graph_id=graph_id, top_k=top_k, trace_id="t-f3")
            run_out = be.run(req)
            ad = RetrievalAdapter(router=router, backend_impl="app.modules.retrieval.retrieval_backend:HybridRetrievalBackend",
                                  backend_kwargs=dict(index_path=str(docs), graph_root=str(groot), bm25_pool_k=30,
                                                      embed_batch=8, graph_window=1, default_top_k=7))
            ad_out = ad.retrieve(req)
            out_cases.append({
                "name": name, "req": req.model_dump() if hasattr(req, "model_dump") else req.dict(),
                "backend_kwargs": {"bm25_pool_k": 30, "embed_batch": 8, "graph_window": 1, "default_top_k": 7,
                                   "alpha_text": be.alpha_text, "alpha_graph": be.alpha_graph,
                                   "alpha_dense": be.alpha_dense},
                "t_hits_raw": rec.get("t_hits_raw"), "g_hits_raw": rec.get("g_hits_raw"),
                "dense_scores_raw": rec.get("dense_scores_raw"),
                "embed_calls": prov.calls, "run_out": run_out,
                "adapter_out": {"hits": [h.model_dump() if hasattr(h, "model_dump") else h.dict() for h in ad_out.hits],
                                "diagnostics": ad_out.diagnostics},
            })
    (HERE / "f3_hybrid_run.json").write_text(json.dumps({"docs_rows": rows, "cases": out_cases}, indent=1))
    print("F3:", len(out_cases), "cases,", len(rows), "rows")


# ------------------------------------------------------------------------------ F4
def f4():
    be = HybridRetrievalBackend.__new__(HybridRetrievalBackend)  # methods below use no state
    mm_in = [{}, {"a": 1.0}, {"a": 2.0, "b": 2.0}, {"a": -1.0, "b": 0.0, "c": 3.0}, {"x": 0.25, "y": 0.75, "z": 0.5},
             {"p": -5.0, "q": -7.5}]
    mm = [{"in": v, "out": be._minmax_norm(dict(v))} for v in mm_in]
    nid_in = [{"id": "sent::A#1::1", "meta": {"doc": "A", "sent_id": 1}}, {"id": "sent::A#0::", "meta": {"doc": "A", "sent_id": 0}},
              {"id": "raw", "meta": {"title": "B", "sid": 4}}, {"id": "raw-only", "meta": {}}, {"id": "", "meta": {}},
              {"id": "x", "meta": {"doc": "C"}}, {"meta": {"doc": "D", "sent_id": 0, "sid": 9}}]
    nid = [{"in": h, "out": be._normalize_id(h)[0]} for h in nid_in]
    ad = RetrievalAdapter.__new__(RetrievalAdapter)
    ad.id_keys = ["id", "doc_id", "docId", "sid", "sent_id"]; ad.score_keys = ["score", "relevance", "sim", "s"]; ad.meta_key = "meta"
    nh_in = [{"id": "a", "score": 0.5, "meta": {"text": "t"}}, {"doc_id": "d#1", "relevance": "0.25", "title": "T"},
             {"sim": "bad", "meta": {"doc": "X", "sent_id": 2}}, {"s": 1, "meta": {"title": "Y"}}, {"score": None, "sid": 7, "extra": 1},
             {"meta": {"doc": "Z", "sid": 3}}, {"id": 0, "score": 2}]
    nh = []
    for r in nh_in:
        h = ad._normalize_hit(dict(r))
        nh.append({"in": r, "out": (h.model_dump() if hasattr(h, "model_dump") else h.dict())})
    rng = np.random.default_rng(99)
    items = [(f"i{i}", float(rng.random()), [float(x) for x in rng.standard_normal(6)]) for i in range(12)]
    items[4] = (items[4][0], items[4][1], None)
    items[7] = (items[7][0], items[2][1], items[7][2])  # tie on score
    mmr = []
    for tk, lam in ((5, 0.7), (12, 0.3), (20, 1.0), (3, 0.0)):
        sel = ref_sim.mmr_diversify(list(items), top_k=tk, lambda_weight=lam)
        mmr.append({"top_k": tk, "lambda": lam, "selected_ids": [s[0] for s in sel]})
    (HERE / "f4_minmax.json").write_text(json.dumps({"minmax": mm, "normalize_id": nid, "normalize_hit": nh,
                                                     "mmr_items": items, "mmr": mmr}, indent=1))
    print("F4 done")


# ------------------------------------------------------------------------------ F5
def f5():
    from oracle.dense_search import make_gaussian, normalize_round
    nq, n, d, k = 100, 5000, 384, 10
    c16 = normalize_round(make_gaussian(n, d, 1234)); q16 = normalize_round(make_gaussian(nq, d, 5678))
    dr = DenseReranker(router=None)
    cl = [list(map(float, r)) for r in c16.astype(np.float64)]
    ids = np.zeros((nq, k), dtype=np.int64); sc = np.zeros((nq, k))
    for i in range(nq):
        qv = list(map(float, q16[i].astype(np.float64)))
        s = np.asarray([dr._cosine(qv, v) for v in cl])
        order = np.argsort(-s, kind="stable")[:k]   # score desc, row asc
        ids[i], sc[i] = order, s[order]
        if i % 20 == 0:
            print("  F5 query", i, flush=True)
    np.savez_compressed(HERE / "f5_bruteforce_c1.npz", nq=nq, n=n, d=d, k=k, corpus_seed=1234, query_seed=5678,
                        ids=ids, scores=sc,
                        corpus_sha256=hashlib.sha256(c16.tobytes()).hexdigest(),
                        query_sha256=hashlib.sha256(q16.tobytes()).hexdigest())
    print("F5 done")


if __name__ == "__main__":
    which = sys.argv[1:] or ["f1", "f2", "f3", "f4", "f5"]
    for w in which:
        globals()[w]()
