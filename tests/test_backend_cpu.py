"""CPU tests of ``DenseRetrievalBackend`` / ``HipDenseReranker`` control flow around the index: the
process-wide registry (no deadlock on the first question), embed-failure handling (nothing poisoned),
cold == warm cache bits, the dense pool above 64, and the re-ranker's lookup by row id.

The GPU index is replaced by a numpy stand-in that plays the kernel with the oracle's arithmetic
(test infrastructure only -- the product has no CPU path); the same scenarios run on the real index in
tests/test_gpu_backend.py."""
import json
import threading
import zlib

import numpy as np
import pytest

import mrag_amd.index as index_mod
from mrag_amd import corpus, fusion
from mrag_amd.backend import DenseRetrievalBackend, HipDenseReranker
from mrag_amd.dto import RetrievalIn
from oracle import dense_search as ods
from oracle import ref_semantics as rs

DIM = 16


def text_vec(text, dim=DIM):
    rng = np.random.default_rng(zlib.crc32(text.encode("utf-8")))
    return [float(x) for x in rng.standard_normal(dim)]


class CpuIndex:
    """numpy stand-in for mrag_amd.index.DenseIndex (fp16 storage, oracle arithmetic)."""
    made = []

    def __init__(self, dim, metric="cosine", dtype="f16", device=0):
        self.dim, self.dtype = dim, dtype
        self._rows = np.zeros((0, dim), dtype=np.float16)
        self.adds = []
        CpuIndex.made.append(self)

    def __len__(self):
        return self._rows.shape[0]

    def add(self, rows, normalize=None):
        r = np.asarray(rows, dtype=np.float32)
        r16 = ods.normalize_round(r) if normalize in (None, True) else r.astype(np.float16)
        self.adds.append(("rows", bool(normalize in (None, True))))
        self._rows = np.concatenate([self._rows, r16])

    def add_stored_bits(self, bits):
        self.adds.append(("bits", False))
        self._rows = np.concatenate([self._rows, np.asarray(bits, dtype=np.uint16).view(np.float16)])

    def stored_bits(self):
        return self._rows.view(np.uint16).copy()

    def max_k(self, nq=1):
        return 256

    def search(self, q, k):
        assert k <= self.max_k()
        return ods.brute_force_topk(ods.normalize_round(np.asarray(q, dtype=np.float32)), self._rows, k)

    def score_rows(self, q, ids):
        q16 = ods.normalize_round(np.asarray(q, dtype=np.float32)[None, :])[0]
        return (self._rows[np.asarray(ids)].astype(np.float64) @ q16.astype(np.float64)).astype(np.float32)

    def close(self):
        pass


@pytest.fixture()
def cpu_index(monkeypatch):
    CpuIndex.made = []
    monkeypatch.setattr(index_mod, "DenseIndex", CpuIndex)
    DenseRetrievalBackend._failed.clear()
    yield CpuIndex
    corpus.drop_shared("dense-index|")
    corpus.drop_shared("encoder|test")
    DenseRetrievalBackend._failed.clear()


class Prov:
    """Provider whose first embed() builds its 'encoder' through the process-wide registry, like
    HipEmbeddingProvider.encoder does (provider.py)."""

    def __init__(self, fail_calls=(), zero_calls=()):
        self.kwargs = {"embed_model": "fake-embed"}
        self.calls, self.fail_calls, self.zero_calls = [], set(fail_calls), set(zero_calls)

    def embed(self, *, model, texts, require):
        corpus.shared("encoder|test", lambda: object())
        i = len(self.calls)
        self.calls.append(len(texts))
        if i in self.fail_calls:
            raise RuntimeError("injected provider failure")
        return {"vectors": [text_vec(t) for t in texts]}


class Router:
    def __init__(self, prov):
        self.providers, self.policy = {"hip": prov}, {"embedding_provider": "hip"}

    def embed(self, *, model_hint, texts, require=None):
        return rs.router_embed(self.providers, self.policy, model_hint=model_hint, texts=texts, require=require)


def write_docs(tmp_path, n=300):
    rows = [{"doc_id": f"T{i // 4}#{i % 4}", "title": f"T{i // 4}", "sent_id": i % 4,
             "text": f"sentence {i} about topic {i % 17} and {i % 5}"} for i in range(n)]
    p = tmp_path / "docs.jsonl"
    corpus.write_docs_jsonl(p, rows)
    return p, rows


def test_first_run_with_a_fresh_provider_does_not_deadlock(tmp_path, cpu_index):
    """the index build ran under the registry's global lock and reached the provider's
    encoder -> shared() again -> hang on the first question of every fresh process."""
    p, rows = write_docs(tmp_path, 40)
    corpus.drop_shared("encoder|test")
    be = DenseRetrievalBackend(Router(Prov()), index_path=str(p), embed_batch=16)
    out = {}
    t = threading.Thread(target=lambda: out.update(r=be.run(RetrievalIn(query="topic 3", graph_id="", top_k=5, trace_id="t"))),
                         daemon=True)
    t.start()
    t.join(30)
    assert not t.is_alive(), "DenseRetrievalBackend.run() hung (registry lock held across build)"
    assert out["r"]["diagnostics"]["dense_error"] is None and len(out["r"]["hits"]) == 5


def test_shared_builder_failure_is_not_cached_and_reentry_is_detected():
    n = []

    def bad():
        n.append(1)
        raise ValueError("boom")
    for _ in range(2):
        with pytest.raises(ValueError):
            corpus.shared("t-fail", bad)
    assert n == [1, 1]
    with pytest.raises(RuntimeError, match="re-entered"):
        corpus.shared("t-self", lambda: corpus.shared("t-self", lambda: 1))
    assert corpus.shared("t-self", lambda: 7) == 7
    corpus.drop_shared("t-")


def test_router_embed_failure_never_builds_or_caches_an_index(tmp_path, cpu_index):
    """llm_router.py:124-129 turns a provider error into [[0.0]*3]*n; a corpus batch like
    that must not become a dim-3 zero index, on disk or in the registry."""
    p, rows = write_docs(tmp_path, 40)
    prov = Prov(fail_calls={2})                     # probe ok, first corpus batch ok, second fails
    be = DenseRetrievalBackend(Router(prov), index_path=str(p), embed_batch=16, cache_dir=str(tmp_path / "cache"))
    req = RetrievalIn(query="topic 3", graph_id="", top_k=5, trace_id="t")
    r1 = be.run(req)
    assert r1["hits"] == [] and "dim 3" in r1["diagnostics"]["dense_error"]
    assert not list((tmp_path / "cache").glob("*.npy")) and not any(len(ix) for ix in CpuIndex.made)
    n_calls = len(prov.calls)
    r2 = be.run(req)                                # inside the backoff window: reported, no re-embed storm
    assert "build failed" in r2["diagnostics"]["dense_error"] and len(prov.calls) == n_calls
    DenseRetrievalBackend._failed.clear()           # backoff over, provider healthy again
    prov.fail_calls = set()
    r3 = be.run(req)
    assert r3["diagnostics"]["dense_error"] is None and len(r3["hits"]) == 5
    # query-time failure (zero vector of the wrong dim) is reported too, and the index stays usable
    prov.fail_calls = {len(prov.calls)}
    r4 = be.run(req)
    assert "query embedding has dim 3" in r4["diagnostics"]["dense_error"]
    assert be.run(req)["diagnostics"]["dense_error"] is None


def test_warm_cache_readds_the_cold_bits_verbatim(tmp_path, cpu_index):
    p, rows = write_docs(tmp_path, 50)
    kw = dict(index_path=str(p), embed_batch=16, cache_dir=str(tmp_path / "cache"))
    req = RetrievalIn(query="topic 3", graph_id="", top_k=7, trace_id="t")
    cold = DenseRetrievalBackend(Router(Prov()), **kw)
    r_cold = cold.run(req)
    ix_cold = CpuIndex.made[-1]
    assert [a[0] for a in ix_cold.adds] == ["rows"]
    corpus.drop_shared("dense-index|")
    prov = Prov()
    warm = DenseRetrievalBackend(Router(prov), **kw)
    r_warm = warm.run(req)
    ix_warm = CpuIndex.made[-1]
    assert ix_warm is not ix_cold and [a[0] for a in ix_warm.adds] == ["bits"]
    assert prov.calls == [1, 1]                      # probe + query: the corpus is not re-embedded
    assert (ix_warm.stored_bits() == ix_cold.stored_bits()).all()
    assert r_warm["hits"] == r_cold["hits"]
    npy = list((tmp_path / "cache").glob("*.npy"))
    assert len(npy) == 1 and np.load(npy[0]).dtype == np.uint16


def test_dense_pool_above_64_with_a_text_channel(tmp_path, cpu_index):
    """ids ranked 65..200 must carry a dense score (retrieval_backend.py:218,245)."""
    p, rows = write_docs(tmp_path, 300)
    t_hits = [{"id": fusion.raw_hit_id(r), "score": float(300 - i), "meta": fusion.row_meta(r, "bm25")}
              for i, r in enumerate(rows[:200])]
    be = DenseRetrievalBackend(Router(Prov()), index_path=str(p), embed_batch=64, dense_pool_k=200,
                               text_search=lambda queries, top_k: t_hits[:top_k])
    out = be.run(RetrievalIn(query="topic 3 and 4", graph_id="", top_k=250, trace_id="t"))
    d = out["diagnostics"]
    assert d["pool"] == {"dense_pool_k": 200, "final_top_k": 250, "dense_pool_k_effective": 250}
    assert d["dense_scored"] == 250
    # oracle: the same fusion over an exact top-250
    q16 = ods.normalize_round(np.asarray([text_vec("topic 3 and 4")], dtype=np.float32))
    c16 = ods.normalize_round(np.asarray([text_vec(r["text"]) for r in rows], dtype=np.float32))
    sv, si = ods.brute_force_topk(q16, c16, 250)
    dense_hits = [{"id": fusion.raw_hit_id(rows[int(i)]), "score": float(s), "meta": fusion.row_meta(rows[int(i)], "dense")}
                  for s, i in zip(sv[0], si[0])]
    dn = rs.norm_map(dense_hits)
    want = rs.fuse(t_hits, [], {k: v["score"] for k, v in dn.items()}, alpha_text=0.4, alpha_graph=0.2, alpha_dense=0.4,
                   top_k=250)
    assert [(h["id"], h["score"]) for h in out["hits"]] == [(h["id"], h["score"]) for h in want]
    by_id = {h["id"]: h for h in out["hits"]}
    deep = [nid for nid in list(dn)[64:-1] if nid in by_id]      # dense ranks 65..249 that made the fused cut
    assert len(deep) > 100 and all(by_id[nid]["meta"]["score_dense_norm"] > 0.0 for nid in deep)


def test_reranker_scores_known_candidates_by_row_id(tmp_path, cpu_index, monkeypatch):
    """candidates that are rows of the cached corpus cost no embed call; the others are embedded
    with the reference's batching (retrieval_backend.py:234); scores agree with the reference arithmetic
    within fp16 storage rounding (1e-3)."""
    monkeypatch.setattr(HipDenseReranker, "_cosines",
                        lambda self, qv, vecs: [rs.cosine(list(qv), list(v)) for v in vecs])
    p, rows = write_docs(tmp_path, 120)
    prov = Prov()
    router = Router(prov)
    be = DenseRetrievalBackend(router, index_path=str(p), embed_batch=64)
    cands = [{"id": fusion.raw_hit_id(r), "score": 1.0, "meta": fusion.row_meta(r, "bm25")} for r in rows[10:70]]
    cands[5]["meta"]["text"] = "edited text that is not the stored row"          # same id, different text -> embedded
    cands.append({"id": "sent::elsewhere::1", "score": 0.5, "meta": {"text": "a passage from another corpus"}})
    want = rs.dense_score(lambda **kw: router.embed(**kw), query="what about topic 7", candidates=cands,
                          trace_id="t", max_pool=200, embed_batch=50, model_hint="fake-embed")
    n_ref_calls = len(prov.calls)
    assert n_ref_calls == 1 + 2                      # the reference's sequence: query + ceil(61/50) batches
    rr = HipDenseReranker(router, max_pool=200, embed_batch=50, corpus=be)
    got = rr.score(query="what about topic 7", candidates=cands, trace_id="t")
    lookup_calls = prov.calls[n_ref_calls:]
    # query, [index build: probe + 2 corpus batches], ONE batch for the two unknown candidates
    assert lookup_calls == [1, 1, 64, 56, 2]
    assert list(got) == list(want)
    assert max(abs(got[k] - want[k]) for k in want) < 1e-3
    prov.calls.clear()
    rr.score(query="second question", candidates=cands, trace_id="t")
    assert prov.calls == [1, 2]                      # index is process-wide: query + the two unknowns
