"""GPU parity for the round-2 boundary fixes: k above 64 (the reference's 200-candidate pool), cold == warm
cache bits on the real index, row-id scoring for the re-ranker.  Oracle = oracle.dense_search (pinned by F5)
and oracle.ref_semantics (pinned by F1-F4)."""
import zlib

import numpy as np
import pytest

from oracle import dense_search as ds
from oracle import ref_semantics as rs

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _check_wide(ix, q16, c16, k, id_base=0):
    sc, ids = ix.search(q16, k, normalize=False)
    rv, ri = ds.brute_force_topk(q16, c16, k)
    ri = np.where(ri >= 0, ri + id_base, ri)
    valid = ri >= 0
    assert ((ids >= 0) == valid).all() and np.isneginf(sc[~valid]).all()
    np.testing.assert_allclose(sc[valid], rv[valid], rtol=0, atol=TOL)
    assert (np.diff(sc, axis=1)[valid[:, 1:]] <= 0).all()
    strict, bad = ds.gap_aware_id_match(ids, sc, ri, rv, tol=TOL)
    assert bad == 0, (strict, bad)
    assert ds.recall_at_k(ids, ri) >= 0.999
    return sc, ids


@pytest.mark.parametrize("nq,n,d,k", [(1, 100_000, 768, 200), (1, 3000, 384, 256), (8, 50_000, 128, 100),
                                      (9, 20_000, 64, 65), (21, 70_000, 256, 200), (3, 150, 32, 200),
                                      (300, 9000, 96, 72)])
def test_k_above_64_matches_oracle(nq, n, d, k):
    """retrieval_backend.py:218,245: the dense pool is 200 per question; 64 < k <= 256 runs the streaming
    kernel with 512-entry lists + the wide merge.  (3, 150, 32, 200): k beyond the corpus -> (-inf, -1) tail."""
    from mrag_amd.index import DenseIndex
    c16 = ds.normalize_round(ds.make_gaussian(n, d, 77))
    q16 = ds.normalize_round(ds.make_gaussian(nq, d, 78))
    ix = DenseIndex(d)
    ix.set_id_base(1000)
    ix.add(c16, normalize=False)
    assert ix.max_k(nq) == 256
    _check_wide(ix, q16, c16, k, id_base=1000)


@pytest.mark.parametrize("k", [100, 200, 256])
def test_wide_batch_k_above_64(k):
    """Batch evaluation with the reference's pool (retrieval_backend.py:218: 200 candidates) -- more than 32 queries with
    64 < k <= 256: the batch kernel at k' = 64 per corpus split + merge + the exactness check, flagged queries redone by the
    streaming kernel (csrc/bf_index.hip bf_wide_verify_kernel).  i.i.d. rows: (almost) nothing is flagged; then rows stored
    cluster by cluster with k neighbours inside ONE cluster -- one split holds a query's whole top k, everything it touches is
    flagged and redone -- and exact integer ties.  Always the oracle's answer."""
    from mrag_amd.index import DenseIndex
    n, nq, d = 70_000, 700, 64
    c16 = ds.normalize_round(ds.make_gaussian(n, d, 81))
    q16 = ds.normalize_round(ds.make_gaussian(nq, d, 82))
    ix = DenseIndex(d)
    ix.set_id_base(5000)
    ix.add(c16, normalize=False)
    _check_wide(ix, q16, c16, k, id_base=5000)
    assert ix.last_wide_redone() <= nq // 20
    # fewer than 33 queries keep the streaming kernel (and reset the counter)
    _check_wide(ix, q16[:9], c16, k, id_base=5000)
    assert ix.last_wide_redone() == 0
    ix.close()
    # clustered storage order: 280 clusters x 250 adjacent rows, every query sits on a cluster centre
    rng = np.random.default_rng(83)
    cent = rng.standard_normal((280, d)).astype(np.float32)
    rows = (np.repeat(cent, 250, axis=0) + 0.05 * rng.standard_normal((280 * 250, d)).astype(np.float32))
    c2 = ds.normalize_round(rows)
    q2 = ds.normalize_round(cent[rng.integers(0, 280, size=200)] + 0.01 * rng.standard_normal((200, d)).astype(np.float32))
    ix2 = DenseIndex(d)
    ix2.add(c2, normalize=False)
    _check_wide(ix2, q2, c2, k, id_base=0)
    assert ix2.last_wide_redone() > 0          # (the check did fire: those answers came from the exact redo)
    ix2.close()
    # integer-valued rows, inner product: exact fp32 sums, real ties on the k-th score -> ids must equal the oracle's
    ci = rng.integers(-2, 3, size=(66_000, d)).astype(np.float16)
    qi = rng.integers(-2, 3, size=(40, d)).astype(np.float16)
    ix3 = DenseIndex(d, metric="ip")
    ix3.add(ci, normalize=False)
    sc, ids = ix3.search(qi, k, normalize=False)
    rv, ri = ds.brute_force_topk(qi, ci, k)
    assert (ids == ri).all() and (sc == rv).all()
    ix3.close()


def test_k_above_64_ties_adversarial_and_limit():
    from mrag_amd.index import DenseIndex
    from mrag_amd._native import MragError
    d, k = 64, 130
    # integer-valued rows: fp32 accumulation is exact, ties are real -> ids must match exactly (row asc)
    rng = np.random.default_rng(9)
    c = rng.integers(-2, 3, size=(6000, d)).astype(np.float16)
    q = rng.integers(-2, 3, size=(5, d)).astype(np.float16)
    ix = DenseIndex(d, metric="ip")
    ix.add(c, normalize=False)
    sc, ids = ix.search(q, k, normalize=False)
    rv, ri = ds.brute_force_topk(q, c, k)
    assert (ids == ri).all() and (sc == rv).all()
    # every later row beats all earlier ones: each 64-row append passes whole, compaction every few chunks
    base = np.zeros((9000, d), np.float16)
    base[:, 0] = np.linspace(0.01, 1.0, 9000).astype(np.float16)
    ix2 = DenseIndex(d, metric="ip")
    ix2.add(base, normalize=False)
    q2 = np.zeros((2, d), np.float16); q2[:, 0] = 1
    sc2, ids2 = ix2.search(q2, 200, normalize=False)
    rv2, ri2 = ds.brute_force_topk(q2, base, 200)
    assert (ids2 == ri2).all() and (sc2 == rv2).all()
    with pytest.raises(MragError):
        ix.search(q, 257, normalize=False)          # above mrag_index_max_k: loud, never clamped


def test_bf16_k200():
    import torch
    from mrag_amd.index import DenseIndex
    n, d, k = 30_000, 128, 200
    c = torch.from_numpy(ds.l2_normalize(ds.make_gaussian(n, d, 1))).to(torch.bfloat16)
    q = torch.from_numpy(ds.l2_normalize(ds.make_gaussian(2, d, 2))).to(torch.bfloat16)
    ix = DenseIndex(d, dtype="bf16")
    ix.add(c, normalize=False)
    sc, ids = ix.search(q, k, normalize=False)
    rv, ri = ds.brute_force_topk(q.float().numpy(), c.float().numpy(), k)
    np.testing.assert_allclose(sc, rv, rtol=0, atol=TOL)
    assert ds.gap_aware_id_match(ids, sc, ri, rv, tol=TOL)[1] == 0


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_stored_bits_roundtrip_is_verbatim(dtype):
    """The embedding cache keeps what K1 produced; re-adding it must reproduce the index bit for bit
    (fp16 -> renormalise -> second rounding moved bits between cold and warm starts)."""
    from mrag_amd.index import DenseIndex
    x = ds.make_gaussian(5000, 200, 3) * 3.7            # un-normalised fp32 "encoder output"
    cold = DenseIndex(200, dtype=dtype)
    cold.add(x)                                          # K1: normalise (fp64 norm) + ONE rounding
    bits = cold.stored_bits()
    assert bits.dtype == np.uint16 and bits.shape == (5000, 200)
    warm = DenseIndex(200, dtype=dtype)
    warm.add_stored_bits(bits[:3000]); warm.add_stored_bits(bits[3000:])
    assert (warm.stored_bits() == bits).all() and (warm.rows() == cold.rows()).all()
    q = ds.make_gaussian(40, 200, 4)
    s1, i1 = cold.search(q, 10)
    s2, i2 = warm.search(q, 10)
    assert (i1 == i2).all() and (s1 == s2).all()
    if dtype == "f16":
        assert (bits.view(np.float16) == ds.normalize_round(x)).all()      # = the oracle's K1


def test_score_rows_matches_oracle_dot():
    from mrag_amd.index import DenseIndex
    n, d = 4000, 384
    c16 = ds.normalize_round(ds.make_gaussian(n, d, 5))
    ix = DenseIndex(d)
    ix.add(c16, normalize=False)
    q = ds.make_gaussian(1, d, 6)[0]
    ids = np.asarray([0, 17, 3999, 17, -1, 4000, 123], dtype=np.int64)
    got = ix.score_rows(q, ids)
    q16 = ds.normalize_round(q[None, :])[0].astype(np.float64)
    want = np.asarray([float(c16[i].astype(np.float64) @ q16) if 0 <= i < n else 0.0 for i in ids])
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-6)
    assert got[4] == 0.0 and got[5] == 0.0
    # agrees with the top-k kernel's scores for the same rows
    sc, top = ix.search(q[None, :], 10)
    np.testing.assert_allclose(ix.score_rows(q, top[0]), sc[0], rtol=0, atol=TOL)


def test_backend_pool_200_cold_warm_and_reranker_lookup(tmp_path):
    """DenseRetrievalBackend on the real index: pool of 200 served in full with a text channel injected,
    warm-cache run == cold run (hits and stored bits), and HipDenseReranker(corpus=...) scoring known
    candidates by row id -- against the reference arithmetic restated in the oracle."""
    from mrag_amd import corpus, fusion
    from mrag_amd.backend import DenseRetrievalBackend, HipDenseReranker
    from mrag_amd.dto import RetrievalIn
    dim = 96

    def text_vec(text):
        rng = np.random.default_rng(zlib.crc32(text.encode("utf-8")))
        return [float(x) for x in rng.standard_normal(dim)]

    class Prov:
        kwargs = {"embed_model": "fake-embed"}
        calls = []
        def embed(self, *, model, texts, require):
            Prov.calls.append(len(texts))
            return {"vectors": [text_vec(t) for t in texts]}

    class Router:
        providers, policy = {"hip": Prov()}, {"embedding_provider": "hip"}
        def embed(self, *, model_hint, texts, require=None):
            return rs.router_embed(self.providers, self.policy, model_hint=model_hint, texts=texts, require=require)

    rows = [{"doc_id": f"T{i // 4}#{i % 4}", "title": f"T{i // 4}", "sent_id": i % 4,
             "text": f"sentence {i} about topic {i % 37} and {i % 11}"} for i in range(1500)]
    docs = tmp_path / "docs.jsonl"
    corpus.write_docs_jsonl(docs, rows)
    t_hits = [{"id": fusion.raw_hit_id(r), "score": float(500 - i), "meta": fusion.row_meta(r, "bm25")}
              for i, r in enumerate(rows[:200])]
    kw = dict(index_path=str(docs), embed_batch=256, dense_pool_k=200, cache_dir=str(tmp_path / "cache"),
              text_search=lambda queries, top_k: t_hits[:top_k])
    router = Router()
    req = RetrievalIn(query="topic 3 and 4", graph_id="", top_k=20, trace_id="t")
    corpus.drop_shared("dense-index|")
    cold = DenseRetrievalBackend(router, **kw)
    r_cold = cold.run(req)
    assert r_cold["diagnostics"]["dense_error"] is None
    assert r_cold["diagnostics"]["pool"]["dense_pool_k_effective"] == 200 and r_cold["diagnostics"]["dense_scored"] == 200
    # oracle: exact top-200 on the oracle's K1 bits, reference fusion
    c16 = ds.normalize_round(np.asarray([text_vec(r["text"]) for r in rows], dtype=np.float32))
    q16 = ds.normalize_round(np.asarray([text_vec(req.query)], dtype=np.float32))
    sv, si = ds.brute_force_topk(q16, c16, 200)
    dense_hits = [{"id": fusion.raw_hit_id(rows[int(i)]), "score": float(s), "meta": fusion.row_meta(rows[int(i)], "dense")}
                  for s, i in zip(sv[0], si[0])]
    dn = rs.norm_map(dense_hits)
    want = rs.fuse(t_hits, [], {k: v["score"] for k, v in dn.items()}, alpha_text=0.4, alpha_graph=0.2, alpha_dense=0.4, top_k=20)
    assert [h["id"] for h in r_cold["hits"]] == [h["id"] for h in want]
    np.testing.assert_allclose([h["score"] for h in r_cold["hits"]], [h["score"] for h in want], rtol=0, atol=1e-4)
    bits_cold = cold._get_state("fake-embed", "t")["index"].stored_bits()
    assert (bits_cold.view(np.float16) == c16).all()
    # warm start in a "new process": registry dropped, cache on disk
    corpus.drop_shared("dense-index|")
    Prov.calls.clear()
    warm = DenseRetrievalBackend(router, **kw)
    r_warm = warm.run(req)
    assert Prov.calls == [1, 1]                                    # probe + query, corpus not re-embedded
    assert (warm._get_state("fake-embed", "t")["index"].stored_bits() == bits_cold).all()
    assert r_warm["hits"] == r_cold["hits"]
    # re-ranker lookup by row id: no embed call for the 60 known candidates
    cands = [{"id": fusion.raw_hit_id(r), "score": 1.0, "meta": fusion.row_meta(r, "bm25")} for r in rows[100:160]]
    cands.append({"id": "sent::elsewhere::1", "score": 0.5, "meta": {"text": "a passage from another corpus"}})
    ref = rs.dense_score(router.embed, query="what about topic 7", candidates=cands, trace_id="t", model_hint="fake-embed")
    Prov.calls.clear()
    got = HipDenseReranker(router, corpus=warm).score(query="what about topic 7", candidates=cands, trace_id="t")
    assert Prov.calls == [1, 1] and list(got) == list(ref)
    assert max(abs(got[k] - ref[k]) for k in ref) < 1e-3            # fp16 storage of the cached rows
    corpus.drop_shared("dense-index|")


def test_online_merge_fallback_path_matches(monkeypatch):
    """The online merge reads only the heads of the per-split sorted lists (K4s) and leaves queries whose loads
    would not fit its LDS budget to the global-memory radix select (K4w).  MRAG_K4S_CAP=0 (read once per process:
    a subprocess) sends every query down the fallback; both must give the oracle's answer."""
    import subprocess, sys, os
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    code = r'''
import sys; sys.path.insert(0, %r)
import numpy as np
from mrag_amd.index import DenseIndex
from oracle import dense_search as ds
c16 = ds.normalize_round(ds.make_gaussian(40000, 64, 1)); q16 = ds.normalize_round(ds.make_gaussian(5, 64, 2))
ix = DenseIndex(64); ix.add(c16, normalize=False)
for k in (10, 64, 200):
    sc, ids = ix.search(q16, k, normalize=False)
    rv, ri = ds.brute_force_topk(q16, c16, k)
    assert np.allclose(sc, rv, rtol=0, atol=1e-5) and ds.gap_aware_id_match(ids, sc, ri, rv, tol=1e-5)[1] == 0, k
print("ok")
''' % str(root)
    for cap in ("0", "64", None):
        env = dict(os.environ)
        if cap is not None:
            env["MRAG_K4S_CAP"] = cap
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "ok" in r.stdout, (cap, r.stdout[-500:], r.stderr[-1500:])
