"""BM25 text channel.  CPU: the oracle restatement and the host-side index builder against F7
(the reference's own BM25LiteIndex / BM25TextSearcher outputs, tests/golden/make_golden_bm25.py).  GPU: the
device search against F7 and the oracle -- fp64 scores bit for bit, order exact up to the reference's
set-iteration tie order (tie groups compared as sets; this library's declared order is doc ascending)."""
import json

import numpy as np
import pytest

from oracle import bm25 as ob


def _tie_groups(ranked):
    groups, cur, last = [], set(), None
    for d, s in ranked:
        if last is not None and s != last:
            groups.append((last, cur)); cur = set()
        cur.add(int(d)); last = s
    if cur:
        groups.append((last, cur))
    return groups


def _same_ranking(got, want_full, top_k):
    """got == the reference's ranking cut at top_k: scores identical position by position; documents identical
    inside complete tie groups; in a tie group cut by top_k any members of the reference's FULL group qualify."""
    want = want_full[:top_k]
    assert [s for _, s in got] == [s for _, s in want]
    full_groups = {s: g for s, g in _tie_groups(want_full)}
    for s, g in _tie_groups(got):
        assert g <= full_groups[s]
        if len(g) == len(full_groups[s]) or s != got[-1][1]:
            assert g == full_groups[s]
    docs = [d for d, _ in got]
    assert len(set(docs)) == len(docs)


def _same_hits(got, want):
    """Hit dicts (id, score, meta) equal position by position wherever the score is not tied."""
    scores = [h["score"] for h in got]
    assert scores == [h["score"] for h in want]
    for i, (g, w) in enumerate(zip(got, want)):
        if scores.count(scores[i]) == 1:
            assert g == w
        assert set(g) == set(w) and set(g["meta"]) == set(w["meta"]) and g["meta"]["source"] == "bm25"


@pytest.fixture(scope="module")
def f7(golden_dir):
    return json.loads((golden_dir / "f7_bm25.json").read_text())


def test_oracle_matches_reference_f7(f7):
    for name, c in f7.items():
        o = ob.Bm25Oracle(c["rows"], k1=c["k1"], b=c["b"])
        assert (o.N, o.avgdl, o.doc_lens) == (c["N"], c["avgdl"], c["doc_lens"]) and dict(sorted(o.df.items())) == c["df"]
        for case in c["cases"]:
            got = o.search(case["queries"], top_k=case["top_k"], alpha_merge=case["alpha_merge"])
            _same_ranking(got, [tuple(x) for x in case["full"]], case["top_k"])
            # every candidate's score, exactly (the reference's left-to-right fp64 sum)
            full = dict(o.search(case["queries"], top_k=10 ** 9, alpha_merge=case["alpha_merge"]))
            assert full == {int(d): s for d, s in case["full"]}
            if "hits" in case:
                _same_hits(o.hits(case["queries"], case["top_k"]), case["hits"])


def test_host_builder_matches_reference_counts_and_csr_is_consistent(f7):
    from mrag_amd.text_index import build_postings, tokenize
    assert tokenize("Café naïve apple-banana_cherry 42") == ob.tokenize("Café naïve apple-banana_cherry 42") == ["caf", "na", "ve", "apple", "banana", "cherry", "42"]
    for name, c in f7.items():
        p = build_postings(c["rows"], c["k1"], c["b"])
        assert (p["N"], p["avgdl"], p["doc_lens"]) == (c["N"], c["avgdl"], c["doc_lens"]) and dict(sorted(p["df"].items())) == c["df"]
        o = ob.Bm25Oracle(c["rows"], k1=c["k1"], b=c["b"])
        for t, i in p["term_id"].items():
            lo, hi = p["indptr"][i], p["indptr"][i + 1]
            docs, tfs = p["post_doc"][lo:hi], p["post_tf"][lo:hi]
            assert (np.diff(docs) > 0).all()                                  # ascending, unique
            assert dict(zip(docs.tolist(), tfs.tolist())) == o.tf[t]
        avg = c["avgdl"] or 1.0
        assert p["doc_norm"].tolist() == [c["k1"] * (1 - c["b"] + c["b"] * (dl / avg)) for dl in c["doc_lens"]]


@pytest.mark.gpu
def test_device_search_matches_reference_f7(f7):
    from mrag_amd.text_index import HipBM25Index, HipBM25TextSearcher
    for name, c in f7.items():
        ix = HipBM25Index(rows=c["rows"], k1=c["k1"], b=c["b"])
        for case in c["cases"]:
            got = ix.search(case["queries"], top_k=case["top_k"], alpha_merge=case["alpha_merge"])
            _same_ranking(got, [tuple(x) for x in case["full"]], case["top_k"])
            full = ix.search(case["queries"], top_k=10 ** 9, alpha_merge=case["alpha_merge"])
            assert dict(full) == {int(d): s for d, s in case["full"]}          # every candidate, bit for bit
            assert full == sorted(full, key=lambda kv: (-kv[1], kv[0]))         # declared order
            if "hits" in case:
                _same_hits(HipBM25TextSearcher(ix).search(queries=case["queries"], top_k=case["top_k"]), case["hits"])
        ix.close()


@pytest.mark.gpu
def test_device_search_matches_oracle_on_a_large_synthetic_corpus():
    """200 000 sentences of a Zipf vocabulary (common words in every other sentence: candidates ~ N, like real text
    without a stop list), expanded query sets, max and sum merges, k up to 4096 and beyond."""
    from mrag_amd.text_index import HipBM25Index
    from mrag_amd._native import MragError
    rng = np.random.default_rng(7)
    vocab = [f"w{i}" for i in range(5000)]
    p = 1.0 / np.arange(1, 5001) ** 1.1
    p /= p.sum()
    n = 200_000
    lens = rng.integers(3, 25, size=n)
    words = rng.choice(5000, size=int(lens.sum()), p=p)
    rows, at = [], 0
    for i, L in enumerate(lens):
        rows.append({"doc_id": f"T{i // 5}#{i % 5}", "title": f"T{i // 5}", "sent_id": i % 5,
                     "text": " ".join(vocab[w] for w in words[at:at + L])})
        at += L
    ix = HipBM25Index(rows=rows)
    o = ob.Bm25Oracle(rows)
    assert (ix.N, ix.avgdl) == (o.N, o.avgdl)
    for queries, k, merge in ((["w0 w3 w77 w4000"], 200, "max"), (["w1 w2", "w9 w9 w1500", "w4999 w17 w2"], 200, "max"),
                              (["w5 w600", "w5 w7"], 1000, "sum"), (["w4998 w4997"], 4096, "max"), (["w0"], 50, "max")):
        got = ix.search(queries, top_k=k, alpha_merge=merge)
        want = o.search(queries, top_k=k, alpha_merge=merge)
        assert got == want, (queries, k, merge)
    with pytest.raises(MragError):
        ix.search(["w0"], top_k=5000)
    ix.close()


@pytest.mark.gpu
def test_backend_with_device_bm25_channel(tmp_path):
    """DenseRetrievalBackend(text_channel="bm25"): text and dense channels both on the device, fused with the
    reference rule (retrieval_backend.py:350-372) -- checked against the oracle end to end."""
    import zlib
    from mrag_amd import corpus, fusion
    from mrag_amd.backend import DenseRetrievalBackend
    from mrag_amd.dto import RetrievalIn
    from oracle import dense_search as ds
    from oracle import ref_semantics as rs

    def text_vec(text, dim=64):
        rng = np.random.default_rng(zlib.crc32(text.encode("utf-8")))
        return [float(x) for x in rng.standard_normal(dim)]

    class Prov:
        kwargs = {"embed_model": "fake-embed"}
        def embed(self, *, model, texts, require):
            return {"vectors": [text_vec(t) for t in texts]}

    class Router:
        providers, policy = {"hip": Prov()}, {"embedding_provider": "hip"}
        def embed(self, *, model_hint, texts, require=None):
            return rs.router_embed(self.providers, self.policy, model_hint=model_hint, texts=texts, require=require)

    rng = np.random.default_rng(3)
    words = ["alpha", "beta", "gamma", "delta", "river", "city", "born", "film", "band", "album", "war", "king"]
    rows = [{"doc_id": f"Title {i // 4}#{i % 4}", "title": f"Title {i // 4}", "sent_id": i % 4,
             "text": " ".join(rng.choice(words, size=int(rng.integers(4, 12))))} for i in range(800)]
    docs = tmp_path / "docs.jsonl"
    corpus.write_docs_jsonl(docs, rows)
    corpus.drop_shared("dense-index|"); corpus.drop_shared("bm25-index|")
    be = DenseRetrievalBackend(Router(), index_path=str(docs), text_channel="bm25", dense_pool_k=100, fuse_on_device=True)
    req = RetrievalIn(query="gamma river born king", graph_id="", top_k=15, trace_id="t")
    out = be.run(req)
    assert out["diagnostics"]["dense_error"] is None and out["diagnostics"]["text_error"] is None
    assert out["diagnostics"]["bm25_candidates"] == 100
    t_hits = ob.Bm25Oracle(rows).hits([req.query], 100)
    c16 = ds.normalize_round(np.asarray([text_vec(r["text"]) for r in rows], dtype=np.float32))
    q16 = ds.normalize_round(np.asarray([text_vec(req.query)], dtype=np.float32))
    sv, si = ds.brute_force_topk(q16, c16, 100)
    dn = rs.norm_map([{"id": fusion.raw_hit_id(rows[int(i)]), "score": float(s), "meta": fusion.row_meta(rows[int(i)], "dense")}
                      for s, i in zip(sv[0], si[0])])
    want = rs.fuse(t_hits, [], {k: v["score"] for k, v in dn.items()}, alpha_text=0.4, alpha_graph=0.2, alpha_dense=0.4, top_k=15)
    np.testing.assert_allclose([h["score"] for h in out["hits"]], [h["score"] for h in want], rtol=0, atol=1e-4)
    assert len({h["id"] for h in out["hits"]} & {h["id"] for h in want}) >= 13     # BM25 ties cut at the pool edge may differ
    corpus.drop_shared("dense-index|"); corpus.drop_shared("bm25-index|")


@pytest.mark.gpu
def test_fusion_on_device_matches_host_fusion_and_f3(golden_dir):
    """a7 on the device (mrag_fuse_topk) vs the host fusion (itself bit-exact to F3 = the reference's run()): ids,
    fused scores and the three score_*_norm values identical, ties included (same declared order)."""
    from mrag_amd import fusion
    data = json.loads((golden_dir / "f3_hybrid_run.json").read_text())
    for c in data["cases"]:
        kw = c["backend_kwargs"]
        args = dict(alpha_text=kw["alpha_text"], alpha_graph=kw["alpha_graph"], alpha_dense=kw["alpha_dense"])
        for top_k in (int(c["req"]["top_k"] or kw["default_top_k"]), 10 ** 9, 1):
            host = fusion.fuse_channels(c["t_hits_raw"], c["g_hits_raw"], c["dense_scores_raw"], top_k=top_k, **args)
            dev = fusion.fuse_channels_device(c["t_hits_raw"], c["g_hits_raw"], c["dense_scores_raw"], top_k=top_k, **args)
            assert dev == host
        want = c["run_out"]["hits"]
        dev = fusion.fuse_channels_device(c["t_hits_raw"], c["g_hits_raw"], c["dense_scores_raw"], top_k=len(want), **args)
        assert [h["score"] for h in dev] == [h["score"] for h in want]
    # synthetic: duplicate normalised ids inside a channel (strictly larger score wins), exact ties, negative and
    # all-equal channels, an empty channel, ids present in one / two / three channels
    rng = np.random.default_rng(5)
    for trial in range(6):
        def hits(n, lo, dup):
            out = []
            for i in range(n):
                t = int(rng.integers(0, 40)) if dup else i
                out.append({"id": f"x{i}", "score": float(np.round(rng.normal(lo, 2.0), 1)),
                            "meta": {"doc": f"T{t}", "sent_id": int(rng.integers(0, 3)), "text": f"s{i}", "m%d" % (i % 3): i}})
            return out
        t_hits, g_hits = hits(int(rng.integers(0, 120)), 5.0, True), hits(int(rng.integers(0, 30)), -3.0, trial % 2 == 0)
        if trial == 3:
            for h in g_hits:
                h["score"] = 2.5                                  # all-equal channel -> norms 0.0
        dense = {f"sent::T{int(rng.integers(0, 60))}::{int(rng.integers(0, 3)) or ''}": float(np.round(rng.uniform(-1, 1), 2))
                 for _ in range(int(rng.integers(0, 200)))}
        for top_k in (7, 10 ** 9):
            host = fusion.fuse_channels(t_hits, g_hits, dense, alpha_text=0.4, alpha_graph=0.2, alpha_dense=0.4, top_k=top_k)
            dev = fusion.fuse_channels_device(t_hits, g_hits, dense, alpha_text=0.4, alpha_graph=0.2, alpha_dense=0.4, top_k=top_k)
            assert dev == host, trial
    assert fusion.fuse_channels_device([], [], {}, alpha_text=1, alpha_graph=1, alpha_dense=1, top_k=5) == []
