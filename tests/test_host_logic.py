"""CPU tests of the host-side boundary code (fusion, hit normalisation, reranker call sequence,
adapter, corpus reader) against the golden fixtures captured from the reference and against the
oracle restatement."""
import json
import math
import zlib

import numpy as np
import pytest

from mrag_amd import fusion, corpus
from mrag_amd.adapter import DenseRetrievalAgent
from mrag_amd.backend import HipDenseReranker, DenseRetrievalBackend, resolve_embed_model
from mrag_amd.dto import RetrievalIn, RetrievalOut, Hit
from oracle import ref_semantics as rs


def text_vec(text, dim=8):
    rng = np.random.default_rng(zlib.crc32(text.encode("utf-8")))
    return [float(x) for x in rng.standard_normal(dim)]


class Prov:
    def __init__(self, spec=None):
        self.kwargs = {"embed_model": "fake-embed"}
        self.spec = spec or {"dim": 8, "fail_calls": [], "bare_list": False}
        self.calls = []

    def embed(self, *, model, texts, require):
        i = len(self.calls)
        self.calls.append({"model": model, "n": len(texts), "require": dict(require)})
        if i in self.spec["fail_calls"]:
            raise RuntimeError("injected")
        v = [text_vec(t, self.spec["dim"]) for t in texts]
        return v if self.spec["bare_list"] else {"vectors": v}


class Router:
    """LLMRouter.embed restated for tests (oracle.ref_semantics.router_embed)."""

    def __init__(self, providers, policy):
        self.providers, self.policy = providers, policy

    def embed(self, *, model_hint, texts, require=None):
        return rs.router_embed(self.providers, self.policy, model_hint=model_hint, texts=texts, require=require)


class RaisingRouter:
    def __init__(self, fail):
        self.policy, self.providers, self.fail, self.n = {}, {}, set(fail), 0

    def embed(self, *, model_hint, texts, require=None):
        i = self.n
        self.n += 1
        if i in self.fail:
            raise RuntimeError("router-level")
        return {"vectors": [text_vec(t) for t in texts]}


@pytest.fixture()
def cpu_cosine(monkeypatch):
    """The reranker's cosine kernel needs a GPU; here the oracle stands in so the CALL
    SEQUENCE / error policy can be checked on CPU (the kernel itself: tests/test_gpu_*)."""
    def fake(self, qv, vecs):
        return [rs.cosine(list(qv), list(v)) if v is not None else 0.0 for v in vecs]
    monkeypatch.setattr(HipDenseReranker, "_cosines", fake)


def test_f2_reranker_call_sequence_and_scores(golden_dir, cpu_cosine):
    cases = json.loads((golden_dir / "f2_dense_score.json").read_text())["cases"]
    for c in cases:
        if c["provider"] is not None:
            prov = Prov(c["provider"])
            policy = {"no_provider_named": {}, "policy_embedding_model_wins":
                      {"embedding_provider": "hip", "embedding": [{"model": "policy-model", "provider": "hip"}]}
                      }.get(c["name"], {"embedding_provider": "hip"})
            router = Router({"hip": prov}, policy)
        else:
            prov, router = None, RaisingRouter(c["router_fail_calls"])
        rr = HipDenseReranker(router, max_pool=c["max_pool"], embed_batch=c["embed_batch"])
        assert rr._resolve_embed_model() == c["resolved_model"], c["name"]
        got = rr.score(query=c["query"], candidates=c["candidates"], trace_id="t-f2")
        assert got == c["out"], c["name"]
        if prov is not None:
            assert prov.calls == c["calls"], c["name"]


def test_f3_fusion_matches_reference_run(golden_dir):
    data = json.loads((golden_dir / "f3_hybrid_run.json").read_text())
    for c in data["cases"]:
        kw = c["backend_kwargs"]
        top_k = int(c["req"]["top_k"] or kw["default_top_k"])
        full = fusion.fuse_channels(c["t_hits_raw"], c["g_hits_raw"], c["dense_scores_raw"], alpha_text=kw["alpha_text"],
                                    alpha_graph=kw["alpha_graph"], alpha_dense=kw["alpha_dense"], top_k=10 ** 9)
        want = c["run_out"]["hits"]
        assert [h["score"] for h in full[:top_k]] == [h["score"] for h in want]
        by_id = {h["id"]: h for h in full}
        for h in want:
            assert by_id[h["id"]]["meta"] == h["meta"] and by_id[h["id"]]["score"] == h["score"]
        # the oracle restatement agrees everywhere, ties included (same declared tie-break)
        assert full == rs.fuse(c["t_hits_raw"], c["g_hits_raw"], c["dense_scores_raw"], alpha_text=kw["alpha_text"],
                               alpha_graph=kw["alpha_graph"], alpha_dense=kw["alpha_dense"], top_k=10 ** 9)
        for h in c["t_hits_raw"]:
            assert fusion.raw_hit_id({"doc_id": h["id"].split("::")[1], "sent_id": h["meta"]["sent_id"]}) == h["id"]


def test_f4_minmax_ids_hits(golden_dir):
    d = json.loads((golden_dir / "f4_minmax.json").read_text())
    for c in d["minmax"]:
        assert fusion.minmax_norm(c["in"]) == c["out"]
    for c in d["normalize_id"]:
        assert fusion.normalize_id(c["in"])[0] == c["out"]
    for c in d["normalize_hit"]:
        assert fusion.normalize_raw_hit(c["in"]) == c["out"]


class _Backend:
    def __init__(self, router=None, sink=None, hits=None, extra=None):
        self.hits, self.router, self.sink, self.extra = hits or [], router, sink, extra

    def run(self, req):
        return {"hits": self.hits, "diagnostics": {"n": len(self.hits)}}


def test_adapter_matches_reference_adapter(golden_dir, monkeypatch):
    data = json.loads((golden_dir / "f3_hybrid_run.json").read_text())
    import mrag_amd.adapter as ad
    for c in data["cases"]:
        monkeypatch.setattr(ad, "import_from_string", lambda path: _Backend)
        agent = DenseRetrievalAgent(router=None, backend_impl="x:y", backend_kwargs={"hits": c["run_out"]["hits"]})
        agent.backend.hits = c["run_out"]["hits"]
        out = agent.retrieve(RetrievalIn(**c["req"]))
        assert isinstance(out, RetrievalOut)
        got = [h.model_dump() if hasattr(h, "model_dump") else h.dict() for h in out.hits]
        assert got == c["adapter_out"]["hits"], c["name"]


def test_adapter_from_settings_filters_kwargs_and_spans():
    class Sink:
        def __init__(self): self.ev = []
        def record(self, e): self.ev.append(e)
    settings = {"modules": {"retrieval": {"type": "mrag_amd.adapter:DenseRetrievalAgent",
                                          "impl": "mrag_amd.backend:DenseRetrievalBackend",
                                          "impl_kwargs": {"index_path": "/nonexistent/docs.jsonl", "alpha_dense": 1.0,
                                                          "dense_pool_k": 50, "bm25_pool_k": 200, "qe_lines": 3},
                                          "kwargs": {"id_keys": ["id"], "score_keys": ["score"]}}}}
    sink = Sink()
    router = Router({"hip": Prov()}, {"embedding_provider": "hip"})
    agent = DenseRetrievalAgent.from_settings(settings, router=router, sink=sink)
    assert isinstance(agent.backend, DenseRetrievalBackend)
    assert agent.backend.alpha_dense == 1.0 and agent.backend.dense_pool_k == 50 and agent.id_keys == ["id"]
    out = agent.retrieve(RetrievalIn(query="q", graph_id="", top_k=5, trace_id="t1"))     # empty corpus: no GPU touched
    assert out.hits == [] and out.diagnostics["resolved_embed_model"] == "fake-embed"
    names = [(e["event"], e["node"]) for e in sink.ev]
    assert ("node_start", "RetrievalAdapter") in names and ("node_end", "Backend/DenseRerank") in names
    assert all("duration_sec" in e for e in sink.ev if e["event"] == "node_end")


def test_resolve_embed_model_paths():
    p = Prov()
    assert resolve_embed_model(Router({"hip": p}, {"embedding_provider": "hip"})) == "fake-embed"
    assert resolve_embed_model(Router({"hip": p}, {"embedding": [{"model": "m"}]})) == "m"
    assert resolve_embed_model(Router({}, {})) == "text-embedding-3-large"
    assert resolve_embed_model(object()) == "text-embedding-3-large"


def test_docs_jsonl_roundtrip_and_cache(tmp_path, golden_dir):
    rows = json.loads((golden_dir / "f3_hybrid_run.json").read_text())["docs_rows"]
    p = tmp_path / "docs.jsonl"
    corpus.write_docs_jsonl(p, rows)
    p.write_text(p.read_text().replace("\n", "\n\n", 3))            # blank lines are skipped
    assert corpus.read_docs_jsonl(p) == rows == rs.read_docs_jsonl(str(p))
    assert corpus.read_docs_jsonl(tmp_path / "missing.jsonl") == []
    cache = corpus.EmbeddingCache(tmp_path / "cache")
    key = cache.key(p, "model-a", 8)
    assert cache.load(key) is None
    m = np.arange(16, dtype=np.float16).reshape(2, 8)
    cache.store(key, m, {"rows": 2})
    assert (np.asarray(cache.load(key)) == m).all()
    assert cache.key(p, "model-b", 8) != key
    n = []
    assert corpus.shared("t-key", lambda: n.append(1) or "obj") == "obj"
    assert corpus.shared("t-key", lambda: n.append(1) or "other") == "obj" and n == [1]
    corpus.drop_shared("t-key")


def test_dto_shapes():
    r = RetrievalIn(query="q", graph_id="g", trace_id="t")
    assert r.top_k == 20
    with pytest.raises(Exception):
        RetrievalIn(query="q")
    h = Hit(id="a", score=1)
    assert h.meta == {} and isinstance(h.score, float)
    assert RetrievalOut().hits == [] and RetrievalOut().diagnostics == {}


def test_f8_segment_context_matches_reference(golden_dir, monkeypatch):
    """segment_context (segmenter.py:10-57) -- rule / embed / passthrough -- against F8, the
    reference's own outputs.  The GPU adjacent-cosine kernel is replaced by the oracle formula here (CPU); the
    kernel itself is checked in tests/test_gpu_bruteforce.py."""
    import mrag_amd.pairwise as pw
    d = json.loads((golden_dir / "f8_segment.json").read_text())
    table = d["table"]

    def fake_adjacent(vectors, eps=1e-9, device=0):
        v = np.asarray(vectors, dtype=np.float64)
        return np.asarray([rs.adjacent_similarity(v[i], v[i + 1]) for i in range(len(v) - 1)])
    monkeypatch.setattr(pw, "cosine_adjacent", fake_adjacent)
    ctx = [(t, s) for t, s in d["ctx"]]
    for c in d["cases"]:
        fn = None if c.get("no_embed_fn") else (lambda s: table[s])
        got = pw.segment_context(ctx, strategy=c["strategy"], embed_fn=fn, sim_threshold=c["sim_threshold"])
        assert [[t, s] for t, s in got] == c["out"], (c["strategy"], c["sim_threshold"])

    class Prov:                                   # BatchedEmbedFn: one provider batch for the whole context
        calls = []
        def embed(self, *, model, texts, require):
            Prov.calls.append(len(texts))
            return {"vectors": [table[t] for t in texts]}
    fn = pw.BatchedEmbedFn(Prov(), model="m")
    got = pw.segment_context(ctx, strategy="embed", embed_fn=fn, sim_threshold=0.65)
    assert [[t, s] for t, s in got] == d["cases"][0]["out"]
    assert len(Prov.calls) == 1 and fn("zero") == table["zero"] and len(Prov.calls) == 1


def test_encode_overlaps_tokenisation_with_the_forward(monkeypatch):
    """the tokeniser thread prepares batch i+1 while the (stubbed) GPU forward of batch i runs;
    results are identical to the serial path, errors surface in the caller."""
    import threading
    import time
    from mrag_amd.encoder import HipSentenceEncoder, EncoderSpec, ARCHS, HashingTokenizer

    enc = object.__new__(HipSentenceEncoder)                 # no GPU: bypass __init__, stub the forward
    enc.spec = EncoderSpec(**ARCHS["tiny"])
    enc.max_length, enc.tokenizer, enc._h = 32, HashingTokenizer(1000), None
    log = []

    def slow_tokenize(texts, _orig=HipSentenceEncoder.tokenize):
        log.append(("tok+", time.perf_counter(), threading.current_thread().name))
        time.sleep(0.03)
        r = _orig(enc, texts)
        log.append(("tok-", time.perf_counter(), threading.current_thread().name))
        return r

    def fake_forward(ids, mask, pool=None, normalize=True):
        log.append(("fwd+", time.perf_counter(), threading.current_thread().name))
        time.sleep(0.05)                                      # the ctypes call releases the GIL like sleep does
        log.append(("fwd-", time.perf_counter(), threading.current_thread().name))
        return np.stack([np.full(64, float(ids[i].sum())) for i in range(ids.shape[0])]).astype(np.float32)
    enc.tokenize, enc.forward = slow_tokenize, fake_forward
    texts = [f"word{i} " * (1 + i % 7) for i in range(40)]
    t0 = time.perf_counter(); a = enc.encode(texts, batch_size=8, overlap=True); t_overlap = time.perf_counter() - t0
    names = {n for k, _, n in log if k.startswith("tok")}
    assert names == {"mrag-tokenizer"}
    # some tokenisation interval lies inside a forward interval
    fwd = [(s[1], e[1]) for s, e in zip([x for x in log if x[0] == "fwd+"], [x for x in log if x[0] == "fwd-"])]
    tok = [(s[1], e[1]) for s, e in zip([x for x in log if x[0] == "tok+"], [x for x in log if x[0] == "tok-"])]
    assert any(fs < ts and te < fe + 0.03 for ts, te in tok for fs, fe in fwd)
    log.clear()
    t0 = time.perf_counter(); b = enc.encode(texts, batch_size=8, overlap=False); t_serial = time.perf_counter() - t0
    assert np.array_equal(a, b) and t_overlap < t_serial - 0.05      # 5 batches: ~4 x 30 ms of tokenisation hidden

    def bad_tokenize(texts):
        raise ValueError("tokenizer failed")
    enc.tokenize = bad_tokenize
    with pytest.raises(ValueError, match="tokenizer failed"):
        enc.encode(texts, batch_size=8)
