"""GPU parity of the HIP sentence encoder (K6) against the CPU oracle (itself pinned to HF BertModel in
tests/test_encoder_oracle.py), within the north-star 1e-3; plus the provider / backend end to end."""
import json
import zlib

import numpy as np
import pytest

from oracle import encoder as oe
from oracle import dense_search as ds
from oracle import ref_semantics as rs

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _batch(spec, B, S, seed, lens=None):
    rng = np.random.default_rng(seed)
    ids = rng.integers(3, spec["vocab_size"], size=(B, S)).astype(np.int32)
    mask = np.ones((B, S), dtype=np.int32)
    lens = lens or [int(x) for x in rng.integers(1, S + 1, size=B)]
    for b, L in enumerate(lens):
        mask[b, L:] = 0
        ids[b, L:] = 0
    return ids, mask


@pytest.mark.parametrize("arch,B,S,pool", [("tiny", 7, 16, "mean"), ("tiny", 3, 100, "cls"), ("small", 9, 80, "cls"),
                                            ("small", 4, 256, "mean")])
def test_small_archs_match_oracle(arch, B, S, pool):
    from mrag_amd.encoder import HipSentenceEncoder, EncoderSpec, ARCHS, seeded_weights
    spec = oe.SPECS[arch]
    w = oe.seeded_weights(spec, 21)
    enc = HipSentenceEncoder(EncoderSpec(**ARCHS[arch]), w)
    ids, mask = _batch(spec, B, S, 3)
    got = enc.forward(ids, mask, pool=pool)
    want = oe.forward(spec, w, ids, mask, pool=pool)
    assert np.isfinite(got).all()
    np.testing.assert_allclose(got, want, rtol=0, atol=TOL)
    raw = enc.forward(ids, mask, pool=pool, normalize=False)
    np.testing.assert_allclose(raw, oe.forward(spec, w, ids, mask, pool=pool, normalize=False), rtol=0, atol=5e-3)


def test_minilm_l6_shape_matches_oracle():
    """BASELINE config 1 encoder shape (6 layers, 384 hidden, 12 heads, 1536 FFN), seeded weights."""
    from mrag_amd.encoder import HipSentenceEncoder
    spec = dict(oe.SPECS["minilm-l6"], vocab_size=4000)          # full 30522-row table not needed for arithmetic parity
    from mrag_amd.encoder import EncoderSpec
    w = oe.seeded_weights(spec, 7)
    enc = HipSentenceEncoder(EncoderSpec(**dict(spec, max_length=128)), w)
    ids, mask = _batch(spec, 16, 64, 9)
    got = enc.forward(ids, mask, pool="mean")
    want = oe.forward(spec, w, ids, mask, pool="mean")
    np.testing.assert_allclose(got, want, rtol=0, atol=TOL)
    assert np.abs(got - want).max() < 5e-4


def test_bge_base_shape_matches_oracle_cls():
    """BASELINE config 3 encoder shape (12 layers, 768 hidden, 12 heads, 3072 FFN), CLS pooling."""
    from mrag_amd.encoder import HipSentenceEncoder, EncoderSpec
    spec = dict(oe.SPECS["bge-base"], vocab_size=3000, max_position=128)
    w = oe.seeded_weights(spec, 8)
    enc = HipSentenceEncoder(EncoderSpec(**dict(spec, max_length=128)), w)
    ids, mask = _batch(spec, 6, 48, 10)
    got = enc.forward(ids, mask, pool="cls")
    want = oe.forward(spec, w, ids, mask, pool="cls")
    np.testing.assert_allclose(got, want, rtol=0, atol=TOL)


@pytest.mark.parametrize("arch,B,S,pool", [("minilm-l6", 1, 20, "mean"), ("minilm-l6", 2, 16, "mean"), ("minilm-l6", 3, 20, "mean"),
                                            ("bge-base", 1, 24, "cls"), ("bge-base", 4, 16, "cls")])
def test_single_question_batches_match_oracle(arch, B, S, pool):
    """<= 64 tokens per call -- the reference embeds one question at a time (retrieval_backend.py:227): the skinny,
    K-split GEMM path (csrc/encoder.hip enc_gemm_skinny_kernel), real layer widths."""
    from mrag_amd.encoder import HipSentenceEncoder, EncoderSpec
    spec = dict(oe.SPECS[arch], vocab_size=3000, max_position=64, layers=3)
    w = oe.seeded_weights(spec, 12)
    enc = HipSentenceEncoder(EncoderSpec(**dict(spec, max_length=64)), w)
    ids, mask = _batch(spec, B, S, 13)
    got = enc.forward(ids, mask, pool=pool)
    want = oe.forward(spec, w, ids, mask, pool=pool)
    np.testing.assert_allclose(got, want, rtol=0, atol=TOL)
    # the same rows inside a larger batch (tiled GEMM path) agree to fp16-activation noise
    ids2, mask2 = _batch(spec, 40, S, 14)
    ids2[:B], mask2[:B] = ids, mask
    big = enc.forward(ids2, mask2, pool=pool)
    np.testing.assert_allclose(big[:B], got, rtol=0, atol=TOL)


@pytest.mark.parametrize("arch,B,S,pool,layers", [("minilm-l6", 512, 32, "mean", 6), ("bge-base", 128, 96, "cls", 3)])
def test_fused_layernorm_flow_matches_oracle(arch, B, S, pool, layers):
    """Batches large enough for the 256 x 256 GEMMs take the fused-LayerNorm flow (csrc/encoder.hip LnArgs: gamma folded into the
    consuming weights, per-token statistics from the producing GEMM's epilogue, residuals normalised on the fly): against the
    oracle on a subsample of the rows, padded / ragged masks included."""
    from mrag_amd.encoder import HipSentenceEncoder, EncoderSpec
    spec = dict(oe.SPECS[arch], vocab_size=3000, max_position=128, layers=layers)
    w = oe.seeded_weights(spec, 31)
    enc = HipSentenceEncoder(EncoderSpec(**dict(spec, max_length=128)), w)
    ids, mask = _batch(spec, B, S, 17)
    got = enc.forward(ids, mask, pool=pool)
    sub = np.arange(0, B, B // 16)
    want = oe.forward(spec, w, ids[sub], mask[sub], pool=pool)
    np.testing.assert_allclose(got[sub], want, rtol=0, atol=TOL)
    assert np.isfinite(got).all()
    if arch == "minilm-l6":                      # the same flow with bf16 weights / activations (coarser rounding: 8e-3, as the plain bf16 test)
        enc16 = HipSentenceEncoder(EncoderSpec(**dict(spec, max_length=128)), w, dtype="bf16")
        np.testing.assert_allclose(enc16.forward(ids, mask, pool=pool)[sub], want, rtol=0, atol=8e-3)


def test_fused_layernorm_flow_on_offset_and_outlier_activations(monkeypatch):
    """the fused flow computes per-token variance as E[x^2] - mean^2 in fp32 from the rounded GEMM outputs and
    evaluates ``rstd (y W'^T - mu s) + c``: both cancel when a token's activations share a large common offset or carry
    outlier dimensions (what trained BERT-family models do).  Same ids through the fused flow, the plain flow
    (MRAG_ENC_NO_FUSED_LN=1) and the oracle, with biases that put +12 on every pre-LayerNorm dimension and +/- 60 on three of
    them: stored rows / query embeddings must agree whichever flow produced them."""
    from mrag_amd.encoder import HipSentenceEncoder, EncoderSpec
    spec = dict(oe.SPECS["minilm-l6"], vocab_size=3000, max_position=128, layers=2)
    w = oe.seeded_weights(spec, 77)
    for i in range(spec["layers"]):
        for name in (f"encoder.layer.{i}.attention.output.dense.bias", f"encoder.layer.{i}.output.dense.bias"):
            b = w[name].copy()
            b += 12.0
            b[[3, 77, 200]] += np.array([60.0, -60.0, 45.0], dtype=np.float32)
            w[name] = b
    enc = HipSentenceEncoder(EncoderSpec(**dict(spec, max_length=128)), w)
    ids, mask = _batch(spec, 512, 32, 23)
    monkeypatch.delenv("MRAG_ENC_NO_FUSED_LN", raising=False)
    fused = enc.forward(ids, mask, pool="mean")
    monkeypatch.setenv("MRAG_ENC_NO_FUSED_LN", "1")
    plain = enc.forward(ids, mask, pool="mean")
    monkeypatch.delenv("MRAG_ENC_NO_FUSED_LN")
    sub = np.arange(0, 512, 32)
    want = oe.forward(spec, w, ids[sub], mask[sub], pool="mean")
    d_fp, d_fo, d_po = np.abs(fused - plain).max(), np.abs(fused[sub] - want).max(), np.abs(plain[sub] - want).max()
    print(f"offset/outlier activations: |fused - plain| {d_fp:.2e}, |fused - oracle| {d_fo:.2e}, |plain - oracle| {d_po:.2e}")
    assert np.isfinite(fused).all() and not np.array_equal(fused, plain)      # (the two flows did run)
    assert d_po <= 2e-3 and d_fo <= 2e-3 and d_fp <= 2e-3


@pytest.mark.parametrize("case", ["minilm2", "bge1"])
def test_f6_hf_golden(golden_dir, case):
    """F6: the committed HF BertModel outputs (tests/golden/make_golden_encoder.py) -- MiniLM-L6 / bge-base layer
    geometry with the full 30 522-row vocabulary and positions up to 272 -- vs the HIP encoder, atol 1e-3."""
    from mrag_amd.encoder import HipSentenceEncoder, EncoderSpec, seeded_weights
    z = np.load(golden_dir / "f6_encoder.npz")
    base = {"minilm2": "minilm-l6", "bge1": "bge-base"}[case]
    spec = EncoderSpec(**dict(oe.SPECS[base], layers=int(z[f"{case}.layers"])))
    enc = HipSentenceEncoder(spec, seeded_weights(spec, int(z[f"{case}.seed"])))
    got = enc.forward(z[f"{case}.ids"], z[f"{case}.mask"], pool=str(z[f"{case}.pool"]))
    np.testing.assert_allclose(got, z[f"{case}.emb"], rtol=0, atol=TOL)


@pytest.mark.parametrize("case", ["minilm6", "bge12"])
def test_f6b_hf_golden_full_depth(golden_dir, case):
    """F6b (round 3): HF BertModel outputs at FULL depth -- all 6 layers of the MiniLM-L6 geometry, all 12 of bge-base, full
    vocabulary -- vs the HIP encoder, atol 1e-3 (tests/golden/make_golden_encoder.py writes both fixture files)."""
    from mrag_amd.encoder import HipSentenceEncoder, EncoderSpec, seeded_weights
    z = np.load(golden_dir / "f6b_encoder_full.npz")
    base = {"minilm6": "minilm-l6", "bge12": "bge-base"}[case]
    assert int(z[f"{case}.layers"]) == oe.SPECS[base]["layers"]
    spec = EncoderSpec(**oe.SPECS[base])
    enc = HipSentenceEncoder(spec, seeded_weights(spec, int(z[f"{case}.seed"])))
    got = enc.forward(z[f"{case}.ids"], z[f"{case}.mask"], pool=str(z[f"{case}.pool"]))
    np.testing.assert_allclose(got, z[f"{case}.emb"], rtol=0, atol=TOL)


def test_bf16_compute_and_errors():
    from mrag_amd.encoder import HipSentenceEncoder, EncoderSpec, ARCHS
    from mrag_amd._native import MragError
    spec = oe.SPECS["tiny"]
    w = oe.seeded_weights(spec, 2)
    enc = HipSentenceEncoder(EncoderSpec(**ARCHS["tiny"]), w, dtype="bf16")
    ids, mask = _batch(spec, 4, 32, 1)
    np.testing.assert_allclose(enc.forward(ids, mask), oe.forward(spec, w, ids, mask), rtol=0, atol=8e-3)
    with pytest.raises(MragError):
        enc.forward(np.zeros((1, 129), np.int32), np.ones((1, 129), np.int32))     # beyond max_position
    w2 = dict(w); w2.pop("encoder.layer.1.output.dense.bias")
    with pytest.raises(ValueError):
        HipSentenceEncoder(EncoderSpec(**ARCHS["tiny"]), w2)


def test_provider_and_backend_end_to_end(tmp_path):
    """B1 + B2 + B3 on a HotpotQA-shaped synthetic docs.jsonl: provider -> router -> DenseRetrievalBackend
    -> DenseRetrievalAgent, checked against the oracle (oracle encoder + brute force + reference fusion)."""
    from mrag_amd import corpus
    from mrag_amd.adapter import DenseRetrievalAgent
    from mrag_amd.backend import HipDenseReranker
    from mrag_amd.dto import RetrievalIn
    from mrag_amd.provider import HipEmbeddingProvider

    rng = np.random.default_rng(4)
    vocab = ["alpha", "beta", "gamma", "delta", "river", "city", "born", "film", "band", "album", "war", "king"]
    rows = []
    for t in range(60):
        for sid in range(int(rng.integers(2, 6))):
            rows.append({"doc_id": f"Title {t}#{sid}", "title": f"Title {t}", "sent_id": sid,
                         "text": " ".join(rng.choice(vocab, size=int(rng.integers(5, 12))))})
    docs = tmp_path / "docs.jsonl"
    corpus.write_docs_jsonl(docs, rows)

    prov = HipEmbeddingProvider(arch="tiny", seed=5, embed_model="tiny-seed5")
    assert prov.kwargs["embed_model"] == "tiny-seed5"

    class Router:                       # LLMRouter.embed restated (oracle.ref_semantics.router_embed)
        providers, policy = {"hip": prov}, {"embedding_provider": "hip"}
        def embed(self, *, model_hint, texts, require=None):
            return rs.router_embed(self.providers, self.policy, model_hint=model_hint, texts=texts, require=require)
    router = Router()

    # provider vectors == oracle encoder on the same token ids
    texts = [r["text"] for r in rows[:9]]
    ids, mask = prov.encoder.tokenize(texts)
    spec = oe.SPECS["tiny"]
    want = oe.forward(spec, oe.seeded_weights(spec, 5), ids, mask, pool="mean")
    got = np.asarray(prov.embed(model="x", texts=texts, require={})["vectors"])
    np.testing.assert_allclose(got, want, rtol=0, atol=TOL)

    settings = {"modules": {"retrieval": {"impl": "mrag_amd.backend:DenseRetrievalBackend",
                                          "impl_kwargs": {"index_path": str(docs), "alpha_dense": 0.4,
                                                          "cache_dir": str(tmp_path / "cache"), "dense_pool_k": 20}}}}
    agent = DenseRetrievalAgent.from_settings(settings, router=router)
    query = "gamma river born king"
    out = agent.retrieve(RetrievalIn(query=query, graph_id="", top_k=5, trace_id="t"))
    assert len(out.hits) == 5 and out.diagnostics["dense_error"] is None and out.diagnostics["dense_scored"] >= 5
    # oracle: encode everything with the oracle encoder, brute force, fuse with the reference rule
    all_ids, all_mask = prov.encoder.tokenize([r["text"] for r in rows])
    E = oe.forward(spec, oe.seeded_weights(spec, 5), all_ids, all_mask, pool="mean")
    qi, qm = prov.encoder.tokenize([query])
    qe = oe.forward(spec, oe.seeded_weights(spec, 5), qi, qm, pool="mean")
    rv, ri = ds.brute_force_topk(ds.normalize_round(qe), ds.normalize_round(E), 20)
    top_rows = [rows[int(i)] for i in ri[0][:5]]
    got_texts = [h.meta["text"] for h in out.hits]
    assert len(set(got_texts) & {r["text"] for r in top_rows}) >= 4          # fp16 encoder noise may swap one near-tie
    assert out.hits[0].meta["score_dense_norm"] == 1.0 and out.hits[0].score == pytest.approx(0.4)
    assert all(h.id.startswith("sent::Title ") for h in out.hits)
    # second agent instance reuses the process-wide index (no re-embedding) and the disk cache exists
    assert any((tmp_path / "cache").glob("*.npy"))
    agent2 = DenseRetrievalAgent.from_settings(settings, router=router)
    assert agent2.backend._get_state("tiny-seed5", "t") is agent.backend._get_state("tiny-seed5", "t")

    # a2 drop-in: GPU cosine re-ranker == reference formula on the provider's vectors
    cands = [{"id": f"sent::{r['doc_id']}::{r['sent_id'] or ''}", "score": 1.0, "meta": {"text": r["text"]}} for r in rows[:30]]
    rr = HipDenseReranker(router, max_pool=25, embed_batch=8)
    scores = rr.score(query=query, candidates=cands, trace_id="t")
    ref = rs.dense_score(router.embed, query=query, candidates=cands, trace_id="t", max_pool=25, embed_batch=8,
                         model_hint="tiny-seed5")
    assert list(scores) == list(ref)
    np.testing.assert_allclose(list(scores.values()), list(ref.values()), rtol=0, atol=1e-12)


def test_f9_provider_from_model_directory(golden_dir):
    """F9: ``HipEmbeddingProvider(model_path=<HF dir>)`` -- config.json, safetensors, vocab.txt WordPiece,
    1_Pooling, max_seq_length truncation, Normalize -- text in, vectors out, against the embeddings the container's
    ``transformers`` produced from the same directory (tests/golden/make_golden_hfdir.py).  Slot:
    app/core/providers/openai_provider.py:96-134."""
    import json
    from mrag_amd.provider import HipEmbeddingProvider
    g = json.loads((golden_dir / "f9_hf.json").read_text())
    prov = HipEmbeddingProvider(model_path=str(golden_dir / "f9_hf_dir"), batch_size=16)
    assert prov.kwargs["embed_model"] == "f9_hf_dir" and prov.dim == 64
    enc = prov.encoder
    assert enc.max_length == g["max_seq_length"] and enc.spec.pool == "mean" and enc.normalize_default is True
    out = prov.embed(model="x", texts=g["texts"], require={})
    got = np.asarray(out["vectors"])
    assert got.shape == (len(g["texts"]), 64) and out["dim"] == 64
    np.testing.assert_allclose(got, np.asarray(g["mean_normalized"]), rtol=0, atol=TOL)
    # the same ids through forward(), CLS pooling
    ids, mask = enc.tokenize(g["texts"])
    for row, m, want in zip(ids, mask, g["ids"]):
        assert row[m.astype(bool)].tolist() == want
    np.testing.assert_allclose(enc.forward(ids, mask, pool="cls"), np.asarray(g["cls_normalized"]), rtol=0, atol=TOL)


def _synthetic_docs(path, n_titles, seed):
    from mrag_amd import corpus
    rng = np.random.default_rng(seed)
    vocab = [f"w{i}" for i in range(400)] + ["alpha", "beta", "gamma", "delta", "river", "city", "born", "film", "band", "album"]
    rows = []
    for t in range(n_titles):
        for sid in range(int(rng.integers(1, 6))):
            rows.append({"doc_id": f"Title {t}#{sid}", "title": f"Title {t}", "sent_id": sid,
                         "text": " ".join(rng.choice(vocab, size=int(rng.integers(3, 40))))})
    corpus.write_docs_jsonl(path, rows)
    return rows


def test_bulk_ingest_matches_the_router_path(tmp_path):
    """the corpus index built from docs.jsonl through the provider's bulk device form (large
    length-sorted batches, device -> device into DenseIndex.add) against the index built through ``router.embed``
    256 texts at a time (the reference-shaped path, kept as the fallback).  Batch composition selects the padded
    sequence length and the GEMM flow, so the two are equal up to the encoder's fp16 noise, not bit for bit: stored
    rows within 1e-3 (the encoder's own tolerance), the same top-10 outside near-ties."""
    from mrag_amd.backend import DenseRetrievalBackend
    from mrag_amd.provider import HipEmbeddingProvider
    docs = tmp_path / "docs.jsonl"
    rows = _synthetic_docs(docs, 1700, 3)
    assert len(rows) >= 5000
    prov = HipEmbeddingProvider(arch="small", seed=7, embed_model="small-seed7")

    class Router:
        providers, policy = {"hip": prov}, {"embedding_provider": "hip"}
        calls = 0
        def embed(self, *, model_hint, texts, require=None):
            Router.calls += 1
            return rs.router_embed(self.providers, self.policy, model_hint=model_hint, texts=texts, require=require)
    fast = DenseRetrievalBackend(Router(), index_path=str(docs), bulk_ingest=True, bulk_batch=2048)
    st_fast = fast._build_state("small-seed7", "t")
    calls_fast = Router.calls
    slow = DenseRetrievalBackend(Router(), index_path=str(docs), bulk_ingest=False)
    st_slow = slow._build_state("small-seed7", "t")
    assert st_fast["ingest"] == "bulk" and st_slow["ingest"] == "router"
    assert calls_fast == 1 and Router.calls - calls_fast >= 1 + len(rows) // 256     # bulk: only the 1-text dim probe
    assert fast.last_build["ingest"] == "bulk" and fast.last_build["rows"] == len(rows)
    a, b = st_fast["index"].rows(), st_slow["index"].rows()
    assert a.shape == b.shape == (len(rows), 128)
    assert float(np.abs(a - b).max()) <= TOL
    np.testing.assert_allclose(np.linalg.norm(a, axis=1), 1.0, atol=2e-3)
    q = prov.embed_array(["gamma river born w17 w3", "film band album w250"])
    sa, ia = st_fast["index"].search(q, 10)
    sb, ib = st_slow["index"].search(q, 10)
    np.testing.assert_allclose(sa, sb, rtol=0, atol=2e-3)
    assert ds.gap_aware_id_match(ia, sa, ib, sb.astype(np.float64), tol=4e-3)[1] == 0
    # a provider whose bulk form fails falls back to the router path and still builds the index
    class Broken(HipEmbeddingProvider):
        def embed_device(self, texts, batch_size=None):
            raise RuntimeError("no bulk today")
    bprov = Broken(arch="small", seed=7, embed_model="small-seed7")
    class Router2(Router):
        providers = {"hip": bprov}
    st_fb = DenseRetrievalBackend(Router2(), index_path=str(docs))._build_state("small-seed7", "t")
    assert st_fb["ingest"] == "router" and len(st_fb["index"]) == len(rows)
    for st in (st_fast, st_slow, st_fb):
        st["index"].close()
