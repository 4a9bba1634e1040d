"""Pin the CPU oracle against golden vectors captured from the reference's own functions
(tests/golden/make_golden.py).  CPU only."""
import hashlib
import json
import math
import zlib

import numpy as np
import pytest

from oracle import ref_semantics as rs
from oracle import dense_search as ds


def text_vec(text, dim=8):
    rng = np.random.default_rng(zlib.crc32(text.encode("utf-8")))
    return [float(x) for x in rng.standard_normal(dim)]


# ------------------------------------------------------------------------------ F1
def test_f1_cosine_known_answers(golden_dir):
    z = np.load(golden_dir / "f1_cosine.npz")
    for a, b, (la, lb), want_dr, want_ut in zip(z["a"], z["b"], z["dims"], z["out_dense_reranker"],
                                               z["out_utils_similarity"]):
        a, b = [float(x) for x in a[:la]], [float(x) for x in b[:lb]]
        got_dr, got_ut = rs.cosine(a, b), rs.cosine_util(a, b)
        # bit-exact fp64; the 1e200 row overflows to inf/inf = NaN in the reference too
        assert got_dr == want_dr or (math.isnan(got_dr) and math.isnan(want_dr))
        assert got_ut == want_ut or (math.isnan(got_ut) and math.isnan(want_ut))
    # the edge rows really are in the fixture
    assert (z["dims"][:, 0] == 0).any() and (z["dims"][:, 0] != z["dims"][:, 1]).any()
    assert (z["out_dense_reranker"] == 0.0).sum() >= 6


# ------------------------------------------------------------------------------ F2
class _Prov:
    def __init__(self, spec):
        self.kwargs = {"embed_model": "fake-embed"}
        self.spec, self.calls = spec, []

    def embed(self, *, model, texts, require):
        i = len(self.calls)
        self.calls.append({"model": model, "n": len(texts), "require": dict(require)})
        if i in self.spec["fail_calls"]:
            raise RuntimeError("injected")
        v = [text_vec(t, self.spec["dim"]) for t in texts]
        return v if self.spec["bare_list"] else {"vectors": v}


def test_f2_dense_score(golden_dir):
    cases = json.loads((golden_dir / "f2_dense_score.json").read_text())["cases"]
    assert len(cases) >= 10
    for c in cases:
        if c["provider"] is not None:
            prov = _Prov(c["provider"])
            if c["name"] == "no_provider_named":
                policy = {}
            elif c["name"] == "policy_embedding_model_wins":
                policy = {"embedding_provider": "hip", "embedding": [{"model": "policy-model", "provider": "hip"}]}
            else:
                policy = {"embedding_provider": "hip"}
            providers = {"hip": prov}
            model = rs.resolve_embed_model(policy, providers)
            embed = lambda *, model_hint, texts, require: rs.router_embed(  # noqa: E731
                providers, policy, model_hint=model_hint, texts=texts, require=require)
        else:
            fails, n = set(c["router_fail_calls"]), [0]
            model = rs.resolve_embed_model({}, {})

            def embed(*, model_hint, texts, require, _f=fails, _n=n):
                i = _n[0]; _n[0] += 1
                if i in _f:
                    raise RuntimeError("router-level")
                return {"vectors": [text_vec(t) for t in texts]}
            prov = None
        assert model == c["resolved_model"], c["name"]
        got = rs.dense_score(embed, query=c["query"], candidates=c["candidates"], trace_id="t-f2",
                             max_pool=c["max_pool"], embed_batch=c["embed_batch"], model_hint=model)
        assert got == c["out"], c["name"]           # same keys, bit-exact fp64 scores
        assert list(got.keys()) == list(c["out"].keys()), c["name"]
        if prov is not None:
            assert prov.calls == c["calls"], c["name"]   # 1 query call + chunks of max(8, bs)


# ------------------------------------------------------------------------------ F3
def _tie_groups(hits):
    groups, cur = [], []
    for h in hits:
        if cur and h["score"] != cur[-1]["score"]:
            groups.append(cur); cur = []
        cur.append(h)
    if cur:
        groups.append(cur)
    return groups


def _same_modulo_ties(got, want, truncated):
    """identical scores position by position; inside a run of equal scores the ids may
    be permuted (the reference's tie order is set-iteration order); the LAST tie group
    of a truncated list may hold any members of the untruncated tie."""
    assert [h["score"] for h in got] == [h["score"] for h in want]
    gg, ww = _tie_groups(got), _tie_groups(want)
    for idx, (g, w) in enumerate(zip(gg, ww)):
        gi, wi = sorted(h["id"] for h in g), sorted(h["id"] for h in w)
        if gi != wi:
            assert truncated and idx == len(ww) - 1, (gi, wi)
    return True


def test_f3_fusion_and_adapter(golden_dir):
    data = json.loads((golden_dir / "f3_hybrid_run.json").read_text())
    for c in data["cases"]:
        kw = c["backend_kwargs"]
        top_k = int(c["req"]["top_k"] or kw["default_top_k"])
        fused_all = rs.fuse(c["t_hits_raw"], c["g_hits_raw"], c["dense_scores_raw"], alpha_text=kw["alpha_text"],
                            alpha_graph=kw["alpha_graph"], alpha_dense=kw["alpha_dense"], top_k=10 ** 9)
        fused = fused_all[:top_k]
        want = c["run_out"]["hits"]
        _same_modulo_ties(fused, want, truncated=len(fused_all) > top_k)
        by_id = {h["id"]: h for h in fused_all}
        for h in want:                                # meta incl. score_*_norm is bit-exact
            assert by_id[h["id"]]["meta"] == h["meta"], c["name"]
            assert by_id[h["id"]]["score"] == h["score"]
        # adapter: normalise + stable sort + truncate (req.top_k == 0 -> no truncation)
        got_ad = rs.adapter_retrieve(c["run_out"], c["req"]["top_k"])
        assert got_ad["hits"] == c["adapter_out"]["hits"], c["name"]
        assert got_ad["diagnostics"] == c["adapter_out"]["diagnostics"]
        # the dense channel is keyed by the RAW bm25 id
        assert set(c["dense_scores_raw"]) <= {h["id"] for h in c["t_hits_raw"]}
        # dense scores themselves follow from the table provider
        embed = lambda *, model_hint, texts, require: {"vectors": [text_vec(t) for t in texts]}  # noqa: E731
        ds_again = rs.dense_score(embed, query=c["req"]["query"], candidates=c["t_hits_raw"], trace_id="t-f3",
                                  max_pool=kw["bm25_pool_k"], embed_batch=kw["embed_batch"])
        assert ds_again == c["dense_scores_raw"]
        n_calls = len(c["embed_calls"]) // 2          # run() and the adapter's own backend each embedded once
        assert [x["n"] for x in c["embed_calls"][:n_calls]] == [1, 8, 8, 8, 6]


def test_f3_bm25_raw_ids_follow_docs_rows(golden_dir):
    data = json.loads((golden_dir / "f3_hybrid_run.json").read_text())
    by_key = {(r["title"], r["sent_id"], r["text"]): r for r in data["docs_rows"]}
    for c in data["cases"]:
        for h in c["t_hits_raw"]:
            m = h["meta"]
            row = by_key[(m["doc"], m["sent_id"], m["text"])]
            assert h["id"] == rs.bm25_hit_id(row)


# ------------------------------------------------------------------------------ F4
def test_f4_minmax_ids_hits_mmr(golden_dir):
    d = json.loads((golden_dir / "f4_minmax.json").read_text())
    for c in d["minmax"]:
        assert rs.minmax_norm(c["in"]) == c["out"]
    for c in d["normalize_id"]:
        assert rs.normalize_id(c["in"])[0] == c["out"]
    for c in d["normalize_hit"]:
        assert rs.normalize_hit(c["in"]) == c["out"]
    items = [(i, s, v) for i, s, v in d["mmr_items"]]
    for c in d["mmr"]:
        sel = rs.mmr_diversify(list(items), top_k=c["top_k"], lambda_weight=c["lambda"])
        assert [s[0] for s in sel] == c["selected_ids"]


# ------------------------------------------------------------------------------ F5
def test_f5_bruteforce_c1_shape(golden_dir):
    z = np.load(golden_dir / "f5_bruteforce_c1.npz")
    nq, n, d, k = int(z["nq"]), int(z["n"]), int(z["d"]), int(z["k"])
    c16 = ds.normalize_round(ds.make_gaussian(n, d, int(z["corpus_seed"])))
    q16 = ds.normalize_round(ds.make_gaussian(nq, d, int(z["query_seed"])))
    assert hashlib.sha256(c16.tobytes()).hexdigest() == str(z["corpus_sha256"])
    assert hashlib.sha256(q16.tobytes()).hexdigest() == str(z["query_sha256"])
    val, ids = ds.brute_force_topk(q16, c16, k)
    # the reference divides by the (fp16-rounded, so not exactly 1) norms; the oracle
    # scores the rounded rows by plain dot.  |norm-1| <= ~5e-4, so compare the reference's
    # cosine to dot/(|q||c|) computed in fp64 here, and ids exactly.
    qn = np.linalg.norm(q16.astype(np.float64), axis=1)
    cn = np.linalg.norm(c16.astype(np.float64), axis=1)
    ref_ids, ref_sc = z["ids"], z["scores"]
    cos = val / (qn[:, None] * cn[ids])
    # ids: exact wherever the reference's own gap to the neighbours exceeds the
    # norm-induced wobble (1e-3 is the north-star tolerance)
    strict, bad = ds.gap_aware_id_match(ids, val, ref_ids, ref_sc, tol=1e-3)
    assert strict > 0.5 * nq * k and bad == 0
    assert ds.recall_at_k(ids, ref_ids) >= 0.999
    same = ids == ref_ids
    np.testing.assert_allclose(cos[same], ref_sc[same], rtol=0, atol=1e-12)
    np.testing.assert_allclose(val[same], ref_sc[same], rtol=0, atol=1e-3)


def test_brute_force_blocks_and_ties():
    rng = np.random.default_rng(0)
    c = rng.integers(-2, 3, size=(700, 16)).astype(np.float16)   # many exact ties
    q = rng.integers(-2, 3, size=(9, 16)).astype(np.float16)
    v1, i1 = ds.brute_force_topk(q, c, 12)
    v2, i2 = ds.brute_force_topk(q, c, 12, block=97)
    assert (i1 == i2).all() and (v1 == v2).all()
    s = q.astype(np.float64) @ c.astype(np.float64).T
    for r in range(9):
        order = sorted(range(700), key=lambda j: (-s[r, j], j))[:12]
        assert list(i1[r]) == order
    # k > n pads with -inf / -1
    v3, i3 = ds.brute_force_topk(q, c[:5], 8)
    assert (i3[:, 5:] == -1).all() and np.isneginf(v3[:, 5:]).all()


def test_l2_normalize_zero_rows_and_ivf_exhaustive_probe():
    x = ds.make_gaussian(300, 32, 3); x[7] = 0
    n = ds.l2_normalize(x)
    assert (n[7] == 0).all()
    np.testing.assert_allclose(np.linalg.norm(n[np.arange(300) != 7], axis=1), 1.0, atol=1e-6)
    c16 = n.astype(np.float16); q16 = ds.normalize_round(ds.make_gaussian(11, 32, 4))
    cen = ds.kmeans_spherical(c16, 8, 3, seed=5)
    a = ds.ivf_assign(c16, cen)
    v_all, i_all = ds.ivf_search(q16, c16, cen, a, nprobe=8, k=5)     # probing every list == brute force
    v_bf, i_bf = ds.brute_force_topk(q16, c16, 5)
    assert (i_all == i_bf).all() and np.allclose(v_all, v_bf, atol=1e-15)
    v_p, i_p = ds.ivf_search(q16, c16, cen, a, nprobe=2, k=5)
    assert ds.recall_at_k(i_p, i_bf) > 0.3


def test_merge_topk_matches_unsharded():
    rng = np.random.default_rng(1)
    c = ds.normalize_round(rng.standard_normal((1000, 24)).astype(np.float32))
    q = ds.normalize_round(rng.standard_normal((13, 24)).astype(np.float32))
    v, i = ds.brute_force_topk(q, c, 7)
    parts_v, parts_i = [], []
    for lo in range(0, 1000, 250):
        pv, pi = ds.brute_force_topk(q, c[lo:lo + 250], 7)
        parts_v.append(pv); parts_i.append(pi + lo)
    mv, mi = ds.merge_topk(parts_v, parts_i, 7)
    assert (mi == i).all() and (mv == v).all()
