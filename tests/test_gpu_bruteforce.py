"""GPU parity: fused MFMA cosine top-k (through the C ABI) vs the CPU oracle on identical
fp16/bf16 input bits.  Integer/index work (ids) is exact outside fp32-accumulation
near-ties; scores within 1e-5 of the fp64 oracle (north-star tolerance: 1e-3)."""
import hashlib

import numpy as np
import pytest

from oracle import dense_search as ds

pytestmark = pytest.mark.gpu

TOL = 1e-5


def _index(c16, **kw):
    from mrag_amd.index import DenseIndex
    ix = DenseIndex(c16.shape[1], **kw)
    ix.add(c16, normalize=False)
    return ix


def _check(ix, q16, c16, k, tol=TOL, id_base=0):
    sc, ids = ix.search(q16, k, normalize=False)
    rv, ri = ds.brute_force_topk(q16, c16, k)
    ri = np.where(ri >= 0, ri + id_base, ri)
    assert sc.shape == (q16.shape[0], k) and ids.dtype == np.int64
    valid = ri >= 0
    assert ((ids >= 0) == valid).all()
    assert np.isneginf(sc[~valid]).all()
    np.testing.assert_allclose(sc[valid], rv[valid], rtol=0, atol=tol)
    assert (np.diff(sc, axis=1)[valid[:, 1:]] <= 0).all()            # descending
    strict, bad = ds.gap_aware_id_match(ids, sc, ri, rv, tol=tol)
    assert bad == 0, (strict, bad)
    assert ds.recall_at_k(ids, ri) >= 0.999
    return sc, ids, rv, ri


def test_c1_shape_against_reference_golden(golden_dir):
    """BASELINE config 1 shape (100 x 5000 x 384, k=10) against F5 = the reference's own
    _cosine over the same rows."""
    z = np.load(golden_dir / "f5_bruteforce_c1.npz")
    c16 = ds.normalize_round(ds.make_gaussian(int(z["n"]), int(z["d"]), int(z["corpus_seed"])))
    q16 = ds.normalize_round(ds.make_gaussian(int(z["nq"]), int(z["d"]), int(z["query_seed"])))
    assert hashlib.sha256(c16.tobytes()).hexdigest() == str(z["corpus_sha256"])
    ix = _index(c16)
    sc, ids, _, _ = _check(ix, q16, c16, int(z["k"]))
    strict, bad = ds.gap_aware_id_match(ids, sc, z["ids"], z["scores"], tol=1e-3)
    assert bad == 0 and strict > 500
    assert ds.recall_at_k(ids, z["ids"]) >= 0.999
    same = ids == z["ids"]
    np.testing.assert_allclose(sc[same], z["scores"][same], rtol=0, atol=1e-3)   # cosine within 1e-3


@pytest.mark.parametrize("nq,n,d,k", [(3, 5, 8, 10), (1, 1, 64, 1), (7, 255, 40, 4), (64, 257, 96, 10),
                                      (300, 20000, 768, 10), (33, 4097, 384, 16), (20, 3000, 128, 17),
                                      (10, 2500, 256, 64)])
def test_shapes_f16(nq, n, d, k):
    c16 = ds.normalize_round(ds.make_gaussian(n, d, 1234))
    q16 = ds.normalize_round(ds.make_gaussian(nq, d, 5678))
    _check(_index(c16), q16, c16, k)


@pytest.mark.parametrize("nq,n", [(600, 20000), (2100, 70000), (257, 131072 + 5)])
def test_many_query_tiles_xcd_map(nq, n):
    d, k = 128, 10
    c16 = ds.normalize_round(ds.make_gaussian(n, d, 11))
    q16 = ds.normalize_round(ds.make_gaussian(nq, d, 12))
    _check(_index(c16), q16, c16, k)


def test_bf16_storage():
    import torch
    n, nq, d, k = 9000, 50, 256, 10
    c = torch.from_numpy(ds.l2_normalize(ds.make_gaussian(n, d, 1))).to(torch.bfloat16)
    q = torch.from_numpy(ds.l2_normalize(ds.make_gaussian(nq, d, 2))).to(torch.bfloat16)
    from mrag_amd.index import DenseIndex
    ix = DenseIndex(d, dtype="bf16")
    ix.add(c, normalize=False)
    sc, ids = ix.search(q, k, normalize=False)
    rv, ri = ds.brute_force_topk(q.to(torch.float64).numpy(), c.to(torch.float64).numpy(), k)
    np.testing.assert_allclose(sc, rv, rtol=0, atol=TOL)
    strict, bad = ds.gap_aware_id_match(ids, sc, ri, rv, tol=TOL)
    assert bad == 0


def test_exact_ties_break_by_row():
    """small-integer rows: every product and partial sum is exact in fp32, so ids must match the
    (score desc, row asc) order exactly, ties included."""
    rng = np.random.default_rng(3)
    c = rng.integers(-2, 3, size=(5000, 64)).astype(np.float16)
    q = rng.integers(-2, 3, size=(40, 64)).astype(np.float16)
    from mrag_amd.index import DenseIndex
    ix = DenseIndex(64, metric="ip")
    ix.add(c, normalize=False)
    for k in (1, 10, 16, 33):
        sc, ids = ix.search(q, k, normalize=False)
        rv, ri = ds.brute_force_topk(q, c, k)
        assert (ids == ri).all()
        assert (sc == rv.astype(np.float32)).all()


@pytest.mark.parametrize("nq,n,d,k", [(300, 40000, 128, 20), (64, 70000, 64, 64), (100, 30000, 96, 33),
                                      (700, 9000, 64, 30)])
def test_large_k_batch_compaction_thresholds(nq, n, d, k):
    """16 < k <= 64 in the batch kernel: no row-count certificate, thresholds come from the
    scheduled radix-select compactions (after tiles 0, 1, 3, 7, ... of a split)."""
    c16 = ds.normalize_round(ds.make_gaussian(n, d, 21))
    q16 = ds.normalize_round(ds.make_gaussian(nq, d, 22))
    _check(_index(c16), q16, c16, k)


def test_large_k_ties_and_identical_rows():
    rng = np.random.default_rng(4)
    c = rng.integers(-1, 2, size=(20000, 64)).astype(np.float16)   # few distinct scores: the radix select is decided by row bits
    q = rng.integers(-1, 2, size=(50, 64)).astype(np.float16)
    from mrag_amd.index import DenseIndex
    ix = DenseIndex(64, metric="ip")
    ix.add(c, normalize=False)
    for k in (17, 40, 64):
        sc, ids = ix.search(q, k, normalize=False)
        rv, ri = ds.brute_force_topk(q, c, k)
        assert (ids == ri).all()
        assert (sc == rv.astype(np.float32)).all()
    same = np.tile(ds.normalize_round(ds.make_gaussian(1, 64, 5)), (5000, 1))
    sc, ids = _index(same).search(ds.normalize_round(ds.make_gaussian(20, 64, 6)), 40, normalize=False)
    assert (ids == np.arange(40)[None, :]).all()


def test_all_rows_identical():
    c = np.tile(ds.normalize_round(ds.make_gaussian(1, 64, 5)), (3000, 1))
    q = ds.normalize_round(ds.make_gaussian(5, 64, 6))
    sc, ids = _index(c).search(q, 10, normalize=False)
    assert (ids == np.arange(10)[None, :]).all()


def test_adversarial_increasing_scores_replay_path():
    """every corpus tile beats the previous one for every query: candidate lists overflow and
    the kernel replays tiles in sub-rounds with compaction -- results must still be exact."""
    _adversarial(10)


def test_adversarial_increasing_scores_large_k():
    _adversarial(48)


def _adversarial(k):
    d, n, nq = 64, 6000, 300
    u = ds.l2_normalize(ds.make_gaussian(1, d, 9))[0]
    a = np.linspace(0.05, 1.0, n, dtype=np.float32)
    c = (a[:, None] * u[None, :]).astype(np.float16)
    q = ds.normalize_round(u[None, :] + 0.01 * ds.make_gaussian(nq, d, 10))
    from mrag_amd.index import DenseIndex
    ix = DenseIndex(d, metric="ip")
    ix.add(c, normalize=False)
    sc, ids = ix.search(q, k, normalize=False)
    rv, ri = ds.brute_force_topk(q, c, k)
    np.testing.assert_allclose(sc, rv, rtol=0, atol=TOL)
    strict, bad = ds.gap_aware_id_match(ids, sc, ri, rv, tol=TOL)
    assert bad == 0
    assert ds.recall_at_k(ids, ri) >= 0.999


def test_normalize_on_device_matches_oracle_rounding():
    """K1: fp64 norm, one rounding to fp32, one to fp16 -- same bits as the oracle's
    normalize_round (allowing 1 fp16 ulp on a vanishing fraction for the fp64 sum order)."""
    x = ds.make_gaussian(3000, 384, 21)
    x[17] = 0
    from mrag_amd.index import DenseIndex
    ix = DenseIndex(384)
    ix.add(x)                       # normalise on the GPU
    got = ix.rows().astype(np.float16)
    want = ds.normalize_round(x)
    diff = got.view(np.uint16).astype(np.int32) - want.view(np.uint16).astype(np.int32)
    assert (np.abs(diff) <= 1).all() and (diff != 0).mean() < 1e-5
    assert (got[17] == 0).all()
    q = ds.make_gaussian(20, 384, 22)
    sc, ids = ix.search(q, 10)      # normalise queries on the GPU too
    rv, ri = ds.brute_force_topk(ds.normalize_round(q), want, 10)
    np.testing.assert_allclose(sc, rv, rtol=0, atol=1e-4)
    assert ds.recall_at_k(ids, ri) >= 0.99
    sc0, _ = ix.search(np.zeros((1, 384), np.float32), 5)   # zero query -> cosine 0.0, never NaN
    assert (sc0 == 0).all()


def test_incremental_add_id_base_and_device_tensors():
    import torch
    d, k = 96, 10
    c16 = ds.normalize_round(ds.make_gaussian(7001, d, 31))
    q16 = ds.normalize_round(ds.make_gaussian(129, d, 32))
    from mrag_amd.index import DenseIndex
    ix = DenseIndex(d)
    for lo in range(0, 7001, 1500):                       # grows + re-allocates
        ix.add(torch.from_numpy(c16[lo:lo + 1500]).cuda(), normalize=False)
    assert len(ix) == 7001
    ix.set_id_base(1_000_000_000_000)
    _check(ix, q16, c16, k, id_base=1_000_000_000_000)
    sc_d, ids_d = ix.search(torch.from_numpy(q16).cuda(), k, normalize=False)
    torch.cuda.synchronize()
    sc_h, ids_h = ix.search(q16, k, normalize=False)
    assert (ids_d.cpu().numpy() == ids_h).all() and (sc_d.cpu().numpy() == sc_h).all()
    g_ms, t_ms = ix.last_timing_ms()
    assert 0 < g_ms <= t_ms


def test_errors_are_loud():
    from mrag_amd.index import DenseIndex
    from mrag_amd._native import MragError
    ix = DenseIndex(32)
    with pytest.raises(MragError):
        ix.search(np.zeros((1, 32), np.float32), 257)     # k above mrag_index_max_k (256)
    with pytest.raises(ValueError):
        ix.add(np.zeros((3, 31), np.float32))
    sc, ids = ix.search(np.ones((2, 32), np.float32), 3)   # empty index
    assert (ids == -1).all() and np.isneginf(sc).all()


def test_cosine_f64_matches_reference_formula(golden_dir):
    from mrag_amd.index import cosine_f64
    from oracle.ref_semantics import cosine
    rng = np.random.default_rng(5)
    q = rng.standard_normal(768)
    c = rng.standard_normal((200, 768))
    c[3] = 0
    got = cosine_f64(q, c)
    want = np.asarray([cosine(list(q), list(v)) for v in c])
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-14)
    assert got[3] == 0.0
    assert (cosine_f64(np.zeros(768), c) == 0).all()


def test_pairwise_next_rows_semantic_edges_and_mmr(golden_dir):
    """SURVEY 8f rows 2 and 4 on the all-pairs kernel: edge_builder's pair rule and similarity.mmr_diversify."""
    import itertools, json
    from mrag_amd.pairwise import cosine_matrix, semantic_edges, mmr_diversify
    from oracle import ref_semantics as rs
    rng = np.random.default_rng(8)
    base = rng.standard_normal((6, 48))
    vecs = np.concatenate([base + 0.05 * rng.standard_normal((6, 48)) for _ in range(7)])     # 42 "sentences", clustered
    vecs[5] = 0
    s = cosine_matrix(vecs)
    want = np.array([[rs.cosine(list(a), list(b)) for b in vecs] for a in vecs])
    np.testing.assert_allclose(s, want, rtol=0, atol=1e-14)
    edges = semantic_edges(vecs, 0.9)
    ref_edges = [(i, j) for i, j in itertools.combinations(range(len(vecs)), 2) if want[i, j] >= 0.9]
    assert [(i, j) for i, j, _ in edges] == ref_edges and len(edges) > 20
    d = json.loads((golden_dir / "f4_minmax.json").read_text())
    items = [(i, sc, v) for i, sc, v in d["mmr_items"]]
    for c in d["mmr"]:
        sel = mmr_diversify(items, top_k=c["top_k"], lambda_weight=c["lambda"])
        assert [x[0] for x in sel] == c["selected_ids"]          # == the reference's own selections (fixture F4)


def test_f8_segmentation_on_the_adjacent_cosine_kernel(golden_dir):
    """embed-mode segment_context (segmenter.py:32-50, eps inside the denominator) on the GPU kernel
    vs F8 = the reference's own segment_context outputs, and the kernel vs the reference formula."""
    import json
    from mrag_amd.pairwise import segment_context, cosine_adjacent
    from oracle import ref_semantics as rs
    d = json.loads((golden_dir / "f8_segment.json").read_text())
    table = d["table"]
    ctx = [(t, s) for t, s in d["ctx"]]
    for c in d["cases"]:
        fn = None if c.get("no_embed_fn") else (lambda s: table[s])
        got = segment_context(ctx, strategy=c["strategy"], embed_fn=fn, sim_threshold=c["sim_threshold"])
        assert [[t, s] for t, s in got] == c["out"], (c["strategy"], c["sim_threshold"])
    v = np.asarray(list(table.values()), dtype=np.float64)
    want = np.asarray([rs.adjacent_similarity(v[i], v[i + 1]) for i in range(len(v) - 1)])
    np.testing.assert_allclose(cosine_adjacent(v), want, rtol=0, atol=1e-14)
    assert cosine_adjacent(v[:1]).shape == (0,)


@pytest.mark.parametrize("nq,n,d,k", [(1, 100000, 768, 20), (16, 70001, 384, 10), (5, 9000, 64, 64), (2, 2048, 128, 1)])
def test_online_regime_streaming_kernel(nq, n, d, k):
    """<= 16 queries take the HBM-streaming kernel (K2s): same exactness bars."""
    c16 = ds.normalize_round(ds.make_gaussian(n, d, 77))
    q16 = ds.normalize_round(ds.make_gaussian(nq, d, 78))
    _check(_index(c16), q16, c16, k)


def test_online_regime_adversarial_and_ties():
    d, n = 64, 8000
    u = ds.l2_normalize(ds.make_gaussian(1, d, 9))[0]
    c = (np.linspace(0.05, 1.0, n, dtype=np.float32)[:, None] * u[None, :]).astype(np.float16)   # every row beats the last
    q = ds.normalize_round(u[None, :] + 0.01 * ds.make_gaussian(3, d, 10))
    from mrag_amd.index import DenseIndex
    ix = DenseIndex(d, metric="ip")
    ix.add(c, normalize=False)
    sc, ids = ix.search(q, 10, normalize=False)
    rv, ri = ds.brute_force_topk(q, c, 10)
    np.testing.assert_allclose(sc, rv, rtol=0, atol=TOL)
    assert ds.gap_aware_id_match(ids, sc, ri, rv, tol=TOL)[1] == 0
    rng = np.random.default_rng(3)
    ci = rng.integers(-2, 3, size=(5000, 64)).astype(np.float16)                                   # exact integer ties
    qi = rng.integers(-2, 3, size=(7, 64)).astype(np.float16)
    ix2 = DenseIndex(64, metric="ip")
    ix2.add(ci, normalize=False)
    for k in (1, 10, 33):
        s2, i2 = ix2.search(qi, k, normalize=False)
        rv2, ri2 = ds.brute_force_topk(qi, ci, k)
        assert (i2 == ri2).all() and (s2 == rv2.astype(np.float32)).all()


@pytest.mark.parametrize("nparts,nq,k,pool", [(8, 300, 10, 200), (2, 17, 1, 5), (3, 64, 64, 100), (64, 9, 32, 3000),
                                              (5, 40, 7, 20)])
def test_device_merge_matches_host_merge(nparts, nq, k, pool):
    """8e on the device: mrag_topk_merge_device == mrag_topk_merge (score desc, id asc, empties last),
    ties and short parts included."""
    import torch
    from mrag_amd.index import topk_merge, topk_merge_device
    rng = np.random.default_rng(nparts * 1000 + k)
    sc = np.full((nparts, nq, k), -np.inf, dtype=np.float32)
    ids = np.full((nparts, nq, k), -1, dtype=np.int64)
    for q in range(nq):
        gid = rng.permutation(pool).astype(np.int64) + 5_000_000_000 * (q % 2)     # ids beyond 2^32 too
        val = rng.integers(0, 12, size=pool).astype(np.float32) / 4                 # many exact ties
        owner = rng.integers(0, nparts, size=pool)
        for p_ in range(nparts):
            m = owner == p_
            order = np.lexsort((gid[m], -val[m]))[:k]
            sc[p_, q, :len(order)] = val[m][order]
            ids[p_, q, :len(order)] = gid[m][order]
    hs, hi = topk_merge(sc, ids)
    ds_, di = topk_merge_device(torch.from_numpy(sc).cuda(), torch.from_numpy(ids).cuda())
    torch.cuda.synchronize()
    assert (di.cpu().numpy() == hi).all()
    assert np.array_equal(ds_.cpu().numpy(), hs)


def test_packed_exchange_kernels_match_the_torch_forms():
    """The exchange's one-kernel forms: ``mrag_pack_partial_device`` == ``sharded.pack_partial`` bit for bit (empties, ids up to
    base + 2^32 - 2, -0.0 / -inf / denormal scores), its flag fires for an id below the base or beyond 32 bits, and
    ``mrag_topk_merge_packed_device`` over gathered words == ``mrag_topk_merge`` over the unpacked arrays."""
    import torch
    from mrag_amd.index import pack_partial_device, topk_merge, topk_merge_packed_device
    from mrag_amd.sharded import pack_partial, unpack_partial
    rng = np.random.default_rng(12)
    nparts, nq, k = 5, 333, 10
    bases = np.array([0, 7_000_000_000, 7_000_000_000 + 2**31, 9_123_456_789, 2**40], dtype=np.int64)
    sc = np.full((nparts, nq, k), -np.inf, dtype=np.float32)
    ids = np.full((nparts, nq, k), -1, dtype=np.int64)
    for p_ in range(nparts):
        for q in range(nq):
            m = int(rng.integers(0, k + 1))
            v = np.sort(rng.integers(-3, 9, size=m).astype(np.float32) / 4)[::-1]
            sc[p_, q, :m] = v
            loc = np.sort(rng.choice(2**32 - 2, size=m, replace=False)) if m else np.empty(0, np.int64)
            order = np.lexsort((loc, -v)) if m else []
            ids[p_, q, :m] = bases[p_] + loc[order] if m else []
            sc[p_, q, :m] = v[order] if m else []
    for (p_, q, val) in ((0, 0, 3.0e-39), (1, 1, -0.0)):          # single-entry parts with a denormal / a negative zero
        sc[p_, q, :] = -np.inf; ids[p_, q, :] = -1
        sc[p_, q, 0] = val; ids[p_, q, 0] = bases[p_] + 17
    words = []
    for p_ in range(nparts):
        s_t, i_t = torch.from_numpy(sc[p_]).cuda(), torch.from_numpy(ids[p_]).cuda()
        flag = torch.zeros(1, dtype=torch.int32, device="cuda")
        w = pack_partial_device(s_t, i_t, int(bases[p_]), bad_flag=flag)
        assert torch.equal(w, pack_partial(s_t, i_t, int(bases[p_]))) and int(flag.item()) == 0
        words.append(w)
    gw = torch.stack(words)
    gs, gi = unpack_partial(gw, torch.from_numpy(bases).cuda().view(nparts, 1, 1))
    assert torch.equal(gi.cpu(), torch.from_numpy(ids)) and torch.equal(gs.cpu().view(torch.int32), torch.from_numpy(sc).view(torch.int32))
    ms, mi = topk_merge_packed_device(gw, torch.from_numpy(bases).cuda())
    hs, hi = topk_merge(sc, ids)
    torch.cuda.synchronize()
    assert (mi.cpu().numpy() == hi).all() and np.array_equal(ms.cpu().numpy(), hs)
    for bad_id in (int(bases[1]) - 1, int(bases[1]) + 2**32 - 1):
        flag = torch.zeros(1, dtype=torch.int32, device="cuda")
        i_bad = torch.from_numpy(ids[1]).cuda().clone(); i_bad[3, 0] = bad_id
        pack_partial_device(torch.from_numpy(sc[1]).cuda(), i_bad, int(bases[1]), bad_flag=flag)
        assert int(flag.item()) == 1


def test_device_merge_limits_are_loud():
    import torch
    from mrag_amd import _native as N
    from mrag_amd.index import topk_merge_device
    with pytest.raises(N.MragError):
        topk_merge_device(torch.zeros((65, 2, 4), device="cuda"), torch.zeros((65, 2, 4), dtype=torch.int64, device="cuda"))
    with pytest.raises(ValueError):
        topk_merge_device(torch.zeros((2, 2, 4)), torch.zeros((2, 2, 4), dtype=torch.int64))


def test_random_shapes_fuzz():
    """seeded sweep over ragged shapes: query counts around the 16 / 256 boundaries (online vs batch
    kernel, partial query tiles), corpus sizes around tile and split boundaries, padded dims, every k regime."""
    rng = np.random.default_rng(2024)
    nqs = [1, 15, 16, 17, 255, 256, 257, 300, 513]
    ns = [1, 7, 255, 256, 257, 2047, 2048, 2049, 5000, 9999, 33000]
    ds_ = [8, 40, 64, 100, 128, 384]
    ks = [1, 2, 7, 10, 15, 16, 17, 31, 33, 64]
    for _ in range(28):
        nq, n, d, k = (int(rng.choice(a)) for a in (nqs, ns, ds_, ks))
        c16 = ds.normalize_round(ds.make_gaussian(n, d, int(rng.integers(1 << 30))))
        q16 = ds.normalize_round(ds.make_gaussian(nq, d, int(rng.integers(1 << 30))))
        ix = _index(c16)
        try:
            _check(ix, q16, c16, k)
        except AssertionError as e:
            raise AssertionError(f"shape nq={nq} n={n} d={d} k={k}: {e}") from e
        finally:
            ix.close()
