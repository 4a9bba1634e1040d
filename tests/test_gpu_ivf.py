"""GPU parity of IVF-flat (BASELINE config 5 shape family).  The reference ships no IVF index:
parity is UNPINNED by the reference and checked against the oracle's ivf_search on identical
centroids + assignments, and against brute force for recall."""
import numpy as np
import pytest

from oracle import dense_search as ds

pytestmark = pytest.mark.gpu


def _data(n, nq, d, seed):
    rows, qs = ds.make_clustered(n, nq, d, seed, n_centroids=64)
    return ds.normalize_round(rows), ds.normalize_round(qs)


def test_assignment_and_search_match_oracle_on_given_centroids():
    from mrag_amd.index import IVFFlatIndex
    n, nq, d, nlist, nprobe, k = 20000, 300, 128, 64, 8, 10
    c16, q16 = _data(n, nq, d, 7)
    cen = ds.kmeans_spherical(c16, nlist, 4, seed=3)
    ix = IVFFlatIndex(d, nlist)
    ix.set_centroids(cen, normalize=False)
    assert np.array_equal(ix.centroids().astype(np.float16), cen)
    for lo in range(0, n, 7000):                            # incremental adds
        ix.add(c16[lo:lo + 7000], normalize=False)
    assert len(ix) == n
    a_ref = ds.ivf_assign(c16, cen)
    a_gpu = ix.assignments()
    # assignment = argmax in fp32-accumulated MFMA vs fp64: only fp-near-ties may differ
    cs = np.sort(c16.astype(np.float64) @ cen.astype(np.float64).T, axis=1)
    near = (cs[:, -1] - cs[:, -2]) < 1e-5
    assert (a_gpu[~near] == a_ref[~near]).all() and near.mean() < 0.01
    sc, ids = ix.search(q16, k, nprobe, normalize=False)
    rv, ri = ds.ivf_search(q16, c16, cen, a_gpu.astype(np.int64), nprobe, k)
    np.testing.assert_allclose(sc, rv, rtol=0, atol=1e-5)
    strict, bad = ds.gap_aware_id_match(ids, sc, ri, rv, tol=1e-5)
    # a query whose nprobe-th and (nprobe+1)-th centroid scores tie within fp32 may probe another list
    assert bad <= 0.002 * nq * k, (strict, bad)
    bv, bi = ds.brute_force_topk(q16, c16, k)
    assert ds.recall_at_k(ids, bi) > 0.8
    # probing every list == brute force
    sc_all, ids_all = ix.search(q16, k, nlist, normalize=False)
    np.testing.assert_allclose(sc_all, bv, rtol=0, atol=1e-5)
    s2, b2 = ds.gap_aware_id_match(ids_all, sc_all, bi, bv, tol=1e-5)
    assert b2 == 0 and ds.recall_at_k(ids_all, bi) >= 0.999


def test_train_on_device_gives_useful_lists():
    from mrag_amd.index import IVFFlatIndex
    n, nq, d, nlist, k = 30000, 200, 64, 128, 10
    c16, q16 = _data(n, nq, d, 11)
    ix = IVFFlatIndex(d, nlist)
    ix.train(c16, iters=8, seed=5, normalize=False)
    cen = ix.centroids()
    np.testing.assert_allclose(np.linalg.norm(cen, axis=1), 1.0, atol=2e-3)
    ix2 = IVFFlatIndex(d, nlist)
    ix2.train(c16, iters=8, seed=5, normalize=False)        # seeded: reproducible up to float-atomic order
    assert np.abs(ix2.centroids() - cen).max() < 5e-3
    ix.add(c16, normalize=False)
    ix.set_id_base(5_000_000)
    counts = np.bincount(ix.assignments(), minlength=nlist)
    assert counts.sum() == n and (counts > 0).mean() > 0.9
    bv, bi = ds.brute_force_topk(q16, c16, k)
    sc, ids = ix.search(q16, k, 16, normalize=False)
    assert (ids[ids >= 0] >= 5_000_000).all()
    assert ds.recall_at_k(ids - 5_000_000, bi) > 0.85
    # found ids carry their exact scores
    got = {(i, int(j)): s for i in range(nq) for j, s in zip(ids[i] - 5_000_000, sc[i]) if j >= 0}
    full = q16.astype(np.float64) @ c16.astype(np.float64).T
    for (i, j), s in list(got.items())[:500]:
        assert abs(full[i, j] - s) < 1e-5


def test_ivf_edge_cases():
    from mrag_amd.index import IVFFlatIndex
    from mrag_amd._native import MragError
    d = 32
    ix = IVFFlatIndex(d, 4)
    with pytest.raises(MragError):
        ix.add(np.zeros((3, d), np.float32))               # no centroids yet
    cen = ds.normalize_round(ds.make_gaussian(4, d, 1))
    ix.set_centroids(cen, normalize=False)
    sc, ids = ix.search(ds.normalize_round(ds.make_gaussian(3, d, 2)), 5, 2, normalize=False)   # empty index
    assert (ids == -1).all() and np.isneginf(sc).all()
    rows = ds.normalize_round(ds.make_gaussian(7, d, 3))
    ix.add(rows, normalize=False)
    sc, ids = ix.search(rows[:2], 10, 4, normalize=False)  # k > n
    assert (ids[:, 7:] == -1).all() and (ids[:, 0] == [0, 1]).all()


def _ivf_with(c16, cen):
    from mrag_amd.index import IVFFlatIndex
    ix = IVFFlatIndex(c16.shape[1], cen.shape[0])
    ix.set_centroids(cen, normalize=False)
    ix.add(c16, normalize=False)
    return ix


def test_select_ties_and_large_candidate_sets():
    """The score-segment regime's per-query selection (csrc/ivf_scan.hip): order is (score desc, original row asc) also
    when hundreds of rows tie on the k-th score (shortlist overflow -> serial path) and when a query has more
    candidates than the register-resident fast path holds (> 8192 -> the two-pass streamed path; with > 512 ties on the
    k-th score in it -> serial path)."""
    d, nlist = 64, 8
    rng = np.random.default_rng(5)
    base = ds.normalize_round(ds.make_gaussian(3000, d, 21))
    dup = np.repeat(base[:3], 1200, axis=0)                    # 3 rows x 1200 identical copies: more ties than the shortlist holds
    dup2 = np.repeat(base[3:6], 300, axis=0)                   # 3 rows x 300: ties ranked inside the shortlist
    c16 = np.concatenate([base, dup, dup2], axis=0)
    perm = rng.permutation(len(c16))
    c16 = c16[perm]
    cen = ds.kmeans_spherical(c16, nlist, 3, seed=1)
    ix = _ivf_with(c16, cen)
    q16 = np.concatenate([base[:6], ds.normalize_round(ds.make_gaussian(26, d, 22))], axis=0)
    for k in (1, 10, 64):
        sc, ids = ix.search(q16, k, nlist // 2, normalize=False)
        rv, ri = ds.ivf_search(q16, c16, cen, ix.assignments().astype(np.int64), nlist // 2, k)
        np.testing.assert_allclose(sc, rv, rtol=0, atol=1e-5)
        # the duplicated rows: the query IS the row, its copies score exactly alike -> lowest original rows first
        for qi in range(6):
            copies = np.sort(np.nonzero((c16 == q16[qi]).all(axis=1))[0])
            assert (ids[qi] == copies[:k]).all(), (k, qi, ids[qi][:8], copies[:8])
        strict, bad = ds.gap_aware_id_match(ids[6:], sc[6:], ri[6:], rv[6:], tol=1e-5)
        assert bad == 0
    # > 8192 candidates per query: 40 000 rows in 4 lists, all probed through the selection kernel (nprobe < nlist)
    big = ds.normalize_round(ds.make_gaussian(40000, d, 23))
    cen5 = ds.kmeans_spherical(big, 5, 3, seed=2)
    ix5 = _ivf_with(big, cen5)
    qs = ds.normalize_round(ds.make_gaussian(16, d, 24))
    sc, ids = ix5.search(qs, 10, 4, normalize=False)
    rv, ri = ds.ivf_search(qs, big, cen5, ix5.assignments().astype(np.int64), 4, 10)
    np.testing.assert_allclose(sc, rv, rtol=0, atol=1e-5)
    strict, bad = ds.gap_aware_id_match(ids, sc, ri, rv, tol=1e-5)
    assert bad == 0
    # the streamed path with exact ties: 40 copies of three rows (ranked inside the shortlist, lowest original rows first)
    # and 900 copies of a fourth (more ties than the shortlist holds -> serial path), k below / across / above the copies
    tied = np.concatenate([big, np.repeat(big[:3], 40, axis=0), np.repeat(big[3:4], 900, axis=0)], axis=0)
    tied = tied[rng.permutation(len(tied))]
    ix6 = _ivf_with(tied, cen5)
    q6 = np.concatenate([big[:4], qs[:4]], axis=0)
    for k in (5, 41, 64):
        sc, ids = ix6.search(q6, k, 4, normalize=False)
        rv, ri = ds.ivf_search(q6, tied, cen5, ix6.assignments().astype(np.int64), 4, k)
        np.testing.assert_allclose(sc, rv, rtol=0, atol=1e-5)
        for qi in range(4):
            copies = np.sort(np.nonzero((tied == q6[qi]).all(axis=1))[0])
            m = min(k, len(copies))
            assert (ids[qi][:m] == copies[:m]).all(), (k, qi, ids[qi][:6], copies[:6])
        assert ds.gap_aware_id_match(ids[4:], sc[4:], ri[4:], rv[4:], tol=1e-5)[1] == 0


def test_fused_regime_gives_the_same_answer():
    """MRAG_IVF_SCORES_MB=0 forces the fused GEMM + top-k kernel in descriptor mode (the regime kept for searches whose
    score segments would not fit), a small buffer forces query chunking: same scores bit for bit, same ids."""
    import os, subprocess, sys, tempfile, textwrap
    code = textwrap.dedent('''
        import sys, numpy as np
        sys.path.insert(0, ".")
        from oracle import dense_search as ds
        from mrag_amd.index import IVFFlatIndex
        rows, qs = ds.make_clustered(30000, 3000, 128, 31, n_centroids=64)
        c16, q16 = ds.normalize_round(rows), ds.normalize_round(qs)
        cen = ds.kmeans_spherical(c16, 96, 3, seed=4)
        ix = IVFFlatIndex(128, 96); ix.set_centroids(cen, normalize=False); ix.add(c16, normalize=False)
        out = {}
        for k, nprobe in ((10, 8), (64, 16), (3, 96)):
            sc, ids = ix.search(q16, k, nprobe, normalize=False)
            out["s%d_%d" % (k, nprobe)] = sc; out["i%d_%d" % (k, nprobe)] = ids
        np.savez(sys.argv[1], **out)
    ''')
    with tempfile.TemporaryDirectory() as td:
        res = {}
        # "chunked": a 48 MB score buffer cuts the 3 000 queries into three chunks (and sends the exhaustive case to the fused kernel)
        for mode, env in (("scores", {}), ("fused", {"MRAG_IVF_SCORES_MB": "0"}), ("chunked", {"MRAG_IVF_SCORES_MB": "48"})):
            path = os.path.join(td, mode + ".npz")
            subprocess.run([sys.executable, "-c", code, path], check=True, env={**os.environ, **env}, cwd=os.path.dirname(os.path.dirname(__file__)))
            res[mode] = np.load(path)
        for key in res["scores"].files:
            for other in ("fused", "chunked"):
                assert np.array_equal(res["scores"][key], res[other][key]), (key, other)


def test_bf16_storage_and_exact_integer_ties():
    """bf16 storage through the score-segment scan, and small-integer rows (inner product): every partial sum is exact
    in fp32, so scores and ids must equal the oracle's exactly -- ties, which are many, included."""
    import torch
    from mrag_amd.index import IVFFlatIndex
    n, nq, d, nlist, nprobe, k = 12000, 100, 128, 32, 8, 10
    rows, qs = ds.make_clustered(n, nq, d, 41, n_centroids=32)
    c = torch.from_numpy(ds.l2_normalize(rows)).to(torch.bfloat16)
    q = torch.from_numpy(ds.l2_normalize(qs)).to(torch.bfloat16)
    cen = torch.from_numpy(ds.kmeans_spherical(c.float().numpy(), nlist, 3, seed=1, dtype=np.float32)).to(torch.bfloat16)
    ix = IVFFlatIndex(d, nlist, dtype="bf16")
    ix.set_centroids(cen, normalize=False)
    ix.add(c, normalize=False)
    sc, ids = ix.search(q, k, nprobe, normalize=False)
    rv, ri = ds.ivf_search(q.double().numpy(), c.double().numpy(), cen.double().numpy(), ix.assignments().astype(np.int64), nprobe, k)
    np.testing.assert_allclose(sc, rv, rtol=0, atol=1e-5)
    strict, bad = ds.gap_aware_id_match(ids, sc, ri, rv, tol=1e-5)
    assert bad <= 0.002 * nq * k

    rng = np.random.default_rng(9)
    ci = rng.integers(-2, 3, size=(6000, 64)).astype(np.float16)
    qi = rng.integers(-2, 3, size=(64, 64)).astype(np.float16)
    ceni = rng.integers(-2, 3, size=(16, 64)).astype(np.float16)
    ixi = IVFFlatIndex(64, 16, metric="ip")
    ixi.set_centroids(ceni, normalize=False)
    ixi.add(ci, normalize=False)
    a = ixi.assignments().astype(np.int64)
    assert np.array_equal(a, ds.ivf_assign(ci, ceni))
    for kk, npb in ((10, 4), (64, 8), (1, 1)):
        sc, ids = ixi.search(qi, kk, npb, normalize=False)
        rv, ri = ds.ivf_search(qi, ci, ceni, a, npb, kk)
        assert np.array_equal(ids, ri), (kk, npb)
        assert np.array_equal(sc, rv.astype(np.float32))


def test_many_probes_and_odd_dim():
    """nprobe = 200 of 512 lists (the probe selection then keeps 200 of 512 per query, the selection walks 200 segments),
    a dimension that is not a multiple of 64, single-query and single-probe calls."""
    from mrag_amd.index import IVFFlatIndex
    n, nq, d, nlist = 40000, 70, 100, 512
    rows, qs = ds.make_clustered(n, nq, d, 51, n_centroids=256)
    c16, q16 = ds.normalize_round(rows), ds.normalize_round(qs)
    cen = ds.kmeans_spherical(c16, nlist, 2, seed=6)
    ix = IVFFlatIndex(d, nlist)
    ix.set_centroids(cen, normalize=False)
    ix.add(c16, normalize=False)
    a = ix.assignments().astype(np.int64)
    for k, nprobe, sl in ((10, 200, slice(0, nq)), (64, 33, slice(0, nq)), (5, 1, slice(0, 1)), (1, 200, slice(3, 4))):
        sc, ids = ix.search(q16[sl], k, nprobe, normalize=False)
        rv, ri = ds.ivf_search(q16[sl], c16, cen, a, nprobe, k)
        np.testing.assert_allclose(sc, rv, rtol=0, atol=1e-5)
        strict, bad = ds.gap_aware_id_match(ids, sc, ri, rv, tol=1e-5)
        assert bad <= max(1, int(0.003 * sc.size)), (k, nprobe, strict, bad)


def test_randomised_shapes_against_oracle():
    """24 seeded random (rows, lists, dim, queries, k, nprobe) combinations -- list lengths around the 128-row tile and the
    4 / 8 / 16-row granules of the scan, query groups around 128, k up to 64 -- against the oracle."""
    from mrag_amd.index import IVFFlatIndex
    rng = np.random.default_rng(2024)
    for case in range(24):
        nlist = int(rng.choice([3, 8, 17, 40, 64]))
        n = int(rng.integers(nlist * 20, nlist * 400))
        d = int(rng.choice([32, 64, 96, 160]))
        nq = int(rng.choice([1, 7, 127, 128, 129, 300]))
        k = int(rng.choice([1, 3, 10, 33, 64]))
        nprobe = int(rng.integers(1, nlist + 1))
        rows, qs = ds.make_clustered(n, nq, d, 100 + case, n_centroids=max(2, nlist // 2))
        c16, q16 = ds.normalize_round(rows), ds.normalize_round(qs)
        cen = ds.kmeans_spherical(c16, nlist, 2, seed=case)
        ix = IVFFlatIndex(d, nlist)
        ix.set_centroids(cen, normalize=False)
        ix.add(c16, normalize=False)
        a = ix.assignments().astype(np.int64)
        sc, ids = ix.search(q16, k, nprobe, normalize=False)
        rv, ri = ds.ivf_search(q16, c16, cen, a, nprobe, k)
        fin = np.isfinite(rv)
        assert np.array_equal(np.isfinite(sc), fin), (case, n, nlist, d, nq, k, nprobe)
        np.testing.assert_allclose(sc[fin], rv[fin], rtol=0, atol=1e-5, err_msg=str((case, n, nlist, d, nq, k, nprobe)))
        strict, bad = ds.gap_aware_id_match(ids, sc, ri, rv, tol=1e-5)
        assert bad <= max(1, int(0.004 * sc.size)), (case, n, nlist, d, nq, k, nprobe, strict, bad)
        ix.close()
