"""GPU parity at BASELINE.json's full sizes through size-independent properties + oracle subsamples:
C2 (1k x 100k x 768), C4 shape (10k x 1M x 768), C3 (bge-base encoder, full vocabulary / 512 positions, 64k+
passages), C5 (IVF-flat nlist 4096 / nprobe 32 over one GPU's 625k x 768 share of the 5M corpus)."""
import numpy as np
import pytest

from oracle import dense_search as ds

pytestmark = pytest.mark.gpu


def _gpu_corpus(n, d, seed):
    import torch
    g = torch.Generator(device="cuda").manual_seed(seed)
    return torch.randn(n, d, device="cuda", generator=g, dtype=torch.float32)


def _properties(sc, ids, n, k):
    assert sc.shape == ids.shape == (sc.shape[0], k)
    assert (np.diff(sc, axis=1) <= 0).all()                          # sorted descending
    assert (ids >= 0).all() and (ids < n).all()
    assert all(len(set(r)) == k for r in ids)                        # no duplicate rows
    tie = np.diff(sc, axis=1) == 0
    assert (np.diff(ids, axis=1)[tie] > 0).all()                     # ties: row ascending


@pytest.mark.parametrize("nq,n", [(1000, 100_000), (10_000, 1_000_000)])
def test_full_size_brute_force(nq, n):
    import torch
    from mrag_amd.index import DenseIndex, topk_merge
    d, k = 768, 10
    ix = DenseIndex(d)
    step = 250_000
    for lo in range(0, n, step):
        ix.add(_gpu_corpus(min(step, n - lo), d, 100 + lo // step))
    q = _gpu_corpus(nq, d, 7)
    sc_d, ids_d = ix.search(q, k)
    torch.cuda.synchronize()
    sc, ids = sc_d.cpu().numpy(), ids_d.cpu().numpy()
    _properties(sc, ids, n, k)
    # idempotence + query-permutation equivariance
    sc2, ids2 = ix.search(q, k)
    perm = torch.randperm(nq, device="cuda")
    sc3, ids3 = ix.search(q[perm], k)
    torch.cuda.synchronize()
    assert (ids2.cpu().numpy() == ids).all() and (sc2.cpu().numpy() == sc).all()
    assert (ids3.cpu().numpy() == ids[perm.cpu().numpy()]).all()
    # row-sharding + host merge == single index (the 8e exchange, on one GPU)
    c32 = ix.rows()
    parts_s, parts_i = [], []
    bounds = [0, n // 3, n // 2, n]
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        sh = DenseIndex(d)
        sh.add(c32[lo:hi].astype(np.float16), normalize=False)
        sh.set_id_base(lo)
        s_, i_ = sh.search(q, k)
        torch.cuda.synchronize()
        parts_s.append(s_.cpu().numpy()); parts_i.append(i_.cpu().numpy())
        sh.close()
    ms, mi = topk_merge(np.stack(parts_s), np.stack(parts_i))
    assert (mi == ids).all() and np.array_equal(ms, sc)
    # oracle on a query subsample (fp64 over the full corpus)
    sub = np.arange(0, nq, max(1, nq // 24))[:24]
    q16 = ds.normalize_round(q[torch.from_numpy(sub).cuda()].cpu().numpy())
    rv, ri = ds.brute_force_topk(q16, c32, k, block=131072)
    np.testing.assert_allclose(sc[sub], rv, rtol=0, atol=1e-5)
    assert ds.gap_aware_id_match(ids[sub], sc[sub], ri, rv, tol=1e-5)[1] == 0
    assert ds.recall_at_k(ids[sub], ri) >= 0.999
    # prefix consistency across k: the k=32 result (no row-count certificate in the batch kernel,
    # thresholds from scheduled compactions) must start with the k=10 result
    s32, i32 = ix.search(q, 32)
    torch.cuda.synchronize()
    s32, i32 = s32.cpu().numpy(), i32.cpu().numpy()
    _properties(s32, i32, n, 32)
    assert (i32[:, :k] == ids).all() and np.array_equal(s32[:, :k], sc)
    rv32, ri32 = ds.brute_force_topk(q16[:6], c32, 32, block=131072)
    assert ds.gap_aware_id_match(i32[sub[:6]], s32[sub[:6]], ri32, rv32, tol=1e-5)[1] == 0
    # one query at a time (online kernel) == the batch kernel
    s1, i1 = ix.search(q[:3], k)
    torch.cuda.synchronize()
    assert (i1.cpu().numpy() == ids[:3]).all()
    np.testing.assert_allclose(s1.cpu().numpy(), sc[:3], rtol=0, atol=1e-6)


def test_c3_encoder_full_shape_streamed_batches_match_oracle_subsample():
    """BASELINE config 3: bge-base shape with the FULL 30 522-row vocabulary and 512 positions, CLS pooling;
    65 536 passages at S = 64 plus 8 192 at S = 256 and 4 096 at S = 512, streamed in batches; the fp64 oracle
    (pinned to HF BertModel by F6) on a subsample of the rows that went through the batches, atol 1e-3."""
    from mrag_amd.encoder import HipSentenceEncoder, EncoderSpec, seeded_weights
    from oracle import encoder as oe
    spec = dict(oe.SPECS["bge-base"])
    assert spec["vocab_size"] == 30522 and spec["max_position"] == 512
    es = EncoderSpec(**dict(spec, max_length=512))
    w = seeded_weights(es, 3)
    enc = HipSentenceEncoder(es, w)
    rng = np.random.default_rng(1)
    for S, n_pass, batch, n_sub in ((64, 65_536, 4096, 6), (256, 8192, 1024, 3), (512, 4096, 512, 2)):
        sub = sorted(int(x) for x in rng.choice(n_pass, size=n_sub, replace=False))
        sub[0] = 0
        sub[-1] = n_pass - 1
        kept = {}
        checked = 0
        for lo in range(0, n_pass, batch):
            brng = np.random.default_rng(1000 * S + lo)        # every batch reproducible on its own
            ids = brng.integers(5, spec["vocab_size"], size=(batch, S)).astype(np.int32)
            lens = brng.integers(2, S + 1, size=batch)
            lens[0] = S                                        # a full-length row in every batch (positions up to S-1)
            mask = (np.arange(S)[None, :] < lens[:, None]).astype(np.int32)
            ids = ids * mask
            out = enc.forward(ids, mask, pool="cls")
            assert out.shape == (batch, 768) and np.isfinite(out).all()
            np.testing.assert_allclose(np.linalg.norm(out, axis=1), 1.0, atol=1e-5)
            for r in sub:
                if lo <= r < lo + batch:
                    kept[r] = (ids[r - lo].copy(), mask[r - lo].copy(), out[r - lo].copy())
            checked += batch
        assert checked == n_pass and sorted(kept) == sub
        sid = np.stack([kept[r][0] for r in sub]); smask = np.stack([kept[r][1] for r in sub])
        want = oe.forward(spec, w, sid, smask, pool="cls")
        got = np.stack([kept[r][2] for r in sub])
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-3)
        # batch composition does not change a row (padding rows are masked out)
        again = enc.forward(sid, smask, pool="cls")
        np.testing.assert_allclose(again, got, rtol=0, atol=5e-4)   # (a small batch takes the plain flow, the streamed ones the fused-LayerNorm flow)


def test_c5_ivf_flat_per_gpu_share():
    """BASELINE config 5 on ONE GPU's share: 625 000 x 768 rows (1/8 of the 5M corpus; SURVEY 8e shards rows and
    replicates the centroids), nlist 4096 trained on the device, nprobe 32, k = 10, clustered data (SURVEY 8d
    generator).  The reference has no IVF: parity is against the oracle's ivf_search on the SAME centroids and
    the GPU's own assignments (query subsample), recall@10 against exact brute force, and nprobe = nlist must
    equal brute force."""
    import torch
    from mrag_amd.index import IVFFlatIndex, DenseIndex
    n, nq, d, nlist, nprobe, k = 625_000, 10_000, 768, 4096, 32, 10
    rows, qs = ds.make_clustered(n, nq, d, seed=2024, n_centroids=4096)
    c16, q16 = ds.normalize_round(rows), ds.normalize_round(qs)
    del rows, qs
    ix = IVFFlatIndex(d, nlist)
    ix.train(c16[:: n // 100_000][:100_000], iters=5, seed=1, normalize=False)
    bf = DenseIndex(d)
    for lo in range(0, n, 125_000):
        ix.add(c16[lo:lo + 125_000], normalize=False)
        bf.add(c16[lo:lo + 125_000], normalize=False)
    assert len(ix) == n
    cen = ix.centroids().astype(np.float16)
    a_gpu = ix.assignments().astype(np.int64)
    counts = np.bincount(a_gpu, minlength=nlist)
    assert counts.sum() == n and (counts > 0).mean() > 0.9
    # assignment parity on a row subsample (argmax over 4096 centroids, fp32 MFMA accumulation vs fp64)
    rsub = np.arange(0, n, n // 2000)[:2000]
    cs = c16[rsub].astype(np.float64) @ cen.astype(np.float64).T
    srt = np.sort(cs, axis=1)
    clear = (srt[:, -1] - srt[:, -2]) > 1e-5
    assert (a_gpu[rsub][clear] == np.argmax(cs, axis=1)[clear]).all() and clear.mean() > 0.99
    sc, ids = ix.search(q16, k, nprobe, normalize=False)
    t = ix.last_timing()
    assert t["n_wg"] > 0 and 0 < t["scan_ms"] <= t["total_ms"] and t["scanned_rows"] > 0
    # oracle on a query subsample
    sub = np.arange(0, nq, nq // 128)[:128]
    rv, ri = ds.ivf_search(q16[sub], c16, cen, a_gpu, nprobe, k)
    np.testing.assert_allclose(sc[sub], rv, rtol=0, atol=1e-5)
    strict, bad = ds.gap_aware_id_match(ids[sub], sc[sub], ri, rv, tol=1e-5)
    assert bad <= 0.002 * len(sub) * k, (strict, bad)     # a probe-boundary near-tie may swap one list
    # recall@10 vs exact brute force, all 10k queries
    bs, bi = bf.search(q16, k, normalize=False)
    bv, bri = ds.brute_force_topk(q16[sub[:32]], c16, k)
    assert ds.gap_aware_id_match(bi[sub[:32]], bs[sub[:32]], bri, bv, tol=1e-5)[1] == 0
    recall = ds.recall_at_k(ids, bi)
    print(f"C5 share: recall@10 vs brute force = {recall:.4f}; scan {t['scan_ms']:.2f} ms, search {t['total_ms']:.2f} ms, "
          f"{t['n_wg']} workgroups, {t['scanned_rows']} rows streamed")
    assert recall >= 0.99
    # nprobe = nlist: exhaustive == brute force (64 queries: every list x every query)
    es, ei = ix.search(q16[:64], k, nlist, normalize=False)
    np.testing.assert_allclose(es, bs[:64], rtol=0, atol=1e-5)
    assert ds.gap_aware_id_match(ei, es, bi[:64], bs[:64], tol=1e-5)[1] == 0
    # nprobe between 64 and 256 goes through the streaming kernel's probe selection
    ws, wi = ix.search(q16[:24], k, 100, normalize=False)
    wv, wri = ds.ivf_search(q16[:24], c16, cen, a_gpu, 100, k)
    np.testing.assert_allclose(ws, wv, rtol=0, atol=1e-5)
    assert ds.gap_aware_id_match(wi, ws, wri, wv, tol=1e-5)[1] <= 1


def test_rccl_exchange_path_single_rank():
    """The N > 1 exchange on the GPU with the real backend: an RCCL process group of ONE rank forced through
    all_gather_into_tensor + the device / host merge must return the plain single-index result."""
    import os
    import torch
    import torch.distributed as dist
    from mrag_amd.sharded import ShardedDenseIndex
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29617")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        n, d, nq, k = 40_000, 256, 700, 10
        sh = ShardedDenseIndex(d, n, rank=0, world=1, device=0)
        sh.add_local(_gpu_corpus(n, d, 31))
        q = _gpu_corpus(nq, d, 32)
        ref_s, ref_i = sh.search(q, k)
        ref_s, ref_i = ref_s.copy(), ref_i.copy()
        for mode in ("device", "host"):
            s_, i_ = sh.search(q, k, merge=mode, force_collective=True)
            assert (i_ == ref_i).all() and np.array_equal(s_, ref_s), mode
            assert {"local_ms", "pack_ms", "gather_ms", "d2h_ms"} <= set(sh.last_phases), sh.last_phases
            # two-half pipeline (all-gather of half A async on RCCL's stream under the local search of half B) == one shot
            s_, i_ = sh.search(q, k, merge=mode, force_collective=True, pipeline=True)
            assert (i_ == ref_i).all() and np.array_equal(s_, ref_s), (mode, "pipeline")
            assert sh.last_phases["local_ms"] > 0
    finally:
        if created:
            dist.destroy_process_group()


def test_c5_ivf_flat_full_5m_on_one_gpu():
    """BASELINE config 5 at its FULL size on one GPU: 5 000 000 x 768 fp16 = 7.7 GB -- the first corpus
    past 2^32 bytes (the batch kernel's and the streaming kernel's 64-bit tile bases, the IVF regrouped copy) -- generated
    on the device, nlist 4096 trained on a 100 k sample, nprobe 32, 10 k queries.  IVF has no reference counterpart
    (parity unpinned by the reference): the oracle's ``ivf_search`` runs on the GPU's own centroids / assignments for a
    query subsample; recall@10 against ``DenseIndex`` brute force over the same rows, itself checked against the fp64
    oracle on a subsample; ``nprobe = nlist`` must equal brute force."""
    import torch
    from mrag_amd.index import IVFFlatIndex, DenseIndex
    n, nq, d, nlist, nprobe, k = 5_000_000, 10_000, 768, 4096, 32, 10
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(55)
    cent = torch.randn(nlist, d, device=dev, generator=g)

    def clustered(m):
        return cent[torch.randint(0, nlist, (m,), device=dev, generator=g)] + 0.3 * torch.randn(m, d, device=dev, generator=g)
    ix, bf = IVFFlatIndex(d, nlist), DenseIndex(d)
    ix.train(clustered(100_000), iters=5, seed=1)
    step = 250_000
    for lo in range(0, n, step):
        rows = clustered(step)
        ix.add(rows)
        bf.add(rows)
    del rows
    assert len(ix) == n and len(bf) == n and n * d * 2 > 2 ** 32
    q = cent[torch.randint(0, nlist, (nq,), device=dev, generator=g)] + 0.3 * torch.randn(nq, d, device=dev, generator=g)
    sc, ids = ix.search(q, k, nprobe)
    bs, bi = bf.search(q, k)
    torch.cuda.synchronize()
    t = ix.last_timing()
    sc, ids, bs, bi = sc.cpu().numpy(), ids.cpu().numpy(), bs.cpu().numpy(), bi.cpu().numpy()
    _properties(bs, bi, n, k)
    assert (ids < n).all() and (ids[:, 0] >= 0).all()
    assert (bi >= 2 ** 32 // (d * 2)).any()                  # answers come from rows stored beyond the 4 GiB mark
    recall = ds.recall_at_k(ids, bi)
    print(f"C5 full size: recall@10 vs brute force = {recall:.4f}; list scan {t['scan_ms']:.2f} ms, search {t['total_ms']:.2f} ms, "
          f"{t['n_wg']} workgroups, {t['scanned_rows']} rows streamed; brute force {bf.last_timing_ms()[1]:.1f} ms")
    assert recall >= 0.99
    # the stored rows on the host (fp16: 7.7 GB), fetched in slices
    c16 = np.empty((n, d), dtype=np.float16)
    for lo in range(0, n, 500_000):
        c16[lo:lo + 500_000] = bf.rows(lo, min(500_000, n - lo))
    q16 = ds.normalize_round(q.cpu().numpy())
    # brute force (batch kernel over 7.7 GB) vs the fp64 oracle, 16 queries
    sub = np.arange(0, nq, nq // 16)[:16]
    rv, ri = ds.brute_force_topk(q16[sub], c16, k, block=131072)
    np.testing.assert_allclose(bs[sub], rv, rtol=0, atol=1e-5)
    assert ds.gap_aware_id_match(bi[sub], bs[sub], ri, rv, tol=1e-5)[1] == 0
    # IVF vs the oracle's ivf_search on the GPU's own centroids / assignments, 32 queries
    cen = ix.centroids().astype(np.float16)
    a_gpu = ix.assignments().astype(np.int64)
    counts = np.bincount(a_gpu, minlength=nlist)
    assert counts.sum() == n and counts.max() < 8 * n // nlist
    sub2 = np.arange(0, nq, nq // 32)[:32]
    iv, ii = ds.ivf_search(q16[sub2], c16, cen, a_gpu, nprobe, k)
    np.testing.assert_allclose(sc[sub2], iv, rtol=0, atol=1e-5)
    strict, bad = ds.gap_aware_id_match(ids[sub2], sc[sub2], ii, iv, tol=1e-5)
    assert bad <= 1, (strict, bad)                            # a probe-boundary near-tie may swap one list
    # nprobe = nlist: exhaustive == brute force (16 queries x every list)
    es, ei = ix.search(q[:16], k, nlist)
    torch.cuda.synchronize()
    es, ei = es.cpu().numpy(), ei.cpu().numpy()
    np.testing.assert_allclose(es, bs[:16], rtol=0, atol=1e-5)
    assert ds.gap_aware_id_match(ei, es, bi[:16], bs[:16], tol=1e-5)[1] == 0
    # the streaming kernel (one query per call) walks the same 7.7 GB: == the batch kernel's answer
    s1, i1 = bf.search(q[sub[-1]:sub[-1] + 1], k)
    torch.cuda.synchronize()
    assert (i1.cpu().numpy()[0] == bi[sub[-1]]).all()
    np.testing.assert_allclose(s1.cpu().numpy()[0], bs[sub[-1]], rtol=0, atol=1e-6)
    ix.close(); bf.close()


MULTI_RANK_WORKER = r"""
import os, sys
sys.path.insert(0, os.environ["MRAG_ROOT"])
import numpy as np, torch, torch.distributed as dist
from mrag_amd.sharded import ShardedDenseIndex, ShardedIVFIndex, shard_bounds
from mrag_amd.index import DenseIndex, IVFFlatIndex
from oracle import dense_search as ds
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")            # gloo moves CUDA tensors too: real kernels on every rank, all ranks on cuda:0
torch.cuda.set_device(0)
n, nq, d, k = 90_001, 700, 256, 10
c16 = ds.normalize_round(ds.make_gaussian(n, d, 1234)); q16 = ds.normalize_round(ds.make_gaussian(nq, d, 5678))
lo, hi = shard_bounds(n, world, rank)
sh = ShardedDenseIndex(d, n, rank, world, device=0)
sh.add_local(torch.from_numpy(c16[lo:hi]).cuda(), normalize=False)
q = torch.from_numpy(q16).cuda()
one = DenseIndex(d); one.add(torch.from_numpy(c16).cuda(), normalize=False)
rs_, ri_ = one.search(q, k, normalize=False); torch.cuda.synchronize()
rs_, ri_ = rs_.cpu().numpy(), ri_.cpu().numpy()
for mode in ("device", "host"):
    for pipe in (False, True):
        s_, i_ = sh.search(q, k, merge=mode, pipeline=pipe, normalize=False)
        assert (i_ == ri_).all() and np.array_equal(s_, rs_), (mode, pipe, "sharded over ranks != one index")
        assert sh.last_phases["local_ms"] > 0 and "gather_ms" in sh.last_phases
ov, oi = ds.brute_force_topk(q16[:64], c16, k)
assert ds.gap_aware_id_match(i_[:64], s_[:64], oi, ov, tol=1e-5)[1] == 0
# the reference's pool (k = 200) through the exchange: wide-batch path per rank, K4w-sized lists in the merge
s2, i2 = sh.search(q, 200, merge="host", normalize=False)
r2s, r2i = one.search(q, 200, normalize=False); torch.cuda.synchronize()
assert (i2 == r2i.cpu().numpy()).all() and np.array_equal(s2, r2s.cpu().numpy())
# IVF: rows sharded, centroids trained on rank 0 and broadcast
nlist, nprobe = 64, 8
iv = ShardedIVFIndex(d, nlist, n, rank, world, device=0)
iv.train(torch.from_numpy(c16[::3]).cuda() if rank == 0 else None, iters=4, seed=3, normalize=False)
iv.add_local(torch.from_numpy(c16[lo:hi]).cuda(), normalize=False)
vs, vi = iv.search(q, k, nprobe)
full = IVFFlatIndex(d, nlist); full.set_centroids(iv.index.centroids(), normalize=False); full.add(torch.from_numpy(c16).cuda(), normalize=False)
fs, fi = full.search(q, k, nprobe); torch.cuda.synchronize()
fs, fi = fs.cpu().numpy(), fi.cpu().numpy()
np.testing.assert_allclose(vs, fs, rtol=0, atol=1e-6)
assert ds.gap_aware_id_match(vi, vs, fi, fs.astype(np.float64), tol=1e-6)[1] == 0
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_three_ranks_on_one_gpu_real_kernels(tmp_path):
    """SURVEY 8e with REAL kernels on every rank: three processes share cuda:0 (no 8-GPU node in reach), each owns a row shard
    in HBM, the exchange runs over gloo on CUDA tensors (same ``all_gather_into_tensor`` call as RCCL), merge on the device
    and on the host, one-shot and two-half pipeline: ids and scores equal the single-index answer bit for bit; k = 200 through
    the exchange; the row-sharded IVF index against one IVF index over all rows with the same centroids."""
    import os, subprocess, sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    script = tmp_path / "ranks.py"
    script.write_text(MULTI_RANK_WORKER)
    env = dict(os.environ, MRAG_ROOT=str(root), MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3",
                        "--master-addr", "127.0.0.1", "--master-port", "29641", str(script)],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("ok") == 3
