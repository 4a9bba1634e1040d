"""GPU parity at BASELINE.json's full sizes through size-independent properties + oracle subsamples:
C2 (1k x 100k x 768), C4 shape (10k x 1M x 768), C3 (bge-base-shaped encoder over thousands of passages)."""
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


def test_c3_encoder_batch_matches_oracle_subsample():
    from mrag_amd.encoder import HipSentenceEncoder, EncoderSpec
    from oracle import encoder as oe
    spec = dict(oe.SPECS["bge-base"], vocab_size=5000, max_position=128)
    w = oe.seeded_weights(spec, 3)
    enc = HipSentenceEncoder(EncoderSpec(**dict(spec, max_length=128)), w)
    rng = np.random.default_rng(1)
    B, S = 4096, 96
    ids = rng.integers(5, 5000, size=(B, S)).astype(np.int32)
    lens = rng.integers(4, S + 1, size=B)
    mask = (np.arange(S)[None, :] < lens[:, None]).astype(np.int32)
    ids = ids * mask
    out = np.concatenate([enc.forward(ids[lo:lo + 1024], mask[lo:lo + 1024], pool="cls") for lo in range(0, B, 1024)])
    assert np.isfinite(out).all()
    np.testing.assert_allclose(np.linalg.norm(out, axis=1), 1.0, atol=1e-5)
    sub = [0, 1, 777, 2048, 4095]
    want = oe.forward(spec, w, ids[sub], mask[sub], pool="cls")
    np.testing.assert_allclose(out[sub], want, rtol=0, atol=1e-3)
    # batch composition does not change a row (padding rows are masked out)
    again = enc.forward(ids[sub], mask[sub], pool="cls")
    np.testing.assert_allclose(again, out[sub], rtol=0, atol=2e-4)


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
    finally:
        if created:
            dist.destroy_process_group()
