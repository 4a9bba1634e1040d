"""Row-sharded corpus search: one process per GPU, each rank owns a
contiguous block of corpus rows, runs the fused similarity + top-k locally, then ONE
all-gather of the per-shard (score, global id) partial top-k (RCCL over xGMI when the
process group is ``nccl``; ``gloo`` in the CPU tests) followed by the merge: on the host
(``mrag_topk_merge``, C++ threads) for host tensors, and by default on the device
(``mrag_topk_merge_device``) when the gathered buffers are in HBM -- the merged [Q,k] result is
then the only thing that crosses PCIe (C4 at 8 GPUs: 1.2 MB instead of 9.6 MB + 0.5 ms of host
merge per step).  Result = the single-GPU result: ids are global, tie-break (score desc, id asc).

The exchange is ONE collective of 8-byte (fp32 score bits, int32 shard-local row) words -- Q*k*8 bytes per
rank (C4: 10 000 x 10 -> 800 KB, 6.4 MB gathered at 8 GPUs) -- latency-bound, so a direct all-gather on the
fully connected xGMI mesh, no ring tuning, no all-reduce; global ids are rebuilt from the rank's row base,
which every rank knows (``shard_bounds``).

``ShardedIVFIndex`` is the same for IVF-flat (BASELINE config 5): rows sharded, centroids trained once and
replicated, so every GPU holds 1/N of every list and does equal work for any probe set.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import numpy as np

from .index import topk_merge, topk_merge_device


def _host_pair(bufs: dict, q: int, k: int):
    """Pinned host result buffers, two sets used alternately: the arrays a call returns are views of
    pinned memory and stay valid until the call AFTER the next one (no per-call allocation or copy)."""
    import torch
    if bufs.get("hkey") != (q, k):
        bufs["hkey"] = (q, k)
        bufs["hring"] = [(torch.empty((q, k), dtype=torch.float32, pin_memory=True),
                          torch.empty((q, k), dtype=torch.int64, pin_memory=True)) for _ in range(2)]
        bufs["hturn"] = 0
    bufs["hturn"] ^= 1
    return bufs["hring"][bufs["hturn"]]


def shard_bounds(n_total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous row block of ``rank``: [lo, hi)."""
    return (n_total * rank) // world, (n_total * (rank + 1)) // world


def gather_and_merge(local_scores, local_ids, group=None, nthreads: int = 0, bufs: Optional[dict] = None,
                     merge: str = "auto", force_collective: bool = False, id_bases=None) -> Tuple[np.ndarray, np.ndarray]:
    """All-gather the per-shard partial top-k and merge.  ``merge``: "host" (mrag_topk_merge),
    "device" (mrag_topk_merge_device, CUDA tensors only) or "auto" (device when it applies).

    ``local_scores`` [Q,k] float32 and ``local_ids`` [Q,k] int64 are torch tensors on the
    backend's device (CUDA for nccl, CPU for gloo).  Every rank returns the full merged
    (scores [Q,k] float32, ids [Q,k] int64) as numpy arrays.  ``bufs`` (a dict the caller
    keeps) caches the gather / pinned staging buffers across calls; with CUDA inputs the returned
    arrays are views of pinned buffers that stay valid until the call after the next one.
    ``id_bases``: first global row of every rank's shard (needed when more than one rank takes part)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    q, k = local_scores.shape
    bufs = bufs if bufs is not None else {}
    if world == 1 and not force_collective:   # (force_collective: run the all-gather + merge even alone -- tests)
        if not local_scores.is_cuda:
            return (local_scores.detach().numpy().astype(np.float32, copy=False),
                    local_ids.detach().numpy().astype(np.int64, copy=False))
        # one pinned D2H per array + one stream sync (pageable .cpu() costs two blocking staged copies)
        hs, hi = _host_pair(bufs, q, k)
        hs.copy_(local_scores, non_blocking=True)
        hi.copy_(local_ids, non_blocking=True)
        torch.cuda.current_stream(local_scores.device).synchronize()
        return hs.numpy(), hi.numpy()
    key = (world, q, k, str(local_scores.device))
    if bufs.get("key") != key:
        bufs.clear()
        bufs["key"] = key
        bufs["gw"] = torch.empty((world, q, k), dtype=torch.int64, device=local_scores.device)
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    bases = bufs.get("bases")
    if bases is None or bases.shape[0] != world:
        if id_bases is None:
            raise ValueError("gather_and_merge needs id_bases (first global row of every rank) for N > 1")
        bases = bufs["bases"] = torch.as_tensor(list(id_bases), dtype=torch.int64, device=local_scores.device).view(world, 1, 1)
    gw = bufs["gw"]
    # ONE collective: (score bits << 32) | shard-local row (0xFFFFFFFF = empty slot); output = the inputs
    # concatenated along dim 0 (the layout both nccl and gloo accept)
    dist.all_gather_into_tensor(gw.view(world * q, k), pack_partial(local_scores, local_ids, int(bases[rank])), group=group)
    gs, gi = unpack_partial(gw, bases)
    return merge_gathered(gs, gi, bufs, merge=merge, nthreads=nthreads)


def pack_partial(scores, ids, id_base: int):
    """[Q,k] fp32 scores + int64 GLOBAL ids (-1 = empty) -> int64 words: score bits in the high half, the
    shard-local row (id - id_base, 32 bits) in the low half."""
    import torch
    local = torch.where(ids >= 0, ids - id_base, torch.full_like(ids, 0xFFFFFFFF))
    return (scores.contiguous().view(torch.int32).to(torch.int64) << 32) | (local & 0xFFFFFFFF)


def unpack_partial(words, bases):
    """Inverse of :func:`pack_partial` for gathered words [world,Q,k]; ``bases`` [world,1,1] int64."""
    import torch
    scores = (words >> 32).to(torch.int32).view(torch.float32)
    low = words & 0xFFFFFFFF
    ids = torch.where(low == 0xFFFFFFFF, torch.full_like(low, -1), low + bases)
    return scores, ids


def merge_gathered(gs, gi, bufs: dict, merge: str = "auto", nthreads: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    """Merge gathered partial top-k ``gs`` / ``gi`` [world, Q, k] (torch tensors, CUDA or CPU) into host
    arrays [Q, k].  CUDA input: merged on the device, then ONE pinned D2H of the result ("device" /
    "auto"), or pinned D2H of everything + the host merge ("host")."""
    import torch
    world, q, k = gs.shape
    if gs.is_cuda:
        on_device = merge == "device" or (merge == "auto" and world <= 64 and world * k <= 2048)
        if on_device:
            if bufs.get("mkey") != (q, k):
                bufs["mkey"] = (q, k)
                bufs["ms"] = torch.empty((q, k), dtype=torch.float32, device=gs.device)
                bufs["mi"] = torch.empty((q, k), dtype=torch.int64, device=gs.device)
            topk_merge_device(gs, gi, bufs["ms"], bufs["mi"])
            hs, hi = _host_pair(bufs, q, k)
            hs.copy_(bufs["ms"], non_blocking=True)
            hi.copy_(bufs["mi"], non_blocking=True)
            torch.cuda.current_stream(gs.device).synchronize()
            return hs.numpy(), hi.numpy()
        if "hs" not in bufs:
            bufs["hs"] = torch.empty((world, q, k), dtype=torch.float32, pin_memory=True)
            bufs["hi"] = torch.empty((world, q, k), dtype=torch.int64, pin_memory=True)
        hs, hi = bufs["hs"], bufs["hi"]
        hs.copy_(gs, non_blocking=True)
        hi.copy_(gi, non_blocking=True)
        torch.cuda.current_stream(gs.device).synchronize()
        return topk_merge(hs.numpy(), hi.numpy(), nthreads)
    if merge == "device":
        raise ValueError("merge='device' needs CUDA tensors")
    return topk_merge(gs.numpy(), gi.numpy(), nthreads)


class ShardedDenseIndex:
    """One rank's view of a row-sharded corpus.

    ``local_search(queries, k) -> (scores, ids)`` defaults to a :class:`DenseIndex` on this
    rank's GPU with ``id_base`` = the shard's first global row; tests inject a CPU searcher
    to exercise the exchange + merge on ``gloo``."""

    def __init__(self, dim: int, n_total: int, rank: int, world: int, device: int = 0, group=None,
                 dtype: str = "f16", metric: str = "cosine",
                 local_search: Optional[Callable] = None):
        self.dim, self.n_total, self.rank, self.world, self.group = dim, n_total, rank, world, group
        self.lo, self.hi = shard_bounds(n_total, world, rank)
        self._bufs: dict = {}
        self.index = None
        if local_search is None:
            from .index import DenseIndex
            self.index = DenseIndex(dim, metric=metric, dtype=dtype, device=device)
            self.index.set_id_base(self.lo)
            self.index.reserve(self.hi - self.lo)
            local_search = self.index.search
        self._local_search = local_search

    def add_local(self, rows, normalize=None):
        """Append rows of THIS shard (global rows lo + len(index) ...)."""
        self.index.add(rows, normalize=normalize)
        if len(self.index) > self.hi - self.lo:
            raise ValueError("more rows than this shard owns")

    def search(self, queries, k: int, nthreads: int = 0, merge: str = "auto", force_collective: bool = False,
               **kw) -> Tuple[np.ndarray, np.ndarray]:
        import torch
        if self.index is not None and torch.is_tensor(queries) and queries.is_cuda and "out" not in kw:
            # reuse the device result tensors across calls (they are consumed by the exchange below)
            key = (int(queries.shape[0]), int(k), queries.device)
            if self._bufs.get("okey") != key:
                self._bufs["okey"] = key
                self._bufs["osc"] = torch.empty((key[0], key[1]), dtype=torch.float32, device=queries.device)
                self._bufs["oid"] = torch.empty((key[0], key[1]), dtype=torch.int64, device=queries.device)
            kw = dict(kw, out=(self._bufs["osc"], self._bufs["oid"]))
        sc, ids = self._local_search(queries, k, **kw)
        if not torch.is_tensor(sc):
            sc, ids = torch.from_numpy(np.ascontiguousarray(sc)), torch.from_numpy(np.ascontiguousarray(ids))
        return gather_and_merge(sc, ids, group=self.group, nthreads=nthreads, bufs=self._bufs, merge=merge,
                                force_collective=force_collective,
                                id_bases=[shard_bounds(self.n_total, self.world, r)[0] for r in range(self.world)])


class ShardedIVFIndex:
    """One rank's view of a row-sharded IVF-flat index (BASELINE config 5): rank r holds rows
    ``shard_bounds(n_total, world, r)`` of EVERY list; the centroids are trained once (rank 0) and broadcast,
    so all ranks probe the same lists and scan equal shares of them; the per-shard top-k meets in the same
    one-collective exchange + merge as the brute-force index.  Result = the single-GPU IVF result for the
    same centroids.

    ``local_search(queries, k, nprobe) -> (scores, ids)`` defaults to an :class:`IVFFlatIndex` on this rank's
    GPU; tests inject a CPU searcher to exercise the exchange on ``gloo``."""

    def __init__(self, dim: int, nlist: int, n_total: int, rank: int, world: int, device: int = 0, group=None,
                 dtype: str = "f16", metric: str = "cosine", local_search: Optional[Callable] = None):
        self.dim, self.nlist, self.n_total, self.rank, self.world, self.group = dim, nlist, n_total, rank, world, group
        self.lo, self.hi = shard_bounds(n_total, world, rank)
        self._bufs: dict = {}
        self.index = None
        if local_search is None:
            from .index import IVFFlatIndex
            self.index = IVFFlatIndex(dim, nlist, metric=metric, dtype=dtype, device=device)
            self.index.set_id_base(self.lo)
            local_search = self.index.search
        self._local_search = local_search

    def train(self, sample_rows, iters: int = 10, seed: int = 0, normalize=None):
        """k-means on rank 0 (``sample_rows`` is ignored elsewhere), centroids broadcast to every rank."""
        import torch
        import torch.distributed as dist
        cen = None
        if self.rank == 0:
            self.index.train(sample_rows, iters=iters, seed=seed, normalize=normalize)
            cen = self.index.centroids()
        if self.world > 1:
            dev = torch.device("cuda", self.index.device) if dist.get_backend(self.group) == "nccl" else torch.device("cpu")
            t = torch.from_numpy(cen).to(dev) if self.rank == 0 else torch.empty((self.nlist, self.dim), dtype=torch.float32, device=dev)
            dist.broadcast(t, src=0, group=self.group)
            if self.rank != 0:
                self.index.set_centroids(t.cpu().numpy(), normalize=False)     # already normalised + rounded
        return self

    def set_centroids(self, centroids, normalize=None):
        self.index.set_centroids(centroids, normalize=normalize)

    def add_local(self, rows, normalize=None):
        self.index.add(rows, normalize=normalize)
        if len(self.index) > self.hi - self.lo:
            raise ValueError("more rows than this shard owns")

    def search(self, queries, k: int, nprobe: int, nthreads: int = 0, merge: str = "auto",
               force_collective: bool = False) -> Tuple[np.ndarray, np.ndarray]:
        import torch
        sc, ids = self._local_search(queries, k, nprobe)
        if not torch.is_tensor(sc):
            sc, ids = torch.from_numpy(np.ascontiguousarray(sc)), torch.from_numpy(np.ascontiguousarray(ids))
        return gather_and_merge(sc, ids, group=self.group, nthreads=nthreads, bufs=self._bufs, merge=merge,
                                force_collective=force_collective,
                                id_bases=[shard_bounds(self.n_total, self.world, r)[0] for r in range(self.world)])
