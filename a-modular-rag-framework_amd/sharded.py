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

from .index import pack_partial_device, topk_merge, topk_merge_device, topk_merge_packed_device


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


class _Phases:
    """Per-phase milliseconds of one exchange.  CUDA inputs: events on the current stream, read after the call's
    final stream sync (no extra sync); host inputs: ``time.perf_counter``."""

    def __init__(self, cuda: bool, device=None):
        self.cuda, self.device, self.marks = cuda, device, []

    def mark(self, name: str):
        if self.cuda:
            import torch
            e = torch.cuda.Event(enable_timing=True)
            e.record(torch.cuda.current_stream(self.device))
            self.marks.append((name, e))
        else:
            import time
            self.marks.append((name, time.perf_counter()))

    def result(self) -> dict:
        out = {}
        for (_, a), (name, b) in zip(self.marks, self.marks[1:]):
            ms = a.elapsed_time(b) if self.cuda else (b - a) * 1e3
            out[name] = out.get(name, 0.0) + float(ms)
        return out


def _bases_tensor(bufs: dict, id_bases, world: int, device):
    """[world,1,1] int64 tensor of the ranks' first global rows, cached per VALUE of ``id_bases`` (another
    ``n_total`` or shard layout with the same world / q / k must never reuse the previous bases)."""
    import torch
    if id_bases is None:
        raise ValueError("gather_and_merge needs id_bases (first global row of every rank) for N > 1")
    key = tuple(int(b) for b in id_bases)
    if len(key) != world:
        raise ValueError(f"id_bases has {len(key)} entries for a world of {world}")
    if bufs.get("bases_key") != key:
        bufs["bases_key"] = key
        bufs["bases"] = torch.as_tensor(list(key), dtype=torch.int64, device=device).view(world, 1, 1)
    return key, bufs["bases"]


def _device_merge_applies(merge: str, world: int, k: int) -> bool:
    return merge == "device" or (merge == "auto" and world <= 64 and world * k <= 2048)


def _raise_if_bad(bad_host) -> None:
    if int(bad_host) != 0:
        raise ValueError("gather_and_merge: a local id lies outside [id_base, id_base + 2^32 - 1) -- the packed exchange "
                         "carries shard-local rows in 32 bits (is the index's id_base set to the shard's first row?)")


def gather_and_merge(local_scores, local_ids, group=None, nthreads: int = 0, bufs: Optional[dict] = None,
                     merge: str = "auto", force_collective: bool = False, id_bases=None,
                     phases: Optional[dict] = None) -> Tuple[np.ndarray, np.ndarray]:
    """All-gather the per-shard partial top-k and merge.  ``merge``: "host" (mrag_topk_merge),
    "device" (mrag_topk_merge_device, CUDA tensors only) or "auto" (device when it applies).

    ``local_scores`` [Q,k] float32 and ``local_ids`` [Q,k] int64 are torch tensors on the
    backend's device (CUDA for nccl, CPU for gloo).  Every rank returns the full merged
    (scores [Q,k] float32, ids [Q,k] int64) as numpy arrays.  ``bufs`` (a dict the caller
    keeps) caches the gather / pinned staging buffers across calls; with CUDA inputs the returned
    arrays are views of pinned buffers that stay valid until the call after the next one.
    ``id_bases``: first global row of every rank's shard (needed when more than one rank takes part);
    ids must lie in [id_base, id_base + 2^32 - 1) of their rank -- anything else raises, it never wraps.
    ``phases`` (a dict) receives pack_ms / gather_ms / merge_ms / d2h_ms of this call."""
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
        ph = _Phases(True, local_scores.device) if phases is not None else None
        if ph: ph.mark("start")
        hs, hi = _host_pair(bufs, q, k)
        hs.copy_(local_scores, non_blocking=True)
        hi.copy_(local_ids, non_blocking=True)
        if ph: ph.mark("d2h_ms")
        torch.cuda.current_stream(local_scores.device).synchronize()
        if ph: phases.update(ph.result())
        return hs.numpy(), hi.numpy()
    key = (world, q, k, str(local_scores.device))
    if bufs.get("key") != key:
        bufs.clear()
        bufs["key"] = key
        bufs["gw"] = torch.empty((world, q, k), dtype=torch.int64, device=local_scores.device)
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    base_list, bases = _bases_tensor(bufs, id_bases, world, local_scores.device)
    gw = bufs["gw"]
    ph = _Phases(local_scores.is_cuda, local_scores.device) if phases is not None else None
    if ph: ph.mark("start")
    # ONE collective: (score bits << 32) | shard-local row (0xFFFFFFFF = empty slot); this rank's base comes from
    # the HOST list (reading bases[rank] back from the device would be a blocking sync inside the exchange);
    # output = the inputs concatenated along dim 0 (the layout both nccl and gloo accept)
    fused = local_scores.is_cuda and _device_merge_applies(merge, world, k)
    if fused:
        # CUDA + device merge: one pack kernel, the collective, one merge kernel that unpacks the gathered words on its way
        # into LDS (the torch forms below are ~8 + ~6 elementwise kernels per step: 0.1-0.15 ms at C4's 10 000 x 10)
        if "wbuf" not in bufs:
            bufs["wbuf"] = torch.empty((q, k), dtype=torch.int64, device=local_scores.device)
            bufs["badi"] = torch.zeros(1, dtype=torch.int32, device=local_scores.device)
        bufs["badi"].zero_()
        words = pack_partial_device(local_scores, local_ids, base_list[rank], out_words=bufs["wbuf"], bad_flag=bufs["badi"])
        if ph: ph.mark("pack_ms")
        dist.all_gather_into_tensor(gw.view(world * q, k), words, group=group)
        if ph: ph.mark("gather_ms")
        out = merge_gathered(None, None, bufs, merge="device", nthreads=nthreads, bad=bufs["badi"], phases=ph,
                             packed=(gw, bases.view(world)))
        if ph: phases.update(ph.result())
        return out
    words, bad = pack_partial(local_scores, local_ids, base_list[rank], return_bad=True)
    if ph: ph.mark("pack_ms")
    dist.all_gather_into_tensor(gw.view(world * q, k), words, group=group)
    if ph: ph.mark("gather_ms")
    if not bad.is_cuda:
        _raise_if_bad(bad.item())
        bad = None
    gs, gi = unpack_partial(gw, bases)
    out = merge_gathered(gs, gi, bufs, merge=merge, nthreads=nthreads, bad=bad, phases=ph)
    if ph: phases.update(ph.result())
    return out


def pack_partial(scores, ids, id_base: int, return_bad: bool = False):
    """[Q,k] fp32 scores + int64 GLOBAL ids (-1 = empty) -> int64 words: score bits in the high half, the
    shard-local row (id - id_base, 32 bits) in the low half.  ``return_bad``: also a 1-element int64 tensor,
    non-zero when some non-empty id is outside [id_base, id_base + 0xFFFFFFFF).  The caller raises AFTER the
    collective (a rank that raised before it would leave the others waiting in it); on CUDA the flag rides along
    with the result copy and is checked after the call's one sync instead of forcing a sync inside the exchange."""
    import torch
    local = ids - id_base
    empty = ids < 0
    words = (scores.contiguous().view(torch.int32).to(torch.int64) << 32) | \
        (torch.where(empty, torch.full_like(ids, 0xFFFFFFFF), local) & 0xFFFFFFFF)
    if not return_bad:
        return words
    bad = (((local < 0) | (local >= 0xFFFFFFFF)) & ~empty).any().to(torch.int64).view(1)
    return words, bad


def unpack_partial(words, bases):
    """Inverse of :func:`pack_partial` for gathered words [world,Q,k]; ``bases`` [world,1,1] int64."""
    import torch
    scores = (words >> 32).to(torch.int32).view(torch.float32)
    low = words & 0xFFFFFFFF
    ids = torch.where(low == 0xFFFFFFFF, torch.full_like(low, -1), low + bases)
    return scores, ids


def merge_gathered(gs, gi, bufs: dict, merge: str = "auto", nthreads: int = 0, bad=None, phases=None, packed=None
                   ) -> Tuple[np.ndarray, np.ndarray]:
    """Merge gathered partial top-k ``gs`` / ``gi`` [world, Q, k] (torch tensors, CUDA or CPU) into host
    arrays [Q, k].  CUDA input: merged on the device, then ONE pinned D2H of the result ("device" /
    "auto"), or pinned D2H of everything + the host merge ("host").  ``bad``: the pack's range flag (a CUDA
    tensor rides along with the result copy and is checked after the one stream sync)."""
    import torch
    if packed is not None:                       # (gathered packed words [world,Q,k] + bases [world], CUDA: merged without unpacking)
        gs = packed[0]
    world, q, k = gs.shape
    if gs.is_cuda:
        hbad = None
        if bad is not None:
            hkey = "hbad32" if bad.dtype == torch.int32 else "hbad"
            if hkey not in bufs:
                bufs[hkey] = torch.zeros(1, dtype=bad.dtype, pin_memory=True)
            hbad = bufs[hkey]
        on_device = packed is not None or _device_merge_applies(merge, world, k)
        if on_device:
            if bufs.get("mkey") != (q, k):
                bufs["mkey"] = (q, k)
                bufs["ms"] = torch.empty((q, k), dtype=torch.float32, device=gs.device)
                bufs["mi"] = torch.empty((q, k), dtype=torch.int64, device=gs.device)
            if packed is not None:
                topk_merge_packed_device(packed[0], packed[1], bufs["ms"], bufs["mi"])
            else:
                topk_merge_device(gs, gi, bufs["ms"], bufs["mi"])
            if phases: phases.mark("merge_ms")
            hs, hi = _host_pair(bufs, q, k)
            hs.copy_(bufs["ms"], non_blocking=True)
            hi.copy_(bufs["mi"], non_blocking=True)
            if hbad is not None:
                hbad.copy_(bad, non_blocking=True)
            if phases: phases.mark("d2h_ms")
            torch.cuda.current_stream(gs.device).synchronize()
            if hbad is not None:
                _raise_if_bad(hbad.item())
            return hs.numpy(), hi.numpy()
        if "hs" not in bufs:
            bufs["hs"] = torch.empty((world, q, k), dtype=torch.float32, pin_memory=True)
            bufs["hi"] = torch.empty((world, q, k), dtype=torch.int64, pin_memory=True)
        hs, hi = bufs["hs"], bufs["hi"]
        hs.copy_(gs, non_blocking=True)
        hi.copy_(gi, non_blocking=True)
        if hbad is not None:
            hbad.copy_(bad, non_blocking=True)
        if phases: phases.mark("d2h_ms")
        torch.cuda.current_stream(gs.device).synchronize()
        if hbad is not None:
            _raise_if_bad(hbad.item())
        return topk_merge(hs.numpy(), hi.numpy(), nthreads)
    if merge == "device":
        raise ValueError("merge='device' needs CUDA tensors")
    out = topk_merge(gs.numpy(), gi.numpy(), nthreads)
    if phases: phases.mark("merge_ms")
    return out


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
        self.last_phases: dict = {}
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
               pipeline: bool = False, **kw) -> Tuple[np.ndarray, np.ndarray]:
        """Local search over this rank's rows, then the exchange + merge; every rank returns the full result.
        ``pipeline=True`` cuts the batch in two halves: the all-gather of half A is in flight (``async_op``, on the
        process group's own stream) while the local search of half B runs -- same result, bit for bit (every query's
        answer depends on its own rows of the batch only).  ``self.last_phases`` holds the call's per-phase
        milliseconds: local_ms, pack_ms, gather_ms (in pipeline mode: the part of the collective the local search
        did NOT cover), merge_ms, d2h_ms."""
        import torch
        import torch.distributed as dist
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        is_cuda = torch.is_tensor(queries) and queries.is_cuda
        id_bases = [shard_bounds(self.n_total, self.world, r)[0] for r in range(self.world)]
        nq = int(queries.shape[0])
        if self.index is not None and is_cuda and "out" not in kw:
            # reuse the device result tensors across calls (they are consumed by the exchange below)
            key = (nq, int(k), queries.device)
            if self._bufs.get("okey") != key:
                self._bufs["okey"] = key
                self._bufs["osc"] = torch.empty((key[0], key[1]), dtype=torch.float32, device=queries.device)
                self._bufs["oid"] = torch.empty((key[0], key[1]), dtype=torch.int64, device=queries.device)
            kw = dict(kw, out=(self._bufs["osc"], self._bufs["oid"]))
        if pipeline and (world > 1 or force_collective) and nq >= 2:
            return self._search_two_halves(queries, k, nthreads, merge, id_bases, world, is_cuda, kw)
        ph = _Phases(is_cuda, queries.device if is_cuda else None)
        ph.mark("start")
        sc, ids = self._local_search(queries, k, **kw)
        ph.mark("local_ms")
        if not torch.is_tensor(sc):
            sc, ids = torch.from_numpy(np.ascontiguousarray(sc)), torch.from_numpy(np.ascontiguousarray(ids))
        rest: dict = {}
        out = gather_and_merge(sc, ids, group=self.group, nthreads=nthreads, bufs=self._bufs, merge=merge,
                               force_collective=force_collective, id_bases=id_bases, phases=rest)
        self.last_phases = dict(ph.result(), **rest)
        return out

    def _search_two_halves(self, queries, k, nthreads, merge, id_bases, world, is_cuda, kw):
        import torch
        import torch.distributed as dist
        nq = int(queries.shape[0])
        h = nq // 2
        if h > 256:
            h = (h + 255) // 256 * 256          # cut on a query-tile boundary of the batch kernel
        cuts = [(0, h), (h, nq)]
        rank = dist.get_rank(self.group) if dist.is_initialized() else 0
        out_pair = kw.pop("out", None)
        ph = _Phases(is_cuda, queries.device if is_cuda else None)
        ph.mark("start")
        inflight = []
        on_device = is_cuda and _device_merge_applies(merge, world, k)
        for a, b in cuts:
            kwh = dict(kw, out=(out_pair[0][a:b], out_pair[1][a:b])) if out_pair is not None else kw
            sc, ids = self._local_search(queries[a:b], k, **kwh)
            ph.mark("local_ms")
            if not torch.is_tensor(sc):
                sc, ids = torch.from_numpy(np.ascontiguousarray(sc)), torch.from_numpy(np.ascontiguousarray(ids))
            bkey = ("gw2", a, b, world, int(k), str(sc.device))
            if bkey not in self._bufs:
                self._bufs[bkey] = torch.empty((world, b - a, k), dtype=torch.int64, device=sc.device)
            gw = self._bufs[bkey]
            base_list, bases = _bases_tensor(self._bufs, id_bases, world, sc.device)
            if on_device:                        # one pack kernel; the merge below unpacks the gathered words itself
                wkey = ("w2", a, b, int(k))
                if wkey not in self._bufs:
                    self._bufs[wkey] = (torch.empty((b - a, k), dtype=torch.int64, device=sc.device),
                                        torch.zeros(1, dtype=torch.int64, device=sc.device).view(torch.int32))
                wbuf, badi = self._bufs[wkey]
                badi.zero_()
                words = pack_partial_device(sc, ids, base_list[rank], out_words=wbuf, bad_flag=badi)
                bad = badi.view(torch.int64)
            else:
                words, bad = pack_partial(sc, ids, base_list[rank], return_bad=True)
            ph.mark("pack_ms")
            work = dist.all_gather_into_tensor(gw.view(world * (b - a), k), words, group=self.group, async_op=True)
            inflight.append((work, words, bad, gw, bases))      # (words stays referenced until the collective is done)
        parts = []
        if on_device:
            mkey = ("m2", nq, int(k))
            if mkey not in self._bufs:
                self._bufs[mkey] = (torch.empty((nq, k), dtype=torch.float32, device=queries.device),
                                    torch.empty((nq, k), dtype=torch.int64, device=queries.device))
            ms, mi = self._bufs[mkey]
        for (a, b), (work, words, bad, gw, bases) in zip(cuts, inflight):
            work.wait()                          # nccl: the current stream waits for the collective; gloo: the host does
            ph.mark("gather_ms")
            if not is_cuda and (a, b) == cuts[-1]:
                _raise_if_bad((inflight[0][2] | inflight[1][2]).item())
            if on_device:
                topk_merge_packed_device(gw, bases.view(world), ms[a:b], mi[a:b])
            else:
                gs, gi = unpack_partial(gw, bases)
                if is_cuda:
                    parts.append((gs.cpu().numpy(), gi.cpu().numpy()))
                else:
                    parts.append(topk_merge(gs.numpy(), gi.numpy(), nthreads))
            ph.mark("merge_ms")
        if is_cuda:
            if "hbad" not in self._bufs:
                self._bufs["hbad"] = torch.zeros(1, dtype=torch.int64, pin_memory=True)
            hbad = self._bufs["hbad"]
            hbad.copy_(inflight[0][2] | inflight[1][2], non_blocking=True)
            if on_device:
                hs, hi = _host_pair(self._bufs, nq, k)
                hs.copy_(ms, non_blocking=True)
                hi.copy_(mi, non_blocking=True)
            ph.mark("d2h_ms")
            torch.cuda.current_stream(queries.device).synchronize()
            _raise_if_bad(hbad.item())
            if on_device:
                out = hs.numpy(), hi.numpy()
            else:
                parts = [topk_merge(s_, i_, nthreads) for s_, i_ in parts]
                out = np.concatenate([p_[0] for p_ in parts]), np.concatenate([p_[1] for p_ in parts])
        else:
            out = np.concatenate([p_[0] for p_ in parts]), np.concatenate([p_[1] for p_ in parts])
        self.last_phases = ph.result()
        return out


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
