"""HBM-resident dense indexes over the C ABI (include/mrag.h).

``DenseIndex`` is the corpus-scale form of the reference's dense scoring
(app/modules/retrieval/retrieval_backend.py:245,371-372: cosine every candidate, sort
descending, truncate): rows are L2-normalised and rounded once to fp16/bf16, kept in
HBM, and every search is one fused MFMA similarity + top-k pass.  Tie-break: score
descending, then row ascending.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from . import _native as N

_NP_DT = {np.dtype(np.float32): N.MRAG_F32, np.dtype(np.float16): N.MRAG_F16, np.dtype(np.float64): N.MRAG_F64}


def _is_torch(x) -> bool:
    return type(x).__module__.split(".")[0] == "torch"


def _torch_dtype_code(t) -> int:
    import torch
    return {torch.float32: N.MRAG_F32, torch.float16: N.MRAG_F16, torch.bfloat16: N.MRAG_BF16,
            torch.float64: N.MRAG_F64}[t.dtype]


def _as_buffer(x, dim: int):
    """-> (keepalive, pointer, n_rows, dtype_code, is_device)."""
    if _is_torch(x):
        if x.dim() != 2 or x.shape[1] != dim:
            raise ValueError(f"expected [n, {dim}] rows, got {tuple(x.shape)}")
        x = x.contiguous()
        if x.is_cuda:
            return x, x.data_ptr(), x.shape[0], _torch_dtype_code(x), 1
        if x.dtype.is_floating_point and str(x.dtype) == "torch.bfloat16":
            return x, x.data_ptr(), x.shape[0], N.MRAG_BF16, 0
        x = x.numpy()
    a = np.ascontiguousarray(x)
    if a.ndim != 2 or a.shape[1] != dim:
        raise ValueError(f"expected [n, {dim}] rows, got {a.shape}")
    if a.dtype not in _NP_DT:
        a = a.astype(np.float32)
    return a, a.ctypes.data, a.shape[0], _NP_DT[a.dtype], 0


def _stream_ptr(stream) -> Optional[int]:
    if stream is None:
        return None
    return int(getattr(stream, "cuda_stream", stream))


class DenseIndex:
    """Brute-force cosine (or inner-product) top-k index on one GPU."""

    def __init__(self, dim: int, metric: str = "cosine", dtype: str = "f16", device: int = 0):
        self._lib = N.load()
        self.dim, self.metric, self.dtype, self.device = int(dim), metric, dtype, int(device)
        h = C.c_uint64(0)
        N.check(self._lib.mrag_index_create(self.dim, N.METRIC_COSINE if metric == "cosine" else N.METRIC_IP,
                                            N.MRAG_F16 if dtype == "f16" else N.MRAG_BF16, self.device, C.byref(h)))
        self._h = h

    # -- lifetime -------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.mrag_index_destroy(self._h)
            self._h = C.c_uint64(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self) -> int:
        n = C.c_int64(0)
        N.check(self._lib.mrag_index_size(self._h, C.byref(n)))
        return n.value

    # -- build ----------------------------------------------------------------------------
    def reserve(self, n_rows: int):
        N.check(self._lib.mrag_index_reserve(self._h, int(n_rows)))

    def set_id_base(self, base: int):
        N.check(self._lib.mrag_index_set_id_base(self._h, int(base)))

    def add(self, rows, normalize: Optional[bool] = None, stream=None):
        """Append rows ([n, dim] numpy / torch, host or device).  ``normalize`` defaults to
        True for the cosine metric; pass False for rows that are already normalised and
        rounded (they are then stored bit-for-bit when their dtype is the storage dtype)."""
        if normalize is None:
            normalize = self.metric == "cosine"
        keep, ptr, n, dt, is_dev = _as_buffer(rows, self.dim)
        if is_dev and stream is None:
            import torch
            stream = torch.cuda.current_stream(self.device)   # stream-ordered with the producer of `rows`
        N.check(self._lib.mrag_index_add(self._h, ptr, n, dt, int(bool(normalize)), is_dev, _stream_ptr(stream)))
        del keep

    def rows(self, row0: int = 0, n: Optional[int] = None) -> np.ndarray:
        n = len(self) - row0 if n is None else n
        out = np.empty((n, self.dim), dtype=np.float32)
        N.check(self._lib.mrag_index_get_rows(self._h, int(row0), int(n), out.ctypes.data, 0, None))
        return out

    # -- search ---------------------------------------------------------------------------
    def search(self, queries, k: int, normalize: Optional[bool] = None, stream=None, out=None
               ) -> Tuple["np.ndarray", "np.ndarray"]:
        """-> (scores [nq,k] fp32 descending, ids [nq,k] int64; (-inf, -1) past the corpus).

        Host queries give numpy results (the call returns when they have landed); CUDA
        tensors give CUDA tensors, written asynchronously on ``stream`` (default: torch's
        current stream).  ``out=(scores, ids)`` reuses caller tensors."""
        if normalize is None:
            normalize = self.metric == "cosine"
        keep, ptr, nq, dt, is_dev = _as_buffer(queries, self.dim)
        if is_dev:
            import torch
            if stream is None:
                stream = torch.cuda.current_stream(self.device)
            if out is None:
                sc = torch.empty((nq, k), dtype=torch.float32, device=keep.device)
                ids = torch.empty((nq, k), dtype=torch.int64, device=keep.device)
            else:
                sc, ids = out
            N.check(self._lib.mrag_index_search(self._h, ptr, nq, dt, int(bool(normalize)), 1, int(k),
                                                sc.data_ptr(), ids.data_ptr(), 1, _stream_ptr(stream)))
            return sc, ids
        sc = np.empty((nq, k), dtype=np.float32)
        ids = np.empty((nq, k), dtype=np.int64)
        N.check(self._lib.mrag_index_search(self._h, ptr, nq, dt, int(bool(normalize)), 0, int(k),
                                            sc.ctypes.data, ids.ctypes.data, 0, _stream_ptr(stream)))
        return sc, ids

    def max_k(self, nq: int = 1) -> int:
        """Largest ``k`` :meth:`search` serves (``mrag_index_max_k``); above it the library raises
        MRAG_ERR_UNSUPPORTED, it never clamps."""
        out = C.c_int(0)
        N.check(self._lib.mrag_index_max_k(int(nq), C.byref(out)))
        return out.value

    def last_wide_redone(self) -> int:
        """Queries the last 64 < k <= 256 batch search redid through the streaming kernel (``mrag_index_last_wide_redone``)."""
        out = C.c_int64(0)
        N.check(self._lib.mrag_index_last_wide_redone(C.byref(out)))
        return out.value

    def score_rows(self, query, row_ids, normalize: Optional[bool] = None) -> np.ndarray:
        """<query, stored row i> for every listed row (``mrag_index_score_rows``): the re-ranker's
        candidate lookup by row id.  Ids outside the index score 0.0."""
        if normalize is None:
            normalize = self.metric == "cosine"
        q = np.ascontiguousarray(np.asarray(query, dtype=np.float32).reshape(-1))
        if q.shape[0] != self.dim:
            raise ValueError(f"query has dim {q.shape[0]}, index has {self.dim}")
        ids = np.ascontiguousarray(row_ids, dtype=np.int64).reshape(-1)
        out = np.empty(ids.shape[0], dtype=np.float32)
        N.check(self._lib.mrag_index_score_rows(self._h, q.ctypes.data, N.MRAG_F32, int(bool(normalize)), ids.ctypes.data,
                                                ids.shape[0], out.ctypes.data, None))
        return out

    def stored_bits(self) -> np.ndarray:
        """The stored rows as their 16-bit patterns ([n, dim] uint16, fp16 or bf16 per ``dtype``):
        what the embedding cache keeps, so that a warm start re-adds the exact bits K1 produced."""
        r = self.rows()
        if self.dtype == "f16":
            return r.astype(np.float16).view(np.uint16)       # exact: every stored value is an fp16 value
        return (r.view(np.uint32) >> 16).astype(np.uint16)     # exact: low 16 bits of a widened bf16 are zero

    def add_stored_bits(self, bits: np.ndarray):
        """Append rows given as the storage dtype's bit patterns, verbatim (no normalise, no rounding)."""
        bits = np.ascontiguousarray(bits, dtype=np.uint16)
        if self.dtype == "f16":
            self.add(bits.view(np.float16), normalize=False)
        else:
            import torch
            self.add(torch.from_numpy(bits.view(np.int16)).view(torch.bfloat16), normalize=False)

    def measure_clock(self, enable: bool = True):
        """Ask the batch kernel's workgroup 0 to stamp s_memtime / s_memrealtime (``mrag_index_measure_clock``)."""
        N.check(self._lib.mrag_index_measure_clock(self._h, int(bool(enable))))

    def last_clock_ghz(self) -> float:
        """Held shader clock during the last batch-kernel search (``mrag_index_last_clock``)."""
        g = C.c_float(0)
        N.check(self._lib.mrag_index_last_clock(self._h, C.byref(g)))
        return g.value

    def last_timing_ms(self) -> Tuple[float, float]:
        """(fused similarity+top-k kernel ms, whole search ms) of the last search, from
        hipEvents recorded on the launch stream."""
        g, t = C.c_float(0), C.c_float(0)
        N.check(self._lib.mrag_index_last_timing(self._h, C.byref(g), C.byref(t)))
        return g.value, t.value


def cosine_f64(query, cands, device: int = 0) -> np.ndarray:
    """One query against n candidates, fp64 on the GPU -- the arithmetic of
    ``DenseReranker._cosine`` (retrieval_backend.py:192-197) for every candidate at once."""
    q = np.ascontiguousarray(query, dtype=np.float64).reshape(-1)
    c = np.ascontiguousarray(cands, dtype=np.float64)
    if c.ndim != 2 or c.shape[1] != q.shape[0]:
        raise ValueError("dimension mismatch")
    out = np.empty(c.shape[0], dtype=np.float64)
    N.check(N.load().mrag_cosine_f64(device, q.ctypes.data, c.ctypes.data, c.shape[0], c.shape[1],
                                     out.ctypes.data, 0, None))
    return out


def topk_merge(scores: np.ndarray, ids: np.ndarray, nthreads: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    """Host merge of per-shard partial top-k: scores/ids [nparts, nq, k] -> [nq, k]
    ordered by (score desc, id asc).  Runs on the CPU (C++ threads)."""
    s = np.ascontiguousarray(scores, dtype=np.float32)
    i = np.ascontiguousarray(ids, dtype=np.int64)
    if s.ndim != 3 or s.shape != i.shape:
        raise ValueError("expected matching [nparts, nq, k] arrays")
    nparts, nq, k = s.shape
    os_, oi = np.empty((nq, k), dtype=np.float32), np.empty((nq, k), dtype=np.int64)
    N.check(N.load().mrag_topk_merge(s.ctypes.data, i.ctypes.data, nparts, nq, k, os_.ctypes.data,
                                     oi.ctypes.data, int(nthreads)))
    return os_, oi


def topk_merge_device(scores, ids, out_scores=None, out_ids=None):
    """The same merge on the GPU (``mrag_topk_merge_device``) for gathered partial top-k that
    already sits in HBM: ``scores`` float32 / ``ids`` int64 CUDA tensors [nparts, nq, k] (each
    part sorted, empties last) -> CUDA tensors [nq, k].  Asynchronous on the current stream."""
    import torch
    if not (torch.is_tensor(scores) and scores.is_cuda and torch.is_tensor(ids) and ids.is_cuda):
        raise ValueError("topk_merge_device expects CUDA tensors (use topk_merge for host arrays)")
    if scores.dim() != 3 or scores.shape != ids.shape or scores.dtype != torch.float32 or ids.dtype != torch.int64:
        raise ValueError("expected matching [nparts, nq, k] float32 / int64 tensors")
    scores, ids = scores.contiguous(), ids.contiguous()
    nparts, nq, k = scores.shape
    dev = scores.device
    if out_scores is None:
        out_scores = torch.empty((nq, k), dtype=torch.float32, device=dev)
    if out_ids is None:
        out_ids = torch.empty((nq, k), dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    N.check(N.load().mrag_topk_merge_device(dev.index or 0, scores.data_ptr(), ids.data_ptr(), nparts, nq, k,
                                            out_scores.data_ptr(), out_ids.data_ptr(), stream))
    return out_scores, out_ids


def pack_partial_device(scores, ids, id_base: int, out_words=None, bad_flag=None):
    """[Q,k] fp32 scores + int64 global ids (CUDA) -> the exchange's packed int64 words in ONE kernel (``mrag_pack_partial_device``;
    the torch form in ``sharded.pack_partial`` is ~8 elementwise kernels).  ``bad_flag``: int32 CUDA tensor [1], OR'ed with 1 when
    an id does not fit 32 bits above ``id_base``.  Asynchronous on the current stream."""
    import torch
    scores, ids = scores.contiguous(), ids.contiguous()
    if out_words is None:
        out_words = torch.empty(scores.shape, dtype=torch.int64, device=scores.device)
    stream = torch.cuda.current_stream(scores.device).cuda_stream
    N.check(N.load().mrag_pack_partial_device(scores.device.index or 0, scores.data_ptr(), ids.data_ptr(), scores.numel(), int(id_base),
                                              out_words.data_ptr(), bad_flag.data_ptr() if bad_flag is not None else None, stream))
    return out_words


def topk_merge_packed_device(words, bases, out_scores=None, out_ids=None):
    """Merge of the GATHERED packed words [nparts, nq, k] (int64 CUDA) with the parts' first global rows ``bases`` (int64 CUDA
    [nparts]) -> CUDA tensors [nq, k] (``mrag_topk_merge_packed_device``): no unpack pass over the gathered buffer."""
    import torch
    words = words.contiguous()
    nparts, nq, k = words.shape
    dev = words.device
    if out_scores is None:
        out_scores = torch.empty((nq, k), dtype=torch.float32, device=dev)
    if out_ids is None:
        out_ids = torch.empty((nq, k), dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    N.check(N.load().mrag_topk_merge_packed_device(dev.index or 0, words.data_ptr(), bases.contiguous().data_ptr(), nparts, nq, k,
                                                   out_scores.data_ptr(), out_ids.data_ptr(), stream))
    return out_scores, out_ids


class IVFFlatIndex:
    """IVF-flat index on one GPU (BASELINE.json config 5): nlist spherical-k-means lists, every
    search probes the ``nprobe`` best lists and scans them exactly.  Same id / tie-break
    conventions as :class:`DenseIndex`."""

    def __init__(self, dim: int, nlist: int, metric: str = "cosine", dtype: str = "f16", device: int = 0):
        self._lib = N.load()
        self.dim, self.nlist, self.metric, self.dtype, self.device = int(dim), int(nlist), metric, dtype, int(device)
        h = C.c_uint64(0)
        N.check(self._lib.mrag_ivf_create(self.dim, self.nlist, N.METRIC_COSINE if metric == "cosine" else N.METRIC_IP,
                                          N.MRAG_F16 if dtype == "f16" else N.MRAG_BF16, self.device, C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.mrag_ivf_destroy(self._h)
            self._h = C.c_uint64(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self) -> int:
        n = C.c_int64(0)
        N.check(self._lib.mrag_ivf_size(self._h, C.byref(n)))
        return n.value

    def _norm(self, normalize):
        return int(bool(self.metric == "cosine" if normalize is None else normalize))

    def train(self, rows, iters: int = 10, seed: int = 0, normalize: Optional[bool] = None):
        keep, ptr, n, dt, is_dev = _as_buffer(rows, self.dim)
        N.check(self._lib.mrag_ivf_train(self._h, ptr, n, dt, self._norm(normalize), is_dev, int(iters), int(seed), None))

    def set_centroids(self, centroids, normalize: Optional[bool] = None):
        keep, ptr, n, dt, is_dev = _as_buffer(centroids, self.dim)
        if n != self.nlist:
            raise ValueError(f"expected {self.nlist} centroids, got {n}")
        N.check(self._lib.mrag_ivf_set_centroids(self._h, ptr, dt, self._norm(normalize), is_dev, None))

    def centroids(self) -> np.ndarray:
        out = np.empty((self.nlist, self.dim), dtype=np.float32)
        N.check(self._lib.mrag_ivf_get_centroids(self._h, out.ctypes.data, 0, None))
        return out

    def add(self, rows, normalize: Optional[bool] = None):
        keep, ptr, n, dt, is_dev = _as_buffer(rows, self.dim)
        N.check(self._lib.mrag_ivf_add(self._h, ptr, n, dt, self._norm(normalize), is_dev, None))

    def set_id_base(self, base: int):
        N.check(self._lib.mrag_ivf_set_id_base(self._h, int(base)))

    def assignments(self) -> np.ndarray:
        out = np.empty(len(self), dtype=np.int32)
        N.check(self._lib.mrag_ivf_get_assignments(self._h, out.ctypes.data, 0, None))
        return out

    def last_timing(self) -> dict:
        """Device timing of the last search: list-scan kernel ms, whole-search ms, rows the scan streamed,
        workgroups (``mrag_ivf_last_timing``)."""
        g, t, rows, nwg = C.c_float(0), C.c_float(0), C.c_int64(0), C.c_int(0)
        N.check(self._lib.mrag_ivf_last_timing(self._h, C.byref(g), C.byref(t), C.byref(rows), C.byref(nwg)))
        return {"scan_ms": g.value, "total_ms": t.value, "scanned_rows": rows.value, "n_wg": nwg.value}

    def search(self, queries, k: int, nprobe: int, normalize: Optional[bool] = None):
        """-> (scores [nq,k], ids [nq,k]); host queries give numpy arrays (complete on return), CUDA tensors give CUDA
        tensors filled asynchronously on the current torch stream, like any torch op.  ``nprobe``: 1..256, or ``nlist``
        (exhaustive)."""
        keep, ptr, nq, dt, is_dev = _as_buffer(queries, self.dim)
        if is_dev:
            import torch
            sc = torch.empty((nq, k), dtype=torch.float32, device=keep.device)
            ids = torch.empty((nq, k), dtype=torch.int64, device=keep.device)
            stream = torch.cuda.current_stream(self.device)
            N.check(self._lib.mrag_ivf_search(self._h, ptr, nq, dt, self._norm(normalize), 1, int(nprobe), int(k),
                                              sc.data_ptr(), ids.data_ptr(), 1, _stream_ptr(stream)))
            return sc, ids
        sc = np.empty((nq, k), dtype=np.float32)
        ids = np.empty((nq, k), dtype=np.int64)
        N.check(self._lib.mrag_ivf_search(self._h, ptr, nq, dt, self._norm(normalize), is_dev, int(nprobe), int(k),
                                          sc.ctypes.data, ids.ctypes.data, 0, None))
        return sc, ids
