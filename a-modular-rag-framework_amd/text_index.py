"""BM25 text channel on the GPU behind the reference's own class shapes.

``HipBM25Index`` mirrors ``BM25LiteIndex`` (app/modules/retrieval/text_index.py:14-101: same constructor,
``N`` / ``avgdl`` / ``doc_lens`` / ``df`` / ``docs``, ``search(queries, top_k, alpha_merge)`` ->
``[(doc_idx, score)]``, ``doc_meta``); ``HipBM25TextSearcher`` mirrors ``BM25TextSearcher``
(retrieval_backend.py:102-128: ``search(queries=, top_k=)`` -> hits with the same ids and meta).

The index is built on the host like the reference builds it (same tokeniser, same counts) and uploaded once
as CSR postings; every search then runs on the device (``mrag_bm25_search``): scores are the reference's
fp64 sums bit for bit, order is (score desc, doc asc).  There is no CPU search path.
"""
from __future__ import annotations

import ctypes as C
import json
import math
import re
from collections import Counter
from pathlib import Path
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _native as N

_SPLIT = re.compile(r"[^a-zA-Z0-9]+")


def tokenize(text: str) -> List[str]:
    """text_index.py:10-11."""
    return [t for t in _SPLIT.split((text or "").lower()) if t]


def build_postings(rows: Sequence[Dict[str, Any]], k1: float, b: float) -> Dict[str, Any]:
    """Host side of the index build: the reference's counts (text_index.py:36-52) as CSR postings --
    ``indptr`` [n_terms+1], ``post_doc`` (ascending inside a term), ``post_tf`` -- plus
    ``doc_norm[d] = k1 * (1 - b + b * dl_d / avgdl)``, the length term of ``_score_doc`` (:66) evaluated with
    the reference's own expression."""
    term_id: Dict[str, int] = {}
    doc_lens: List[int] = []
    t_idx: List[int] = []
    d_idx: List[int] = []
    tfs: List[int] = []
    for d, obj in enumerate(rows):
        toks = tokenize(obj.get("text", ""))
        doc_lens.append(len(toks))
        for t, c in Counter(toks).items():
            t_idx.append(term_id.setdefault(t, len(term_id)))
            d_idx.append(d)
            tfs.append(c)
    n_docs, n_terms = len(rows), len(term_id)
    avgdl = (sum(doc_lens) / n_docs) if n_docs else 0.0
    t_arr = np.asarray(t_idx, dtype=np.int64)
    order = np.argsort(t_arr, kind="stable")                     # stable: docs stay ascending inside a term
    counts = np.bincount(t_arr, minlength=n_terms) if n_terms else np.zeros(0, dtype=np.int64)
    indptr = np.zeros(n_terms + 1, dtype=np.int64)
    np.cumsum(counts, out=indptr[1:])
    inv: List[Optional[str]] = [None] * n_terms
    for t, i in term_id.items():
        inv[i] = t
    avg = avgdl or 1.0
    return {"term_id": term_id, "doc_lens": doc_lens, "N": n_docs, "avgdl": avgdl,
            "df": {inv[i]: int(counts[i]) for i in range(n_terms)},
            "indptr": indptr, "post_doc": np.ascontiguousarray(np.asarray(d_idx, dtype=np.int32)[order]),
            "post_tf": np.ascontiguousarray(np.asarray(tfs, dtype=np.int32)[order]),
            "doc_norm": np.asarray([k1 * (1 - b + b * (dl / avg)) for dl in doc_lens], dtype=np.float64)}


class HipBM25Index:
    def __init__(self, path: Optional[str] = None, k1: float = 1.5, b: float = 0.75, device: int = 0,
                 rows: Optional[Sequence[Dict[str, Any]]] = None):
        self.path = Path(path) if path is not None else None
        self.k1, self.b, self.device = k1, b, int(device)
        self.docs: List[Dict[str, Any]] = []
        self.N = 0
        self.avgdl = 0.0
        self.df: Dict[str, int] = {}
        self.doc_lens: List[int] = []
        self._term_id: Dict[str, int] = {}
        self._h = None
        self._lib = None
        if rows is not None:
            self._build(list(rows))
        elif self.path is not None and self.path.exists():       # text_index.py:32-33: a missing file is an empty index
            with self.path.open("r", encoding="utf-8") as f:
                self._build([json.loads(line) for line in (ln.strip() for ln in f) if line])

    # -- build (text_index.py:36-52) -------------------------------------------------------------------
    def _build(self, rows: List[Dict[str, Any]]):
        self.docs = rows
        p = build_postings(rows, self.k1, self.b)
        self._term_id, self.doc_lens, self.N, self.avgdl, self.df = p["term_id"], p["doc_lens"], p["N"], p["avgdl"], p["df"]
        if self.N:
            self._lib = N.load()
            h = C.c_uint64(0)
            N.check(self._lib.mrag_bm25_create(self.device, self.N, len(self._term_id), p["indptr"].ctypes.data,
                                               p["post_doc"].ctypes.data, p["post_tf"].ctypes.data, p["doc_norm"].ctypes.data,
                                               float(self.k1 + 1), C.byref(h)))
            self._h = h

    def close(self):
        if self._h is not None and self._h.value:
            self._lib.mrag_bm25_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- query side ------------------------------------------------------------------------------------
    def _idf(self, term: str) -> float:
        """text_index.py:54-56."""
        n = self.df.get(term, 0)
        return math.log((self.N - n + 0.5) / (n + 0.5) + 1.0) if self.N > 0 else 0.0

    def search(self, queries: List[str], top_k: int = 20, alpha_merge: str = "max") -> List[Tuple[int, float]]:
        """text_index.py:71-97 on the device.  ``[(doc_idx, score)]``, best first."""
        if not self.N or not queries or top_k <= 0:
            return []
        q_ptr, q_terms, q_idf = [0], [], []
        for q in queries:
            for t in tokenize(q):
                q_terms.append(self._term_id.get(t, -1))
                q_idf.append(self._idf(t))
            q_ptr.append(len(q_terms))
        k = int(min(top_k, self.N))
        qp = np.asarray(q_ptr, dtype=np.int32)
        qt = np.asarray(q_terms if q_terms else [0], dtype=np.int32)
        qi = np.asarray(q_idf if q_idf else [0.0], dtype=np.float64)
        docs = np.empty(k, dtype=np.int64)
        scores = np.empty(k, dtype=np.float64)
        n = C.c_int(0)
        N.check(self._lib.mrag_bm25_search(self._h, len(queries), qp.ctypes.data, qt.ctypes.data, qi.ctypes.data,
                                           1 if alpha_merge == "sum" else 0, k, docs.ctypes.data, scores.ctypes.data,
                                           C.byref(n), None))
        return [(int(d), float(s)) for d, s in zip(docs[:n.value], scores[:n.value])]

    def doc_meta(self, doc_idx: int) -> Dict[str, Any]:
        return dict(self.docs[doc_idx]) if 0 <= doc_idx < len(self.docs) else {}


class HipBM25TextSearcher:
    """``BM25TextSearcher`` (retrieval_backend.py:102-128) over a :class:`HipBM25Index`; usable as the
    ``text_search`` channel of :class:`mrag_amd.backend.DenseRetrievalBackend`."""

    def __init__(self, index: HipBM25Index):
        self.index = index

    def search(self, *, queries: List[str], top_k: int) -> List[Dict[str, Any]]:
        if self.index.N <= 0:
            return []
        out = []
        for doc_idx, s in self.index.search(queries, top_k=top_k, alpha_merge="max"):
            meta = self.index.doc_meta(doc_idx)
            out.append({"id": "sent::%s::%s" % (meta.get("doc_id") or meta.get("title") or "doc", str(meta.get("sent_id") or "")),
                        "score": float(s),
                        "meta": {"kind": "sentence", "text": meta.get("text"), "doc": meta.get("title"),
                                 "sent_id": meta.get("sent_id"), "source": "bm25"}})
        return out

    __call__ = search
