""""Next" rows of SURVEY section 8f that reuse the cosine kernels on small all-pairs problems.

``semantic_edges``  the semantic_sim edge pass of graph construction
                    (app/modules/graph_construction/edge_builder.py:146-169): every unordered pair of
                    sentence embeddings, edge when cosine >= threshold (settings.yaml:66: 0.9).
``mmr_diversify``   greedy MMR (app/utils/similarity.py:33-62) with the candidate x candidate
                    similarities taken from ONE all-pairs launch instead of O(k*n) Python cosines.
Both run ``mrag_cosine_matrix_f64`` (fp64 on the GPU, zero-norm rows -> 0.0).

``segment_context`` embed-mode segmentation (app/modules/graph_construction/segmenter.py:10-57): sentences of a
                    title are merged while the cosine of ADJACENT sentence embeddings stays >= the threshold.
                    All sentences of the context are embedded in one batch and the adjacent cosines come from
                    one launch (``mrag_cosine_adjacent_f64``) instead of one embed call + numpy per sentence.
``BatchedEmbedFn``  the ``embed_fn: text -> vector`` callable the reference's graph-construction hooks expect
                    (edge_builder.py:26,146-152; node_builder.py:56 ``policy["embed_fn"]``; segmenter.py:14) but
                    nothing in the reference ever supplies, backed by this package's provider.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _native as N


def cosine_matrix(vectors, device: int = 0) -> np.ndarray:
    x = np.ascontiguousarray(vectors, dtype=np.float64)
    if x.ndim != 2:
        raise ValueError("expected [n, d] vectors")
    n, d = x.shape
    out = np.empty((n, n), dtype=np.float64)
    if n:
        N.check(N.load().mrag_cosine_matrix_f64(device, x.ctypes.data, n, d, out.ctypes.data, 0, None))
    return out


def semantic_edges(vectors, threshold: float = 0.9, device: int = 0) -> List[Tuple[int, int, float]]:
    """[(i, j, sim)] for i < j with sim >= threshold, in (i, j) order -- the pairs
    ``itertools.combinations`` visits in edge_builder.py:152."""
    s = cosine_matrix(vectors, device)
    iu, ju = np.triu_indices(s.shape[0], k=1)
    keep = s[iu, ju] >= threshold
    return [(int(i), int(j), float(v)) for i, j, v in zip(iu[keep], ju[keep], s[iu, ju][keep])]


def mmr_diversify(items: Sequence[Tuple[str, float, Optional[Sequence[float]]]], *, top_k: int = 20,
                  lambda_weight: float = 0.7, device: int = 0):
    """Same selection rule as the reference: first pick = best score; afterwards
    ``lambda*score - (1-lambda)*max(0, max sim to the selected)``; the first strictly greater value
    wins; a missing vector has similarity 0; selected ids are removed by id equality."""
    items = list(items)
    has = [i for i, it in enumerate(items) if it[2] is not None]
    sim = np.zeros((len(items), len(items)))
    if has:
        dim = max(len(items[i][2]) for i in has)
        same = [i for i in has if len(items[i][2]) == dim]
        if len(same) == len(has):       # the reference zips (truncates) mismatched lengths; keep that case on the host
            sub = cosine_matrix(np.asarray([items[i][2] for i in has], dtype=np.float64), device)
            sim[np.ix_(has, has)] = sub
        else:
            from math import sqrt
            for a in has:
                for b in has:
                    u, v = items[a][2], items[b][2]
                    du, dv = sqrt(sum(x * x for x in u)), sqrt(sum(x * x for x in v))
                    sim[a, b] = 0.0 if du == 0 or dv == 0 else sum(x * y for x, y in zip(u, v)) / (du * dv)
    chosen: List[int] = []
    pool = list(range(len(items)))
    while pool and len(chosen) < top_k:
        best, best_val = None, -1e9
        for c in pool:
            if not chosen:
                val = items[c][1]
            else:
                worst = max(0.0, max(sim[c, s] for s in chosen))
                val = lambda_weight * items[c][1] - (1 - lambda_weight) * worst
            if val > best_val:
                best_val, best = val, c
        chosen.append(best)
        bid = items[best][0]
        pool = [c for c in pool if items[c][0] != bid]
    return [items[c] for c in chosen]


def cosine_adjacent(vectors, eps: float = 1e-9, device: int = 0) -> np.ndarray:
    """out[i] = dot(x_i, x_{i+1}) / (|x_i| |x_{i+1}| + eps) (segmenter.py:40-42), fp64 on the GPU."""
    x = np.ascontiguousarray(vectors, dtype=np.float64)
    if x.ndim != 2:
        raise ValueError("expected [n, d] vectors")
    out = np.empty(max(0, x.shape[0] - 1), dtype=np.float64)
    if x.shape[0] >= 2:
        N.check(N.load().mrag_cosine_adjacent_f64(device, x.ctypes.data, x.shape[0], x.shape[1], float(eps), out.ctypes.data, None))
    return out


class BatchedEmbedFn:
    """``embed_fn(text) -> List[float]`` for the reference's graph-construction hooks, served from a cache that
    :meth:`prime` fills with ONE provider batch (the hooks call it once per sentence; an encoder forward per
    sentence would waste the GPU).  ``provider``: anything with ``embed(model=, texts=, require=)`` (this
    package's ``HipEmbeddingProvider``) or a router-like object with ``embed(model_hint=, texts=, require=)``."""

    def __init__(self, provider, model: Optional[str] = None, trace_id: str = "graph-construction"):
        self.provider, self.model, self.trace_id = provider, model, trace_id
        self.cache = {}

    def _embed(self, texts: List[str]) -> List[List[float]]:
        if hasattr(self.provider, "policy"):       # an LLMRouter
            ret = self.provider.embed(model_hint=self.model or "", texts=texts, require={"trace_id": self.trace_id})
        else:
            ret = self.provider.embed(model=self.model, texts=texts, require={"trace_id": self.trace_id})
        return ret.get("vectors") if isinstance(ret, dict) else ret

    def prime(self, texts: Sequence[str]) -> None:
        todo = [t for t in dict.fromkeys(texts) if t not in self.cache]
        if todo:
            for t, v in zip(todo, self._embed(todo)):
                self.cache[t] = [float(x) for x in v]

    def __call__(self, text: str) -> List[float]:
        if text not in self.cache:
            self.prime([text])
        return self.cache[text]


def segment_context(ctx: Sequence[Tuple[str, List[str]]], *, strategy: str = "rule", embed_fn=None,
                    sim_threshold: float = 0.65, device: int = 0) -> List[Tuple[str, List[str]]]:
    """``segment_context`` of the reference (segmenter.py:10-57), same arguments and results.  ``rule``: split on
    sentence punctuation (:5-7); ``embed`` with an ``embed_fn``: start a new segment where the adjacent-sentence
    similarity falls below ``sim_threshold``; anything else: sentences unchanged.  In embed mode every sentence
    of the context is embedded up front (one batch when ``embed_fn`` is a :class:`BatchedEmbedFn`) and all
    adjacent similarities come from one GPU launch."""
    import re
    ctx = [(title, list(sents)) for title, sents in ctx]
    sims = None
    if strategy == "embed" and embed_fn:
        flat = [s for _, sents in ctx for s in sents]
        if isinstance(embed_fn, BatchedEmbedFn):
            embed_fn.prime(flat)
        vecs = [embed_fn(s) for s in flat]
        if len(flat) >= 2:
            sims = cosine_adjacent(np.asarray(vecs, dtype=np.float64), 1e-9, device)
    out: List[Tuple[str, List[str]]] = []
    at = 0
    for title, sents in ctx:
        if strategy == "rule":
            new = [p.strip() for s in sents for p in re.split(r"[。！？.!?]", s) if p.strip()]
        elif strategy == "embed" and embed_fn:
            new, batch = [], []
            for i, s in enumerate(sents):
                if i > 0 and float(sims[at + i - 1]) < sim_threshold:     # :43-46 (never cuts before a title's first sentence)
                    new.append(" ".join(batch))
                    batch = []
                batch.append(s)
            if batch:
                new.append(" ".join(batch))
            at += len(sents)
        else:
            new = list(sents)
        out.append((title, new))
    return out
