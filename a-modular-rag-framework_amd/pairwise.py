""""Next" rows of SURVEY section 8f that reuse the cosine kernels on small all-pairs problems.

``semantic_edges``  the semantic_sim edge pass of graph construction
                    (app/modules/graph_construction/edge_builder.py:146-169): every unordered pair of
                    sentence embeddings, edge when cosine >= threshold (settings.yaml:66: 0.9).
``mmr_diversify``   greedy MMR (app/utils/similarity.py:33-62) with the candidate x candidate
                    similarities taken from ONE all-pairs launch instead of O(k*n) Python cosines.
Both run ``mrag_cosine_matrix_f64`` (fp64 on the GPU, zero-norm rows -> 0.0).
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
