"""CPU oracle for corpus-wide cosine top-k (brute force and IVF-flat).

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  numpy only.

The reference scores ``dot/(|a||b|)`` in fp64 per pair
(app/modules/retrieval/retrieval_backend.py:192-197), sorts by score
descending and truncates (``:371-372``).  At corpus scale the same arithmetic
is restated as: L2-normalise rows once, round ONCE to the storage type
(fp16), then score the *rounded* rows in fp64 -- so CPU and GPU see identical
input bits and the only difference left is the GPU's
fp32 accumulation.  Declared tie-break: score descending, then corpus row
ascending (the reference's own tie order is hash-seed dependent,
``retrieval_backend.py:357-359``).

``brute_force_topk`` is pinned by fixture F5 (reference ``_cosine`` over the
C1 shape).  ``ivf_*``: the reference has no IVF index -- PARITY UNPINNED by the
reference; pinned against ``brute_force_topk`` only.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np


# ------------------------------------------------------------------ synthetic data
def make_gaussian(n: int, d: int, seed: int) -> np.ndarray:
    """Rows ~ N(0,1)^d, fp32 (corpus seed 1234, queries 5678)."""
    return np.random.default_rng(seed).standard_normal((n, d), dtype=np.float32)


def make_clustered(n: int, nq: int, d: int, seed: int, n_centroids: int = 4096,
                   sigma_row: float = 0.3, sigma_query: float = 0.1) -> Tuple[np.ndarray, np.ndarray]:
    """Clustered set of SURVEY.md section 8d: rows = centroid + 0.3 N(0,1), queries =
    (random corpus row) + 0.1 N(0,1)."""
    rng = np.random.default_rng(seed)
    cent = rng.standard_normal((n_centroids, d), dtype=np.float32)
    which = rng.integers(0, n_centroids, size=n)
    rows = cent[which] + np.float32(sigma_row) * rng.standard_normal((n, d), dtype=np.float32)
    pick = rng.integers(0, n, size=nq)
    qs = rows[pick] + np.float32(sigma_query) * rng.standard_normal((nq, d), dtype=np.float32)
    return rows.astype(np.float32), qs.astype(np.float32)


# ------------------------------------------------------------------ K1 restated
def l2_normalize(x: np.ndarray) -> np.ndarray:
    """Row L2-normalise: fp64 sum of squares, fp64 divide, ONE rounding to fp32.
    Zero-norm rows stay all-zero, so their cosine is 0.0 exactly like
    ``_cosine``'s ``na and nb`` guard (retrieval_backend.py:197)."""
    x64 = np.asarray(x, dtype=np.float64)
    n = np.sqrt(np.einsum("ij,ij->i", x64, x64))
    n[n == 0] = 1.0
    return (x64 / n[:, None]).astype(np.float32)


def normalize_round(x: np.ndarray, dtype=np.float16) -> np.ndarray:
    """normalise in fp32-from-fp64, then round once to the storage dtype."""
    return l2_normalize(x).astype(dtype)


# ------------------------------------------------------------------ top-k
def topk_desc_rowasc(scores: np.ndarray, k: int) -> Tuple[np.ndarray, np.ndarray]:
    """Per row of ``scores`` [nq, n]: top-k by (score desc, column asc)."""
    nq, n = scores.shape
    k = min(k, n)
    if n > 4 * k + 64:
        # kth-largest value per row; keep everything >= it, then order exactly
        part = np.partition(scores, n - k, axis=1)[:, n - k]
        ids = np.empty((nq, k), dtype=np.int64)
        val = np.empty((nq, k), dtype=scores.dtype)
        for i in range(nq):
            cand = np.nonzero(scores[i] >= part[i])[0]
            order = np.argsort(-scores[i, cand], kind="stable")[:k]
            ids[i] = cand[order]
            val[i] = scores[i, ids[i]]
        return val, ids
    order = np.argsort(-scores, axis=1, kind="stable")[:, :k]
    return np.take_along_axis(scores, order, axis=1), order.astype(np.int64)


def merge_topk(val_parts, id_parts, k: int) -> Tuple[np.ndarray, np.ndarray]:
    """Merge per-shard partial top-k: concatenate, order by
    (score desc, global id asc), keep k.  ids < 0 mark empty slots."""
    v = np.concatenate(val_parts, axis=1)
    g = np.concatenate(id_parts, axis=1)
    nq = v.shape[0]
    out_v = np.full((nq, k), -np.inf, dtype=v.dtype)
    out_i = np.full((nq, k), -1, dtype=np.int64)
    for i in range(nq):
        ok = g[i] >= 0
        vi, gi = v[i][ok], g[i][ok]
        order = np.lexsort((gi, -vi))[:k]
        out_v[i, :len(order)] = vi[order]
        out_i[i, :len(order)] = gi[order]
    return out_v, out_i


def brute_force_topk(q: np.ndarray, c: np.ndarray, k: int, block: int = 262144
                     ) -> Tuple[np.ndarray, np.ndarray]:
    """Exact top-k of ``q @ c.T`` with fp64 accumulation over the GIVEN (already
    normalised + rounded) rows.  Returns (scores fp64 [nq,k], rows int64 [nq,k])."""
    q64 = np.asarray(q, dtype=np.float64)
    nq, n = q64.shape[0], c.shape[0]
    vals, ids = [], []
    for lo in range(0, n, block):
        s = q64 @ np.asarray(c[lo:lo + block], dtype=np.float64).T
        v, i = topk_desc_rowasc(s, k)
        vals.append(v)
        ids.append(i + lo)
    if len(vals) == 1:
        kk = vals[0].shape[1]
        if kk < k:
            pv = np.full((nq, k - kk), -np.inf); pi = np.full((nq, k - kk), -1, dtype=np.int64)
            return np.concatenate([vals[0], pv], 1), np.concatenate([ids[0], pi], 1)
        return vals[0], ids[0]
    return merge_topk(vals, ids, k)


def recall_at_k(got_ids: np.ndarray, ref_ids: np.ndarray) -> float:
    """mean over queries of |got ∩ ref| / |ref| (ids < 0 ignored)."""
    hit = tot = 0
    for g, r in zip(got_ids, ref_ids):
        r = set(int(x) for x in r if x >= 0)
        hit += len(r & set(int(x) for x in g if x >= 0))
        tot += len(r)
    return hit / max(tot, 1)


def gap_aware_id_match(got_ids, got_scores, ref_ids, ref_scores, tol: float) -> Tuple[int, int]:
    """Doc-id parity rule of SURVEY.md section 7 ("hard parts"): position p must hold the
    reference id unless the reference score at p is within ``tol`` of a neighbouring
    reference score (a near-tie the fp32 accumulation order may legally flip), in
    which case the id only has to appear in the reference list or score within tol
    of the reference k-th.  Returns (#strict positions checked, #mismatches)."""
    strict = bad = 0
    nq, k = ref_ids.shape
    for i in range(nq):
        rs, ri = ref_scores[i], ref_ids[i]
        for p in range(k):
            if ri[p] < 0:
                continue
            near = (p > 0 and abs(rs[p - 1] - rs[p]) <= tol) or (p + 1 < k and abs(rs[p] - rs[p + 1]) <= tol) \
                or (p == k - 1)
            if not near:
                strict += 1
                bad += int(got_ids[i, p] != ri[p])
            else:
                ok = (got_ids[i, p] in ri) or abs(float(got_scores[i, p]) - rs[k - 1]) <= tol
                bad += int(not ok)
    return strict, bad


# ------------------------------------------------------------------ IVF-flat
def ivf_assign(c: np.ndarray, centroids: np.ndarray, block: int = 65536) -> np.ndarray:
    """Row -> list id: argmax inner product with the (normalised, rounded) centroids;
    ties go to the lowest centroid id.  fp64 accumulation."""
    cen = np.asarray(centroids, dtype=np.float64)
    out = np.empty(c.shape[0], dtype=np.int64)
    for lo in range(0, c.shape[0], block):
        s = np.asarray(c[lo:lo + block], dtype=np.float64) @ cen.T
        out[lo:lo + block] = np.argmax(s, axis=1)
    return out


def kmeans_spherical(x: np.ndarray, nlist: int, iters: int, seed: int, dtype=np.float16) -> np.ndarray:
    """Seeded spherical k-means on normalised rows: init = ``nlist`` distinct rows
    drawn with ``default_rng(seed)``, assign by max inner product, centroid = the
    re-normalised mean (an empty list keeps its previous centroid).  Centroids are
    rounded to the storage dtype after every update."""
    rng = np.random.default_rng(seed)
    pick = rng.choice(x.shape[0], size=nlist, replace=False)
    cen = normalize_round(np.asarray(x[pick], dtype=np.float32), dtype)
    x64 = np.asarray(x, dtype=np.float64)
    for _ in range(iters):
        a = ivf_assign(x, cen)
        sums = np.zeros((nlist, x.shape[1]), dtype=np.float64)
        np.add.at(sums, a, x64)
        cnt = np.bincount(a, minlength=nlist)
        new = normalize_round(sums.astype(np.float32), dtype)
        cen = np.where((cnt > 0)[:, None], new, cen)
    return cen


def ivf_search(q: np.ndarray, c: np.ndarray, centroids: np.ndarray, assign: np.ndarray,
               nprobe: int, k: int) -> Tuple[np.ndarray, np.ndarray]:
    """IVF-flat: probe the ``nprobe`` best lists per query (score desc, list id asc),
    exact fp64 scan of their rows, top-k by (score desc, row asc)."""
    q64 = np.asarray(q, dtype=np.float64)
    cs = q64 @ np.asarray(centroids, dtype=np.float64).T
    _, probes = topk_desc_rowasc(cs, nprobe)
    nlist = centroids.shape[0]
    order = np.argsort(assign, kind="stable")
    starts = np.searchsorted(assign[order], np.arange(nlist + 1))
    nq = q.shape[0]
    out_v = np.full((nq, k), -np.inf)
    out_i = np.full((nq, k), -1, dtype=np.int64)
    c64 = None
    for i in range(nq):
        rows = np.concatenate([order[starts[p]:starts[p + 1]] for p in probes[i]]) if nprobe else np.empty(0, int)
        if rows.size == 0:
            continue
        rows = np.sort(rows)
        s = np.asarray(c[rows], dtype=np.float64) @ q64[i]
        o = np.argsort(-s, kind="stable")[:k]
        out_v[i, :len(o)] = s[o]
        out_i[i, :len(o)] = rows[o]
    return out_v, out_i


# ------------------------------------------------------------------ CPU baseline ("port")
def brute_force_topk_f32(q32: np.ndarray, c32: np.ndarray, k: int, block: int = 131072
                         ) -> Tuple[np.ndarray, np.ndarray]:
    """The restated CPU path timed as ``cpu_baseline``: fp32 BLAS
    ``Q @ C.T`` on pre-normalised rows, argpartition + exact ordering of the survivors
    (score desc, row asc).  Multi-threaded through numpy's BLAS."""
    nq = q32.shape[0]
    best_v = np.full((nq, 0), -np.inf, dtype=np.float32)
    best_i = np.zeros((nq, 0), dtype=np.int64)
    for lo in range(0, c32.shape[0], block):
        s = q32 @ c32[lo:lo + block].T
        kk = min(k, s.shape[1])
        part = np.argpartition(-s, kk - 1, axis=1)[:, :kk]
        v = np.concatenate([best_v, np.take_along_axis(s, part, axis=1)], axis=1)
        i = np.concatenate([best_i, part.astype(np.int64) + lo], axis=1)
        order = np.lexsort((i, -v), axis=1)[:, :k]
        best_v, best_i = np.take_along_axis(v, order, axis=1), np.take_along_axis(i, order, axis=1)
    return best_v, best_i
