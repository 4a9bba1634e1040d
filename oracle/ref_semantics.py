"""CPU restatement of the reference's dense re-rank / fusion semantics.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Plain Python, small inputs.
All ``file:line`` citations are relative to the reference checkout
(``/root/reference``); nothing here imports it.  Pinned by fixtures F1-F4 in
``tests/golden`` (captured from the reference's own functions).
"""
from __future__ import annotations

import json
import math
from typing import Any, Callable, Dict, Iterable, List, Optional, Sequence, Tuple

DEFAULT_EMBED_MODEL = "text-embedding-3-large"
ID_KEYS = ("id", "doc_id", "docId", "sid", "sent_id")
SCORE_KEYS = ("score", "relevance", "sim", "s")


# --------------------------------------------------------------------------- a1
def cosine(a: Sequence[float], b: Sequence[float]) -> float:
    """``DenseReranker._cosine`` -- app/modules/retrieval/retrieval_backend.py:192-197.

    fp64, left-to-right accumulation; 0.0 when either side is empty, the
    lengths differ, or either norm is zero.  (Squares that overflow fp64 give
    inf/inf = NaN in the reference as well -- fixture F1 holds such a row.)
    """
    if len(a) == 0 or len(b) == 0 or len(a) != len(b):
        return 0.0
    dot = 0
    saa = 0
    sbb = 0
    for x, y in zip(a, b):
        dot = dot + x * y
    for x in a:
        saa = saa + x * x
    for y in b:
        sbb = sbb + y * y
    na = math.sqrt(saa)
    nb = math.sqrt(sbb)
    if not na or not nb:
        return 0.0
    return float(dot / (na * nb))


# -------------------------------------------------------------------------- a10
def cosine_util(u: Sequence[float], v: Sequence[float]) -> float:
    """``app/utils/similarity.py:11-16``: like :func:`cosine` but with no length
    check -- the dot product silently truncates to the shorter vector while
    each norm runs over its full vector."""
    if len(u) == 0 or len(v) == 0:
        return 0.0
    du = math.sqrt(sum(x * x for x in u))
    dv = math.sqrt(sum(x * x for x in v))
    if du == 0 or dv == 0:
        return 0.0
    return sum(x * y for x, y in zip(u, v)) / (du * dv)


def mmr_diversify(items: List[Tuple[str, float, Optional[Sequence[float]]]], *,
                  top_k: int = 20, lambda_weight: float = 0.7):
    """Greedy MMR -- ``app/utils/similarity.py:33-62``.

    value = score for the first pick, afterwards
    ``lambda*score - (1-lambda)*max_sim_to_selected`` (sim 0 when a vector is
    missing; max_sim floor is 0.0).  First strictly-greater value wins, i.e.
    ties keep the earliest candidate; selected ids are removed by id equality.
    """
    chosen: List[Tuple[str, float, Optional[Sequence[float]]]] = []
    pool = list(items)
    while pool and len(chosen) < top_k:
        best_item, best_val = None, -1e9
        for cid, cscore, cvec in pool:
            if not chosen:
                val = cscore
            else:
                worst = 0.0
                for _sid, _ss, svec in chosen:
                    s = cosine_util(cvec, svec) if (cvec is not None and svec is not None) else 0.0
                    worst = max(worst, s)
                val = lambda_weight * cscore - (1 - lambda_weight) * worst
            if val > best_val:
                best_val, best_item = val, (cid, cscore, cvec)
        chosen.append(best_item)
        pool = [x for x in pool if x[0] != best_item[0]]
    return chosen


# --------------------------------------------------------------------------- a3
def resolve_embed_model(policy: Optional[Dict[str, Any]], providers: Optional[Dict[str, Any]]) -> str:
    """``DenseReranker._resolve_embed_model`` -- retrieval_backend.py:199-213.

    1. ``policy["embedding"][0]["model"]``; 2. ``.kwargs["embed_model"]`` of the
    provider named by ``policy["embedding_provider"]``; 3. the OpenAI default.
    Any exception falls through to the default.
    """
    try:
        policy = policy or {}
        emb = policy.get("embedding") or []
        if emb and isinstance(emb[0], dict) and emb[0].get("model"):
            return emb[0]["model"]
        name = policy.get("embedding_provider")
        prov = (providers or {}).get(name) if name else None
        if prov and hasattr(prov, "kwargs"):
            m = (getattr(prov, "kwargs") or {}).get("embed_model")
            if m:
                return m
    except Exception:
        pass
    return DEFAULT_EMBED_MODEL


# --------------------------------------------------------------------------- a4
def router_embed(providers: Dict[str, Any], policy: Dict[str, Any], *, model_hint: str,
                 texts: List[str], require: Optional[Dict[str, Any]] = None):
    """``LLMRouter.embed`` dispatch + fallbacks -- app/core/llm_router.py:103-130.

    Provider = ``policy["embedding_provider"]`` (default name ``"mock"``).  A
    real provider is called as ``embed(model=, texts=, require=)`` and its
    return value is passed through untouched.  Missing provider, the ``mock``
    name or ANY exception give ``[[0.0]*3]*len(texts)``.
    """
    name = (policy or {}).get("embedding_provider") or "mock"
    prov = providers.get(name)
    try:
        if prov and name != "mock":
            return prov.embed(model=model_hint, texts=texts, require=require or {})
        return [[0.0] * 3 for _ in texts]
    except Exception:
        return [[0.0] * 3 for _ in texts]


# --------------------------------------------------------------------------- a2
def _vectors_of(ret):
    return ret.get("vectors") if isinstance(ret, dict) else ret


def dense_score(embed: Callable[..., Any], *, query: str, candidates: List[Dict[str, Any]],
                trace_id: str, max_pool: int = 200, embed_batch: int = 50,
                model_hint: str = DEFAULT_EMBED_MODEL) -> Dict[str, float]:
    """``DenseReranker.score`` -- retrieval_backend.py:215-247.

    ``embed(model_hint=, texts=, require=)`` plays ``LLMRouter.embed``.  First
    ``max_pool`` candidates; rows with empty ``meta.text`` are skipped; the
    query is embedded alone (failure -> ``{}``); texts go in chunks of
    ``max(8, embed_batch)``; a failed chunk is zero-filled at the query's
    dimension; ids and vectors are zipped (so a short provider answer truncates
    and duplicate ids keep the LAST score).
    """
    if not candidates:
        return {}
    ids, texts = [], []
    for h in candidates[:max_pool]:
        t = (h.get("meta") or {}).get("text") or ""
        if t:
            ids.append(h["id"])
            texts.append(t)
    if not texts:
        return {}
    try:
        qv = _vectors_of(embed(model_hint=model_hint, texts=[query], require={"trace_id": trace_id}))[0]
    except Exception:
        return {}
    vecs: List[Sequence[float]] = []
    step = max(8, int(embed_batch))
    for lo in range(0, len(texts), step):
        chunk = texts[lo:lo + step]
        try:
            got = _vectors_of(embed(model_hint=model_hint, texts=chunk, require={"trace_id": trace_id}))
            vecs.extend(got or [])
        except Exception:
            vecs.extend([[0.0] * len(qv) for _ in chunk])
    return {i: cosine(qv, v) for i, v in zip(ids, vecs)}


# ---------------------------------------------------------------------- a6 / a7
def bm25_hit_id(meta: Dict[str, Any]) -> str:
    """Raw text-channel id -- retrieval_backend.py:116-119: ``sent::{doc_id or
    title or 'doc'}::{sent_id or ''}`` (so ``sent_id == 0`` renders empty)."""
    doc_part = meta.get("doc_id") or meta.get("title") or "doc"
    return f"sent::{doc_part}::{str(meta.get('sent_id') or '')}"


def normalize_id(hit: Dict[str, Any]) -> Tuple[str, Dict[str, Any]]:
    """``HybridRetrievalBackend._normalize_id`` -- retrieval_backend.py:283-294."""
    meta = hit.get("meta") or {}
    doc = meta.get("doc") or meta.get("title")
    sid = meta.get("sent_id") or meta.get("sid")
    if doc is not None and sid is not None:
        return f"sent::{doc}::{sid}", meta
    if doc is not None:
        return f"sent::{doc}::{sid or ''}", meta
    return (hit.get("id") or "") or "sent::unknown::", meta


def minmax_norm(values: Dict[str, float]) -> Dict[str, float]:
    """``_minmax_norm`` -- retrieval_backend.py:296-301: empty -> {}, all-equal -> 0.0."""
    if not values:
        return {}
    lo, hi = min(values.values()), max(values.values())
    if hi <= lo:
        return {k: 0.0 for k in values}
    return {k: (v - lo) / (hi - lo) for k, v in values.items()}


def norm_map(hits: Optional[Iterable[Dict[str, Any]]]) -> Dict[str, Dict[str, Any]]:
    """Dedupe by normalised id -- retrieval_backend.py:336-348: a strictly larger
    score replaces the entry (and its meta); otherwise the newcomer's meta only
    fills keys the kept entry lacks."""
    out: Dict[str, Dict[str, Any]] = {}
    for h in hits or []:
        nid, meta = normalize_id(h)
        sc = float(h.get("score") or 0.0)
        prev = out.get(nid)
        if prev is None or sc > float(prev.get("score") or 0.0):
            out[nid] = {"id": nid, "score": sc, "meta": dict(meta or {})}
        else:
            pm = prev.get("meta") or {}
            for k, v in (meta or {}).items():
                pm.setdefault(k, v)
            prev["meta"] = pm
    return out


def fuse(t_hits, g_hits, dense_scores: Dict[str, float], *, alpha_text: float, alpha_graph: float,
         alpha_dense: float, top_k: int) -> List[Dict[str, Any]]:
    """Fusion + top-k -- retrieval_backend.py:350-372.

    Scores per channel are min-max normalised, summed with the alphas over the
    union of ids, ``meta`` = text meta overlaid by graph meta plus the three
    ``score_*_norm`` keys; sorted by score descending, truncated to ``top_k``.
    The reference iterates a ``set`` of ids, so the order INSIDE a tie is
    hash-seed dependent there; here ties are left in ascending id order (the
    build's declared tie-break), and tests compare tie groups as sets.
    """
    tmap, gmap = norm_map(t_hits), norm_map(g_hits)
    nt = minmax_norm({k: float(v["score"]) for k, v in tmap.items()})
    ng = minmax_norm({k: float(v["score"]) for k, v in gmap.items()})
    nd = minmax_norm(dense_scores)
    fused = []
    for nid in sorted(set(tmap) | set(gmap) | set(nd)):
        ts, gs, ds = nt.get(nid, 0.0), ng.get(nid, 0.0), nd.get(nid, 0.0)
        meta: Dict[str, Any] = {}
        if nid in tmap and isinstance(tmap[nid].get("meta"), dict):
            meta.update(tmap[nid]["meta"])
        if nid in gmap and isinstance(gmap[nid].get("meta"), dict):
            meta.update(gmap[nid]["meta"])
        meta["score_text_norm"], meta["score_graph_norm"], meta["score_dense_norm"] = ts, gs, ds
        fused.append({"id": nid, "score": float(alpha_text * ts + alpha_graph * gs + alpha_dense * ds),
                      "meta": meta})
    fused.sort(key=lambda h: h["score"], reverse=True)
    return fused[:top_k]


# --------------------------------------------------------------------------- a9
def normalize_hit(raw: Any, id_keys: Sequence[str] = ID_KEYS, score_keys: Sequence[str] = SCORE_KEYS,
                  meta_key: Optional[str] = "meta") -> Optional[Dict[str, Any]]:
    """``RetrievalAdapter._normalize_hit`` -- retrieval_adapter.py:71-109 (dict
    input only): first non-None id / score alias, bad score -> 0.0, ``meta``
    taken from ``meta_key`` when it is a dict else every non-alias field, and
    a missing id rebuilt as ``sent::{doc|title|'doc'}::{sent_id|sid|''}``."""
    if raw is None:
        return None
    d = dict(raw)
    _id = next((d[k] for k in id_keys if k in d and d[k] is not None), None)
    sc = next((d[k] for k in score_keys if k in d and d[k] is not None), None)
    try:
        sc = float(sc) if sc is not None else 0.0
    except Exception:
        sc = 0.0
    if meta_key and isinstance(d.get(meta_key), dict):
        meta = dict(d[meta_key])
    else:
        meta = {k: v for k, v in d.items() if k not in id_keys and k not in score_keys}
    if not _id:
        doc = meta.get("doc") or meta.get("title") or "doc"
        sid = meta.get("sent_id") or meta.get("sid") or ""
        _id = f"sent::{doc}::{sid}"
    return {"id": str(_id), "score": sc, "meta": meta}


def adapter_retrieve(run_out: Any, top_k: int) -> Dict[str, Any]:
    """``RetrievalAdapter.retrieve`` -- retrieval_adapter.py:112-133: normalise,
    STABLE sort by score descending, truncate when ``top_k`` is truthy."""
    raw = run_out.get("hits", []) if isinstance(run_out, dict) else []
    diag = run_out.get("diagnostics", {}) if isinstance(run_out, dict) else {}
    hits = [h for h in (normalize_hit(r) for r in raw) if h]
    hits = sorted(hits, key=lambda h: h["score"], reverse=True)
    if top_k:
        hits = hits[:top_k]
    return {"hits": hits, "diagnostics": diag}


# -------------------------------------------------------------------------- a11
def read_docs_jsonl(path: str) -> List[Dict[str, Any]]:
    """Corpus reader -- app/modules/retrieval/text_index.py:36-46; row format
    written by my_code/ingest_hotpotqa.py:73-81.  Blank lines are skipped and
    the corpus row index is the index among the remaining lines."""
    rows = []
    with open(path, "r", encoding="utf-8") as f:
        for line in f:
            line = line.strip()
            if line:
                rows.append(json.loads(line))
    return rows


# ---------------------------------------------------------------------- 8f-2 (segmentation)
def adjacent_similarity(va: Sequence[float], vb: Sequence[float]) -> float:
    """The cut test of embed-mode segmentation -- app/modules/graph_construction/segmenter.py:40-42:
    ``dot / (|a| * |b| + 1e-9)`` (epsilon inside the denominator; zero vectors give 0.0)."""
    import numpy as np
    a, b = np.array(va), np.array(vb)
    return float(np.dot(a, b) / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-9))
