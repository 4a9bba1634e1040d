"""Score fusion and hit normalisation of the retrieval boundary -- host side, runs on the k
results AFTER the GPU top-k.  Semantics follow the reference line by line (cited per
function); implementation is this package's own.

Declared tie-break where the reference leaves the order to ``set`` iteration
(retrieval_backend.py:357-359): equal fused scores are ordered by ascending id.
"""
from __future__ import annotations

from typing import Any, Dict, Iterable, List, Optional, Sequence, Tuple

ID_KEYS = ["id", "doc_id", "docId", "sid", "sent_id"]          # retrieval_adapter.py:33
SCORE_KEYS = ["score", "relevance", "sim", "s"]                # retrieval_adapter.py:34


def raw_hit_id(row: Dict[str, Any]) -> str:
    """id of a corpus row as the reference's text channel spells it
    (retrieval_backend.py:116-119): ``sent::{doc_id|title|'doc'}::{sent_id or ''}``."""
    return "sent::%s::%s" % (row.get("doc_id") or row.get("title") or "doc", str(row.get("sent_id") or ""))


def row_meta(row: Dict[str, Any], source: str) -> Dict[str, Any]:
    """meta block of a hit (retrieval_backend.py:121-127); ``text``/``doc``/``sent_id`` are what
    reasoning and graph bootstrap read."""
    return {"kind": "sentence", "text": row.get("text"), "doc": row.get("title"), "sent_id": row.get("sent_id"),
            "source": source}


def normalize_id(hit: Dict[str, Any]) -> Tuple[str, Dict[str, Any]]:
    """retrieval_backend.py:283-294."""
    meta = hit.get("meta") or {}
    doc = meta.get("doc") or meta.get("title")
    sid = meta.get("sent_id") or meta.get("sid")
    if doc is not None:
        return ("sent::%s::%s" % (doc, sid if sid is not None else "")), meta
    return (hit.get("id") or "sent::unknown::"), meta


def minmax_norm(values: Dict[str, float]) -> Dict[str, float]:
    """retrieval_backend.py:296-301: empty -> {}, max <= min -> all 0.0."""
    if not values:
        return {}
    seq = list(values.values())
    lo, hi = min(seq), max(seq)
    if hi <= lo:
        return dict.fromkeys(values, 0.0)
    span = hi - lo
    return {key: (val - lo) / span for key, val in values.items()}


def dedupe_by_norm_id(hits: Optional[Iterable[Dict[str, Any]]]) -> Dict[str, Dict[str, Any]]:
    """retrieval_backend.py:336-348: strictly higher score wins the slot; a loser only
    contributes meta keys the winner does not have."""
    table: Dict[str, Dict[str, Any]] = {}
    for hit in hits or []:
        nid, meta = normalize_id(hit)
        score = float(hit.get("score") or 0.0)
        cur = table.get(nid)
        if cur is None or score > float(cur.get("score") or 0.0):
            table[nid] = {"id": nid, "score": score, "meta": dict(meta or {})}
            continue
        kept = cur.get("meta") or {}
        for key, val in (meta or {}).items():
            if key not in kept:
                kept[key] = val
        cur["meta"] = kept
    return table


def fuse_channels(text_hits, graph_hits, dense_scores: Dict[str, float], *, alpha_text: float,
                  alpha_graph: float, alpha_dense: float, top_k: int) -> List[Dict[str, Any]]:
    """retrieval_backend.py:350-372: per-channel min-max, alpha-weighted sum over the union of
    ids, meta = text meta overlaid by graph meta + the three ``score_*_norm`` keys, sort by
    score descending (ties: id ascending), keep ``top_k``."""
    tmap, gmap = dedupe_by_norm_id(text_hits), dedupe_by_norm_id(graph_hits)
    n_text = minmax_norm({key: float(val["score"]) for key, val in tmap.items()})
    n_graph = minmax_norm({key: float(val["score"]) for key, val in gmap.items()})
    n_dense = minmax_norm(dense_scores)
    out = []
    for nid in sorted(set(tmap).union(gmap, n_dense)):
        ts, gs, ds = n_text.get(nid, 0.0), n_graph.get(nid, 0.0), n_dense.get(nid, 0.0)
        meta: Dict[str, Any] = {}
        for src in (tmap, gmap):
            if nid in src and isinstance(src[nid].get("meta"), dict):
                meta.update(src[nid]["meta"])
        meta["score_text_norm"], meta["score_graph_norm"], meta["score_dense_norm"] = ts, gs, ds
        out.append({"id": nid, "score": float(alpha_text * ts + alpha_graph * gs + alpha_dense * ds), "meta": meta})
    out.sort(key=lambda h: h["score"], reverse=True)        # stable: ties stay in id order
    return out[:top_k]


def fuse_channels_device(text_hits, graph_hits, dense_scores: Dict[str, float], *, alpha_text: float, alpha_graph: float,
                         alpha_dense: float, top_k: int, device: int = 0) -> List[Dict[str, Any]]:
    """:func:`fuse_channels` with the arithmetic on the GPU (``mrag_fuse_topk``): ids are normalised and ranked
    on the host (a key = the id's position among the call's sorted ids), the dedupe / min-max / weighted sum /
    sort / truncate of retrieval_backend.py:336-372 run in one launch, and the host re-attaches ids and meta to
    the ``top_k`` survivors.  Same values bit for bit (fp64, same operation order), same declared tie order."""
    import ctypes as C
    import numpy as np
    from . import _native as N
    chans = []
    for hits in (text_hits, graph_hits):
        ent = []
        for hit in hits or []:
            nid, _ = normalize_id(hit)
            ent.append((nid, float(hit.get("score") or 0.0)))
        chans.append(ent)
    chans.append([(nid, float(v)) for nid, v in dense_scores.items()])
    ids = sorted({nid for ent in chans for nid, _ in ent})
    if not ids or top_k <= 0:
        return []
    key_of = {nid: i for i, nid in enumerate(ids)}
    keys = np.asarray([key_of[nid] for ent in chans for nid, _ in ent], dtype=np.int32)
    scores = np.asarray([v for ent in chans for _, v in ent], dtype=np.float64)
    kk = int(min(top_k, len(ids)))
    ok, osc = np.empty(kk, dtype=np.int32), np.empty(kk, dtype=np.float64)
    nt, ng, nd = (np.empty(kk, dtype=np.float64) for _ in range(3))
    n_out = C.c_int(0)
    N.check(N.load().mrag_fuse_topk(device, keys.ctypes.data, scores.ctypes.data, len(chans[0]), len(chans[1]), len(chans[2]),
                                    float(alpha_text), float(alpha_graph), float(alpha_dense), kk, ok.ctypes.data,
                                    osc.ctypes.data, nt.ctypes.data, ng.ctypes.data, nd.ctypes.data, C.byref(n_out), None))
    tmap, gmap = dedupe_by_norm_id(text_hits), dedupe_by_norm_id(graph_hits)        # meta merge of the survivors (host dicts)
    out = []
    for i in range(n_out.value):
        nid = ids[int(ok[i])]
        meta: Dict[str, Any] = {}
        for src in (tmap, gmap):
            if nid in src and isinstance(src[nid].get("meta"), dict):
                meta.update(src[nid]["meta"])
        meta["score_text_norm"], meta["score_graph_norm"], meta["score_dense_norm"] = float(nt[i]), float(ng[i]), float(nd[i])
        out.append({"id": nid, "score": float(osc[i]), "meta": meta})
    return out


def normalize_raw_hit(raw: Any, id_keys: Sequence[str] = ID_KEYS, score_keys: Sequence[str] = SCORE_KEYS,
                      meta_key: Optional[str] = "meta") -> Optional[Dict[str, Any]]:
    """retrieval_adapter.py:71-109: first non-None id/score alias, unparsable score -> 0.0,
    meta from ``meta_key`` when it is a dict else all non-alias fields, missing id rebuilt
    as ``sent::{doc|title|'doc'}::{sent_id|sid|''}``."""
    if raw is None:
        return None
    if isinstance(raw, dict):
        d = dict(raw)
    else:
        d = {k: getattr(raw, k) for k in dir(raw) if not k.startswith("_") and hasattr(raw, k)}

    def first(keys):
        for k in keys:
            if d.get(k) is not None:
                return d[k]
        return None

    hid, score = first(id_keys), first(score_keys)
    try:
        score = 0.0 if score is None else float(score)
    except Exception:
        score = 0.0
    if meta_key and isinstance(d.get(meta_key), dict):
        meta = dict(d[meta_key])
    else:
        meta = {k: v for k, v in d.items() if k not in id_keys and k not in score_keys}
    if not hid:
        hid = "sent::%s::%s" % (meta.get("doc") or meta.get("title") or "doc", meta.get("sent_id") or meta.get("sid") or "")
    return {"id": str(hid), "score": score, "meta": meta}
