"""Boundary B2 -- retrieval backends selected by ``modules.retrieval.impl``
(config/settings.yaml:83; built by RetrievalAgentFlow.from_settings,
app/modules/retrieval/flow.py:91-108: kwargs filtered to ``__init__``, ``router``/``sink``
injected when declared).

``HipDenseReranker``      drop-in for ``DenseReranker`` (retrieval_backend.py:186-247): same
                          embed-call sequence and error policy, the N cosines run as one fp64
                          GPU kernel instead of a Python loop.
``DenseRetrievalBackend`` corpus-wide dense retrieval: the whole docs.jsonl is embedded once
                          (cached on disk and process-wide), kept in HBM as fp16, and every
                          query is one fused MFMA cosine top-k -- ``run(req) -> {"hits",
                          "diagnostics"}`` and ``retrieve(req) -> RetrievalOut`` like
                          ``HybridRetrievalBackend`` (:303-390), hit ids / meta / fusion keys
                          unchanged.  Optional text / graph channels are injected callables
                          with the reference's hit shape and fused with the reference's
                          min-max + alpha rule.
"""
from __future__ import annotations

import logging
from typing import Any, Callable, Dict, List, Optional

import numpy as np

from . import corpus as _corpus
from . import fusion as _fusion
from .dto import Hit, RetrievalIn, RetrievalOut
from .telemetry import span

logger = logging.getLogger(__name__)
DEFAULT_EMBED_MODEL = "text-embedding-3-large"      # retrieval_backend.py:213


def resolve_embed_model(router) -> str:
    """retrieval_backend.py:199-213."""
    try:
        policy = getattr(router, "policy", {}) or {}
        emb = policy.get("embedding") or []
        if emb and isinstance(emb[0], dict) and emb[0].get("model"):
            return emb[0]["model"]
        name = policy.get("embedding_provider")
        prov = (getattr(router, "providers", {}) or {}).get(name) if name else None
        if prov is not None and hasattr(prov, "kwargs"):
            model = (getattr(prov, "kwargs") or {}).get("embed_model")
            if model:
                return model
    except Exception:
        pass
    return DEFAULT_EMBED_MODEL


def _vectors(ret):
    return ret.get("vectors") if isinstance(ret, dict) else ret


class HipDenseReranker:
    """``DenseReranker`` with the cosine loop on the GPU.  Same constructor fields."""

    def __init__(self, router, max_pool: int = 200, embed_batch: int = 50, device: int = 0):
        self.router, self.max_pool, self.embed_batch, self.device = router, max_pool, embed_batch, device

    def _resolve_embed_model(self) -> str:
        return resolve_embed_model(self.router)

    def _cosines(self, qv, vecs) -> List[float]:
        """retrieval_backend.py:192-197 for every candidate: 0.0 when a side is empty, the
        lengths differ (:193) or a norm is zero (:197); everything else in one kernel."""
        from .index import cosine_f64
        d = len(qv)
        out = [0.0] * len(vecs)
        rows = [i for i, v in enumerate(vecs) if d and v is not None and len(v) == d]
        if rows:
            got = cosine_f64(np.asarray(qv, dtype=np.float64),
                             np.asarray([vecs[i] for i in rows], dtype=np.float64), device=self.device)
            for i, g in zip(rows, got):
                out[i] = float(g)
        return out

    def score(self, *, query: str, candidates: List[Dict[str, Any]], trace_id: str) -> Dict[str, float]:
        if not candidates:
            return {}
        ids, texts = [], []
        for h in candidates[: self.max_pool]:
            t = (h.get("meta") or {}).get("text") or ""
            if t:
                ids.append(h["id"])
                texts.append(t)
        if not texts:
            return {}
        model_hint = self._resolve_embed_model()
        try:
            qv = _vectors(self.router.embed(model_hint=model_hint, texts=[query], require={"trace_id": trace_id}))[0]
        except Exception as e:                      # :229-231
            logger.error("[HipDenseReranker] query embed error: %s", e)
            return {}
        vecs: List[Any] = []
        step = max(8, int(self.embed_batch))        # :234
        for lo in range(0, len(texts), step):
            chunk = texts[lo:lo + step]
            try:
                got = _vectors(self.router.embed(model_hint=model_hint, texts=chunk, require={"trace_id": trace_id}))
                vecs.extend(got or [])
            except Exception as e:                  # :241-243
                logger.error("[HipDenseReranker] batch embed error: %s", e)
                vecs.extend([[0.0] * len(qv) for _ in chunk])
        n = min(len(ids), len(vecs))                # zip() truncation of :245
        return dict(zip(ids[:n], self._cosines(qv, vecs[:n])))


class DenseRetrievalBackend:
    """Corpus-wide dense retrieval behind ``run`` / ``retrieve`` (see module docstring)."""

    def __init__(self, router, sink=None, index_path: str = "data/hotpotqa/docs.jsonl",
                 alpha_text: float = 0.4, alpha_graph: float = 0.2, alpha_dense: float = 0.4,
                 default_top_k: int = 20, dense_pool_k: int = 200, embed_batch: int = 256,
                 cache_dir: Optional[str] = None, device: int = 0, index_dtype: str = "f16",
                 text_search: Optional[Callable[..., List[Dict[str, Any]]]] = None,
                 graph_expand: Optional[Callable[..., List[Dict[str, Any]]]] = None):
        self.router, self.sink = router, sink
        self.index_path = index_path
        self.alpha_text, self.alpha_graph, self.alpha_dense = float(alpha_text), float(alpha_graph), float(alpha_dense)
        self.default_top_k, self.dense_pool_k, self.embed_batch = int(default_top_k), int(dense_pool_k), int(embed_batch)
        self.cache_dir, self.device, self.index_dtype = cache_dir, int(device), index_dtype
        self.text_search, self.graph_expand = text_search, graph_expand
        self._state = None

    # -- corpus index: built once per (file signature, model), shared process-wide --------------
    def _embed_texts(self, texts: List[str], model_hint: str, trace_id: str) -> np.ndarray:
        out = []
        for lo in range(0, len(texts), self.embed_batch):
            got = _vectors(self.router.embed(model_hint=model_hint, texts=texts[lo:lo + self.embed_batch],
                                             require={"trace_id": trace_id}))
            out.append(np.asarray(got, dtype=np.float32))
        return np.concatenate(out, axis=0) if out else np.zeros((0, 0), dtype=np.float32)

    def _build_state(self, model_hint: str, trace_id: str):
        from .index import DenseIndex
        rows = _corpus.read_docs_jsonl(self.index_path)
        state = {"rows": rows, "index": None, "dim": 0}
        if not rows:
            return state
        emb, cache, key = None, None, None
        if self.cache_dir:
            cache = _corpus.EmbeddingCache(self.cache_dir)
        texts = [r.get("text") or "" for r in rows]
        if cache is not None:
            probe = self._embed_texts(texts[:1], model_hint, trace_id)
            key = cache.key(self.index_path, model_hint, probe.shape[1], self.index_dtype)
            emb = cache.load(key)
            if emb is not None and emb.shape != (len(rows), probe.shape[1]):
                emb = None
        if emb is None:
            emb = self._embed_texts(texts, model_hint, trace_id)
            if cache is not None:
                cache.store(key, emb.astype(np.float16), {"model": model_hint, "rows": len(rows), "dim": int(emb.shape[1]),
                                                          "docs": _corpus.file_signature(self.index_path)})
        ix = DenseIndex(int(emb.shape[1]), metric="cosine", dtype=self.index_dtype, device=self.device)
        step = 131072
        for lo in range(0, len(rows), step):
            ix.add(np.ascontiguousarray(emb[lo:lo + step]))
        state["index"], state["dim"] = ix, int(emb.shape[1])
        return state

    def _get_state(self, model_hint: str, trace_id: str):
        if self._state is None:
            try:
                sig = _corpus.file_signature(self.index_path)
            except OSError:
                sig = f"{self.index_path}|missing"
            key = f"dense-index|{sig}|{model_hint}|{self.device}|{self.index_dtype}"
            self._state = _corpus.shared(key, lambda: self._build_state(model_hint, trace_id))
        return self._state

    # -- the backend protocol ---------------------------------------------------------------------
    def run(self, req) -> Dict[str, Any]:
        trace_id = getattr(req, "trace_id", None) or "trace-demo"                       # :304
        top_k = int(getattr(req, "top_k", None) or self.default_top_k)                  # :305
        model_hint = resolve_embed_model(self.router)
        pool = max(top_k, self.dense_pool_k)

        t_hits: List[Dict[str, Any]] = []
        g_hits: List[Dict[str, Any]] = []
        if self.text_search is not None:
            with span("Backend/TextSearch", self.sink, trace_id):
                t_hits = list(self.text_search(queries=[req.query], top_k=pool) or [])
        if self.graph_expand is not None:
            with span("Backend/GraphExpand", self.sink, trace_id):
                g_hits = list(self.graph_expand(query=req.query, graph_id=getattr(req, "graph_id", "") or "",
                                                top_k=pool) or [])

        dense_hits: List[Dict[str, Any]] = []
        dense_error = None
        with span("Backend/DenseRerank", self.sink, trace_id):      # same span name as :332
            try:
                state = self._get_state(model_hint, trace_id)
                if state["index"] is not None:
                    qv = np.asarray(_vectors(self.router.embed(model_hint=model_hint, texts=[req.query],
                                                               require={"trace_id": trace_id}))[0], dtype=np.float32)
                    if qv.shape[0] != state["dim"]:
                        raise ValueError(f"query embedding has dim {qv.shape[0]}, corpus has {state['dim']}")
                    k = min(pool, len(state["rows"]), 64)
                    sc, ids = state["index"].search(qv[None, :], k)
                    for s, i in zip(sc[0], ids[0]):
                        if i < 0:
                            continue
                        row = state["rows"][int(i)]
                        dense_hits.append({"id": _fusion.raw_hit_id(row), "score": float(s),
                                           "meta": _fusion.row_meta(row, "dense")})
            except Exception as e:     # reference policy: degrade, never raise out of run() (:229-231)
                dense_error = repr(e)
                logger.error("[DenseRetrievalBackend] dense channel failed: %s", e)

        # dense hits double as candidates (their meta gives the normalised id); scores fused per :350-372
        dense_norm_hits = _fusion.dedupe_by_norm_id(dense_hits)
        dense_scores = {nid: h["score"] for nid, h in dense_norm_hits.items()}
        fused = _fusion.fuse_channels(t_hits, g_hits, dense_scores, alpha_text=self.alpha_text,
                                      alpha_graph=self.alpha_graph, alpha_dense=self.alpha_dense, top_k=10 ** 9)
        for h in fused:                                   # dense-only ids carry their own meta
            if not h["meta"].get("text") and h["id"] in dense_norm_hits:
                meta = dict(dense_norm_hits[h["id"]]["meta"])
                meta.update({k: v for k, v in h["meta"].items() if k.startswith("score_")})
                h["meta"] = meta
        fused = fused[:top_k]
        diagnostics = {
            "queries": [req.query],
            "bm25_candidates": len(t_hits),
            "graph_candidates": len(g_hits),
            "dense_scored": len(dense_scores),
            "weights": {"alpha_text": self.alpha_text, "alpha_graph": self.alpha_graph, "alpha_dense": self.alpha_dense},
            "pool": {"dense_pool_k": self.dense_pool_k, "final_top_k": top_k},
            "resolved_embed_model": model_hint,
            "dense_backend": "mrag_amd.DenseIndex",
            "dense_error": dense_error,
        }
        return {"hits": fused, "diagnostics": diagnostics}

    def retrieve(self, req) -> RetrievalOut:
        result = self.run(req)
        hits = [Hit(id=h["id"], score=h["score"], meta=h.get("meta") or {}) for h in result["hits"]]
        return RetrievalOut(hits=hits, diagnostics=result.get("diagnostics", {}))
