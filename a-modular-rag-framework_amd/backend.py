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

    def __init__(self, router, max_pool: int = 200, embed_batch: int = 50, device: int = 0, corpus=None):
        """``corpus`` (optional): a :class:`DenseRetrievalBackend` over the docs.jsonl the candidates come
        from.  Candidates whose id is a row of its index are then scored by row id against the stored
        embedding -- no embed call for them (the reference pays 1 + ceil(N/bs) embedding
        round-trips per question, retrieval_backend.py:227-243); candidates it does not know, or whose
        text differs from the stored row's, are embedded as before."""
        self.router, self.max_pool, self.embed_batch, self.device = router, max_pool, embed_batch, device
        self.corpus = corpus

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
        known: Dict[int, float] = {}                # candidate position -> score from the stored row
        if self.corpus is not None:
            known = self._lookup(ids, texts, qv, model_hint, trace_id)
        todo = [i for i in range(len(texts)) if i not in known]
        vecs: List[Any] = []
        step = max(8, int(self.embed_batch))        # :234
        for lo in range(0, len(todo), step):
            chunk = [texts[i] for i in todo[lo:lo + step]]
            try:
                got = _vectors(self.router.embed(model_hint=model_hint, texts=chunk, require={"trace_id": trace_id}))
                vecs.extend(got or [])
            except Exception as e:                  # :241-243
                logger.error("[HipDenseReranker] batch embed error: %s", e)
                vecs.extend([[0.0] * len(qv) for _ in chunk])
        n = min(len(todo), len(vecs))               # zip() truncation of :245
        scored = dict(zip(todo[:n], self._cosines(qv, vecs[:n])))
        scored.update(known)
        return {ids[i]: scored[i] for i in range(len(ids)) if i in scored}

    def _lookup(self, ids: List[str], texts: List[str], qv, model_hint: str, trace_id: str) -> Dict[int, float]:
        """Scores of the candidates that are rows of ``self.corpus``'s index (same embedding model, same
        text), by row id on the GPU.  Any failure degrades to "nothing known" -- the embed path then
        scores everything, as the reference does."""
        try:
            state = self.corpus._get_state(model_hint, trace_id)
            ix, row_of, rows = state["index"], state["row_of"], state["rows"]
            if ix is None or len(qv) != state["dim"]:
                return {}
            pos, rid = [], []
            for i, (hid, text) in enumerate(zip(ids, texts)):
                r = row_of.get(hid)
                if r is not None and (rows[r].get("text") or "") == text:
                    pos.append(i)
                    rid.append(r)
            if not pos:
                return {}
            sc = ix.score_rows(np.asarray(qv, dtype=np.float32), np.asarray(rid, dtype=np.int64))
            return {i: float(v) for i, v in zip(pos, sc)}
        except Exception as e:
            logger.error("[HipDenseReranker] row lookup failed, embedding every candidate: %s", e)
            return {}


class DenseRetrievalBackend:
    """Corpus-wide dense retrieval behind ``run`` / ``retrieve`` (see module docstring)."""

    def __init__(self, router, sink=None, index_path: str = "data/hotpotqa/docs.jsonl",
                 alpha_text: float = 0.4, alpha_graph: float = 0.2, alpha_dense: float = 0.4,
                 default_top_k: int = 20, dense_pool_k: int = 200, embed_batch: int = 256,
                 cache_dir: Optional[str] = None, device: int = 0, index_dtype: str = "f16",
                 text_search: Optional[Callable[..., List[Dict[str, Any]]]] = None,
                 graph_expand: Optional[Callable[..., List[Dict[str, Any]]]] = None,
                 text_channel: Optional[str] = None, bm25_k1: float = 1.5, bm25_b: float = 0.75,
                 fuse_on_device: bool = False, bulk_ingest: bool = True, bulk_batch: int = 4096):
        self.router, self.sink = router, sink
        self.index_path = index_path
        self.alpha_text, self.alpha_graph, self.alpha_dense = float(alpha_text), float(alpha_graph), float(alpha_dense)
        self.default_top_k, self.dense_pool_k, self.embed_batch = int(default_top_k), int(dense_pool_k), int(embed_batch)
        self.cache_dir, self.device, self.index_dtype = cache_dir, int(device), index_dtype
        self.text_search, self.graph_expand = text_search, graph_expand
        # text_channel="bm25": the reference's BM25 channel (text_index.py, retrieval_backend.py:102-128) over the same
        # docs.jsonl, on the device (mrag_amd.text_index), built once per file and shared process-wide
        self.text_channel, self.bm25_k1, self.bm25_b = text_channel, float(bm25_k1), float(bm25_b)
        self.fuse_on_device = bool(fuse_on_device)     # a7's arithmetic in one launch (mrag_fuse_topk) instead of host Python
        # corpus ingest: when the provider the router resolves can embed in bulk on the device (``embed_device``), the
        # index is built from docs.jsonl through it in ``bulk_batch``-passage batches (text_index.py:32-53: the reference
        # builds its index from the file in one go); the router path below stays as the fallback / error-policy reference
        self.bulk_ingest, self.bulk_batch = bool(bulk_ingest), int(bulk_batch)
        self.last_build: Dict[str, Any] = {}
        if text_channel not in (None, "bm25"):
            raise ValueError(f"unknown text_channel {text_channel!r} (supported: 'bm25')")
        self._state = None

    # -- corpus index: built once per (file signature, model), shared process-wide --------------
    def _embed_texts(self, texts: List[str], model_hint: str, trace_id: str, dim: Optional[int] = None) -> np.ndarray:
        """Embed through the router, one call per ``embed_batch`` texts, and VALIDATE every batch.  The
        router answers ``[[0.0]*3]*n`` instead of raising when the provider fails (llm_router.py:119-129):
        such a batch (wrong row count, a dim that differs from the first batch / ``dim``, or all zeros)
        must never be built into -- or cached as -- the corpus index, so it raises here."""
        out = []
        for lo in range(0, len(texts), self.embed_batch):
            chunk = texts[lo:lo + self.embed_batch]
            got = np.asarray(_vectors(self.router.embed(model_hint=model_hint, texts=chunk,
                                                        require={"trace_id": trace_id})), dtype=np.float32)
            if got.ndim != 2 or got.shape[0] != len(chunk):
                raise ValueError(f"embed batch at row {lo}: expected {len(chunk)} vectors, got shape {got.shape}")
            if dim is None:
                dim = int(got.shape[1])
            if got.shape[1] != dim or dim == 0:
                raise ValueError(f"embed batch at row {lo}: dim {got.shape[1]}, expected {dim}")
            if not np.isfinite(got).all():
                raise ValueError(f"embed batch at row {lo}: non-finite values")
            if not got.any():
                raise ValueError(f"embed batch at row {lo}: all-zero vectors (the router's failure fallback)")
            out.append(got)
        return np.concatenate(out, axis=0) if out else np.zeros((0, dim or 0), dtype=np.float32)

    def _bulk_provider(self):
        """The provider ``router.embed`` would call (llm_router.py:103-115 / retrieval_backend.py:203-206:
        ``router.providers[policy["embedding_provider"]]``) when it offers the bulk device form, else None."""
        if not self.bulk_ingest:
            return None
        try:
            policy = getattr(self.router, "policy", {}) or {}
            name = policy.get("embedding_provider")
            prov = (getattr(self.router, "providers", {}) or {}).get(name) if name else None
            return prov if prov is not None and hasattr(prov, "embed_device") else None
        except Exception:
            return None

    def _bulk_build(self, prov, ix, texts: List[str], dim: int) -> None:
        """texts -> encoder (large length-sorted batches) -> DenseIndex.add, device to device.  Validated like the
        router path (row count, dim, finite, not all-zero) chunk by chunk BEFORE the rows enter the index."""
        import torch
        step = 65536
        for lo in range(0, len(texts), step):
            chunk = texts[lo:lo + step]
            emb = prov.embed_device(chunk, batch_size=self.bulk_batch)
            if tuple(emb.shape) != (len(chunk), dim):
                raise ValueError(f"bulk embed at row {lo}: expected {(len(chunk), dim)}, got {tuple(emb.shape)}")
            ok = torch.stack([torch.isfinite(emb).all(), emb.any()]).cpu().tolist()      # one sync per 65 536 rows
            if not ok[0]:
                raise ValueError(f"bulk embed at row {lo}: non-finite values")
            if not ok[1]:
                raise ValueError(f"bulk embed at row {lo}: all-zero vectors")
            ix.add(emb)                                                                    # K1: fp64 norm, ONE rounding

    def _build_state(self, model_hint: str, trace_id: str):
        import time
        from .index import DenseIndex
        t_start = time.perf_counter()
        rows = _corpus.read_docs_jsonl(self.index_path)
        state = {"rows": rows, "index": None, "dim": 0, "row_of": {}}
        if not rows:
            return state
        texts = [r.get("text") or "" for r in rows]
        probe = self._embed_texts(texts[:1], model_hint, trace_id)
        dim = int(probe.shape[1])
        cache = _corpus.EmbeddingCache(self.cache_dir) if self.cache_dir else None
        bits, key = None, None
        if cache is not None:
            key = cache.key(self.index_path, model_hint, dim, self.index_dtype)
            bits = cache.load(key)
            if bits is not None and (bits.shape != (len(rows), dim) or bits.dtype != np.uint16):
                bits = None
        ix = DenseIndex(dim, metric="cosine", dtype=self.index_dtype, device=self.device)
        step = 131072
        if bits is not None:
            # warm start: the cache holds the 16-bit patterns K1 produced in the cold build; they are
            # re-added verbatim, so cold and warm indexes are bit-identical (doc-id order near ties
            # cannot depend on the cache state)
            for lo in range(0, len(rows), step):
                ix.add_stored_bits(np.ascontiguousarray(bits[lo:lo + step]))
        else:
            prov, how = self._bulk_provider(), "router"
            if prov is not None:
                try:
                    self._bulk_build(prov, ix, texts, dim)
                    how = "bulk"
                except Exception as e:          # the router path is the reference behaviour: fall back to it, loudly
                    logger.error("[DenseRetrievalBackend] bulk ingest failed (%s); rebuilding through router.embed", e)
                    ix.close()
                    ix = DenseIndex(dim, metric="cosine", dtype=self.index_dtype, device=self.device)
            if how == "router":
                emb = self._embed_texts(texts, model_hint, trace_id, dim=dim)
                for lo in range(0, len(rows), step):
                    ix.add(np.ascontiguousarray(emb[lo:lo + step]))       # K1: fp64 norm, ONE rounding
            state["ingest"] = how
            if cache is not None:
                cache.store(key, ix.stored_bits(), {"model": model_hint, "rows": len(rows), "dim": dim,
                                                    "dtype": self.index_dtype, "format": "storage bits (uint16)",
                                                    "docs": _corpus.file_signature(self.index_path)})
        state["index"], state["dim"] = ix, dim
        # raw hit id (retrieval_backend.py:116-119) -> corpus row, for the re-ranker's lookup by id; ids that
        # occur twice (duplicate doc_id lines) are ambiguous and left out: such candidates are embedded as before
        seen: Dict[str, int] = {}
        for r in rows:
            hid = _fusion.raw_hit_id(r)
            seen[hid] = seen.get(hid, 0) + 1
        row_of: Dict[str, int] = {}
        for i, r in enumerate(rows):
            hid = _fusion.raw_hit_id(r)
            if seen[hid] == 1:
                row_of[hid] = i
        state["row_of"] = row_of
        self.last_build = {"rows": len(rows), "dim": dim, "ingest": state.get("ingest", "cache"),
                           "seconds": time.perf_counter() - t_start}
        return state

    def _bm25_searcher(self):
        from .text_index import HipBM25Index, HipBM25TextSearcher
        try:
            sig = _corpus.file_signature(self.index_path)
        except OSError:
            sig = f"{self.index_path}|missing"
        key = f"bm25-index|{sig}|{self.device}|{self.bm25_k1}|{self.bm25_b}"
        ix = _corpus.shared(key, lambda: HipBM25Index(self.index_path, k1=self.bm25_k1, b=self.bm25_b, device=self.device))
        return HipBM25TextSearcher(ix)

    _FAIL_BACKOFF_S = 30.0
    _failed: Dict[str, Any] = {}        # key -> (monotonic deadline, error text): no re-embed storm after a failed build

    def _get_state(self, model_hint: str, trace_id: str):
        if self._state is None:
            import time
            try:
                sig = _corpus.file_signature(self.index_path)
            except OSError:
                sig = f"{self.index_path}|missing"
            key = f"dense-index|{sig}|{model_hint}|{self.device}|{self.index_dtype}"
            bad = DenseRetrievalBackend._failed.get(key)
            if bad and time.monotonic() < bad[0]:
                raise RuntimeError(f"corpus index build failed {self._FAIL_BACKOFF_S:.0f}s ago or less: {bad[1]}")
            try:
                self._state = _corpus.shared(key, lambda: self._build_state(model_hint, trace_id))
            except Exception as e:
                DenseRetrievalBackend._failed[key] = (time.monotonic() + self._FAIL_BACKOFF_S, repr(e))
                raise
            DenseRetrievalBackend._failed.pop(key, None)
        return self._state

    # -- the backend protocol ---------------------------------------------------------------------
    def run(self, req) -> Dict[str, Any]:
        trace_id = getattr(req, "trace_id", None) or "trace-demo"                       # :304
        top_k = int(getattr(req, "top_k", None) or self.default_top_k)                  # :305
        model_hint = resolve_embed_model(self.router)
        pool = max(top_k, self.dense_pool_k)

        t_hits: List[Dict[str, Any]] = []
        g_hits: List[Dict[str, Any]] = []
        text_error = None
        if self.text_search is None and self.text_channel == "bm25":
            try:
                self.text_search = self._bm25_searcher()
            except Exception as e:     # degrade like the dense channel: report, keep serving the other channels
                text_error = repr(e)
                logger.error("[DenseRetrievalBackend] BM25 channel unavailable: %s", e)
        if self.text_search is not None:
            with span("Backend/TextSearch", self.sink, trace_id):
                t_hits = list(self.text_search(queries=[req.query], top_k=pool) or [])
        if self.graph_expand is not None:
            with span("Backend/GraphExpand", self.sink, trace_id):
                g_hits = list(self.graph_expand(query=req.query, graph_id=getattr(req, "graph_id", "") or "",
                                                top_k=pool) or [])

        dense_hits: List[Dict[str, Any]] = []
        dense_error = None
        k_eff = 0
        with span("Backend/DenseRerank", self.sink, trace_id):      # same span name as :332
            try:
                state = self._get_state(model_hint, trace_id)
                if state["index"] is not None:
                    qv = np.asarray(_vectors(self.router.embed(model_hint=model_hint, texts=[req.query],
                                                               require={"trace_id": trace_id}))[0], dtype=np.float32)
                    if qv.shape[0] != state["dim"]:
                        raise ValueError(f"query embedding has dim {qv.shape[0]}, corpus has {state['dim']}")
                    # the reference scores its whole pool (retrieval_backend.py:218,245); the index serves
                    # k up to max_k() (256) -- anything beyond is reported, never silently dropped
                    k = min(pool, len(state["rows"]))
                    k_eff = min(k, state["index"].max_k(1))
                    sc, ids = state["index"].search(qv[None, :], k_eff)
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
        fuse = _fusion.fuse_channels
        if self.fuse_on_device and len(t_hits) + len(g_hits) + len(dense_scores) <= 4096:
            fuse = lambda *a, **kw: _fusion.fuse_channels_device(*a, device=self.device, **kw)    # noqa: E731
        fused = fuse(t_hits, g_hits, dense_scores, alpha_text=self.alpha_text,
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
            "pool": {"dense_pool_k": self.dense_pool_k, "final_top_k": top_k, "dense_pool_k_effective": k_eff},
            "resolved_embed_model": model_hint,
            "dense_backend": "mrag_amd.DenseIndex",
            "dense_error": dense_error,
            "text_error": text_error,
        }
        return {"hits": fused, "diagnostics": diagnostics}

    def retrieve(self, req) -> RetrievalOut:
        result = self.run(req)
        hits = [Hit(id=h["id"], score=h["score"], meta=h.get("meta") or {}) for h in result["hits"]]
        return RetrievalOut(hits=hits, diagnostics=result.get("diagnostics", {}))
