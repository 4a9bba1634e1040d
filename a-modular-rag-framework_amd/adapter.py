"""Boundary B3 -- a ``RetrievalAgent`` (app/core/interfaces.py:17-18) selected by
``modules.retrieval.type``; mirrors ``RetrievalAdapter`` (app/modules/retrieval/
retrieval_adapter.py:13-133): build the backend from a ``"pkg.mod:Class"`` string, call
``backend.run(req)``, normalise raw hits through the id / score alias lists, re-sort by
score descending (stable) and truncate to ``req.top_k``.
"""
from __future__ import annotations

import importlib
from typing import Any, Dict, List, Optional

from . import fusion as _fusion
from .dto import Hit, RetrievalIn, RetrievalOut
from .telemetry import span

DEFAULT_BACKEND = "mrag_amd.backend:DenseRetrievalBackend"


def import_from_string(path: str):
    """``"pkg.mod:Class"`` -> object (app/di/factory.py:12-16)."""
    mod, name = path.split(":")
    return getattr(importlib.import_module(mod), name)


class DenseRetrievalAgent:
    def __init__(self, router, sink=None, backend_impl: str = DEFAULT_BACKEND,
                 backend_kwargs: Optional[Dict[str, Any]] = None, id_keys: Optional[List[str]] = None,
                 score_keys: Optional[List[str]] = None, meta_key: Optional[str] = "meta"):
        self.router, self.sink = router, sink
        self.id_keys = list(id_keys or _fusion.ID_KEYS)
        self.score_keys = list(score_keys or _fusion.SCORE_KEYS)
        self.meta_key = meta_key
        cls = import_from_string(backend_impl)
        kw = dict(backend_kwargs or {})
        try:
            self.backend = cls(router=router, sink=sink, **kw)      # retrieval_adapter.py:38-43
        except TypeError:
            self.backend = cls(**kw)

    @classmethod
    def from_settings(cls, settings: Dict[str, Any], router, sink=None) -> "DenseRetrievalAgent":
        """Reads ``modules.retrieval`` like both reference adapters do: ``impl`` / ``impl_kwargs``
        (flow.py:83-88) or ``kwargs.backend.{impl,kwargs}`` (retrieval_adapter.py:50-62)."""
        cfg = ((settings or {}).get("modules") or {}).get("retrieval") or {}
        own = dict(cfg.get("kwargs") or {}) if isinstance(cfg, dict) else {}
        bspec = own.pop("backend", {}) or {}
        impl = (cfg.get("impl") if isinstance(cfg, dict) else None) or bspec.get("impl") or DEFAULT_BACKEND
        impl_kwargs = dict((cfg.get("impl_kwargs") if isinstance(cfg, dict) else None) or bspec.get("kwargs") or {})
        import inspect
        valid = set(inspect.signature(import_from_string(impl).__init__).parameters)       # flow.py:98-106
        impl_kwargs = {k: v for k, v in impl_kwargs.items() if k in valid and k not in ("self", "router", "sink")}
        own = {k: v for k, v in own.items() if k in ("id_keys", "score_keys", "meta_key")}
        return cls(router=router, sink=sink, backend_impl=impl, backend_kwargs=impl_kwargs, **own)

    def retrieve(self, req) -> RetrievalOut:
        trace_id = getattr(req, "trace_id", None) or "trace-demo"
        with span("RetrievalAdapter", self.sink, trace_id):
            out = self.backend.run(req)
            raw = out.get("hits", []) if isinstance(out, dict) else []
            diagnostics = out.get("diagnostics", {}) if isinstance(out, dict) else {}
            hits = []
            for r in raw:
                h = _fusion.normalize_raw_hit(r, self.id_keys, self.score_keys, self.meta_key)
                if h:
                    hits.append(Hit(id=h["id"], score=h["score"], meta=h["meta"]))
            hits.sort(key=lambda h: h.score, reverse=True)          # stable, retrieval_adapter.py:129
            if getattr(req, "top_k", None):
                hits = hits[: req.top_k]
            return RetrievalOut(hits=hits, diagnostics=diagnostics)
