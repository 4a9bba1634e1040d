"""Contract types of the retrieval boundary (reference: app/core/dto.py:38-55).

``RetrievalIn{query, graph_id, top_k=20, trace_id}``, ``Hit{id, score, meta}``,
``RetrievalOut{hits, diagnostics}``.  When the reference package is importable (the
drop-in case: this code runs inside the reference app) its own pydantic models are used,
so ``isinstance`` checks and serialisers on the orchestrator side keep working; otherwise
field-for-field equivalents are defined here.
"""
from __future__ import annotations

from typing import Any, Dict, List

try:  # pragma: no cover - exercised only inside the reference app
    from app.core.dto import Hit, RetrievalIn, RetrievalOut  # type: ignore
    USING_REFERENCE_DTO = True
except Exception:  # reference not on sys.path (tests, GPU box, bench)
    from pydantic import BaseModel, Field

    USING_REFERENCE_DTO = False

    class RetrievalIn(BaseModel):
        query: str
        graph_id: str
        top_k: int = 20
        trace_id: str

    class Hit(BaseModel):
        id: str
        score: float
        meta: Dict[str, Any] = Field(default_factory=dict)

    class RetrievalOut(BaseModel):
        hits: List[Hit] = Field(default_factory=list)
        diagnostics: Dict[str, Any] = Field(default_factory=dict)

__all__ = ["RetrievalIn", "Hit", "RetrievalOut", "USING_REFERENCE_DTO"]
