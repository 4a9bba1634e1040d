"""Drop-in check against the REAL reference classes (CPU, this container only: skipped wherever
/root/reference is absent, e.g. on the GPU box).  The reference's own factory / router / adapter are
driven with this repo's provider and backend selected by "pkg.mod:Class" strings, exactly as
config/settings.yaml would."""
import sys
from pathlib import Path

import pytest

REF = Path("/root/reference")
pytestmark = pytest.mark.skipif(not (REF / "app" / "di" / "factory.py").exists(), reason="reference checkout not present")


@pytest.fixture()
def ref_path(monkeypatch):
    monkeypatch.syspath_prepend(str(REF))
    monkeypatch.setenv("PYTHONDONTWRITEBYTECODE", "1")
    sys.dont_write_bytecode = True
    yield
    for m in [m for m in sys.modules if m == "app" or m.startswith("app.")]:
        sys.modules.pop(m, None)


def test_reference_factory_builds_our_provider_and_router_calls_it(ref_path, monkeypatch):
    from app.di.factory import build_providers, build_router         # reference code
    from mrag_amd.provider import HipEmbeddingProvider
    settings = {"providers": {"hip": {"type": "mrag_amd.provider:HipEmbeddingProvider",
                                      "kwargs": {"arch": "tiny", "embed_model": "tiny-seed0", "seed": 0}}},
                "llm_policy": {"embedding_provider": "hip"}}
    providers = build_providers(settings)                             # factory.py:32-58 (tries from_settings first)
    assert isinstance(providers["hip"], HipEmbeddingProvider)
    assert providers["hip"].kwargs["embed_model"] == "tiny-seed0"
    router = build_router(settings, providers)
    seen = {}

    def fake_embed_array(self, texts):                               # no GPU here: stub only the forward pass
        import numpy as np
        seen["texts"] = list(texts)
        return np.ones((len(texts), 4), dtype=np.float32)
    monkeypatch.setattr(HipEmbeddingProvider, "embed_array", fake_embed_array)
    out = router.embed(model_hint="tiny-seed0", texts=["a", "b"], require={"trace_id": "t"})   # llm_router.py:115
    assert seen["texts"] == ["a", "b"] and len(out["vectors"]) == 2 and out["vectors"][0] == [1.0] * 4
    from app.modules.retrieval.retrieval_backend import DenseReranker                          # reference re-ranker
    assert DenseReranker(router)._resolve_embed_model() == "tiny-seed0"                       # via provider.kwargs
    scores = DenseReranker(router, max_pool=5, embed_batch=8).score(
        query="q", candidates=[{"id": "x", "score": 1.0, "meta": {"text": "hello"}}], trace_id="t")
    assert scores == {"x": pytest.approx(1.0)}


def test_reference_adapter_drives_our_backend(ref_path, tmp_path):
    from app.core.dto import RetrievalIn, RetrievalOut
    from app.core.llm_router import LLMRouter
    from app.modules.retrieval.retrieval_adapter import RetrievalAdapter   # the reference's own adapter
    router = LLMRouter(providers={}, policy={})
    ad = RetrievalAdapter(router=router, backend_impl="mrag_amd.backend:DenseRetrievalBackend",
                          backend_kwargs={"index_path": str(tmp_path / "no-docs.jsonl")})
    out = ad.retrieve(RetrievalIn(query="q", graph_id="", top_k=3, trace_id="t"))
    assert isinstance(out, RetrievalOut) and out.hits == []
    assert out.diagnostics["resolved_embed_model"] == "text-embedding-3-large" and out.diagnostics["dense_error"] is None
    # our agent returns the REFERENCE's DTO classes when they are importable
    import importlib, mrag_amd.dto as d
    importlib.reload(d)
    assert d.USING_REFERENCE_DTO and d.RetrievalOut is RetrievalOut
    importlib.reload(d)
