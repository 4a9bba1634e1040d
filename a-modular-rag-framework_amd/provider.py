"""Boundary B1 -- the Embedder: a provider for the reference's ``LLMRouter``.

Selected exactly like the shipped providers (config/settings.yaml:2-9,16;
app/di/factory.py:32-58):

    providers:
      hip:
        type: "mrag_amd.provider:HipEmbeddingProvider"
        kwargs: { arch: "minilm-l6", model_path: "/models/all-MiniLM-L6-v2", embed_model: "all-MiniLM-L6-v2" }
    llm_policy:
      embedding_provider: hip

Called by the router as ``embed(model=, texts=, require=)`` (app/core/llm_router.py:115) and
answers ``{"vectors": [[float]*d]*n}`` like the shipped providers
(app/core/providers/openai_provider.py:96-134); the Protocol's positional form
``embed(texts, **kw)`` (app/core/providers/base.py:6) works too.  ``.kwargs["embed_model"]``
is what ``DenseReranker._resolve_embed_model`` looks up (retrieval_backend.py:206-210).

The forward pass is the HIP encoder (``mrag_amd.encoder``).  Errors are raised as Python
exceptions; the router's own handler degrades them to zero vectors exactly as it does for
the shipped providers (llm_router.py:124-129).  Vectors are never NaN.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional

import numpy as np

from . import corpus as _corpus


class HipEmbeddingProvider:
    def __init__(self, arch: str = "minilm-l6", model_path: Optional[str] = None,
                 embed_model: Optional[str] = None, device: int = 0, batch_size: int = 256,
                 max_length: Optional[int] = None, seed: int = 0, normalize: Optional[bool] = None, **extra):
        self.kwargs: Dict[str, Any] = dict(arch=arch, model_path=model_path, device=device, batch_size=batch_size,
                                           max_length=max_length, seed=seed, normalize=normalize, **extra)
        self.kwargs["embed_model"] = embed_model or (model_path.rstrip("/").split("/")[-1] if model_path else arch)
        self._encoder = None

    @classmethod
    def from_settings(cls, settings: Dict[str, Any]) -> "HipEmbeddingProvider":
        """factory.py:49-55 tries ``Cls.from_settings(settings)`` before ``Cls(**kwargs)``: find this
        class's own block under ``providers`` and use its kwargs."""
        me = f"{cls.__module__}:{cls.__name__}"
        for cfg in ((settings or {}).get("providers") or {}).values():
            if isinstance(cfg, dict) and str(cfg.get("type", "")).endswith(me.split(".")[-1]):
                return cls(**dict(cfg.get("kwargs") or {}))
        return cls()

    # -- encoder (process-wide: app/system.py:36 rebuilds providers for every question) --------
    @property
    def encoder(self):
        if self._encoder is None:
            kw = self.kwargs
            key = f"encoder|{kw['arch']}|{kw['model_path']}|{kw['device']}|{kw['seed']}"

            def build():
                from .encoder import HipSentenceEncoder
                if kw["model_path"]:
                    return HipSentenceEncoder.from_pretrained_dir(kw["model_path"], device=kw["device"],
                                                                  max_length=kw["max_length"])
                return HipSentenceEncoder.from_seed(kw["arch"], seed=kw["seed"], device=kw["device"],
                                                    max_length=kw["max_length"])
            self._encoder = _corpus.shared(key, build)
        return self._encoder

    @property
    def dim(self) -> int:
        return self.encoder.spec.hidden

    # -- the provider protocol -----------------------------------------------------------------
    def embed_array(self, texts: List[str]) -> np.ndarray:
        """[n, d] float32, L2-normalised rows (zero rows for empty input text never occur: the
        tokenizer always emits [CLS] [SEP])."""
        if not texts:
            return np.zeros((0, self.dim), dtype=np.float32)
        enc = self.encoder
        norm = self.kwargs["normalize"]      # None: what the model directory's pipeline does (Normalize module), else True
        return enc.encode([t if isinstance(t, str) else str(t) for t in texts], batch_size=int(self.kwargs["batch_size"]),
                          normalize=bool(enc.normalize_default if norm is None else norm))

    def embed_device(self, texts: List[str], batch_size: Optional[int] = None):
        """Bulk form for corpus ingest (``DenseRetrievalBackend``'s fast path): [n, d] float32 rows as a CUDA tensor in
        input order, embedded in large length-sorted batches (default 4096 passages) -- no ``.tolist()`` round trip,
        no host copy; the rows go device -> device into ``DenseIndex.add``."""
        enc = self.encoder
        norm = self.kwargs["normalize"]
        return enc.encode_device([t if isinstance(t, str) else str(t) for t in texts],
                                 batch_size=int(batch_size or self.kwargs.get("bulk_batch_size") or 4096),
                                 normalize=bool(enc.normalize_default if norm is None else norm))

    def embed(self, texts: Optional[List[str]] = None, *, model: Optional[str] = None,
              require: Optional[Dict[str, Any]] = None, **kw) -> Dict[str, Any]:
        vecs = self.embed_array(list(texts or []))
        return {"vectors": vecs.astype(np.float64).tolist(), "model": model or self.kwargs["embed_model"],
                "dim": int(vecs.shape[1])}

    def complete(self, prompt: str = "", *, model: Optional[str] = None, require: Optional[Dict[str, Any]] = None,
                 **kw) -> Dict[str, Any]:
        """Embedding-only provider.  Same shape as the shipped providers' mock answers
        (app/core/providers/ollama_provider.py:24) so a mis-routed completion degrades, not crashes."""
        return {"text": f"[HipEmbeddingProvider: no completion model] {str(prompt)[:200]}", "tokens": 0}
