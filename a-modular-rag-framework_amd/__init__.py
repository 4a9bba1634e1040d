"""MI355X-native dense-retrieval hot path behind the Retriever / Embedder plug-in API of
AndyUkJ/A-Modular-RAG-Framework.  Import as ``mrag_amd`` (see ``mrag_amd/__init__.py``).

Layout
  csrc/            HIP kernels + the C ABI (include/mrag.h) -> libmrag_hip.so
  _native.py       ctypes binding of the C ABI (fails loudly when the .so is missing)
  index.py         DenseIndex / IVFFlatIndex: HBM-resident corpus, cosine top-k
  encoder.py       HipSentenceEncoder: BERT-family forward in HIP
  provider.py      HipEmbeddingProvider  -- boundary B1 (LLMProvider.embed slot)
  backend.py       DenseRetrievalBackend / HipDenseReranker -- boundary B2
  adapter.py       DenseRetrievalAgent -- boundary B3 (RetrievalAgent.retrieve)
  corpus.py        docs.jsonl reader, id table, on-disk embedding cache
  sharded.py       row-sharded search: RCCL all-gather of partial top-k + host merge
  dto.py           RetrievalIn / Hit / RetrievalOut shapes
"""
__version__ = "0.1.0"
