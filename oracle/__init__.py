"""CPU oracle for the dense-retrieval hot path.  TEST INFRASTRUCTURE ONLY.

Everything under ``oracle/`` is a CPU restatement of the reference's algorithm
(AndyUkJ/A-Modular-RAG-Framework) for the path named in BASELINE.json, each
function citing the reference file:line it follows.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it -- as the checker, never as the thing measured or shipped.  The product
package (``a-modular-rag-framework_amd`` / ``mrag_amd``) never imports it.

Pinning:
  * ``ref_semantics`` is pinned by golden fixtures F1-F4 captured from the
    reference's own functions (``tests/golden/make_golden.py``).
  * ``dense_search.brute_force_topk`` is pinned by F5 (reference ``_cosine``
    run over the C1 shape).
  * ``dense_search.ivf_*`` and ``encoder`` have no counterpart in the
    reference (it ships neither an IVF index nor an encoder): PARITY UNPINNED
    by the reference; pinned only against this repo's own brute force / the
    container's ``transformers.BertModel``.
"""
