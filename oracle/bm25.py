"""CPU oracle of the BM25 text channel.  TEST INFRASTRUCTURE (oracle/__init__.py): only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.

Restates ``BM25LiteIndex`` (app/modules/retrieval/text_index.py) and the hit shaping of
``BM25TextSearcher.search`` (app/modules/retrieval/retrieval_backend.py:106-128) in plain Python, fp64,
same operation order.  PINNED by fixture F7 (tests/golden/f7_bm25.json, captured from the reference's
own classes by tests/golden/make_golden_bm25.py).

Declared tie-break where the reference's order is the iteration order of a ``set`` of ints
(text_index.py:79,88,95): equal scores are ordered by ascending document index.
"""
from __future__ import annotations

import math
import re
from typing import Any, Dict, List, Sequence, Tuple


def tokenize(text: str) -> List[str]:
    """text_index.py:10-11."""
    return [t for t in re.split(r"[^a-zA-Z0-9]+", (text or "").lower()) if t]


class Bm25Oracle:
    def __init__(self, rows: Sequence[Dict[str, Any]], k1: float = 1.5, b: float = 0.75):
        """text_index.py:36-52 over already-parsed docs.jsonl rows."""
        self.k1, self.b = k1, b
        self.docs = list(rows)
        self.tf: Dict[str, Dict[int, int]] = {}
        self.doc_lens: List[int] = []
        for i, obj in enumerate(self.docs):
            toks = tokenize(obj.get("text", ""))
            self.doc_lens.append(len(toks))
            for t in toks:
                bucket = self.tf.setdefault(t, {})
                bucket[i] = bucket.get(i, 0) + 1
        self.N = len(self.docs)
        self.avgdl = (sum(self.doc_lens) / self.N) if self.N else 0.0
        self.df = {t: len(p) for t, p in self.tf.items()}

    def idf(self, term: str) -> float:
        """text_index.py:54-56."""
        n = self.df.get(term, 0)
        return math.log((self.N - n + 0.5) / (n + 0.5) + 1.0) if self.N > 0 else 0.0

    def score_doc(self, q_terms: Sequence[str], d: int) -> float:
        """text_index.py:59-69: every token of the query (repeats included) adds its term, left to right."""
        score = 0.0
        dl = self.doc_lens[d] if d < len(self.doc_lens) else 0
        for t in q_terms:
            f = self.tf.get(t, {}).get(d, 0)
            if f == 0:
                continue
            denom = f + self.k1 * (1 - self.b + self.b * (dl / (self.avgdl or 1.0)))
            score += self.idf(t) * (f * (self.k1 + 1)) / (denom or 1.0)
        return score

    def search(self, queries: Sequence[str], top_k: int = 20, alpha_merge: str = "max") -> List[Tuple[int, float]]:
        """text_index.py:71-97; ties ordered by ascending doc (declared, see module docstring)."""
        if not self.N:
            return []
        q_terms_list = [tokenize(q) for q in queries]
        cands = set()
        for q_terms in q_terms_list:
            for t in set(q_terms):
                cands.update(self.tf.get(t, {}).keys())
        scores: Dict[int, float] = {}
        for d in cands:
            s_list = [self.score_doc(q_terms, d) for q_terms in q_terms_list]
            s = sum(s_list) if alpha_merge == "sum" else (max(s_list) if s_list else 0.0)
            if s > 0:
                scores[d] = s
        return sorted(scores.items(), key=lambda kv: (-kv[1], kv[0]))[:top_k]

    def hits(self, queries: Sequence[str], top_k: int) -> List[Dict[str, Any]]:
        """BM25TextSearcher.search, retrieval_backend.py:106-128."""
        out = []
        for d, s in self.search(queries, top_k=top_k, alpha_merge="max"):
            meta = dict(self.docs[d])
            out.append({"id": "sent::%s::%s" % (meta.get("doc_id") or meta.get("title") or "doc", str(meta.get("sent_id") or "")),
                        "score": float(s),
                        "meta": {"kind": "sentence", "text": meta.get("text"), "doc": meta.get("title"),
                                 "sent_id": meta.get("sent_id"), "source": "bm25"}})
        return out
