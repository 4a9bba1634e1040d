#!/usr/bin/env python3
"""F7: BM25 golden vectors from the REFERENCE's own BM25LiteIndex / BM25TextSearcher
(app/modules/retrieval/text_index.py, retrieval_backend.py:102-128) on the F3 corpus (the 193 HotpotQA-shaped
rows kept in f3_hybrid_run.json) plus a second corpus with repeated tokens, empty texts and a unicode row.

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference python tests/golden/make_golden_bm25.py

The script only CALLS the reference; the fixture holds inputs and its outputs (every candidate, full ranking).
"""
import json
import sys
import tempfile
from pathlib import Path

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent.parent))

from app.modules.retrieval.text_index import BM25LiteIndex  # noqa: E402  (reference)
from app.modules.retrieval.retrieval_backend import BM25TextSearcher  # noqa: E402


def write_jsonl(path, rows):
    with open(path, "w", encoding="utf-8") as f:
        for r in rows:
            f.write(json.dumps(r, ensure_ascii=False) + "\n")


def capture(rows, query_sets, k1=1.5, b=0.75):
    with tempfile.TemporaryDirectory() as td:
        p = Path(td) / "docs.jsonl"
        write_jsonl(p, rows)
        ix = BM25LiteIndex(str(p), k1=k1, b=b)
        out = {"rows": rows, "k1": k1, "b": b, "N": ix.N, "avgdl": ix.avgdl, "doc_lens": ix.doc_lens,
               "df": dict(sorted(ix.df.items())), "cases": []}
        for queries, top_k, merge in query_sets:
            ranked = ix.search(queries, top_k=top_k, alpha_merge=merge)
            full = ix.search(queries, top_k=10 ** 9, alpha_merge=merge)
            case = {"queries": queries, "top_k": top_k, "alpha_merge": merge,
                    "ranked": [[int(d), float(s)] for d, s in ranked], "full": [[int(d), float(s)] for d, s in full]}
            if merge == "max":
                case["hits"] = BM25TextSearcher(index=ix).search(queries=queries, top_k=top_k)
            out["cases"].append(case)
        return out


def main():
    f3 = json.loads((HERE / "f3_hybrid_run.json").read_text())
    rows_a = f3["docs_rows"]
    sets_a = [(["alpha river city"], 20, "max"),
              (["alpha river city", "river born king", "film band album war"], 50, "max"),
              (["alpha river city", "river born king"], 50, "sum"),
              (["born born born city"], 10, "max"),                     # repeated query token adds repeatedly (:62)
              (["zzzunknown alpha", ""], 10, "max"),                     # unknown token, empty query
              (["the quick brown fox"], 10, "max"),                      # no candidate at all
              (["king"], 200, "max")]
    rows_b = [{"doc_id": f"D{i}#{i % 3}", "title": f"D{i}", "sent_id": i % 3, "text": t} for i, t in enumerate([
        "apple apple apple banana", "banana cherry", "", "apple", "cherry cherry banana apple apple",
        "Café naïve apple-banana_cherry 42", "42 42 answers", "APPLE Banana", "x", "apple banana cherry date egg fig"])]
    sets_b = [(["apple banana"], 10, "max"), (["apple", "banana", "cherry 42"], 4, "max"), (["apple", "apple banana"], 10, "sum"),
              (["cafe caf na ve"], 10, "max"), (["x"], 1, "max")]
    out = {"f3_corpus": capture(rows_a, sets_a), "small_corpus": capture(rows_b, sets_b, k1=1.2, b=0.5)}
    dst = HERE / "f7_bm25.json"
    dst.write_text(json.dumps(out, ensure_ascii=False))
    print("wrote", dst, dst.stat().st_size, "bytes;", sum(len(c["cases"]) for c in out.values()), "cases")


if __name__ == "__main__":
    main()
