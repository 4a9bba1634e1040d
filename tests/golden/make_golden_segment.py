#!/usr/bin/env python3
"""F8: embed-mode / rule-mode segmentation golden vectors from the REFERENCE's own segment_context
(app/modules/graph_construction/segmenter.py:10-57), driven with a table embed_fn (the vectors are stored in
the fixture).

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference python tests/golden/make_golden_segment.py
"""
import json
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
from app.modules.graph_construction.segmenter import segment_context  # noqa: E402  (reference)


def main():
    rng = np.random.default_rng(12)
    topics = rng.standard_normal((4, 24))
    ctx, table = [], {}
    for t in range(5):
        sents = []
        for i in range(int(rng.integers(1, 8))):
            s = f"Title{t} sentence {i}. It has two parts! Really?"
            topic = topics[(t + i // 2) % 4]
            table[s] = [float(x) for x in topic + 0.35 * rng.standard_normal(24)]
            sents.append(s)
        ctx.append((f"Title {t}", sents))
    ctx.append(("Empty", []))
    table["zero"] = [0.0] * 24
    ctx.append(("Zeros", ["zero", "zero", list(table)[0]]))      # zero vectors: 0 / (0 + 1e-9) = 0 -> always cut
    cases = []
    for strategy, thr in (("embed", 0.65), ("embed", 0.2), ("embed", 0.95), ("embed", -1.0), ("rule", 0.65), ("other", 0.65)):
        out = segment_context([(t, list(s)) for t, s in ctx], strategy=strategy, embed_fn=lambda s: table[s], sim_threshold=thr)
        cases.append({"strategy": strategy, "sim_threshold": thr, "out": [[t, s] for t, s in out]})
    no_fn = segment_context([(t, list(s)) for t, s in ctx], strategy="embed", embed_fn=None)
    cases.append({"strategy": "embed", "sim_threshold": 0.65, "no_embed_fn": True, "out": [[t, s] for t, s in no_fn]})
    dst = HERE / "f8_segment.json"
    dst.write_text(json.dumps({"ctx": [[t, s] for t, s in ctx], "table": table, "cases": cases}))
    print("wrote", dst, dst.stat().st_size, "bytes")


if __name__ == "__main__":
    main()
