"""Trace spans with the reference's event shapes (app/telemetry/sinks.py:105-116): a sink
is anything with ``record(evt: dict)``; ``span`` emits ``node_start`` then ``node_end``
(with ``duration_sec``) or ``error`` (re-raised).  ``sink=None`` is a no-op, like the
reference backends' ``_null_ctx`` (retrieval_backend.py:394-397)."""
from __future__ import annotations

import contextlib
import time


@contextlib.contextmanager
def span(node: str, sink, trace_id: str):
    if sink is None:
        yield
        return
    t0 = time.time()
    sink.record({"trace_id": trace_id, "ts": t0, "event": "node_start", "node": node, "status": "running",
                 "payload": {}})
    try:
        yield
    except Exception as e:
        t1 = time.time()
        sink.record({"trace_id": trace_id, "ts": t1, "event": "error", "node": node, "status": "error",
                     "duration_sec": t1 - t0, "error": repr(e), "payload": {}})
        raise
    t1 = time.time()
    sink.record({"trace_id": trace_id, "ts": t1, "event": "node_end", "node": node, "status": "ok",
                 "duration_sec": t1 - t0, "payload": {}})
