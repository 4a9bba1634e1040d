"""Corpus side of the boundary: the ``docs.jsonl`` the reference's ingest writes
(my_code/ingest_hotpotqa.py:73-81, one ``{"doc_id": "{title}#{sid}", "title", "sent_id",
"text"}`` per sentence) is consumed unchanged; corpus row index = line order among the
non-blank lines (app/modules/retrieval/text_index.py:36-46).

Also the "next" row f1 of SURVEY section 8: an on-disk embedding cache keyed by (file, mtime,
size, model, dim, dtype) and a process-wide index registry -- the reference rebuilds every
module per question (app/system.py:36), so GPU state must outlive its constructors.
"""
from __future__ import annotations

import hashlib
import json
import os
import threading
from pathlib import Path
from typing import Any, Callable, Dict, List, Optional

import numpy as np


def read_docs_jsonl(path) -> List[Dict[str, Any]]:
    rows: List[Dict[str, Any]] = []
    p = Path(path)
    if not p.exists():          # text_index.py:32-33: a missing file is an empty corpus
        return rows
    with p.open("r", encoding="utf-8") as f:
        for line in f:
            line = line.strip()
            if line:
                rows.append(json.loads(line))
    return rows


def write_docs_jsonl(path, rows) -> None:
    """Same row writer as the ingest (ingest_hotpotqa.py:73-81) -- used for synthetic corpora."""
    p = Path(path)
    p.parent.mkdir(parents=True, exist_ok=True)
    with p.open("w", encoding="utf-8") as f:
        for r in rows:
            f.write(json.dumps({"doc_id": r["doc_id"], "title": r["title"], "sent_id": r["sent_id"],
                                "text": r["text"]}, ensure_ascii=False) + "\n")


def file_signature(path) -> str:
    st = os.stat(path)
    return f"{Path(path).resolve()}|{st.st_mtime_ns}|{st.st_size}"


class EmbeddingCache:
    """Embedding matrix of a docs.jsonl under ``cache_dir`` as ``<key>.npy`` + ``<key>.json``: the 16-bit
    patterns (uint16) of the index's storage dtype exactly as K1 produced them, so that an index rebuilt
    from the cache is bit-identical to the one first built from the encoder's output."""

    def __init__(self, cache_dir):
        self.dir = Path(cache_dir)

    def key(self, docs_path, model: str, dim: int, dtype: str = "f16") -> str:
        return hashlib.sha1(f"{file_signature(docs_path)}|{model}|{dim}|{dtype}".encode()).hexdigest()[:24]

    def load(self, key: str) -> Optional[np.ndarray]:
        f = self.dir / f"{key}.npy"
        if not f.exists():
            return None
        return np.load(f, mmap_mode="r", allow_pickle=False)

    def store(self, key: str, matrix: np.ndarray, info: Dict[str, Any]) -> None:
        self.dir.mkdir(parents=True, exist_ok=True)
        tmp = self.dir / f"{key}.tmp.npy"
        np.save(tmp, np.ascontiguousarray(matrix), allow_pickle=False)
        os.replace(tmp, self.dir / f"{key}.npy")
        (self.dir / f"{key}.json").write_text(json.dumps(info))


class _Slot:
    """One registry entry: the value once built, and the event its builder sets."""
    __slots__ = ("done", "value", "error", "owner")

    def __init__(self):
        self.done = threading.Event()
        self.value: Any = None
        self.error: Optional[BaseException] = None
        self.owner = threading.get_ident()


_REGISTRY: Dict[str, _Slot] = {}
_LOCK = threading.Lock()          # guards the dict only; never held while a build() runs


def shared(key: str, build: Callable[[], Any]):
    """Process-wide singleton: ``build()`` runs once per key (SURVEY 8b "Threading").

    ``build()`` runs OUTSIDE the registry lock, so a builder may itself call ``shared`` for
    another key (the corpus index builder reaches the provider's encoder through the router);
    other callers of the SAME key wait for the builder.  A failed build is not cached: the
    error propagates to everyone waiting and the next caller builds again."""
    with _LOCK:
        slot = _REGISTRY.get(key)
        mine = slot is None
        if mine:
            slot = _REGISTRY[key] = _Slot()
    if mine:
        try:
            slot.value = build()
        except BaseException as e:
            slot.error = e
            with _LOCK:
                if _REGISTRY.get(key) is slot:
                    del _REGISTRY[key]
            raise
        finally:
            slot.done.set()
        return slot.value
    if not slot.done.is_set() and slot.owner == threading.get_ident():
        raise RuntimeError(f"shared({key!r}): build() re-entered its own key")
    slot.done.wait()
    if slot.error is not None:
        raise slot.error
    return slot.value


def drop_shared(prefix: str = "") -> None:
    with _LOCK:
        slots = [_REGISTRY.pop(k) for k in [k for k in _REGISTRY if k.startswith(prefix)]]
    for slot in slots:
        slot.done.wait()
        close = getattr(slot.value, "close", None)
        if callable(close):
            try:
                close()
            except Exception:
                pass
