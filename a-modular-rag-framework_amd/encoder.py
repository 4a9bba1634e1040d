"""HIP sentence encoder behind the provider boundary (a5: the ``LLMProvider.embed`` slot,
app/core/providers/base.py:6).  BERT-family forward in HIP (csrc/encoder.hip) through the C ABI;
tokenisation stays on the host.

Weights: a local HF directory (``config.json`` + ``model.safetensors`` / ``pytorch_model.bin``
loaded with ``weights_only=True`` + ``vocab.txt``) when one exists on the box; otherwise seeded
synthetic parameters of the named architecture and a hashing tokenizer -- the container has no
checkpoints and no network.
"""
from __future__ import annotations

import ctypes as C
import json
import re
import zlib
from pathlib import Path
from typing import Dict, List, Optional

import numpy as np

from . import _native as N

# public model-card shapes (not in the reference); overridden by config.json when weights are local
ARCHS = {
    "minilm-l6": dict(vocab_size=30522, hidden=384, layers=6, heads=12, intermediate=1536, max_position=512,
                      type_vocab_size=2, layer_norm_eps=1e-12, pool="mean", max_length=256),
    "bge-base": dict(vocab_size=30522, hidden=768, layers=12, heads=12, intermediate=3072, max_position=512,
                     type_vocab_size=2, layer_norm_eps=1e-12, pool="cls", max_length=512),
    "tiny": dict(vocab_size=1000, hidden=64, layers=2, heads=2, intermediate=256, max_position=128,
                 type_vocab_size=2, layer_norm_eps=1e-12, pool="mean", max_length=64),
    "small": dict(vocab_size=2000, hidden=128, layers=2, heads=2, intermediate=512, max_position=256,
                  type_vocab_size=2, layer_norm_eps=1e-12, pool="cls", max_length=128),
}


class EncoderSpec:
    def __init__(self, **kw):
        self.vocab_size = int(kw["vocab_size"]); self.hidden = int(kw["hidden"]); self.layers = int(kw["layers"])
        self.heads = int(kw["heads"]); self.intermediate = int(kw["intermediate"])
        self.max_position = int(kw["max_position"]); self.type_vocab_size = int(kw.get("type_vocab_size", 2))
        self.layer_norm_eps = float(kw.get("layer_norm_eps", 1e-12))
        self.pool = kw.get("pool", "mean"); self.max_length = int(kw.get("max_length") or self.max_position)

    def as_dict(self):
        return dict(vocab_size=self.vocab_size, hidden=self.hidden, layers=self.layers, heads=self.heads,
                    intermediate=self.intermediate, max_position=self.max_position,
                    type_vocab_size=self.type_vocab_size, layer_norm_eps=self.layer_norm_eps, pool=self.pool)


def param_shapes(spec: EncoderSpec) -> Dict[str, tuple]:
    """HF ``BertModel`` state-dict names (no pooler) -> shapes, in state-dict order."""
    H, I = spec.hidden, spec.intermediate
    shapes: Dict[str, tuple] = {"embeddings.word_embeddings.weight": (spec.vocab_size, H),
                                "embeddings.position_embeddings.weight": (spec.max_position, H),
                                "embeddings.token_type_embeddings.weight": (spec.type_vocab_size, H),
                                "embeddings.LayerNorm.weight": (H,), "embeddings.LayerNorm.bias": (H,)}
    for i in range(spec.layers):
        p = f"encoder.layer.{i}."
        for n in ("attention.self.query", "attention.self.key", "attention.self.value", "attention.output.dense"):
            shapes[p + n + ".weight"], shapes[p + n + ".bias"] = (H, H), (H,)
        shapes[p + "intermediate.dense.weight"], shapes[p + "intermediate.dense.bias"] = (I, H), (I,)
        shapes[p + "output.dense.weight"], shapes[p + "output.dense.bias"] = (H, I), (H,)
        for n in ("attention.output.LayerNorm", "output.LayerNorm"):
            shapes[p + n + ".weight"], shapes[p + n + ".bias"] = (H,), (H,)
    return shapes


def seeded_weights(spec: EncoderSpec, seed: int) -> Dict[str, np.ndarray]:
    """Deterministic fp32 parameters (N(0,0.05) matrices, 0.02 N(0,1) biases, LayerNorm gains
    1 + 0.1 N(0,1)), drawn in state-dict order from ``default_rng(seed)`` -- the generator the test
    fixtures were captured with (tests/golden/f6_encoder.npz stores the SHA-256 of its output)."""
    rng = np.random.default_rng(seed)
    out: Dict[str, np.ndarray] = {}
    for name, shape in param_shapes(spec).items():
        if name.endswith("LayerNorm.weight"):
            out[name] = (1.0 + 0.1 * rng.standard_normal(shape)).astype(np.float32)
        elif name.endswith(".bias"):
            out[name] = (0.02 * rng.standard_normal(shape)).astype(np.float32)
        else:
            out[name] = (0.05 * rng.standard_normal(shape)).astype(np.float32)
    return out


class HashingTokenizer:
    """Stand-in for WordPiece when no vocab.txt is on disk: lower-case, split on non-alphanumerics
    (the reference's own tokeniser for BM25, text_index.py:10-11), token -> crc32 bucket.
    [PAD]=0, [CLS]=101, [SEP]=102 like BERT."""

    PAD, CLS, SEP = 0, 101, 102

    _WORD = re.compile(r"[a-z0-9]+")

    def __init__(self, vocab_size: int):
        self.vocab_size = vocab_size
        self.lo = min(1000, max(3, vocab_size // 10))
        self._ids: Dict[str, int] = {}          # token -> id memo (natural text repeats its words: corpus ingest is tokeniser-bound)

    def _id(self, t: str) -> int:
        v = self._ids.get(t)
        if v is None:
            v = self.lo + zlib.crc32(t.encode()) % (self.vocab_size - self.lo)
            if len(self._ids) < 1_000_000:
                self._ids[t] = v
        return v

    def encode(self, text: str, max_length: int) -> List[int]:
        # (same tokens as re.split(r"[^a-zA-Z0-9]+", text.lower()) minus the empty strings)
        ids = [self._id(t) for t in self._WORD.findall((text or "").lower())[: max_length - 2]]
        cls, sep = min(self.CLS, self.vocab_size - 1), min(self.SEP, self.vocab_size - 1)
        return [cls] + ids + [sep]


class WordPieceTokenizer:
    """BERT WordPiece over a ``vocab.txt`` (``tokenizers.BertWordPieceTokenizer``: basic tokenisation -- clean-up,
    lower-casing + accent stripping, punctuation and CJK splitting -- then greedy longest-match pieces); ids equal
    ``transformers.BertTokenizer``'s on fixture F9.  Over-long sequences are cut the sentence-transformers way: the
    first ``max_length - 1`` ids + the final [SEP]."""

    def __init__(self, vocab_file: str, lowercase: bool = True):
        from tokenizers import BertWordPieceTokenizer
        self._tok = BertWordPieceTokenizer(vocab_file, lowercase=lowercase)

    def encode(self, text: str, max_length: int) -> List[int]:
        return self._truncate(self._tok.encode(text or "").ids, max_length)

    def encode_batch(self, texts: List[str], max_length: int) -> List[List[int]]:
        """Many texts at once through the library's multi-threaded ``encode_batch`` (bulk ingest)."""
        return [self._truncate(e.ids, max_length) for e in self._tok.encode_batch([t or "" for t in texts])]

    @staticmethod
    def _truncate(ids: List[int], max_length: int) -> List[int]:
        return ids if len(ids) <= max_length else ids[: max_length - 1] + [ids[-1]]


def read_pretrained_dir(path: str, max_length: Optional[int] = None):
    """Parse a local HF / sentence-transformers model directory (no GPU needed):

    * ``config.json`` -> :class:`EncoderSpec`;
    * ``model.safetensors`` (``safetensors``) or ``pytorch_model.bin`` (``torch.load(weights_only=True)``) -> fp32 arrays
      under HF's ``BertModel`` names (a ``bert.`` prefix is dropped by the constructor);
    * ``vocab.txt`` -> :class:`WordPieceTokenizer` (``tokenizer_config.json:do_lower_case``, default true);
    * ``1_Pooling/config.json`` -> "cls" / "mean"; ``modules.json`` -> whether a Normalize module closes the pipeline
      (no ``modules.json``: a plain HF directory, vectors are normalised as the cosine index needs);
    * ``sentence_bert_config.json:max_seq_length`` caps the sequence length (sentence-transformers truncates there),
      then ``max_position_embeddings``; an explicit ``max_length`` wins over both.

    -> (spec, weights, tokenizer or None, {"normalize": bool, "max_seq_length": int})"""
    p = Path(path)
    cfg = json.loads((p / "config.json").read_text())
    pool = "mean"
    st_cfg = p / "1_Pooling" / "config.json"
    if st_cfg.exists():
        pc = json.loads(st_cfg.read_text())
        if pc.get("pooling_mode_cls_token"):
            pool = "cls"
        elif not pc.get("pooling_mode_mean_tokens", True):
            raise ValueError(f"{st_cfg}: only CLS and mean pooling are implemented")
    max_seq = int(cfg["max_position_embeddings"])
    sb = p / "sentence_bert_config.json"
    if sb.exists():
        ms = json.loads(sb.read_text()).get("max_seq_length")
        if ms:
            max_seq = min(max_seq, int(ms))
    normalize = True
    mods = p / "modules.json"
    if mods.exists():
        normalize = any(str(m.get("type", "")).endswith("Normalize") for m in json.loads(mods.read_text()))
    spec = EncoderSpec(vocab_size=cfg["vocab_size"], hidden=cfg["hidden_size"], layers=cfg["num_hidden_layers"],
                       heads=cfg["num_attention_heads"], intermediate=cfg["intermediate_size"],
                       max_position=cfg["max_position_embeddings"], type_vocab_size=cfg.get("type_vocab_size", 2),
                       layer_norm_eps=cfg.get("layer_norm_eps", 1e-12), pool=pool,
                       max_length=min(int(max_length), int(cfg["max_position_embeddings"])) if max_length else min(512, max_seq))
    if cfg.get("hidden_act", "gelu") not in ("gelu",):
        raise ValueError(f"hidden_act {cfg.get('hidden_act')!r}: the HIP encoder implements erf-GELU only")
    if (p / "model.safetensors").exists():
        from safetensors.numpy import load_file
        weights = {k: np.asarray(v, dtype=np.float32) for k, v in load_file(str(p / "model.safetensors")).items()}
    else:
        import torch
        sd = torch.load(str(p / "pytorch_model.bin"), map_location="cpu", weights_only=True)
        weights = {k: v.float().numpy() for k, v in sd.items()}
    tok = None
    if (p / "vocab.txt").exists():
        lower = True
        tc = p / "tokenizer_config.json"
        if tc.exists():
            lower = bool(json.loads(tc.read_text()).get("do_lower_case", True))
        tok = WordPieceTokenizer(str(p / "vocab.txt"), lowercase=lower)
    return spec, weights, tok, {"normalize": normalize, "max_seq_length": max_seq}


class HipSentenceEncoder:
    normalize_default = True      # from_pretrained_dir: whether the directory's pipeline ends in a Normalize module

    def __init__(self, spec: EncoderSpec, weights: Dict[str, np.ndarray], tokenizer=None, device: int = 0,
                 dtype: str = "f16", max_length: Optional[int] = None):
        self._lib = N.load()
        self.spec, self.device = spec, int(device)
        self.max_length = int(max_length or spec.max_length)
        self.tokenizer = tokenizer or HashingTokenizer(spec.vocab_size)
        cfg = N.EncoderConfig(spec.vocab_size, spec.hidden, spec.layers, spec.heads, spec.intermediate, spec.max_position,
                              spec.type_vocab_size, spec.layer_norm_eps, N.MRAG_F16 if dtype == "f16" else N.MRAG_BF16)
        h = C.c_uint64(0)
        N.check(self._lib.mrag_encoder_create(C.byref(cfg), self.device, C.byref(h)))
        self._h = h
        for name, arr in weights.items():
            name = name[5:] if name.startswith("bert.") else name
            if name.startswith("pooler.") or name.endswith("position_ids"):
                continue
            a = np.ascontiguousarray(arr, dtype=np.float32)
            N.check(self._lib.mrag_encoder_set_param(self._h, name.encode(), a.ctypes.data, a.size, 0, None))
        miss = C.c_int(0)
        N.check(self._lib.mrag_encoder_missing_params(self._h, C.byref(miss)))
        if miss.value:
            raise ValueError(f"encoder weights incomplete: {self._lib.mrag_last_error().decode()}")

    # -- construction -----------------------------------------------------------------------------
    @classmethod
    def from_seed(cls, arch: str = "minilm-l6", seed: int = 0, **kw) -> "HipSentenceEncoder":
        spec = EncoderSpec(**ARCHS[arch])
        return cls(spec, seeded_weights(spec, seed), **kw)

    @classmethod
    def from_pretrained_dir(cls, path: str, **kw) -> "HipSentenceEncoder":
        """A local HF / sentence-transformers directory -> encoder (the real-weights route of the Embedder slot;
        pinned by fixture F9, tests/golden/make_golden_hfdir.py)."""
        spec, weights, tok, info = read_pretrained_dir(path, max_length=kw.pop("max_length", None))
        enc = cls(spec, weights, tokenizer=tok, **kw)
        enc.normalize_default = info["normalize"]
        return enc

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.mrag_encoder_destroy(self._h)
            self._h = C.c_uint64(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- forward ----------------------------------------------------------------------------------
    def forward(self, ids: np.ndarray, mask: np.ndarray, pool: Optional[str] = None, normalize: bool = True) -> np.ndarray:
        """ids/mask int32 [B,S] -> float32 [B,hidden]."""
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        mask = np.ascontiguousarray(mask, dtype=np.int32)
        B, S = ids.shape
        out = np.empty((B, self.spec.hidden), dtype=np.float32)
        pool = pool or self.spec.pool
        N.check(self._lib.mrag_encoder_forward(self._h, ids.ctypes.data, mask.ctypes.data, B, S, out.ctypes.data,
                                               N.POOL_CLS if pool == "cls" else N.POOL_MEAN, int(bool(normalize)), 0, None))
        return out

    def forward_device(self, ids, mask, out=None, pool: Optional[str] = None, normalize: bool = True, stream=None):
        """The same forward with ids / mask (int32 CUDA tensors [B,S]) and the result (float32 CUDA tensor
        [B,hidden]) in HBM -- ``io_is_device`` of ``mrag_encoder_forward``: nothing crosses PCIe but the ids, and
        the rows can go straight into ``DenseIndex.add`` (bulk ingest).  Asynchronous on ``stream`` (default: torch's
        current stream)."""
        import torch
        ids = ids.contiguous(); mask = mask.contiguous()
        if ids.dtype != torch.int32 or mask.dtype != torch.int32 or not ids.is_cuda or not mask.is_cuda:
            raise ValueError("forward_device expects int32 CUDA tensors")
        B, S = ids.shape
        if out is None:
            out = torch.empty((B, self.spec.hidden), dtype=torch.float32, device=ids.device)
        if stream is None:
            stream = torch.cuda.current_stream(self.device)
        pool = pool or self.spec.pool
        N.check(self._lib.mrag_encoder_forward(self._h, ids.data_ptr(), mask.data_ptr(), B, S, out.data_ptr(),
                                               N.POOL_CLS if pool == "cls" else N.POOL_MEAN, int(bool(normalize)), 1,
                                               int(getattr(stream, "cuda_stream", stream))))
        return out

    def last_timing_ms(self) -> float:
        """Device ms of the last forward's kernels (``mrag_encoder_last_timing``)."""
        ms = C.c_float(0)
        N.check(self._lib.mrag_encoder_last_timing(self._h, C.byref(ms)))
        return ms.value

    def tokenize(self, texts: List[str]):
        if hasattr(self.tokenizer, "encode_batch") and len(texts) > 8:
            seqs = self.tokenizer.encode_batch(texts, self.max_length)
        else:
            seqs = [self.tokenizer.encode(t, self.max_length) for t in texts]
        S = max(16, -(-max(len(s) for s in seqs) // 16) * 16)
        S = min(S, self.spec.max_position)
        ids = np.zeros((len(seqs), S), dtype=np.int32)
        mask = np.zeros((len(seqs), S), dtype=np.int32)
        for i, s in enumerate(seqs):
            s = s[:S]
            ids[i, :len(s)] = s
            mask[i, :len(s)] = 1
        return ids, mask

    def encode_device(self, texts: List[str], batch_size: int = 4096, normalize: bool = True, out=None):
        """Texts -> float32 CUDA tensor [n, hidden], rows in input order (bulk ingest: large length-sorted batches so
        that the 256 x 256 / fused-LayerNorm GEMM flow runs, no host copy of the embeddings, no Python float lists).
        A tokeniser thread prepares batch i+1 while the GPU runs batch i; every batch's rows are scattered to their
        input positions on the device.  ``out``: an existing [n, hidden] float32 CUDA tensor to fill."""
        import queue
        import threading
        import torch
        n = len(texts)
        dev = torch.device("cuda", self.device)
        if out is None:
            out = torch.empty((n, self.spec.hidden), dtype=torch.float32, device=dev)
        if n == 0:
            return out
        order = sorted(range(n), key=lambda i: len(texts[i]))
        batches = [order[lo:lo + batch_size] for lo in range(0, n, batch_size)]
        ready: "queue.Queue" = queue.Queue(maxsize=2)

        def produce():
            try:
                for sel in batches:
                    ids, mask = self.tokenize([texts[i] for i in sel])
                    ready.put((sel, ids, mask))
                ready.put(None)
            except BaseException as e:
                ready.put(e)
        t = threading.Thread(target=produce, name="mrag-tokenizer", daemon=True)
        t.start()
        try:
            while True:
                item = ready.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                sel, ids, mask = item
                rows = self.forward_device(torch.from_numpy(ids).to(dev, non_blocking=True),
                                           torch.from_numpy(mask).to(dev, non_blocking=True), normalize=normalize)
                out.index_copy_(0, torch.as_tensor(sel, dtype=torch.int64, device=dev), rows)
        finally:
            while t.is_alive():
                try:
                    ready.get_nowait()
                except queue.Empty:
                    t.join(0.01)
        return out

    def encode(self, texts: List[str], batch_size: int = 256, normalize: bool = True, overlap: bool = True) -> np.ndarray:
        """Texts -> [n, hidden] float32.  Batches are formed in length order (less padding) and the rows are
        returned in input order.  ``overlap``: a tokeniser thread prepares batch i+1 (WordPiece /
        hashing, padding) while the GPU runs batch i -- the forward is one ctypes call, which releases the GIL for
        its whole duration -- so host tokenisation leaves the timeline when the GPU is the slower side."""
        n = len(texts)
        out = np.empty((n, self.spec.hidden), dtype=np.float32)
        order = sorted(range(n), key=lambda i: len(texts[i]))
        batches = [order[lo:lo + batch_size] for lo in range(0, n, batch_size)]
        if not overlap or len(batches) < 2:
            for sel in batches:
                ids, mask = self.tokenize([texts[i] for i in sel])
                out[sel] = self.forward(ids, mask, normalize=normalize)
            return out
        import queue
        import threading
        ready: "queue.Queue" = queue.Queue(maxsize=2)       # at most two tokenised batches ahead of the GPU

        def produce():
            try:
                for sel in batches:
                    ready.put((sel,) + self.tokenize([texts[i] for i in sel]))
                ready.put(None)
            except BaseException as e:                        # surface tokeniser errors in the caller
                ready.put(e)
        t = threading.Thread(target=produce, name="mrag-tokenizer", daemon=True)
        t.start()
        try:
            while True:
                item = ready.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                sel, ids, mask = item
                out[sel] = self.forward(ids, mask, normalize=normalize)
        finally:
            while t.is_alive():                               # unblock the producer if the consumer bailed out early
                try:
                    ready.get_nowait()
                except queue.Empty:
                    t.join(0.01)
        return out
