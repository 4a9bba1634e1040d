"""CPU oracle of the sentence-encoder forward pass.  TEST INFRASTRUCTURE (oracle/__init__.py).

The reference contains no encoder: its embedding step is a provider
slot (app/core/providers/base.py:6).  This restates the HF ``BertModel`` arithmetic
(embeddings + LayerNorm, L x {multi-head attention, erf-GELU FFN, post-LN residuals}) with
sentence-transformers pooling (masked mean, or CLS) and L2 normalisation, in numpy fp64.
PARITY UNPINNED by the reference; pinned against the container's ``transformers.BertModel``
(tests/test_encoder_oracle.py) on seeded weights.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np

SPECS = {
    # public model-card shapes; re-check against config.json when weights exist
    "minilm-l6": dict(vocab_size=30522, hidden=384, layers=6, heads=12, intermediate=1536, max_position=512,
                      type_vocab_size=2, layer_norm_eps=1e-12, pool="mean"),
    "bge-base": dict(vocab_size=30522, hidden=768, layers=12, heads=12, intermediate=3072, max_position=512,
                     type_vocab_size=2, layer_norm_eps=1e-12, pool="cls"),
    "tiny": dict(vocab_size=1000, hidden=64, layers=2, heads=2, intermediate=256, max_position=128,
                 type_vocab_size=2, layer_norm_eps=1e-12, pool="mean"),
    "small": dict(vocab_size=2000, hidden=128, layers=2, heads=2, intermediate=512, max_position=256,
                  type_vocab_size=2, layer_norm_eps=1e-12, pool="cls"),
}


def param_shapes(spec) -> Dict[str, tuple]:
    H, I = spec["hidden"], spec["intermediate"]
    s = {"embeddings.word_embeddings.weight": (spec["vocab_size"], H),
         "embeddings.position_embeddings.weight": (spec["max_position"], H),
         "embeddings.token_type_embeddings.weight": (spec["type_vocab_size"], H),
         "embeddings.LayerNorm.weight": (H,), "embeddings.LayerNorm.bias": (H,)}
    for i in range(spec["layers"]):
        p = f"encoder.layer.{i}."
        for n in ("attention.self.query", "attention.self.key", "attention.self.value", "attention.output.dense"):
            s[p + n + ".weight"], s[p + n + ".bias"] = (H, H), (H,)
        s[p + "intermediate.dense.weight"], s[p + "intermediate.dense.bias"] = (I, H), (I,)
        s[p + "output.dense.weight"], s[p + "output.dense.bias"] = (H, I), (H,)
        for n in ("attention.output.LayerNorm", "output.LayerNorm"):
            s[p + n + ".weight"], s[p + n + ".bias"] = (H,), (H,)
    return s


def seeded_weights(spec, seed: int) -> Dict[str, np.ndarray]:
    """Deterministic fp32 parameters: N(0, 0.05) matrices, small random biases, LayerNorm gains
    around 1 -- the same generator on the oracle and the GPU side (real checkpoints are not in the
    container)."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shape in param_shapes(spec).items():
        if name.endswith("LayerNorm.weight"):
            out[name] = (1.0 + 0.1 * rng.standard_normal(shape)).astype(np.float32)
        elif name.endswith(".bias"):
            out[name] = (0.02 * rng.standard_normal(shape)).astype(np.float32)
        else:
            out[name] = (0.05 * rng.standard_normal(shape)).astype(np.float32)
    return out


def _ln(x, g, b, eps):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * g + b


_erf = np.vectorize(math.erf)


def forward(spec, w: Dict[str, np.ndarray], ids: np.ndarray, mask: np.ndarray, pool: str = "mean",
            normalize: bool = True) -> np.ndarray:
    """ids/mask int [B,S] -> embeddings fp64 [B,H]."""
    W = {k: np.asarray(v, dtype=np.float64) for k, v in w.items()}
    B, S = ids.shape
    H, nh, eps = spec["hidden"], spec["heads"], spec["layer_norm_eps"]
    dh = H // nh
    x = W["embeddings.word_embeddings.weight"][ids] + W["embeddings.position_embeddings.weight"][:S][None] \
        + W["embeddings.token_type_embeddings.weight"][0][None, None]
    x = _ln(x, W["embeddings.LayerNorm.weight"], W["embeddings.LayerNorm.bias"], eps)
    bias = np.where(mask[:, None, None, :] != 0, 0.0, -1e30)
    for i in range(spec["layers"]):
        p = f"encoder.layer.{i}."
        lin = lambda t, n: t @ W[p + n + ".weight"].T + W[p + n + ".bias"]     # noqa: E731
        q = lin(x, "attention.self.query").reshape(B, S, nh, dh).transpose(0, 2, 1, 3)
        k = lin(x, "attention.self.key").reshape(B, S, nh, dh).transpose(0, 2, 1, 3)
        v = lin(x, "attention.self.value").reshape(B, S, nh, dh).transpose(0, 2, 1, 3)
        sc = q @ k.transpose(0, 1, 3, 2) / math.sqrt(dh) + bias
        sc = sc - sc.max(-1, keepdims=True)
        pr = np.exp(sc)
        pr = pr / pr.sum(-1, keepdims=True)
        ctx = (pr @ v).transpose(0, 2, 1, 3).reshape(B, S, H)
        x = _ln(lin(ctx, "attention.output.dense") + x, W[p + "attention.output.LayerNorm.weight"],
                W[p + "attention.output.LayerNorm.bias"], eps)
        h = lin(x, "intermediate.dense")
        h = 0.5 * h * (1.0 + _erf(h / math.sqrt(2.0)))
        x = _ln(lin(h, "output.dense") + x, W[p + "output.LayerNorm.weight"], W[p + "output.LayerNorm.bias"], eps)
    if pool == "cls":
        e = x[:, 0]
    else:
        m = (mask != 0).astype(np.float64)[:, :, None]
        e = (x * m).sum(1) / np.maximum(m.sum(1), 1e-9)
    if normalize:
        e = e / np.maximum(np.linalg.norm(e, axis=1, keepdims=True), 1e-12)
    return e
