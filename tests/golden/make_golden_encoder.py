#!/usr/bin/env python3
"""F6: encoder golden vectors from the container's ``transformers.BertModel``.

The reference ships no encoder (its embedding step is a provider slot, app/core/providers/base.py:6), and
pretrained MiniLM / bge weights are not on disk, so the fixture pins the ARCHITECTURE ARITHMETIC: seeded
weights of the two BASELINE shapes (MiniLM-L6 and bge-base layer geometry, full 30 522-row vocabulary and
512 positions, 2 layers / 1 layer to keep the HF run short), ragged token-id batches, and the embeddings
HF's BertModel (fp32, eager attention, erf GELU) + sentence-transformers pooling + L2 normalisation give.
The weights themselves are NOT stored: both sides regenerate them from the seed
(``oracle.encoder.seeded_weights`` == ``mrag_amd.encoder.seeded_weights``); their SHA-256 is.

    python tests/golden/make_golden_encoder.py        # writes tests/golden/f6_encoder.npz

Needs torch + transformers (this container); the GPU box only loads the .npz.
"""
import hashlib
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from oracle import encoder as oe  # noqa: E402

CASES = {
    # name: (spec overrides on top of the public shape, seed, pool, B, S)
    "minilm2": (dict(oe.SPECS["minilm-l6"], layers=2), 41, "mean", 10, 160),
    "bge1": (dict(oe.SPECS["bge-base"], layers=1), 42, "cls", 6, 272),
}
# round 3: the same at FULL depth -- all 6 / 12 layers of the two BASELINE models -- in a second file,
# so that f6_encoder.npz stays byte-identical to what round 2 committed
CASES_FULL = {
    "minilm6": (dict(oe.SPECS["minilm-l6"]), 43, "mean", 8, 128),
    "bge12": (dict(oe.SPECS["bge-base"]), 44, "cls", 5, 160),
}


def weights_sha(w):
    h = hashlib.sha256()
    for k in w:
        h.update(k.encode())
        h.update(np.ascontiguousarray(w[k]).tobytes())
    return h.hexdigest()


def batch(spec, B, S, seed):
    rng = np.random.default_rng(seed)
    ids = rng.integers(3, spec["vocab_size"], size=(B, S)).astype(np.int32)
    ids[0, :4] = [101, spec["vocab_size"] - 1, 30000, 102]          # top of the table
    lens = [S, 1, 2, S - 1, 129, 17][:B] + [int(x) for x in rng.integers(3, S, size=max(0, B - 6))]
    mask = np.zeros((B, S), dtype=np.int32)
    for b, L in enumerate(lens):
        mask[b, :L] = 1
    return ids * mask, mask


def hf_forward(spec, w, ids, mask, pool):
    import torch
    import transformers as tr
    cfg = tr.BertConfig(vocab_size=spec["vocab_size"], hidden_size=spec["hidden"], num_hidden_layers=spec["layers"],
                        num_attention_heads=spec["heads"], intermediate_size=spec["intermediate"],
                        max_position_embeddings=spec["max_position"], type_vocab_size=spec["type_vocab_size"],
                        layer_norm_eps=spec["layer_norm_eps"], hidden_act="gelu", hidden_dropout_prob=0.0,
                        attention_probs_dropout_prob=0.0)
    m = tr.BertModel(cfg, add_pooling_layer=False).eval()
    sd = m.state_dict()
    for k, v in w.items():
        assert k in sd and tuple(sd[k].shape) == v.shape, k
        sd[k] = torch.from_numpy(v)
    m.load_state_dict(sd, strict=False)
    with torch.no_grad():
        h = m(input_ids=torch.from_numpy(ids).long(), attention_mask=torch.from_numpy(mask).long()).last_hidden_state.double()
    if pool == "cls":
        e = h[:, 0]
    else:
        mm = torch.from_numpy(mask).double()[:, :, None]
        e = (h * mm).sum(1) / mm.sum(1).clamp(min=1e-9)
    return e.numpy(), torch.nn.functional.normalize(e, p=2, dim=1).numpy()


def main():
    import torch
    import transformers as tr
    for cases, fname in ((CASES, "f6_encoder.npz"), (CASES_FULL, "f6b_encoder_full.npz")):
      out = {"torch_version": np.array(torch.__version__), "transformers_version": np.array(tr.__version__)}
      for name, (spec, seed, pool, B, S) in cases.items():
        w = oe.seeded_weights(spec, seed)
        ids, mask = batch(spec, B, S, seed + 100)
        raw, emb = hf_forward(spec, w, ids, mask, pool)
        out.update({f"{name}.ids": ids, f"{name}.mask": mask, f"{name}.raw": raw.astype(np.float32),
                    f"{name}.emb": emb.astype(np.float32), f"{name}.seed": np.array(seed), f"{name}.pool": np.array(pool),
                    f"{name}.layers": np.array(spec["layers"]), f"{name}.weights_sha256": np.array(weights_sha(w))})
        print(name, ids.shape, "emb", emb.shape, "sha", weights_sha(w)[:16])
      dst = Path(__file__).resolve().parent / fname
      np.savez_compressed(dst, **out)
      print("wrote", dst, dst.stat().st_size, "bytes")


if __name__ == "__main__":
    main()
