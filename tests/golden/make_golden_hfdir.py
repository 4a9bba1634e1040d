#!/usr/bin/env python3
"""F9: the real-weights route of the Embedder slot -- text in, vectors out, from a local
HF / sentence-transformers directory (what app/core/providers/openai_provider.py:96-134 is for the remote model).

The reference ships no encoder and the container holds no pretrained checkpoint (no network), so this script BUILDS a
small HF-format model directory in the container -- nothing is downloaded:

    tests/golden/f9_hf_dir/
        config.json                      BertConfig: hidden 64, 2 layers, 2 heads, FFN 256, 96 positions
        vocab.txt                        synthetic WordPiece vocabulary (specials, letters, digits, punctuation,
                                         words, ``##`` continuation pieces, a few CJK characters)
        model.safetensors                seeded weights (written with ``safetensors``)
        modules.json, 1_Pooling/config.json, sentence_bert_config.json (max_seq_length 24), 2_Normalize/

and runs the container's ``transformers`` ``BertTokenizer`` + ``BertModel.from_pretrained(<that dir>)`` + mean / CLS
pooling + L2 normalisation on ~50 texts (unicode, accents, empty string, unknown words, punctuation runs, texts longer
than max_seq_length).  Stored in f9_hf.json: the texts, the token ids HF's tokenizer produced (truncated to
max_seq_length the sentence-transformers way) and the embeddings.

    python tests/golden/make_golden_hfdir.py        # rewrites the directory and f9_hf.json

CPU test: ``mrag_amd.encoder.WordPieceTokenizer`` ids == the stored ids.  GPU test: ``HipEmbeddingProvider(model_path=dir)``
-> ``embed(...)`` within 1e-3 of the stored embeddings.  The GPU box only reads the directory and the JSON.
"""
import json
import shutil
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
OUT_DIR = HERE / "f9_hf_dir"
MAX_SEQ = 24

WORDS = ("the of and a to in is was for on that with as by at from it his an were are which this be or had first one their "
         "has its new after who also they two her she been other when time during there into school more years over city "
         "some world would where later up such used many can state about national out known university united then made "
         "film album band born american english french german river lake mountain island north south east west king queen "
         "war battle army team season league club game player music song series book novel author director actor actress "
         "company station railway road bridge church museum park county district village town capital population founded "
         "released written directed produced starring located named called member president party election government "
         "what which who when where how did does is are was were year name country").split()
PIECES = "s ed ing er est ly ion tion al ic ous ive ment ness ity an en es y a e i o u n t r l d".split()
TEXTS = [
    "Were Scott Derrickson and Ed Wood of the same nationality?",
    "The film was directed by an american director and released in 1998.",
    "",
    " ",
    "king",
    "Kings and queens of the united kingdoms",
    "Réné Müller wrote the novel in the café near the Fjörd",          # accents stripped by the basic tokenizer
    "naïve coöperation — façade, jalapeño; Ångström",
    "unknownword xyzzyq qwrtpsdf 12345 67890",
    "What is the population of the city where the university is located?",
    "river,lake;mountain:island!north?south(east)west[king]{queen}",
    "a" * 150,                                                          # > 100 characters: [UNK] per BertTokenizer
    "the " * 40,                                                        # longer than max_seq_length: truncated
    "first second third fourth fifth sixth seventh eighth ninth tenth eleventh twelfth thirteenth fourteenth fifteenth",
    "东京 is the capital of 日本",                                      # CJK characters are split one by one
    "tabs\tand\nnewlines\r\nand   runs   of   spaces",
    "MiXeD CaSe WoRdS aNd UPPERCASE",
    "hyphen-ated well-known state-of-the-art",
    "don't can't it's o'clock",
    "3.14159 2,000,000 1998-2004 $5 50% #1",
    "released releasing releases release",
    "national nationality nationalities internationally",
    "​zero​width﻿ and control\x00chars\x7f",
    "##ing ##ed literal hashes",
    "[CLS] literal specials [SEP] [MASK] [PAD] [UNK]",
]


def build_vocab():
    v = ["[PAD]"] + [f"[unused{i}]" for i in range(1, 100)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"]
    v += list("!\"#$%&'()*+,-./:;<=>?@[\\]^_`{|}~")
    v += [str(d) for d in range(10)] + [chr(c) for c in range(ord("a"), ord("z") + 1)]
    v += ["东", "京", "日", "本"]
    seen = set(v)
    for w in WORDS:
        if w not in seen:
            v.append(w); seen.add(w)
    for p in PIECES + [str(d) for d in range(10)] + [chr(c) for c in range(ord("a"), ord("z") + 1)]:
        if "##" + p not in seen:
            v.append("##" + p); seen.add("##" + p)
    return v


def main():
    import torch
    import transformers as tr
    from safetensors.numpy import save_file
    sys.path.insert(0, str(HERE.parent.parent))
    rng = np.random.default_rng(9)
    texts = list(TEXTS)
    # random word salads (known words, known words + suffix pieces, unknown tokens) to ~50 texts
    while len(texts) < 50:
        n = int(rng.integers(1, 30))
        toks = []
        for _ in range(n):
            r = rng.random()
            w = WORDS[int(rng.integers(0, len(WORDS)))]
            toks.append(w if r < 0.6 else w + PIECES[int(rng.integers(0, len(PIECES)))] if r < 0.85 else
                        "".join(chr(int(c)) for c in rng.integers(97, 123, size=int(rng.integers(2, 9)))))
        texts.append(" ".join(toks))

    if OUT_DIR.exists():
        shutil.rmtree(OUT_DIR)
    (OUT_DIR / "1_Pooling").mkdir(parents=True)
    (OUT_DIR / "2_Normalize").mkdir()
    vocab = build_vocab()
    (OUT_DIR / "vocab.txt").write_text("\n".join(vocab) + "\n", encoding="utf-8")
    cfg = tr.BertConfig(vocab_size=len(vocab), hidden_size=64, num_hidden_layers=2, num_attention_heads=2,
                        intermediate_size=256, max_position_embeddings=96, type_vocab_size=2, layer_norm_eps=1e-12,
                        hidden_act="gelu", hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    d = cfg.to_dict()
    d["architectures"] = ["BertModel"]
    (OUT_DIR / "config.json").write_text(json.dumps(d, indent=1, sort_keys=True))
    (OUT_DIR / "tokenizer_config.json").write_text(json.dumps({"do_lower_case": True, "tokenizer_class": "BertTokenizer"}))
    (OUT_DIR / "modules.json").write_text(json.dumps([
        {"idx": 0, "name": "0", "path": "", "type": "sentence_transformers.models.Transformer"},
        {"idx": 1, "name": "1", "path": "1_Pooling", "type": "sentence_transformers.models.Pooling"},
        {"idx": 2, "name": "2", "path": "2_Normalize", "type": "sentence_transformers.models.Normalize"}], indent=1))
    (OUT_DIR / "1_Pooling" / "config.json").write_text(json.dumps(
        {"word_embedding_dimension": 64, "pooling_mode_cls_token": False, "pooling_mode_mean_tokens": True,
         "pooling_mode_max_tokens": False, "pooling_mode_mean_sqrt_len_tokens": False}, indent=1))
    (OUT_DIR / "sentence_bert_config.json").write_text(json.dumps({"max_seq_length": MAX_SEQ, "do_lower_case": False}))
    # seeded weights in HF's own parameter names (no pooler), through this repo's generator
    from oracle import encoder as oe
    spec = dict(vocab_size=len(vocab), hidden=64, layers=2, heads=2, intermediate=256, max_position=96,
                type_vocab_size=2, layer_norm_eps=1e-12)
    w = oe.seeded_weights(spec, 99)
    save_file({k: np.ascontiguousarray(v) for k, v in w.items()}, str(OUT_DIR / "model.safetensors"))

    # ---- the reference side of the fixture: HF tokenizer + BertModel loaded FROM THE DIRECTORY ----
    tok = tr.BertTokenizer(str(OUT_DIR / "vocab.txt"), do_lower_case=True)
    model = tr.BertModel.from_pretrained(str(OUT_DIR), add_pooling_layer=False).eval()
    enc = tok(texts, padding=True, truncation=True, max_length=MAX_SEQ, return_tensors="pt")
    with torch.no_grad():
        h = model(input_ids=enc["input_ids"], attention_mask=enc["attention_mask"]).last_hidden_state.double()
    mm = enc["attention_mask"].double()[:, :, None]
    mean = torch.nn.functional.normalize((h * mm).sum(1) / mm.sum(1).clamp(min=1e-9), p=2, dim=1).numpy()
    cls = torch.nn.functional.normalize(h[:, 0], p=2, dim=1).numpy()
    ids = [[int(t) for t, m in zip(row, mrow) if m] for row, mrow in zip(enc["input_ids"].tolist(), enc["attention_mask"].tolist())]
    full = [tok(t)["input_ids"] for t in texts]                      # untruncated ids, for the tokenizer test
    out = {"max_seq_length": MAX_SEQ, "texts": texts, "ids": ids, "ids_untruncated": full,
           "mean_normalized": mean.tolist(), "cls_normalized": cls.tolist(),
           "torch_version": torch.__version__, "transformers_version": tr.__version__, "vocab_size": len(vocab)}
    (HERE / "f9_hf.json").write_text(json.dumps(out))
    print("vocab", len(vocab), "texts", len(texts), "longest", max(len(x) for x in full), "dir bytes",
          sum(f.stat().st_size for f in OUT_DIR.rglob("*") if f.is_file()))


if __name__ == "__main__":
    main()
