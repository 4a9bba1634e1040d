"""CPU: pin the encoder oracle (numpy restatement of BERT + sentence-transformers pooling) against
the container's transformers.BertModel on seeded weights.  The reference ships no encoder, so this
is the only pin the encoder path has (parity unpinned by the reference -- SURVEY.md section 8c)."""
import numpy as np
import pytest

from oracle import encoder as oe


def _hf_forward(spec, w, ids, mask, pool):
    torch = pytest.importorskip("torch")
    tr = pytest.importorskip("transformers")
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
    e = torch.nn.functional.normalize(e, p=2, dim=1)
    return e.numpy()


@pytest.mark.parametrize("arch,pool", [("tiny", "mean"), ("small", "cls"), ("tiny", "cls")])
def test_oracle_matches_hf_bert(arch, pool):
    spec = oe.SPECS[arch]
    w = oe.seeded_weights(spec, seed=11)
    rng = np.random.default_rng(5)
    B, S = 5, 24
    ids = rng.integers(3, spec["vocab_size"], size=(B, S)).astype(np.int64)
    mask = np.ones((B, S), dtype=np.int64)
    for b, L in enumerate((24, 7, 16, 1, 23)):
        mask[b, L:] = 0
        ids[b, L:] = 0
    got = oe.forward(spec, w, ids, mask, pool=pool)
    want = _hf_forward(spec, w, ids, mask, pool)
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-5)        # HF runs fp32
    np.testing.assert_allclose(np.linalg.norm(got, axis=1), 1.0, atol=1e-12)


def test_param_shapes_cover_hf_names():
    spec = oe.SPECS["tiny"]
    names = set(oe.param_shapes(spec))
    assert "encoder.layer.1.output.LayerNorm.bias" in names and len(names) == 5 + 16 * spec["layers"]


def _sha(w):
    import hashlib
    h = hashlib.sha256()
    for k in w:
        h.update(k.encode())
        h.update(np.ascontiguousarray(w[k]).tobytes())
    return h.hexdigest()


@pytest.mark.parametrize("case", ["minilm2", "bge1"])
def test_oracle_and_weight_generators_match_f6_fixture(golden_dir, case):
    """F6 = HF BertModel outputs captured by tests/golden/make_golden_encoder.py (MiniLM-L6 / bge-base layer
    geometry, full vocabulary, 512 positions).  Pins the oracle WITHOUT transformers at run time, and pins
    the product's weight generator to the one the fixture was made with."""
    from mrag_amd.encoder import EncoderSpec, seeded_weights
    z = np.load(golden_dir / "f6_encoder.npz")
    base = {"minilm2": "minilm-l6", "bge1": "bge-base"}[case]
    spec = dict(oe.SPECS[base], layers=int(z[f"{case}.layers"]))
    w = oe.seeded_weights(spec, int(z[f"{case}.seed"]))
    assert _sha(w) == str(z[f"{case}.weights_sha256"])
    assert _sha(seeded_weights(EncoderSpec(**spec), int(z[f"{case}.seed"]))) == str(z[f"{case}.weights_sha256"])
    ids, mask, pool = z[f"{case}.ids"], z[f"{case}.mask"], str(z[f"{case}.pool"])
    np.testing.assert_allclose(oe.forward(spec, w, ids, mask, pool=pool), z[f"{case}.emb"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(oe.forward(spec, w, ids, mask, pool=pool, normalize=False), z[f"{case}.raw"], rtol=0, atol=2e-4)


# ---- F9: the real-weights route (a builder-generated HF / sentence-transformers directory) ----------
def _f9(golden_dir):
    import json
    return json.loads((golden_dir / "f9_hf.json").read_text()), golden_dir / "f9_hf_dir"


def test_f9_wordpiece_ids_equal_hf_tokenizer(golden_dir):
    """``WordPieceTokenizer`` ids == ``transformers.BertTokenizer`` ids (stored by make_golden_hfdir.py), bit for bit:
    accents, CJK, control characters, > 100-character words, literal specials, truncation at max_seq_length."""
    from mrag_amd.encoder import WordPieceTokenizer
    g, d = _f9(golden_dir)
    tok = WordPieceTokenizer(str(d / "vocab.txt"))
    assert len(g["texts"]) >= 50 and any(len(x) > g["max_seq_length"] for x in g["ids_untruncated"])
    for text, ids, full in zip(g["texts"], g["ids"], g["ids_untruncated"]):
        assert tok.encode(text, g["max_seq_length"]) == ids, text
        assert tok.encode(text, 10 ** 6) == full, text


def test_f9_directory_is_read_as_sentence_transformers_would(golden_dir):
    """config.json / 1_Pooling / modules.json (Normalize) / sentence_bert_config.json:max_seq_length / safetensors."""
    from mrag_amd.encoder import read_pretrained_dir
    g, d = _f9(golden_dir)
    spec, w, tok, info = read_pretrained_dir(str(d))
    assert (spec.hidden, spec.layers, spec.heads, spec.intermediate, spec.max_position) == (64, 2, 2, 256, 96)
    assert spec.vocab_size == g["vocab_size"] and spec.pool == "mean"
    assert spec.max_length == g["max_seq_length"] == info["max_seq_length"] and info["normalize"] is True
    assert tok is not None and w["embeddings.word_embeddings.weight"].shape == (g["vocab_size"], 64)
    assert read_pretrained_dir(str(d), max_length=16)[0].max_length == 16          # an explicit cap wins
    assert read_pretrained_dir(str(d), max_length=4096)[0].max_length == 96        # ... up to the position table


def test_f9_oracle_matches_hf_from_pretrained(golden_dir):
    """The oracle on the directory's weights and the stored HF token ids reproduces HF's
    ``BertModel.from_pretrained(dir)`` + pooling + normalise (both poolings)."""
    from mrag_amd.encoder import read_pretrained_dir
    g, d = _f9(golden_dir)
    spec, w, _, _ = read_pretrained_dir(str(d))
    S = max(len(x) for x in g["ids"])
    ids = np.zeros((len(g["ids"]), S), dtype=np.int64); mask = np.zeros_like(ids)
    for i, x in enumerate(g["ids"]):
        ids[i, :len(x)] = x; mask[i, :len(x)] = 1
    sp = dict(spec.as_dict())
    for pool, key in (("mean", "mean_normalized"), ("cls", "cls_normalized")):
        np.testing.assert_allclose(oe.forward(sp, w, ids, mask, pool=pool), np.asarray(g[key]), rtol=0, atol=2e-5)


@pytest.mark.parametrize("case", ["minilm6", "bge12"])
def test_f6b_oracle_matches_hf_at_full_depth(golden_dir, case):
    """The fp64 oracle against the committed HF outputs at full depth (6 / 12 layers): the pin the GPU tests lean on."""
    import hashlib
    z = np.load(golden_dir / "f6b_encoder_full.npz")
    base = {"minilm6": "minilm-l6", "bge12": "bge-base"}[case]
    spec = dict(oe.SPECS[base])
    w = oe.seeded_weights(spec, int(z[f"{case}.seed"]))
    h = hashlib.sha256()
    for k in w:
        h.update(k.encode()); h.update(np.ascontiguousarray(w[k]).tobytes())
    assert h.hexdigest() == str(z[f"{case}.weights_sha256"])
    sub = slice(0, 3)
    got = oe.forward(spec, w, z[f"{case}.ids"][sub].astype(np.int64), z[f"{case}.mask"][sub].astype(np.int64), pool=str(z[f"{case}.pool"]))
    np.testing.assert_allclose(got, z[f"{case}.emb"][sub], rtol=0, atol=5e-5)
