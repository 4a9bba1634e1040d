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
