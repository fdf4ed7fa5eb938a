"""GPU parity: BERT-style text classifier (Taiyi Chinese text tower) vs the oracle and the
HF BertForSequenceClassification goldens."""
import os

import numpy as np
import pytest
import torch

import mmr_amd
from mmr_amd import weights
from mmr_amd.config import get_bert_config

pytestmark = pytest.mark.gpu


def _cos(a, b):
    a, b = a.double(), b.double()
    return (a * b).sum(-1) / (a.norm(dim=-1) * b.norm(dim=-1))


def _ids(cfg, n, T, seed=7):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(1, cfg.vocab, (n, T), generator=g, dtype=torch.int32)
    ids[:, 0] = 101 % cfg.vocab
    ids[1:, T - 3:] = 0
    return ids


@pytest.mark.parametrize("name", ["tiny-bert-test", "Taiyi-CLIP-Roberta-large-326M-Chinese"])
def test_bert_logits_vs_hf_golden_and_oracle(device, golden_dir, name):
    from oracle import bert_ref
    g = np.load(os.path.join(golden_dir, f"bert_{name}.npz"))
    cfg = get_bert_config(name)
    enc = mmr_amd.load_text_encoder(name, device=device, seed=int(g["weight_seed"]))
    ids = _ids(cfg, int(g["n_txt"]), int(g["T"]), int(g["ids_seed"]))
    N, T = ids.shape
    tap = torch.zeros(N * T, cfg.width, device=device)
    out = enc.logits(ids.to(device), tap_after=0, tap=tap).cpu()
    gold = torch.from_numpy(g["logits"])
    w = weights.make_bert_weights(cfg, seed=int(g["weight_seed"]))
    st = {}
    with torch.no_grad():
        ref = bert_ref.bert_logits(w, cfg, ids, stages=st)
    cos_g, cos_o = _cos(out, gold).min().item(), _cos(out, ref).min().item()
    print(f"{name}: cos vs golden {cos_g:.6f}, vs oracle {cos_o:.6f}")
    assert cos_g >= 1 - 1e-3 and cos_o >= 1 - 1e-3
    l0 = st["layer0"]
    assert (tap.cpu().view(N, T, -1) - l0).abs().max().item() <= 3e-2 * l0.abs().max().item()
    if "layer0" in g:
        assert np.abs(tap.cpu().view(N, T, -1).numpy() - g["layer0"]).max() <= 3e-2 * l0.abs().max().item()
    # the reference's call surface: text_encoder(text).logits, then normalise + cosine logits
    feats = enc(ids.to(device)).logits
    assert torch.equal(feats.cpu(), out)
    feats = feats / feats.norm(dim=1, keepdim=True)
    assert torch.allclose(feats.norm(dim=1), torch.ones(N, device=device), atol=1e-5)


def test_bert_shapes_and_errors(device):
    from oracle import bert_ref
    cfg = get_bert_config("tiny-bert-test")
    enc = mmr_amd.load_text_encoder("tiny-bert-test", device=device)
    w = weights.make_bert_weights(cfg)
    for n, T in [(1, 1), (5, 7), (2, 64), (9, 33)]:          # ragged lengths, max positions
        ids = _ids(cfg, n, T, seed=n * 100 + T) if T > 3 else torch.ones(n, T, dtype=torch.int32)
        out = enc.logits(ids.to(device)).cpu()
        with torch.no_grad():
            ref = bert_ref.bert_logits(w, cfg, ids)
        assert out.shape == (n, cfg.embed_dim) and _cos(out, ref).min().item() >= 1 - 1e-3, (n, T)
    with pytest.raises(ValueError):
        enc.logits(torch.ones(1, 65, dtype=torch.int32))      # longer than max positions
    with pytest.raises(IndexError):
        enc.logits(torch.full((1, 4), 5000, dtype=torch.int32))
    with pytest.raises(ValueError):
        enc(None)
    with pytest.raises(RuntimeError):
        mmr_amd.load_text_encoder("bert-base-uncased", device=device)
