"""Checkpoint name maps (mmr_amd.checkpoint): OpenAI CLIP / HF CLIPModel / HF BERT -> this package's names.

CPU only.  The HF maps are checked against ``transformers`` itself (the backend the reference calls,
code/test_taiyi.py:12-24): a randomly initialised HF model's own state dict is converted, and the
oracle run on the converted weights must reproduce that model's outputs.  The OpenAI map has no
importable counterpart here (the ``clip`` package is absent): it is checked structurally, against the
tensor names and layouts of that package's ``CLIP`` module."""
import os

import pytest
import torch

import mmr_amd
from mmr_amd import checkpoint, config, synth, weights

transformers = pytest.importorskip("transformers")


def _tiny_hf_clip():
    from oracle.hf_adapter import build_hf_clip
    ccfg = mmr_amd.get_config("tiny-test")
    w = weights.make_clip_weights(ccfg, seed=3)
    return ccfg, w, build_hf_clip(ccfg, w)


def test_hf_clip_state_dict_converts_and_reproduces_hf_outputs():
    from oracle import clip_ref
    from oracle.hf_adapter import hf_image_features, hf_text_features
    ccfg, w, hf = _tiny_hf_clip()
    conv = checkpoint.from_hf_clip_state_dict(hf.state_dict())
    assert set(conv) == set(w)
    for k in w:
        assert torch.equal(conv[k], w[k]), k
    inferred = checkpoint.infer_clip_config(conv, "tiny-test")
    assert (inferred.vision, inferred.text, inferred.embed_dim) == (ccfg.vision, ccfg.text, ccfg.embed_dim)
    px = synth.synth_images(2, ccfg.vision.image_size, seed=1)
    ids = synth.synth_token_ids(3, ccfg.text.tokens, ccfg.text.vocab, seed=2)
    with torch.no_grad():
        assert torch.allclose(clip_ref.encode_image(conv, inferred.vision, px), hf_image_features(hf, px), atol=2e-5)
        assert torch.allclose(clip_ref.encode_text(conv, inferred.text, ids), hf_text_features(hf, ids), atol=2e-5)
    # and the inverse map loads back into transformers without missing/unexpected weights
    back = checkpoint.to_hf_clip_state_dict(conv)
    missing, unexpected = hf.load_state_dict(back, strict=False)
    assert not unexpected and all("position_ids" in m for m in missing)


def test_openai_naming_round_trip_and_layout():
    ccfg = mmr_amd.get_config("tiny-test")
    w = weights.make_clip_weights(ccfg, seed=4)
    sd = checkpoint.to_openai_state_dict(w)
    d, E = ccfg.vision.width, ccfg.embed_dim
    # names and layouts of the OpenAI `clip` package's CLIP module
    assert sd["visual.proj"].shape == (d, E) and sd["text_projection"].shape == (ccfg.text.width, E)   # [d,E], x @ proj
    assert sd["visual.conv1.weight"].shape == (d, 3, ccfg.vision.patch, ccfg.vision.patch)
    assert sd["visual.transformer.resblocks.1.attn.in_proj_weight"].shape == (3 * d, d)
    assert sd["transformer.resblocks.0.mlp.c_fc.weight"].shape == (ccfg.text.mlp, ccfg.text.width)
    assert sd["positional_embedding"].shape == (ccfg.text.tokens, ccfg.text.width)
    assert len(sd) == 14 + 12 * (ccfg.vision.layers + ccfg.text.layers)
    # wrapped in DataParallel-style prefixes and fp16, as clip.load leaves them on CUDA
    wrapped = {"module." + k: v.half() for k, v in sd.items()}
    back = checkpoint.from_openai_state_dict(wrapped)
    assert set(back) == set(w)
    for k in w:
        assert back[k].dtype == torch.float32
        assert torch.equal(back[k], w[k].half().float()), k
    kind, auto = checkpoint.convert_state_dict(sd)
    assert kind == "clip" and torch.equal(auto["v.proj"], w["v.proj"])


def test_hf_bert_state_dict_converts_and_reproduces_hf_logits():
    from oracle import bert_ref
    from oracle.hf_adapter import build_hf_bert
    cfg = config.get_bert_config("tiny-bert-test")
    w = weights.make_bert_weights(cfg, seed=5)
    hf = build_hf_bert(cfg, w)
    kind, conv = checkpoint.convert_state_dict(hf.state_dict())
    assert kind == "bert" and set(conv) == set(w)
    for k in w:
        assert torch.equal(conv[k], w[k]), k
    inferred = checkpoint.infer_bert_config(conv, "tiny-bert-test")
    assert (inferred.width, inferred.layers, inferred.mlp, inferred.vocab, inferred.embed_dim) == \
           (cfg.width, cfg.layers, cfg.mlp, cfg.vocab, cfg.embed_dim)
    ids = torch.randint(1, cfg.vocab, (3, 12), generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        ref = hf(input_ids=ids).logits
        got = bert_ref.bert_logits(conv, inferred, ids)
    assert torch.allclose(got, ref, atol=2e-5)


def test_checkpoint_files_safe_loaders(tmp_path):
    ccfg, w, hf = _tiny_hf_clip()
    from safetensors.torch import save_file
    st = tmp_path / "model.safetensors"
    save_file({k: v.contiguous() for k, v in hf.state_dict().items()}, str(st))
    cfg1, w1 = checkpoint.load_clip_checkpoint(str(st), "tiny-test")
    pt = tmp_path / "tiny.state.pt"
    torch.save({"state_dict": checkpoint.to_openai_state_dict(w)}, str(pt))
    cfg2, w2 = checkpoint.load_clip_checkpoint(str(pt))
    assert cfg1.vision == cfg2.vision == ccfg.vision
    for k in w:
        assert torch.equal(w1[k], w[k]) and torch.equal(w2[k], w[k]), k
    # a pickled module (what a TorchScript / full-model save looks like to the safe loader) is refused, not executed
    bad = tmp_path / "module.pt"
    torch.save(torch.nn.Linear(2, 2), str(bad))
    with pytest.raises(RuntimeError, match="not a plain tensor state dict"):
        checkpoint.read_state_dict(str(bad))
    with pytest.raises(KeyError, match="unrecognised checkpoint"):
        checkpoint.convert_state_dict({"foo": torch.zeros(1)})
    with pytest.raises(FileNotFoundError):
        checkpoint.read_state_dict(str(tmp_path / "absent.pt"))


@pytest.mark.gpu
def test_load_from_checkpoint_file_matches_weight_dict(tmp_path):
    """clip.load(<file>) and clip.load(name, download_root=...) run the same towers as the weight dict."""
    device = torch.device("cuda:0")
    ccfg = mmr_amd.get_config("tiny-test")
    w = weights.make_clip_weights(ccfg, seed=6)
    from safetensors.torch import save_file
    save_file({k: v.contiguous() for k, v in checkpoint.to_hf_clip_state_dict(w).items()}, str(tmp_path / "tiny-test.safetensors"))
    px = synth.synth_images(3, ccfg.vision.image_size, seed=1).to(device)
    ref_model, _ = mmr_amd.load("tiny-test", device=device, weights=w)
    ref = ref_model.encode_image(px)
    m1, _ = mmr_amd.load(str(tmp_path / "tiny-test.safetensors"), device=device)
    m2, _ = mmr_amd.load("tiny-test", device=device, download_root=str(tmp_path))
    m3, _ = mmr_amd.load("tiny-test", device=device, weights=checkpoint.to_openai_state_dict(w))
    for m in (m1, m2, m3):
        assert torch.equal(m.encode_image(px), ref)
    with pytest.raises(ValueError, match="does not match"):
        mmr_amd.load("ViT-B/32", device=device, weights=w)
