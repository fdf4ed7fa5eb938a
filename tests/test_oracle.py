"""CPU: the oracle against the committed golden vectors (no GPU, no transformers import).

The goldens are outputs of transformers.CLIPModel / the reference's torch expression,
written by oracle/make_golden.py in the dev container.
"""
import os

import numpy as np
import pytest
import torch

import mmr_amd
from mmr_amd import synth, weights
from oracle import clip_ref, search_ref


def _gold(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


@pytest.mark.parametrize("name,fn", [("tiny-test", "encoder_tiny-test.npz"), ("ViT-B/32", "encoder_ViT-B-32.npz"),
                                     ("ViT-B/16", "encoder_ViT-B-16.npz")])
def test_encoder_oracle_matches_hf_golden(golden_dir, name, fn):
    g = _gold(golden_dir, fn)
    ccfg = mmr_amd.get_config(name)
    w = weights.make_clip_weights(ccfg, seed=int(g["weight_seed"]))
    px = synth.synth_images(int(g["n_img"]), ccfg.vision.image_size, seed=int(g["image_seed"]))
    ids = synth.synth_token_ids(int(g["n_txt"]), ccfg.text.tokens, ccfg.text.vocab, seed=int(g["text_seed"]))
    st = {}
    with torch.no_grad():
        img = clip_ref.encode_image(w, ccfg.vision, px, stages=st)
        txt = clip_ref.encode_text(w, ccfg.text, ids)
        lpi, lpt = clip_ref.clip_forward(w, ccfg, px, ids)
    assert np.abs(img.numpy() - g["image_features"]).max() <= 3e-5
    assert np.abs(txt.numpy() - g["text_features"]).max() <= 3e-5
    assert np.abs(lpi.numpy() - g["logits_per_image"]).max() <= 1e-4
    assert np.abs(lpt.numpy() - g["logits_per_text"]).max() <= 1e-4
    if "v_ln_pre" in g:
        assert np.abs(st["ln_pre"].numpy() - g["v_ln_pre"]).max() <= 3e-5
        assert np.abs(st["layer0"].numpy() - g["v_layer0"]).max() <= 3e-5
        assert np.abs(st["pooled"].numpy() - g["v_pooled"]).max() <= 5e-5


def test_encoder_oracle_rejects_wrong_image_size():
    ccfg = mmr_amd.get_config("tiny-test")
    w = weights.make_vision_weights(ccfg.vision)
    with pytest.raises(ValueError):
        clip_ref.encode_image(w, ccfg.vision, torch.zeros(1, 3, 32, 32))


def test_text_pooling_is_first_eot():
    ids = synth.synth_token_ids(16, 77, 1024, seed=9)
    assert (ids[:, 0] == 1022).all()
    eot = ids.long().argmax(-1)
    for i in range(16):
        assert ids[i, eot[i]] == 1023 and (ids[i, eot[i] + 1:] == 0).all() and (ids[i, 1:eot[i]] < 1022).all()


@pytest.mark.parametrize("N", [1000, 10000])
@pytest.mark.parametrize("Q", [1, 7, 128])
@pytest.mark.parametrize("tag", ["f32", "bf16"])
def test_search_oracle_matches_golden(golden_dir, N, Q, tag):
    g = _gold(golden_dir, "search.npz")
    key = f"N{N}_Q{Q}_{tag}"
    gal = synth.synth_unit_rows(N, 512, seed=int(g["meta"][0]))
    q = synth.synth_unit_rows(Q, 512, seed=int(g[key + "_qseed"]))
    if tag == "bf16":
        gal, q = gal.bfloat16(), q.bfloat16()
    idx, score, s64 = search_ref.cosine_topk(q, gal, 10, scale=100.0)
    assert np.array_equal(idx.astype(np.int32), g[key + "_idx"])
    assert np.array_equal(score, g[key + "_score"])
    assert np.array_equal(s64, g[key + "_dot64"])
    # the reference's own expression agrees on these tie-free fixtures
    ridx, rval = search_ref.reference_expression_topk(gal.float(), q.float(), 10, 100.0)
    assert np.array_equal(ridx.numpy(), idx)
    assert np.abs(rval.numpy() - score).max() < 1e-3


def test_search_oracle_tie_rule(golden_dir):
    g = _gold(golden_dir, "search.npz")
    gal = synth.synth_unit_rows(2048, 512, seed=int(g["meta"][1])).bfloat16()
    q = synth.synth_unit_rows(5, 512, seed=int(g["meta"][2])).bfloat16()
    gal[1500:1520] = gal[7]
    gal[40] = gal[900]
    q[0] = gal[7]
    q[1] = gal[900]
    idx, score, s64 = search_ref.cosine_topk(q, gal, 10, scale=100.0)
    assert np.array_equal(idx.astype(np.int32), g["ties_idx"])
    assert list(idx[0]) == [7] + list(range(1500, 1509))      # ties -> lowest index first
    assert np.array_equal(s64, g["ties_dot64"])


def test_search_oracle_edge_cases():
    gal = synth.synth_unit_rows(5, 128, seed=3)
    q = synth.synth_unit_rows(2, 128, seed=4)
    idx, score, _ = search_ref.cosine_topk(q, gal, 8)          # k > N
    assert (idx[:, 5:] == -1).all() and np.isinf(score[:, 5:]).all()
    assert sorted(idx[0, :5].tolist()) == [0, 1, 2, 3, 4]
    sim = search_ref.similarity(q, gal, scale=100.0)
    assert np.allclose(sim, 100.0 * (q @ gal.t()).numpy(), atol=1e-4)
    # merge of two shards == search over the concatenation
    a, b = gal[:3], gal[3:]
    ia, _, sa = search_ref.cosine_topk(q, a, 4)
    ib, _, sb = search_ref.cosine_topk(q, b, 4)
    ib = np.where(ib >= 0, ib + 3, ib)
    mi, ms, _ = search_ref.topk_merge(np.stack([ia, ib]), np.stack([sa, sb]))
    fi, fs, _ = search_ref.cosine_topk(q, gal, 4)
    assert np.array_equal(mi, fi) and np.array_equal(ms, fs)
    n = search_ref.l2norm_rows(torch.randn(4, 128) * 3)
    assert np.allclose(np.linalg.norm(n, axis=1), 1.0, atol=1e-6)
