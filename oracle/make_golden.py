"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/*.npz (run in the dev container).

    python oracle/make_golden.py [--skip-large]

What it pins
  * encoder: outputs of ``transformers.CLIPModel`` (5.15.0, eager attention, fp32, CPU) --
    the third-party backend the reference calls (code/test_taiyi.py:17-30,
    CLIP-Chinese/lab_chinese.py:83-114) -- built from a LOCAL config and this repo's seeded
    weights.  Before writing, it asserts oracle/clip_ref.py reproduces them to <= 2e-5.
  * search: oracle/search_ref.c outputs on seeded galleries, cross-checked here against the
    reference's own torch expression ``100. * features @ ref.t()`` + ``topk``
    (code/search_image.py:107, code/utils.py:17) on tie-free data, with the minimum top-(k+1)
    gap recorded so the fixtures cannot flip under fp32 summation-order noise.

Inputs are NOT stored: tests regenerate them from the seeds recorded in each file via
``mmr_amd.synth`` / ``mmr_amd.weights``.  The reference's own known-answer value
(code/test_clip.py:17) needs pretrained weights + CLIP.png (absent offline): parity at that
boundary stays unpinned by the reference itself, as SURVEY.md section 4 records.
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import mmr_amd  # noqa: E402
from mmr_amd import synth, weights  # noqa: E402
from oracle import clip_ref, hf_adapter, search_ref  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def _maxdiff(a, b):
    return float((a - b).abs().max())


@torch.no_grad()
def encoder_golden(name, n_img, n_txt, with_stages, with_text=True, wseed=0):
    ccfg = mmr_amd.get_config(name)
    w = weights.make_clip_weights(ccfg, seed=wseed)
    hf = hf_adapter.build_hf_clip(ccfg, w)
    px = synth.synth_images(n_img, ccfg.vision.image_size, seed=0)
    out = {"weight_seed": wseed, "image_seed": 0, "n_img": n_img}

    img_hf = hf_adapter.hf_image_features(hf, px)
    st = {}
    img_or = clip_ref.encode_image(w, ccfg.vision, px, stages=st)
    d = _maxdiff(img_hf, img_or)
    print(f"[{name}] image features oracle-vs-HF max|diff| = {d:.3e} (|f|~{float(img_hf.norm(dim=-1).mean()):.2f})")
    assert d <= 2e-5 * max(1.0, float(img_hf.abs().max())), d
    out["image_features"] = img_hf.numpy()

    if with_stages:
        vo = hf.vision_model(pixel_values=px, output_hidden_states=True)
        hs = vo.hidden_states
        # hidden_states[0] = after pre_layrnorm; [i] = after layer i
        assert _maxdiff(hs[0], st["ln_pre"]) <= 2e-5
        assert _maxdiff(hs[1], st["layer0"]) <= 2e-5
        assert _maxdiff(hs[-1], st[f"layer{ccfg.vision.layers - 1}"]) <= 5e-5
        out["v_ln_pre"] = hs[0].numpy()
        out["v_layer0"] = hs[1].numpy()
        out["v_layer_last"] = hs[-1].numpy()
        out["v_pooled"] = vo.pooler_output.numpy()
        assert _maxdiff(vo.pooler_output, st["pooled"]) <= 5e-5

    if with_text:
        ids = synth.synth_token_ids(n_txt, ccfg.text.tokens, ccfg.text.vocab, seed=5)
        txt_hf = hf_adapter.hf_text_features(hf, ids.long())
        tst = {}
        txt_or = clip_ref.encode_text(w, ccfg.text, ids, stages=tst)
        d = _maxdiff(txt_hf, txt_or)
        print(f"[{name}] text features  oracle-vs-HF max|diff| = {d:.3e}")
        assert d <= 2e-5 * max(1.0, float(txt_hf.abs().max())), d
        out["text_seed"] = 5
        out["n_txt"] = n_txt
        out["text_features"] = txt_hf.numpy()
        if with_stages:
            to = hf.text_model(input_ids=ids.long(), output_hidden_states=True)
            assert _maxdiff(to.hidden_states[1], tst["layer0"]) <= 2e-5
            out["t_layer0"] = to.hidden_states[1].numpy()
            out["t_pooled"] = to.pooler_output.numpy()
        full = hf(pixel_values=px, input_ids=ids.long())
        lpi, lpt = clip_ref.clip_forward(w, ccfg, px, ids)
        assert _maxdiff(full.logits_per_image, lpi) <= 1e-4
        out["logits_per_image"] = full.logits_per_image.numpy()
        out["logits_per_text"] = full.logits_per_text.numpy()

    fn = os.path.join(GOLD, "encoder_" + name.replace("/", "-").replace("@", "_") + ".npz")
    np.savez_compressed(fn, **out)
    print("wrote", fn, os.path.getsize(fn), "bytes")


def search_golden():
    out = {}
    E, k = 512, 10
    for N in (1000, 10000):
        g32 = synth.synth_unit_rows(N, E, seed=1)
        for Q in (1, 7, 128):
            # pick the first query seed whose fixture has no near-tie in either dtype
            for qseed in range(4, 64):
                q32 = synth.synth_unit_rows(Q, E, seed=qseed)
                gaps = []
                for g, q in ((g32, q32), (g32.bfloat16(), q32.bfloat16())):
                    _, _, s11 = search_ref.cosine_topk(q, g, k + 1, scale=100.0)
                    gaps.append(float(np.min(s11[:, :-1] - s11[:, 1:])))
                if min(gaps) > 2e-6:
                    break
            else:
                raise AssertionError("no tie-free query seed found")
            for tag, g, q in (("f32", g32, q32), ("bf16", g32.bfloat16(), q32.bfloat16())):
                idx, score, s64 = search_ref.cosine_topk(q, g, k, scale=100.0)
                # cross-check with the reference's own expression (fp32 torch matmul + topk)
                ridx, rval = search_ref.reference_expression_topk(g.float(), q.float(), k, 100.0)
                _, _, s11 = search_ref.cosine_topk(q, g, k + 1, scale=100.0)
                gap = float(np.min(s11[:, :-1] - s11[:, 1:]))
                assert np.array_equal(ridx.numpy(), idx), "oracle disagrees with reference expression"
                assert np.abs(rval.numpy() - score).max() < 1e-3
                key = f"N{N}_Q{Q}_{tag}"
                out[key + "_idx"] = idx.astype(np.int32)
                out[key + "_score"] = score
                out[key + "_dot64"] = s64
                out[key + "_mingap"] = np.float64(gap)
                out[key + "_qseed"] = np.int64(qseed)
                print(f"search {key}: qseed {qseed}, min top-(k+1) gap {gap:.3e}; matches reference expression")
    # adversarial tie fixture: duplicated gallery rows pin the (-score,+index) rule
    N, Q = 2048, 5
    g = synth.synth_unit_rows(N, E, seed=11).bfloat16()
    q = synth.synth_unit_rows(Q, E, seed=12).bfloat16()
    g[1500:1520] = g[7]          # 21 copies of row 7 in total
    g[40] = g[900]               # a pair
    q[0] = g[7]                  # query 0's best match is the duplicated row
    q[1] = g[900]
    idx, score, s64 = search_ref.cosine_topk(q, g, k, scale=100.0)
    assert list(idx[0][:10]) == [7] + list(range(1500, 1509)), idx[0]
    assert list(idx[1][:2]) == [40, 900], idx[1]
    out["ties_idx"] = idx.astype(np.int32)
    out["ties_score"] = score
    out["ties_dot64"] = s64
    out["meta"] = np.array([1, 11, 12])  # gallery seed, tie gallery seed, tie query seed
    fn = os.path.join(GOLD, "search.npz")
    np.savez_compressed(fn, **out)
    print("wrote", fn, os.path.getsize(fn), "bytes")


@torch.no_grad()
def bert_golden(name, n_txt, T, with_stages):
    from mmr_amd.config import get_bert_config
    from oracle import bert_ref
    cfg = get_bert_config(name)
    w = weights.make_bert_weights(cfg, seed=0)
    hf = hf_adapter.build_hf_bert(cfg, w)
    g = torch.Generator().manual_seed(7)
    ids = torch.randint(1, cfg.vocab, (n_txt, T), generator=g, dtype=torch.int32)
    ids[:, 0] = 101 % cfg.vocab                      # [CLS]-like first token
    ids[1:, T - 3:] = 0                              # trailing [PAD]s: attended to, as in the reference's call
    out_hf = hf(ids.long())                          # ids only, exactly like code/test_taiyi.py:24
    st = {}
    out_or = bert_ref.bert_logits(w, cfg, ids, stages=st)
    d = _maxdiff(out_hf.logits, out_or)
    print(f"[{name}] BERT logits oracle-vs-HF max|diff| = {d:.3e} (|f|~{float(out_hf.logits.norm(dim=-1).mean()):.2f})")
    assert d <= 5e-5 * max(1.0, float(out_hf.logits.abs().max())), d
    out = {"weight_seed": 0, "ids_seed": 7, "n_txt": n_txt, "T": T, "logits": out_hf.logits.numpy()}
    # the tokenizer-output form (reference CLIP/union_dataset.py:312-314): padded keys masked, some type-1 tokens
    mask = (ids != 0).long()
    types = torch.zeros_like(ids).long()
    types[:, T // 2:] = 1
    types = types * mask
    out_m = hf(input_ids=ids.long(), attention_mask=mask, token_type_ids=types).logits
    or_m = bert_ref.bert_logits(w, cfg, ids, attention_mask=mask, token_type_ids=types)
    dm = _maxdiff(out_m, or_m)
    print(f"[{name}] BERT masked logits oracle-vs-HF max|diff| = {dm:.3e}; masked-vs-unmasked {_maxdiff(out_m, out_hf.logits):.3e}")
    assert dm <= 5e-5 * max(1.0, float(out_m.abs().max())), dm
    out["logits_masked"] = out_m.numpy()
    if with_stages:
        hs = hf.bert(ids.long(), output_hidden_states=True).hidden_states
        assert _maxdiff(hs[0], st["embed"]) <= 2e-5 and _maxdiff(hs[1], st["layer0"]) <= 5e-5
        out["embed"] = hs[0].numpy()
        out["layer0"] = hs[1].numpy()
    fn = os.path.join(GOLD, "bert_" + name + ".npz")
    np.savez_compressed(fn, **out)
    print("wrote", fn, os.path.getsize(fn), "bytes")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-large", action="store_true", help="skip the ViT-L/14 goldens (slow, ~3 GB RAM)")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(8)
    if a.only in ("", "search"):
        search_golden()
    if a.only in ("", "tiny"):
        encoder_golden("tiny-test", 4, 3, with_stages=True)
    if a.only in ("", "b32"):
        encoder_golden("ViT-B/32", 4, 3, with_stages=False)
    if a.only in ("", "b16"):
        encoder_golden("ViT-B/16", 2, 2, with_stages=False)
    if not a.skip_large and a.only in ("", "l14"):
        encoder_golden("ViT-L/14", 1, 2, with_stages=False, with_text=True)
    if a.only in ("", "bert"):
        bert_golden("tiny-bert-test", 3, 19, with_stages=True)
        if not a.skip_large:
            bert_golden("Taiyi-CLIP-Roberta-large-326M-Chinese", 2, 12, with_stages=False)
    if not a.skip_large and a.only in ("", "l14_336"):
        encoder_golden("ViT-L/14@336px", 1, 0, with_stages=False, with_text=False)


if __name__ == "__main__":
    main()
