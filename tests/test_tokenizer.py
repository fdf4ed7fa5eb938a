"""Host tokenizers pinned against the transformers tokenizers (the library the reference calls) on synthetic
vocabularies -- the real merge table / vocab.txt are not reachable offline."""
import json
import os

import pytest
import torch

from mmr_amd import tokenizer as T

MERGES = ["a n", "an d</w>", "t h", "th e</w>", "c a", "ca t</w>", "d o", "do g</w>", "p h", "ph o", "pho t", "phot o</w>",
          "o f</w>", "' s</w>", "4 2</w>", "! !", "h é", "l l", "ll o</w>"]
TEXTS = ["a photo of the cat and dog", "A Photo, of  THE cat's dog!!  42", "héllo wörld", "the   the\tthe", "", "x",
         "caterpillar dog <b>", "日本語 and cat"]


def _hf_clip(tmp_path):
    transformers = pytest.importorskip("transformers")
    tok = T.ClipBPETokenizer(merges=MERGES)
    vocab = dict(tok.encoder)
    return transformers.CLIPTokenizer(vocab=vocab, merges=[tuple(m.split()) for m in MERGES]), tok


def test_clip_bpe_matches_hf_tokenizer(tmp_path):
    hf, mine = _hf_clip(tmp_path)
    for t in TEXTS:
        want = hf(t)["input_ids"]
        got = [mine.sot] + mine.encode(t) + [mine.eot]
        assert got == want, (t, got, want)


def test_clip_tokenize_surface(tmp_path):
    mine = T.ClipBPETokenizer(merges=MERGES)
    # the clip package (unlike the HF tokenizer) html-unescapes first: basic_clean in clip/simple_tokenizer.py
    assert mine.encode("cat&amp;dog") == mine.encode("cat&dog")
    out = mine(["a photo of the cat", "dog"])
    assert out.shape == (2, 77) and out.dtype == torch.int32
    assert out[0, 0] == mine.sot and (out[0] == mine.eot).nonzero()[0, 0] == out[0].argmax()   # EOT is the largest id
    assert out[1, 3:].sum() == 0
    with pytest.raises(RuntimeError):
        mine("cat " * 100)
    assert mine("cat " * 100, truncate=True)[0, -1] == mine.eot
    p = tmp_path / "merges.txt"
    p.write_text("#version: 0.2\n" + "\n".join(MERGES) + "\n")
    assert torch.equal(T.ClipBPETokenizer(str(p))(TEXTS[:2]), mine(TEXTS[:2]))
    with pytest.raises(RuntimeError):
        T.ClipBPETokenizer()                                   # no table offline


BERT_VOCAB = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]", "荔", "枝", "羽", "毛", "球", "拍", "a", "photo", "##s", "of", "t",
              "##shirt", "-", ",", "shirt", "un", "##believ", "##able", "cafe", "!", "的", "图", "片"]
BERT_TEXTS = ["荔枝", "羽毛球拍 photos of t-shirt", "一张荔枝的图片", "Unbelievable, CAFÉ!", "a  photo\tof", "", "photoss"]


def test_bert_wordpiece_matches_hf_tokenizer(tmp_path):
    transformers = pytest.importorskip("transformers")
    vp = tmp_path / "vocab.txt"
    vp.write_text("\n".join(BERT_VOCAB) + "\n", encoding="utf-8")
    hf = transformers.BertTokenizer(str(vp))
    mine = T.BertWordPieceTokenizer(str(vp))
    want = hf(BERT_TEXTS, padding=True)["input_ids"]
    got = mine(BERT_TEXTS)
    assert got.tolist() == want, (got.tolist(), want)
    assert got.dtype == torch.int32 and got[0, 0] == mine.cls
    with pytest.raises(RuntimeError):
        T.BertWordPieceTokenizer()
