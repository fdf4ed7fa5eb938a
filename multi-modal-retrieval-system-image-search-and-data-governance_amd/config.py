"""Model geometry for the CLIP towers on the hot path.

Shapes follow the backbones the reference scripts request:
``clip.load("ViT-B/32")`` (reference code/search_image.py:327, code/test_clip.py:6) and
``CLIPModel.from_pretrained("openai/clip-vit-large-patch14")`` (code/test_taiyi.py:17,
CLIP-Chinese/lab_chinese.py:83).  ViT-L/14@336px is BASELINE.json configs[4].
Head dim is 64 for every tower (width / heads), which the attention kernel relies on.
"""
from dataclasses import dataclass, field


@dataclass(frozen=True)
class TowerConfig:
    kind: str            # "vision" | "text"
    width: int           # d
    layers: int          # L
    heads: int           # h (width // 64)
    mlp: int             # m
    tokens: int          # T (vision: 1 + grid^2, text: context length)
    embed_dim: int       # E (projection output)
    image_size: int = 0  # vision only
    patch: int = 0       # vision only
    vocab: int = 0       # text only
    ln_eps: float = 1e-5

    @property
    def grid(self) -> int:
        return self.image_size // self.patch if self.patch else 0

    @property
    def patch_k(self) -> int:
        """Contraction length of the patch-embed conv-as-GEMM (3*P*P)."""
        return 3 * self.patch * self.patch

    @property
    def patch_k_pad(self) -> int:
        """patch_k rounded up to the GEMM K granule (64)."""
        return (self.patch_k + 63) // 64 * 64


@dataclass(frozen=True)
class ClipConfig:
    name: str
    vision: TowerConfig
    text: TowerConfig
    embed_dim: int
    logit_scale_init: float = 2.6592  # ln(1/0.07), HF/OpenAI initial value


def _clip(name, image, patch, vw, vl, vmlp, tw, tl, tmlp, embed, ctx=77, vocab=49408):
    g = image // patch
    v = TowerConfig("vision", vw, vl, vw // 64, vmlp, 1 + g * g, embed, image_size=image, patch=patch)
    t = TowerConfig("text", tw, tl, tw // 64, tmlp, ctx, embed, vocab=vocab)
    return ClipConfig(name, v, t, embed)


MODEL_CONFIGS = {
    # OpenAI CLIP ViT-B/32: the backbone every EN script in the reference loads.
    "ViT-B/32": _clip("ViT-B/32", 224, 32, 768, 12, 3072, 512, 12, 2048, 512),
    # OpenAI CLIP ViT-B/16: not loaded by the reference's scripts, listed by `clip.available_models()`; same towers
    # as B/32 with 14x14 patches per side (197 tokens -> the streaming attention kernel).
    "ViT-B/16": _clip("ViT-B/16", 224, 16, 768, 12, 3072, 512, 12, 2048, 512),
    # openai/clip-vit-large-patch14 (HF), used by the CN pipeline's image tower.
    "ViT-L/14": _clip("ViT-L/14", 224, 14, 1024, 24, 4096, 768, 12, 3072, 768),
    "ViT-L/14@336px": _clip("ViT-L/14@336px", 336, 14, 1024, 24, 4096, 768, 12, 3072, 768),
    # Small geometry for fast parity tests (same code paths, 2 layers).
    "tiny-test": _clip("tiny-test", 64, 16, 128, 2, 512, 128, 2, 512, 128, ctx=77, vocab=1024),
}

@dataclass(frozen=True)
class BertTextConfig:
    """BERT-style sequence classifier used as a text tower (its logits ARE the text embedding):
    ``BertForSequenceClassification.from_pretrained("IDEA-CCNL/Taiyi-CLIP-Roberta-large-326M-Chinese")``
    (reference code/test_taiyi.py:12,24; CLIP-Chinese/lab_chinese.py:81-93)."""
    name: str
    width: int
    layers: int
    heads: int
    mlp: int
    max_positions: int
    vocab: int
    embed_dim: int       # num_labels = dimension of the logits
    ln_eps: float = 1e-12
    kind: str = "bert"

    @property
    def tokens(self) -> int:
        return self.max_positions


BERT_CONFIGS = {
    # Taiyi-CLIP-Roberta-large-326M-Chinese: BERT-large geometry, 21128-token Chinese vocabulary, 768-d logits
    "Taiyi-CLIP-Roberta-large-326M-Chinese": BertTextConfig("Taiyi-CLIP-Roberta-large-326M-Chinese", 1024, 24, 16,
                                                            4096, 512, 21128, 768),
    "tiny-bert-test": BertTextConfig("tiny-bert-test", 128, 2, 2, 512, 64, 1000, 128),
}
BERT_ALIASES = {"IDEA-CCNL/Taiyi-CLIP-Roberta-large-326M-Chinese": "Taiyi-CLIP-Roberta-large-326M-Chinese"}


def get_bert_config(name: str) -> BertTextConfig:
    name = BERT_ALIASES.get(name, name)
    if name not in BERT_CONFIGS:
        raise RuntimeError(f"Text encoder {name} not found; available = {list(BERT_CONFIGS)}")
    return BERT_CONFIGS[name]


# HF hub ids the reference passes to from_pretrained, mapped to the same geometry.
MODEL_ALIASES = {
    "openai/clip-vit-base-patch32": "ViT-B/32",
    "openai/clip-vit-base-patch16": "ViT-B/16",
    "openai/clip-vit-large-patch14": "ViT-L/14",
    "openai/clip-vit-large-patch14-336": "ViT-L/14@336px",
}


def available_models():
    """Mirror of ``clip.available_models()``."""
    return [k for k in MODEL_CONFIGS if k != "tiny-test"]


def get_config(name: str) -> ClipConfig:
    name = MODEL_ALIASES.get(name, name)
    if name not in MODEL_CONFIGS:
        raise RuntimeError(f"Model {name} not found; available models = {available_models()}")
    return MODEL_CONFIGS[name]
