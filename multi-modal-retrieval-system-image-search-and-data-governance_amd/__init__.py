"""MI355X-native CLIP encode + cosine top-k, behind the call surface the reference uses.

Public surface (mirrors ``import clip`` as used in reference code/test_clip.py:6-16 and
code/search_image.py:327-338, plus the HF flavour of code/test_taiyi.py:17-30):

    model, preprocess = load("ViT-B/32", device="cuda")
    tokens = tokenize(...)                       # ids pass-through (BPE vocab absent offline)
    model.encode_image(x) / model.encode_text(t) / model(image, text)
    model.get_image_features(pixel_values=x)     # HF spelling
    similarity(features, ref, scale=100.)        # code/search_image.py:107
    cosine_topk(queries, gallery, k)             # code/utils.py:17 generalised
    GalleryIndex / ShardedGalleryIndex           # row-sharded gallery, RCCL all-gather of top-k

Everything that computes runs in hand-written HIP kernels from ``csrc/libmmr_hip.so``
through the C ABI declared in ``include/mmr.h``; there is no CPU fallback.
"""
from .config import MODEL_CONFIGS, ClipConfig, TowerConfig, available_models, get_config  # noqa: F401

__all__ = [
    "available_models", "get_config", "load", "tokenize", "similarity", "cosine_topk", "l2_normalize",
    "GalleryIndex", "ShardedGalleryIndex", "CLIP",
]

_LAZY = {
    "load": "clip", "tokenize": "clip", "CLIP": "clip",
    "similarity": "search", "cosine_topk": "search", "l2_normalize": "search",
    "GalleryIndex": "search", "ShardedGalleryIndex": "search", "merge_topk": "search",
    "tip_adapter_logits": "search", "load_text_encoder": "bert", "BertTextEncoder": "bert", "encode_gallery": "gallery", "build_cache": "gallery",
}


def __getattr__(name):
    # Lazy so that config/weights/synth import on a CPU-only box without touching the GPU lib.
    if name in _LAZY:
        import importlib

        mod = importlib.import_module(f"{__name__}.{_LAZY[name]}")
        return getattr(mod, name)
    if name in ("synth", "weights", "search", "clip", "config", "_lib", "gallery", "preprocess", "bert", "tokenizer", "checkpoint"):
        import importlib

        return importlib.import_module(f"{__name__}.{name}")
    raise AttributeError(name)
