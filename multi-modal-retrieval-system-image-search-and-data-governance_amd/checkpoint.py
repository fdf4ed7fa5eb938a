"""Real checkpoints -> this package's flat weight names (weights.py) and back.

The reference gets its weights from ``clip.load("ViT-B/32")`` (an OpenAI CLIP state dict,
reference code/search_image.py:327, code/test_clip.py:6) and from
``CLIPModel.from_pretrained("openai/clip-vit-large-patch14")`` /
``BertForSequenceClassification.from_pretrained("IDEA-CCNL/Taiyi-CLIP-Roberta-large-326M-Chinese")``
(HuggingFace state dicts, code/test_taiyi.py:12,17).  Neither can be downloaded here, but a user who
has the files can hand them over: this module maps the three naming schemes onto the names the
towers are packed from, infers the geometry from tensor shapes, and reads checkpoint files with
loaders that execute nothing from the file (safetensors, ``torch.load(weights_only=True)``).

Pure tensor bookkeeping -- no arithmetic, no GPU.  The name maps are the ones in SURVEY.md
Appendix A; ``tests/test_checkpoint.py`` checks the HF maps against ``transformers`` itself.
"""
import math
import os
import re
from typing import Dict, Tuple

import torch

from .config import BertTextConfig, ClipConfig, TowerConfig

Weights = Dict[str, torch.Tensor]

# per-block tensors: (ours, OpenAI CLIP resblock suffix, HF CLIP encoder-layer suffix)
_BLOCK = (
    ("ln1.w", "ln_1.weight", "layer_norm1.weight"), ("ln1.b", "ln_1.bias", "layer_norm1.bias"),
    ("out.w", "attn.out_proj.weight", "self_attn.out_proj.weight"), ("out.b", "attn.out_proj.bias", "self_attn.out_proj.bias"),
    ("ln2.w", "ln_2.weight", "layer_norm2.weight"), ("ln2.b", "ln_2.bias", "layer_norm2.bias"),
    ("fc1.w", "mlp.c_fc.weight", "mlp.fc1.weight"), ("fc1.b", "mlp.c_fc.bias", "mlp.fc1.bias"),
    ("fc2.w", "mlp.c_proj.weight", "mlp.fc2.weight"), ("fc2.b", "mlp.c_proj.bias", "mlp.fc2.bias"),
)


def _f32(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to("cpu", torch.float32).contiguous()


def _count_layers(sd, pattern: str) -> int:
    rx = re.compile(pattern)
    idx = [int(m.group(1)) for k in sd for m in [rx.match(k)] if m]
    if not idx:
        raise KeyError(f"no tensors match {pattern!r}")
    return max(idx) + 1


def _strip_prefix(sd):
    """Drop wrapper prefixes (``module.``, ``model.``, ``clip_model.``) some training scripts add."""
    for pre in ("module.", "model.", "clip_model."):
        if sd and all(k.startswith(pre) for k in sd):
            sd = {k[len(pre):]: v for k, v in sd.items()}
    return sd


# --------------------------------------------------------------------------------------------- CLIP
def infer_clip_config(w: Weights, name: str = "checkpoint") -> ClipConfig:
    """Geometry of a weight dict in this package's naming (head dim is 64 in every CLIP tower)."""
    d, P = int(w["v.cls"].numel()), int(w["v.patch_w"].shape[-1])
    T = int(w["v.pos"].shape[0])
    g = math.isqrt(T - 1)
    if g * g != T - 1:
        raise ValueError(f"vision position table has {T} rows: not 1 + a square grid")
    vl = _count_layers(w, r"v\.l(\d+)\.qkv\.w$")
    E = int(w["v.proj"].shape[0])
    vision = TowerConfig("vision", d, vl, d // 64, int(w["v.l0.fc1.w"].shape[0]), T, E, image_size=g * P, patch=P)
    if "t.tok" in w:
        td = int(w["t.tok"].shape[1])
        tl = _count_layers(w, r"t\.l(\d+)\.qkv\.w$")
        text = TowerConfig("text", td, tl, td // 64, int(w["t.l0.fc1.w"].shape[0]), int(w["t.pos"].shape[0]), E,
                           vocab=int(w["t.tok"].shape[0]))
    else:
        raise KeyError("checkpoint has no text tower (t.tok)")
    for t in (vision, text):
        if t.width % 64:
            raise ValueError(f"tower width {t.width} is not a multiple of the head dim 64")
    scale = float(w["logit_scale"]) if "logit_scale" in w else 2.6592
    return ClipConfig(name, vision, text, E, logit_scale_init=scale)


def from_openai_state_dict(sd) -> Weights:
    """OpenAI ``clip`` package state dict (``clip.load(...)[0].state_dict()``) -> this package's names.
    ``visual.proj`` / ``text_projection`` are stored [d,E] there and [E,d] (nn.Linear layout) here."""
    sd = _strip_prefix(dict(sd))
    if "visual.conv1.weight" not in sd:
        raise KeyError("not an OpenAI CLIP ViT state dict (visual.conv1.weight missing; ResNet CLIPs are not supported)")
    w: Weights = {
        "v.patch_w": _f32(sd["visual.conv1.weight"]), "v.cls": _f32(sd["visual.class_embedding"]),
        "v.pos": _f32(sd["visual.positional_embedding"]),
        "v.ln_pre.w": _f32(sd["visual.ln_pre.weight"]), "v.ln_pre.b": _f32(sd["visual.ln_pre.bias"]),
        "v.ln_post.w": _f32(sd["visual.ln_post.weight"]), "v.ln_post.b": _f32(sd["visual.ln_post.bias"]),
        "v.proj": _f32(sd["visual.proj"]).t().contiguous(),
        "t.tok": _f32(sd["token_embedding.weight"]), "t.pos": _f32(sd["positional_embedding"]),
        "t.ln_final.w": _f32(sd["ln_final.weight"]), "t.ln_final.b": _f32(sd["ln_final.bias"]),
        "t.proj": _f32(sd["text_projection"]).t().contiguous(),
        "logit_scale": _f32(sd["logit_scale"]).reshape(()),
    }
    for pre, root in (("v", "visual.transformer"), ("t", "transformer")):
        for i in range(_count_layers(sd, re.escape(root) + r"\.resblocks\.(\d+)\.attn\.in_proj_weight$")):
            src = f"{root}.resblocks.{i}"
            w[f"{pre}.l{i}.qkv.w"] = _f32(sd[f"{src}.attn.in_proj_weight"])
            w[f"{pre}.l{i}.qkv.b"] = _f32(sd[f"{src}.attn.in_proj_bias"])
            for ours, oa, _ in _BLOCK:
                w[f"{pre}.l{i}.{ours}"] = _f32(sd[f"{src}.{oa}"])
    return w


def to_openai_state_dict(w: Weights) -> Weights:
    """Inverse of :func:`from_openai_state_dict` (export / tests)."""
    sd: Weights = {
        "visual.conv1.weight": w["v.patch_w"], "visual.class_embedding": w["v.cls"],
        "visual.positional_embedding": w["v.pos"], "visual.ln_pre.weight": w["v.ln_pre.w"],
        "visual.ln_pre.bias": w["v.ln_pre.b"], "visual.ln_post.weight": w["v.ln_post.w"],
        "visual.ln_post.bias": w["v.ln_post.b"], "visual.proj": w["v.proj"].t().contiguous(),
        "token_embedding.weight": w["t.tok"], "positional_embedding": w["t.pos"],
        "ln_final.weight": w["t.ln_final.w"], "ln_final.bias": w["t.ln_final.b"],
        "text_projection": w["t.proj"].t().contiguous(), "logit_scale": w["logit_scale"],
    }
    for pre, root in (("v", "visual.transformer"), ("t", "transformer")):
        for i in range(_count_layers(w, re.escape(pre) + r"\.l(\d+)\.qkv\.w$")):
            dst = f"{root}.resblocks.{i}"
            sd[f"{dst}.attn.in_proj_weight"] = w[f"{pre}.l{i}.qkv.w"]
            sd[f"{dst}.attn.in_proj_bias"] = w[f"{pre}.l{i}.qkv.b"]
            for ours, oa, _ in _BLOCK:
                sd[f"{dst}.{oa}"] = w[f"{pre}.l{i}.{ours}"]
    return sd


def from_hf_clip_state_dict(sd) -> Weights:
    """``transformers.CLIPModel.state_dict()`` -> this package's names (q/k/v fused row-wise into
    qkv [3d,d], the order the QKV GEMM and the attention kernel expect)."""
    sd = _strip_prefix(dict(sd))
    if "vision_model.embeddings.patch_embedding.weight" not in sd:
        raise KeyError("not a HF CLIPModel state dict (vision_model.embeddings.patch_embedding.weight missing)")
    w: Weights = {
        "v.patch_w": _f32(sd["vision_model.embeddings.patch_embedding.weight"]),
        "v.cls": _f32(sd["vision_model.embeddings.class_embedding"]),
        "v.pos": _f32(sd["vision_model.embeddings.position_embedding.weight"]),
        "v.ln_pre.w": _f32(sd["vision_model.pre_layrnorm.weight"]), "v.ln_pre.b": _f32(sd["vision_model.pre_layrnorm.bias"]),
        "v.ln_post.w": _f32(sd["vision_model.post_layernorm.weight"]), "v.ln_post.b": _f32(sd["vision_model.post_layernorm.bias"]),
        "v.proj": _f32(sd["visual_projection.weight"]),
        "t.tok": _f32(sd["text_model.embeddings.token_embedding.weight"]),
        "t.pos": _f32(sd["text_model.embeddings.position_embedding.weight"]),
        "t.ln_final.w": _f32(sd["text_model.final_layer_norm.weight"]), "t.ln_final.b": _f32(sd["text_model.final_layer_norm.bias"]),
        "t.proj": _f32(sd["text_projection.weight"]),
        "logit_scale": _f32(sd["logit_scale"]).reshape(()),
    }
    for pre, root in (("v", "vision_model"), ("t", "text_model")):
        for i in range(_count_layers(sd, re.escape(root) + r"\.encoder\.layers\.(\d+)\.self_attn\.q_proj\.weight$")):
            src = f"{root}.encoder.layers.{i}"
            w[f"{pre}.l{i}.qkv.w"] = torch.cat([_f32(sd[f"{src}.self_attn.{n}_proj.weight"]) for n in "qkv"], dim=0)
            w[f"{pre}.l{i}.qkv.b"] = torch.cat([_f32(sd[f"{src}.self_attn.{n}_proj.bias"]) for n in "qkv"], dim=0)
            for ours, _, hf in _BLOCK:
                w[f"{pre}.l{i}.{ours}"] = _f32(sd[f"{src}.{hf}"])
    return w


def to_hf_clip_state_dict(w: Weights) -> Weights:
    """Inverse of :func:`from_hf_clip_state_dict` (``CLIPModel.load_state_dict(..., strict=False)``:
    the ``position_ids`` buffers are not weights)."""
    sd: Weights = {
        "vision_model.embeddings.patch_embedding.weight": w["v.patch_w"],
        "vision_model.embeddings.class_embedding": w["v.cls"],
        "vision_model.embeddings.position_embedding.weight": w["v.pos"],
        "vision_model.pre_layrnorm.weight": w["v.ln_pre.w"], "vision_model.pre_layrnorm.bias": w["v.ln_pre.b"],
        "vision_model.post_layernorm.weight": w["v.ln_post.w"], "vision_model.post_layernorm.bias": w["v.ln_post.b"],
        "visual_projection.weight": w["v.proj"],
        "text_model.embeddings.token_embedding.weight": w["t.tok"],
        "text_model.embeddings.position_embedding.weight": w["t.pos"],
        "text_model.final_layer_norm.weight": w["t.ln_final.w"], "text_model.final_layer_norm.bias": w["t.ln_final.b"],
        "text_projection.weight": w["t.proj"], "logit_scale": w["logit_scale"],
    }
    for pre, root in (("v", "vision_model"), ("t", "text_model")):
        for i in range(_count_layers(w, re.escape(pre) + r"\.l(\d+)\.qkv\.w$")):
            dst = f"{root}.encoder.layers.{i}"
            d = w[f"{pre}.l{i}.qkv.w"].shape[1]
            for j, n in enumerate("qkv"):
                sd[f"{dst}.self_attn.{n}_proj.weight"] = w[f"{pre}.l{i}.qkv.w"][j * d:(j + 1) * d]
                sd[f"{dst}.self_attn.{n}_proj.bias"] = w[f"{pre}.l{i}.qkv.b"][j * d:(j + 1) * d]
            for ours, _, hf in _BLOCK:
                sd[f"{dst}.{hf}"] = w[f"{pre}.l{i}.{ours}"]
    return sd


# --------------------------------------------------------------------------------------------- BERT
_BERT_BLOCK = (
    ("out.w", "attention.output.dense.weight"), ("out.b", "attention.output.dense.bias"),
    ("ln1.w", "attention.output.LayerNorm.weight"), ("ln1.b", "attention.output.LayerNorm.bias"),
    ("fc1.w", "intermediate.dense.weight"), ("fc1.b", "intermediate.dense.bias"),
    ("fc2.w", "output.dense.weight"), ("fc2.b", "output.dense.bias"),
    ("ln2.w", "output.LayerNorm.weight"), ("ln2.b", "output.LayerNorm.bias"),
)


def from_hf_bert_state_dict(sd) -> Weights:
    """``BertForSequenceClassification.state_dict()`` (the Taiyi text tower, reference
    code/test_taiyi.py:12) -> names b.* of weights.make_bert_weights."""
    sd = _strip_prefix(dict(sd))
    if "bert.embeddings.word_embeddings.weight" not in sd or "classifier.weight" not in sd:
        raise KeyError("not a BertForSequenceClassification state dict")
    w: Weights = {
        "b.tok": _f32(sd["bert.embeddings.word_embeddings.weight"]),
        "b.pos": _f32(sd["bert.embeddings.position_embeddings.weight"]),
        "b.type": _f32(sd["bert.embeddings.token_type_embeddings.weight"]),
        "b.ln_emb.w": _f32(sd["bert.embeddings.LayerNorm.weight"]), "b.ln_emb.b": _f32(sd["bert.embeddings.LayerNorm.bias"]),
        "b.pool.w": _f32(sd["bert.pooler.dense.weight"]), "b.pool.b": _f32(sd["bert.pooler.dense.bias"]),
        "b.cls.w": _f32(sd["classifier.weight"]), "b.cls.b": _f32(sd["classifier.bias"]),
    }
    for i in range(_count_layers(sd, r"bert\.encoder\.layer\.(\d+)\.attention\.self\.query\.weight$")):
        src = f"bert.encoder.layer.{i}"
        w[f"b.l{i}.qkv.w"] = torch.cat([_f32(sd[f"{src}.attention.self.{n}.weight"]) for n in ("query", "key", "value")], dim=0)
        w[f"b.l{i}.qkv.b"] = torch.cat([_f32(sd[f"{src}.attention.self.{n}.bias"]) for n in ("query", "key", "value")], dim=0)
        for ours, hf in _BERT_BLOCK:
            w[f"b.l{i}.{ours}"] = _f32(sd[f"{src}.{hf}"])
    return w


def infer_bert_config(w: Weights, name: str = "checkpoint", ln_eps: float = 1e-12) -> BertTextConfig:
    d = int(w["b.tok"].shape[1])
    layers = _count_layers(w, r"b\.l(\d+)\.qkv\.w$")
    return BertTextConfig(name, d, layers, d // 64, int(w["b.l0.fc1.w"].shape[0]), int(w["b.pos"].shape[0]),
                          int(w["b.tok"].shape[0]), int(w["b.cls.w"].shape[0]), ln_eps=ln_eps)


# --------------------------------------------------------------------------------------------- files
def read_state_dict(path: str):
    """Read a checkpoint file WITHOUT executing anything from it: ``.safetensors`` through safetensors,
    everything else through ``torch.load(weights_only=True)``.  OpenAI's published ``ViT-*.pt`` files are
    TorchScript archives, which the safe loader refuses; re-save them once where the ``clip`` package is
    installed: ``torch.save(clip.load(name)[0].state_dict(), "ViT-B-32.state.pt")``."""
    if not os.path.isfile(path):
        raise FileNotFoundError(path)
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file

        return load_file(path, device="cpu")
    try:
        obj = torch.load(path, map_location="cpu", weights_only=True)
    except Exception as e:  # TorchScript archive / pickled module
        raise RuntimeError(f"{path}: not a plain tensor state dict ({type(e).__name__}: {e}). "
                           "Only state dicts are loaded (weights_only=True); see read_state_dict.__doc__.") from e
    if isinstance(obj, dict) and "state_dict" in obj and isinstance(obj["state_dict"], dict):
        obj = obj["state_dict"]
    if not isinstance(obj, dict):
        raise RuntimeError(f"{path}: expected a dict of tensors, got {type(obj).__name__}")
    return obj


def convert_state_dict(sd) -> Tuple[str, Weights]:
    """Detect the naming scheme and convert.  Returns (kind, weights), kind in {"clip", "bert"}."""
    keys = _strip_prefix(dict(sd)).keys()
    if "v.patch_w" in keys:
        return "clip", {k: _f32(v) for k, v in sd.items()}
    if "b.tok" in keys:
        return "bert", {k: _f32(v) for k, v in sd.items()}
    if "visual.conv1.weight" in keys:
        return "clip", from_openai_state_dict(sd)
    if "vision_model.embeddings.patch_embedding.weight" in keys:
        return "clip", from_hf_clip_state_dict(sd)
    if "bert.embeddings.word_embeddings.weight" in keys:
        return "bert", from_hf_bert_state_dict(sd)
    raise KeyError("unrecognised checkpoint: expected OpenAI CLIP (visual.conv1.weight), HF CLIPModel "
                   "(vision_model.*), HF BertForSequenceClassification (bert.*) or this package's own names")


def load_clip_checkpoint(path: str, name: str = None) -> Tuple[ClipConfig, Weights]:
    kind, w = convert_state_dict(read_state_dict(path))
    if kind != "clip":
        raise ValueError(f"{path} holds a {kind} model, not a CLIP")
    return infer_clip_config(w, name or os.path.basename(path)), w


def load_bert_checkpoint(path: str, name: str = None) -> Tuple[BertTextConfig, Weights]:
    kind, w = convert_state_dict(read_state_dict(path))
    if kind != "bert":
        raise ValueError(f"{path} holds a {kind} model, not a BERT text encoder")
    return infer_bert_config(w, name or os.path.basename(path)), w
