"""Deterministic synthetic CLIP weights (no checkpoint is reachable offline).

The reference obtains weights by download inside ``clip.load`` / ``from_pretrained``
(reference code/search_image.py:327, code/test_taiyi.py:17).  There is no network here or
on the GPU box, so weights are generated from a per-tensor seeded CPU generator with the
per-tensor standard deviations of the HF CLIP initialiser
(transformers/models/clip/modeling_clip.py:404-437).  The same call reproduces the same
tensors in this container and on the GPU box (same torch build), which is what lets the
committed golden vectors pin the encoder.

Every tensor is rounded to a bf16-representable fp32 value, so the fp32 oracle and the
bf16 device path consume *identical* weight values.

Naming (this repo's own flat names; ``oracle/hf_adapter.py`` maps them to HF names):
  vision:  v.patch_w[d,3,P,P] v.cls[d] v.pos[T,d] v.ln_pre.{w,b}
           v.l{i}.ln1.{w,b} v.l{i}.qkv.{w[3d,d],b[3d]} v.l{i}.out.{w[d,d],b}
           v.l{i}.ln2.{w,b} v.l{i}.fc1.{w[m,d],b} v.l{i}.fc2.{w[d,m],b}
           v.ln_post.{w,b} v.proj[E,d]
  text:    t.tok[V,d] t.pos[T,d] t.l{i}.* (same) t.ln_final.{w,b} t.proj[E,d]
  joint:   logit_scale (0-d)
"""
import zlib
from typing import Dict

import torch

from .config import ClipConfig, TowerConfig


def _bf16_round(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.bfloat16).to(torch.float32)


def _randn(name: str, shape, std: float, seed: int, mean: float = 0.0) -> torch.Tensor:
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    x = torch.randn(*shape, generator=g, dtype=torch.float32) * std + mean
    return _bf16_round(x)


def _tower_layers(prefix: str, cfg: TowerConfig, seed: int, out: Dict[str, torch.Tensor]):
    d, m, L = cfg.width, cfg.mlp, cfg.layers
    in_proj_std = d ** -0.5 * (2 * L) ** -0.5
    out_proj_std = d ** -0.5
    fc_std = (2 * d) ** -0.5
    for i in range(L):
        p = f"{prefix}.l{i}"
        out[f"{p}.ln1.w"] = _randn(f"{p}.ln1.w", (d,), 0.1, seed, mean=1.0)
        out[f"{p}.ln1.b"] = _randn(f"{p}.ln1.b", (d,), 0.05, seed)
        out[f"{p}.qkv.w"] = _randn(f"{p}.qkv.w", (3 * d, d), in_proj_std, seed)
        out[f"{p}.qkv.b"] = _randn(f"{p}.qkv.b", (3 * d,), 0.02, seed)
        out[f"{p}.out.w"] = _randn(f"{p}.out.w", (d, d), out_proj_std, seed)
        out[f"{p}.out.b"] = _randn(f"{p}.out.b", (d,), 0.02, seed)
        out[f"{p}.ln2.w"] = _randn(f"{p}.ln2.w", (d,), 0.1, seed, mean=1.0)
        out[f"{p}.ln2.b"] = _randn(f"{p}.ln2.b", (d,), 0.05, seed)
        out[f"{p}.fc1.w"] = _randn(f"{p}.fc1.w", (m, d), fc_std, seed)
        out[f"{p}.fc1.b"] = _randn(f"{p}.fc1.b", (m,), 0.02, seed)
        out[f"{p}.fc2.w"] = _randn(f"{p}.fc2.w", (d, m), in_proj_std, seed)
        out[f"{p}.fc2.b"] = _randn(f"{p}.fc2.b", (d,), 0.02, seed)


def make_vision_weights(cfg: TowerConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    d, P, T, E = cfg.width, cfg.patch, cfg.tokens, cfg.embed_dim
    w: Dict[str, torch.Tensor] = {}
    w["v.patch_w"] = _randn("v.patch_w", (d, 3, P, P), 0.02, seed)
    w["v.cls"] = _randn("v.cls", (d,), d ** -0.5, seed)
    w["v.pos"] = _randn("v.pos", (T, d), 0.02, seed)
    w["v.ln_pre.w"] = _randn("v.ln_pre.w", (d,), 0.1, seed, mean=1.0)
    w["v.ln_pre.b"] = _randn("v.ln_pre.b", (d,), 0.05, seed)
    _tower_layers("v", cfg, seed, w)
    w["v.ln_post.w"] = _randn("v.ln_post.w", (d,), 0.1, seed, mean=1.0)
    w["v.ln_post.b"] = _randn("v.ln_post.b", (d,), 0.05, seed)
    w["v.proj"] = _randn("v.proj", (E, d), d ** -0.5, seed)
    return w


def make_text_weights(cfg: TowerConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    d, T, E, V = cfg.width, cfg.tokens, cfg.embed_dim, cfg.vocab
    w: Dict[str, torch.Tensor] = {}
    w["t.tok"] = _randn("t.tok", (V, d), 0.02, seed)
    w["t.pos"] = _randn("t.pos", (T, d), 0.02, seed)
    _tower_layers("t", cfg, seed, w)
    w["t.ln_final.w"] = _randn("t.ln_final.w", (d,), 0.1, seed, mean=1.0)
    w["t.ln_final.b"] = _randn("t.ln_final.b", (d,), 0.05, seed)
    w["t.proj"] = _randn("t.proj", (E, d), d ** -0.5, seed)
    return w


def make_clip_weights(cfg: ClipConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    w = make_vision_weights(cfg.vision, seed)
    w.update(make_text_weights(cfg.text, seed))
    w["logit_scale"] = torch.tensor(cfg.logit_scale_init, dtype=torch.float32)
    return w


def make_bert_weights(cfg, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Seeded weights for the BERT-style text classifier (names b.*; std 0.02 like BertPreTrainedModel
    _init_weights, with perturbed LayerNorm / bias tensors so they are exercised).

      b.tok[V,d] b.pos[P,d] b.type[2,d] b.ln_emb.{w,b}
      b.l{i}.qkv.{w[3d,d],b} b.l{i}.out.{w,b} b.l{i}.ln1.{w,b}   (ln1 = attention-output LayerNorm)
      b.l{i}.fc1.{w[m,d],b} b.l{i}.fc2.{w[d,m],b} b.l{i}.ln2.{w,b} (ln2 = output LayerNorm)
      b.pool.{w[d,d],b} b.cls.{w[E,d],b}
    """
    d, m, E = cfg.width, cfg.mlp, cfg.embed_dim
    w: Dict[str, torch.Tensor] = {}
    w["b.tok"] = _randn("b.tok", (cfg.vocab, d), 0.02, seed)
    w["b.pos"] = _randn("b.pos", (cfg.max_positions, d), 0.02, seed)
    w["b.type"] = _randn("b.type", (2, d), 0.02, seed)
    w["b.ln_emb.w"] = _randn("b.ln_emb.w", (d,), 0.1, seed, mean=1.0)
    w["b.ln_emb.b"] = _randn("b.ln_emb.b", (d,), 0.05, seed)
    for i in range(cfg.layers):
        p = f"b.l{i}"
        w[f"{p}.qkv.w"] = _randn(f"{p}.qkv.w", (3 * d, d), 0.02, seed)
        w[f"{p}.qkv.b"] = _randn(f"{p}.qkv.b", (3 * d,), 0.02, seed)
        w[f"{p}.out.w"] = _randn(f"{p}.out.w", (d, d), 0.02, seed)
        w[f"{p}.out.b"] = _randn(f"{p}.out.b", (d,), 0.02, seed)
        w[f"{p}.ln1.w"] = _randn(f"{p}.ln1.w", (d,), 0.1, seed, mean=1.0)
        w[f"{p}.ln1.b"] = _randn(f"{p}.ln1.b", (d,), 0.05, seed)
        w[f"{p}.fc1.w"] = _randn(f"{p}.fc1.w", (m, d), 0.02, seed)
        w[f"{p}.fc1.b"] = _randn(f"{p}.fc1.b", (m,), 0.02, seed)
        w[f"{p}.fc2.w"] = _randn(f"{p}.fc2.w", (d, m), 0.02, seed)
        w[f"{p}.fc2.b"] = _randn(f"{p}.fc2.b", (d,), 0.02, seed)
        w[f"{p}.ln2.w"] = _randn(f"{p}.ln2.w", (d,), 0.1, seed, mean=1.0)
        w[f"{p}.ln2.b"] = _randn(f"{p}.ln2.b", (d,), 0.05, seed)
    w["b.pool.w"] = _randn("b.pool.w", (d, d), 0.02, seed)
    w["b.pool.b"] = _randn("b.pool.b", (d,), 0.02, seed)
    w["b.cls.w"] = _randn("b.cls.w", (E, d), 0.02, seed)
    w["b.cls.b"] = _randn("b.cls.b", (E,), 0.02, seed)
    return w
