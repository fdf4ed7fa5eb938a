"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the CLIP encode path.

A plain-torch fp32 restatement of the encoder arithmetic the reference obtains from its
third-party model packages (``clip`` / ``transformers``; the arithmetic is not in the
reference tree -- SURVEY.md section 0.2).  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this module; the product package never does.

Pinned: ``oracle/make_golden.py`` checks every function here against
``transformers==5.15.0`` ``CLIPModel`` (the backend the reference calls in
code/test_taiyi.py:17-30 and CLIP-Chinese/lab_chinese.py:83-114) on the same weights and
commits that model's outputs as ``tests/golden/*.npz``; ``tests/test_oracle.py`` re-checks
this file against those fixtures.  The reference's only known-answer value
(code/test_clip.py:17) needs pretrained weights + CLIP.png, neither available offline.

Source lines followed (transformers/models/clip/modeling_clip.py @5.15.0):
  vision embeddings :138-218   attention :259-335   MLP + block :338-383
  vision tower :594-656        text tower + EOT pooling :494-586
  projections :660-753         normalise + logits :806-817
  QuickGELU: transformers/activations.py:117-123
"""
from typing import Dict

import torch
import torch.nn.functional as F


def quick_gelu(x: torch.Tensor) -> torch.Tensor:
    # activations.py:117-123
    return x * torch.sigmoid(1.702 * x)


def _block(h, w: Dict[str, torch.Tensor], p: str, heads: int, mask, eps: float):
    """One pre-LN transformer block (modeling_clip.py:362-383)."""
    B, T, d = h.shape
    dh = d // heads
    r = h
    x = F.layer_norm(h, (d,), w[f"{p}.ln1.w"], w[f"{p}.ln1.b"], eps)
    qkv = x @ w[f"{p}.qkv.w"].t() + w[f"{p}.qkv.b"]
    q, k, v = qkv.split(d, dim=-1)
    q = q.view(B, T, heads, dh).transpose(1, 2)
    k = k.view(B, T, heads, dh).transpose(1, 2)
    v = v.view(B, T, heads, dh).transpose(1, 2)
    # eager_attention_forward, modeling_clip.py:259-277: scale after QK^T, fp32 softmax.
    att = (q @ k.transpose(-1, -2)) * dh ** -0.5
    if mask is not None:
        att = att + mask
    att = torch.softmax(att, dim=-1, dtype=torch.float32).to(q.dtype)
    o = (att @ v).transpose(1, 2).reshape(B, T, d)
    h = r + (o @ w[f"{p}.out.w"].t() + w[f"{p}.out.b"])
    r = h
    x = F.layer_norm(h, (d,), w[f"{p}.ln2.w"], w[f"{p}.ln2.b"], eps)
    x = quick_gelu(x @ w[f"{p}.fc1.w"].t() + w[f"{p}.fc1.b"])
    x = x @ w[f"{p}.fc2.w"].t() + w[f"{p}.fc2.b"]
    return r + x


def encode_image(w: Dict[str, torch.Tensor], cfg, pixels: torch.Tensor, stages: dict = None):
    """Vision tower forward; ``cfg`` is a TowerConfig-like object (kind == "vision").

    pixels [B,3,H,W] fp32 -> features [B,E] fp32 (un-normalised, like ``encode_image``).
    ``stages`` (optional dict) receives intermediate activations for stage-level parity.
    """
    B, C, H, W = pixels.shape
    if H != cfg.image_size or W != cfg.image_size:
        # modeling_clip.py:204-207
        raise ValueError(f"Input image size ({H}*{W}) doesn't match model ({cfg.image_size}*{cfg.image_size}).")
    d, P = cfg.width, cfg.patch
    x = pixels.to(torch.float32)
    patch = F.conv2d(x, w["v.patch_w"], bias=None, stride=P)          # :209
    patch = patch.flatten(2).transpose(1, 2)                           # :210 [B,G*G,d]
    cls = w["v.cls"].expand(B, 1, d)                                   # :212
    h = torch.cat([cls, patch], dim=1) + w["v.pos"]                    # :213-217
    if stages is not None:
        stages["embed"] = h.clone()
    h = F.layer_norm(h, (d,), w["v.ln_pre.w"], w["v.ln_pre.b"], cfg.ln_eps)   # :642 pre_layrnorm
    if stages is not None:
        stages["ln_pre"] = h.clone()
    for i in range(cfg.layers):
        h = _block(h, w, f"v.l{i}", cfg.heads, None, cfg.ln_eps)
        if stages is not None and (i == 0 or i == cfg.layers - 1):
            stages[f"layer{i}"] = h.clone()
    pooled = F.layer_norm(h[:, 0, :], (d,), w["v.ln_post.w"], w["v.ln_post.b"], cfg.ln_eps)  # :650-651
    if stages is not None:
        stages["pooled"] = pooled.clone()
    return pooled @ w["v.proj"].t()                                    # :674,751 (no bias)


def encode_text(w: Dict[str, torch.Tensor], cfg, ids: torch.Tensor, stages: dict = None):
    """Text tower forward; ids [N,T] integer -> features [N,E] fp32.

    Pooled row = position of the EOT token = ``argmax(ids)`` (OpenAI ``text.argmax(-1)``,
    equal to HF's "first eos_token_id" rule when EOT is the largest id,
    modeling_clip.py:561-581).
    """
    N, T = ids.shape
    d = cfg.width
    ids = ids.long()
    h = w["t.tok"][ids] + w["t.pos"][:T]                               # :251-254
    mask = torch.full((T, T), float("-inf")).triu(1)                   # causal, :543-556
    for i in range(cfg.layers):
        h = _block(h, w, f"t.l{i}", cfg.heads, mask, cfg.ln_eps)
        if stages is not None and (i == 0 or i == cfg.layers - 1):
            stages[f"layer{i}"] = h.clone()
    h = F.layer_norm(h, (d,), w["t.ln_final.w"], w["t.ln_final.b"], cfg.ln_eps)  # :559
    pooled = h[torch.arange(N), ids.argmax(dim=-1)]                    # :561-581
    if stages is not None:
        stages["pooled"] = pooled.clone()
    return pooled @ w["t.proj"].t()                                    # :675,713


def l2_normalize(x: torch.Tensor) -> torch.Tensor:
    """``f /= f.norm(dim=-1, keepdim=True)`` -- reference code/search_image.py:133,157; no epsilon."""
    return x / x.norm(dim=-1, keepdim=True)


def clip_forward(w, ccfg, pixels, ids):
    """``model(image, text)`` -> (logits_per_image, logits_per_text), modeling_clip.py:806-817."""
    i = l2_normalize(encode_image(w, ccfg.vision, pixels))
    t = l2_normalize(encode_text(w, ccfg.text, ids))
    logits_per_text = t @ i.t() * w["logit_scale"].exp()
    return logits_per_text.t(), logits_per_text
