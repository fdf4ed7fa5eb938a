"""Seeded synthetic inputs of the shapes BASELINE.json names (no datasets are reachable offline).

All generators run on the CPU generator so that the dev container (golden fixtures), the
CPU oracle and the GPU box see bit-identical inputs; callers move the result to the device.
"""
import torch


def synth_images(batch: int, image_size: int, seed: int = 0) -> torch.Tensor:
    """[B,3,S,S] fp32 ~ N(0,1): stands in for ``preprocess(Image)`` output (already normalised)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randn(batch, 3, image_size, image_size, generator=g, dtype=torch.float32)


def synth_token_ids(n: int, context: int = 77, vocab: int = 49408, seed: int = 5) -> torch.Tensor:
    """[N,ctx] int32 shaped like ``clip.tokenize`` output: SOT, random ids, EOT, zero pad.

    SOT = vocab-2, EOT = vocab-1 (49406 / 49407 for the real vocabulary); the EOT position
    is uniform in [2, ctx-1]; ids after EOT are 0 (reference code/search_image.py:334).
    """
    g = torch.Generator(device="cpu").manual_seed(seed)
    ids = torch.zeros(n, context, dtype=torch.int32)
    eot_pos = torch.randint(2, context, (n,), generator=g)
    body = torch.randint(1, vocab - 2, (n, context), generator=g, dtype=torch.int32)
    for i in range(n):
        p = int(eot_pos[i])
        ids[i, 0] = vocab - 2
        ids[i, 1:p] = body[i, 1:p]
        ids[i, p] = vocab - 1
    return ids


def synth_unit_rows(n: int, dim: int, seed: int, dtype=torch.float32, chunk: int = 131072) -> torch.Tensor:
    """[n,dim] L2-normalised N(0,1) rows (gallery / query stand-ins), generated in chunks."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    out = torch.empty(n, dim, dtype=dtype)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        x = torch.randn(e - s, dim, generator=g, dtype=torch.float32)
        x = x / x.norm(dim=-1, keepdim=True)
        out[s:e] = x.to(dtype)
    return out
