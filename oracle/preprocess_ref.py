"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the preprocessing step.

Restates what the reference's ``preprocess`` does to one PIL image (reference code/search_image.py:127,155;
constants code/custom.py:28): torchvision ``Resize(n_px, BICUBIC)`` -> ``CenterCrop(n_px)`` -> ``ToTensor`` ->
``Normalize``.  The resize IS Pillow's ``Image.resize`` (Pillow 12.2.0 is importable here and is the
third-party code the reference itself calls), so that stage is pinned on the real implementation; the
geometry / crop / normalise steps restate torchvision (absent offline: those three lines are unpinned
by a runnable reference and follow torchvision's documented semantics).
Only tests/ may import this.
"""
import numpy as np
import torch
from PIL import Image

MEAN = np.array([0.48145466, 0.4578275, 0.40821073], np.float32)
STD = np.array([0.26862954, 0.26130258, 0.27577711], np.float32)


def geometry(h, w, n_px):
    if w <= h:
        nw, nh = n_px, int(n_px * h / w)
    else:
        nh, nw = n_px, int(n_px * w / h)
    return nh, nw, int(round((nh - n_px) / 2.0)), int(round((nw - n_px) / 2.0))


def preprocess_u8(img: np.ndarray, n_px: int) -> np.ndarray:
    """uint8 [H,W,3] -> resized + centre-cropped uint8 [n_px,n_px,3] via Pillow."""
    h, w = img.shape[:2]
    nh, nw, top, left = geometry(h, w, n_px)
    r = Image.fromarray(img, "RGB").resize((nw, nh), Image.BICUBIC)
    return np.asarray(r)[top:top + n_px, left:left + n_px].copy()


def preprocess(img: np.ndarray, n_px: int) -> torch.Tensor:
    """-> fp32 [3,n_px,n_px]: ToTensor (x/255 in fp32) then Normalize ((x-mean)/std in fp32)."""
    u8 = preprocess_u8(img, n_px)
    x = torch.from_numpy(u8).permute(2, 0, 1).to(torch.float32).div(255)
    mean = torch.from_numpy(MEAN).view(3, 1, 1)
    std = torch.from_numpy(STD).view(3, 1, 1)
    return x.sub(mean).div(std)
