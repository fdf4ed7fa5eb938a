"""On-device CLIP preprocessing: bicubic resize of the shorter side, centre crop, /255, normalise.

The step immediately before ``encode_image``.  The reference gets it
from ``clip.load``'s ``preprocess`` -- torchvision ``Resize(n_px, BICUBIC)`` -> ``CenterCrop(n_px)`` ->
``ToTensor`` -> ``Normalize(mean, std)`` on a PIL image (use sites code/search_image.py:127,155;
constants code/custom.py:28).  The resize inside that is Pillow's ``Image.resize(..., BICUBIC)``:
8-bit fixed-point separable convolution, horizontal pass then vertical pass, each rounded back to
uint8 (Pillow src/libImaging/Resample.c: precompute_coeffs, normalize_coeffs_8bpc,
ImagingResampleHorizontal_8bpc / Vertical_8bpc).  That is integer/byte arithmetic, so the device
path reproduces it BIT-EXACTLY: the coefficient tables are built here on the host with the same
double-precision steps Pillow takes, and csrc/preprocess.hip applies them in int32.

Only the crop window is computed (same pixels as resize-then-crop, less work).
"""
import math
from functools import lru_cache
from typing import Sequence, Tuple

import numpy as np
import torch

from . import _lib

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
PRECISION_BITS = 32 - 8 - 2      # Pillow Resample.c


def _bicubic(x: float) -> float:
    a = -0.5                      # Pillow's bicubic_filter
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


@lru_cache(maxsize=256)
def resample_tables(in_size: int, out_size: int, first: int, count: int) -> Tuple[np.ndarray, np.ndarray, int]:
    """Pillow's precompute_coeffs + normalize_coeffs_8bpc for outputs [first, first+count) of a
    full-box resize in_size -> out_size.  Returns (bounds int32[count,2] = (xmin, taps),
    coeffs int32[count, ksize], ksize); ksize is Pillow's, rounded up to a multiple of 4 (zero taps)."""
    scale = float(np.float32(in_size) - np.float32(0.0)) / out_size      # (double)(in1 - in0) / outSize, box is float
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale                                          # bicubic support = 2
    ksize = int(math.ceil(support)) * 2 + 1
    ksize = (ksize + 3) // 4 * 4          # rows padded with zero taps to whole groups of 4: the kernels read 4 taps per load
    bounds = np.zeros((count, 2), np.int32)
    coeffs = np.zeros((count, ksize), np.int32)
    ss = 1.0 / filterscale
    for j in range(count):
        xx = first + j
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for w in k:
            ww += w
        if ww != 0.0:
            k = [w / ww for w in k]
        for x, w in enumerate(k):
            coeffs[j, x] = int(-0.5 + w * (1 << PRECISION_BITS)) if w < 0 else int(0.5 + w * (1 << PRECISION_BITS))
        bounds[j] = (xmin, xmax)
    # the kernels multiply with the 24-bit integer multiplier: |coefficient| <= 2^22 * |weight|, |weight| <= ~1.2 for bicubic
    assert int(np.abs(coeffs).max(initial=0)) < (1 << 23), "resample coefficient outside the 24-bit multiplier's range"
    return bounds, coeffs, ksize


def resize_geometry(h: int, w: int, n_px: int):
    """torchvision Resize(int) + CenterCrop(int) geometry: shorter side -> n_px, the other
    int(n_px * long / short); crop offsets int(round((size - n_px) / 2.0)) (Python round)."""
    if w <= h:
        nw, nh = n_px, int(n_px * h / w)
    else:
        nh, nw = n_px, int(n_px * w / h)
    top = int(round((nh - n_px) / 2.0))
    left = int(round((nw - n_px) / 2.0))
    return nh, nw, top, left


_table_cache = {}


def _device_tables(h, w, n_px, device):
    key = (h, w, n_px, str(device))
    hit = _table_cache.get(key)
    if hit is not None:
        return hit
    nh, nw, top, left = resize_geometry(h, w, n_px)
    hb, hc, hk = resample_tables(w, nw, left, n_px)       # horizontal: columns of the crop window
    vb, vc, vk = resample_tables(h, nh, top, n_px)        # vertical: rows of the crop window
    row0 = int(vb[:, 0].min())
    row1 = int((vb[:, 0] + vb[:, 1]).max())               # input rows the vertical pass touches
    t = dict(hb=torch.from_numpy(hb).to(device), hc=torch.from_numpy(hc).to(device), hk=hk,
             vb=torch.from_numpy(vb).to(device), vc=torch.from_numpy(vc).to(device), vk=vk, row0=row0, row1=row1)
    if len(_table_cache) > 512:
        _table_cache.clear()
    _table_cache[key] = t
    return t


@torch.no_grad()
def preprocess_image(img_u8: torch.Tensor, n_px: int = 224, out: torch.Tensor = None,
                     out_dtype: torch.dtype = torch.float32, mean: Sequence[float] = CLIP_MEAN,
                     std: Sequence[float] = CLIP_STD, return_u8: bool = False):
    """uint8 RGB image [H,W,3] on the GPU -> normalised [3,n_px,n_px] (fp32 or bf16).

    Equals ``Normalize(ToTensor(CenterCrop(Resize(PIL image))))`` bit for bit in the uint8 stage and in
    fp32 arithmetic after it.  ``return_u8`` also returns the resized+cropped uint8 [n_px,n_px,3].
    """
    if img_u8.dtype != torch.uint8 or img_u8.dim() != 3 or img_u8.shape[2] != 3:
        raise ValueError(f"expected a uint8 [H,W,3] tensor, got {img_u8.dtype} {tuple(img_u8.shape)}")
    if not img_u8.is_cuda:
        raise RuntimeError("image must live on the GPU (there is no CPU path); upload the decoded bytes first")
    img_u8 = img_u8.contiguous()
    h, w = int(img_u8.shape[0]), int(img_u8.shape[1])
    dev = img_u8.device
    t = _device_tables(h, w, n_px, dev)
    rows = t["row1"] - t["row0"]
    tmp = torch.empty(rows, n_px, 3, dtype=torch.uint8, device=dev)
    if out is None:
        out = torch.empty(3, n_px, n_px, dtype=out_dtype, device=dev)
    u8 = torch.empty(n_px, n_px, 3, dtype=torch.uint8, device=dev) if return_u8 else None
    L = _lib.lib()
    _lib.check(L.mmr_preprocess_image(img_u8.data_ptr(), h, w, n_px, t["row0"], rows,
                                      t["hb"].data_ptr(), t["hc"].data_ptr(), t["hk"],
                                      t["vb"].data_ptr(), t["vc"].data_ptr(), t["vk"],
                                      float(mean[0]), float(mean[1]), float(mean[2]),
                                      float(std[0]), float(std[1]), float(std[2]),
                                      tmp.data_ptr(), out.data_ptr(), _lib.dtype_code(out.dtype), _lib.ptr(u8),
                                      _lib.stream_ptr(dev)))
    return (out, u8) if return_u8 else out


def preprocess_batch(images: Sequence[torch.Tensor], n_px: int = 224, out_dtype: torch.dtype = torch.float32,
                     mean: Sequence[float] = CLIP_MEAN, std: Sequence[float] = CLIP_STD) -> torch.Tensor:
    """A list of uint8 [H_i,W_i,3] GPU tensors (any sizes) -> [B,3,n_px,n_px] in ONE launch pair
    (mmr_preprocess_batch): per-image descriptors (pointers, geometry, coefficient tables) are built on
    the host and uploaded as one small array."""
    import numpy as np

    if not images:
        raise ValueError("empty batch")
    dev = images[0].device
    imgs, tabs, rows = [], [], []
    for im in images:
        if im.dtype != torch.uint8 or im.dim() != 3 or im.shape[2] != 3:
            raise ValueError(f"expected uint8 [H,W,3] tensors, got {im.dtype} {tuple(im.shape)}")
        if not im.is_cuda:
            raise RuntimeError("images must live on the GPU (there is no CPU path)")
        im = im.contiguous()
        t = _device_tables(int(im.shape[0]), int(im.shape[1]), n_px, dev)
        imgs.append(im)
        tabs.append(t)
        rows.append(t["row1"] - t["row0"])
    B = len(imgs)
    offs = np.concatenate([[0], np.cumsum([r * n_px * 3 for r in rows])]).astype(np.int64)
    tmp = torch.empty(int(offs[-1]) + 16, dtype=torch.uint8, device=dev)
    desc = np.zeros((B, 9), dtype=np.int64)              # 6 pointers + 6 int32 packed as 3 int64 = 72 bytes
    for i, (im, t) in enumerate(zip(imgs, tabs)):
        desc[i, 0] = im.data_ptr()
        desc[i, 1], desc[i, 2] = t["hb"].data_ptr(), t["hc"].data_ptr()
        desc[i, 3], desc[i, 4] = t["vb"].data_ptr(), t["vc"].data_ptr()
        desc[i, 5] = tmp.data_ptr() + int(offs[i])
        ints = np.array([im.shape[0], im.shape[1], t["row0"], rows[i], t["hk"], t["vk"]], dtype=np.int32)
        desc[i, 6:9] = ints.view(np.int64)
    desc_d = torch.from_numpy(desc).to(dev)
    out = torch.empty(B, 3, n_px, n_px, dtype=out_dtype, device=dev)
    L = _lib.lib()
    _lib.check(L.mmr_preprocess_batch_ex(desc_d.data_ptr(), B, n_px, int(max(rows)), int(max(im.shape[1] for im in imgs)),
                                         float(mean[0]), float(mean[1]), float(mean[2]), float(std[0]), float(std[1]),
                                         float(std[2]), out.data_ptr(), _lib.dtype_code(out_dtype), _lib.stream_ptr(dev)))
    # descriptors, tables and images must outlive the launch: tie them to the output's lifetime
    out._mmr_keepalive = (desc_d, tmp, imgs)
    return out


class UniformBatchPreprocessor:
    """``preprocess`` for batches of SAME-SIZE images held as one uint8 tensor [B,H,W,3] on the GPU (a decoded video
    frame stack, a dataset stored at one resolution): the per-image descriptors differ only in two pointers, so they are
    built with three numpy array ops instead of a Python loop, staged in PINNED host memory and uploaded asynchronously --
    the call returns at once and can run on a side stream under the previous batch's ``encode_image``
    (``gallery.build_gallery_overlapped``).  Same kernels (``mmr_preprocess_batch``), same bytes as
    ``preprocess_image`` / Pillow.  One instance owns ``slots`` sets of buffers (descriptors, vertical-pass scratch,
    output pixels): slot ``i`` may be reused once the consumer of its previous output has finished."""

    def __init__(self, batch: int, height: int, width: int, n_px: int = 224, out_dtype: torch.dtype = torch.bfloat16,
                 device="cuda", slots: int = 2, mean: Sequence[float] = CLIP_MEAN, std: Sequence[float] = CLIP_STD):
        self.B, self.H, self.W, self.S = int(batch), int(height), int(width), int(n_px)
        self.device = torch.device(device)
        self.mean, self.std, self.out_dtype = tuple(mean), tuple(std), out_dtype
        self.t = _device_tables(self.H, self.W, self.S, self.device)
        self.rows = self.t["row1"] - self.t["row0"]
        self.tmp_stride = self.rows * self.S * 3
        self.slots = []
        for _ in range(slots):
            self.slots.append(dict(
                desc_h=torch.zeros(self.B, 9, dtype=torch.int64).pin_memory(),
                desc_d=torch.zeros(self.B, 9, dtype=torch.int64, device=self.device),
                tmp=torch.empty(self.B * self.tmp_stride + 16, dtype=torch.uint8, device=self.device),
                out=torch.empty(self.B, 3, self.S, self.S, dtype=out_dtype, device=self.device)))
        ints = np.array([self.H, self.W, self.t["row0"], self.rows, self.t["hk"], self.t["vk"]], dtype=np.int32).view(np.int64)
        for sl in self.slots:                                  # everything but the image pointers is fixed per slot
            d = sl["desc_h"].numpy()
            d[:, 1], d[:, 2] = self.t["hb"].data_ptr(), self.t["hc"].data_ptr()
            d[:, 3], d[:, 4] = self.t["vb"].data_ptr(), self.t["vc"].data_ptr()
            d[:, 5] = sl["tmp"].data_ptr() + np.arange(self.B, dtype=np.int64) * self.tmp_stride
            d[:, 6:9] = ints

    @torch.no_grad()
    def __call__(self, images_u8: torch.Tensor, slot: int = 0) -> torch.Tensor:
        """[b,H,W,3] uint8 (b <= batch) -> the slot's pixel buffer [b,3,S,S]; enqueued on the CURRENT stream."""
        if images_u8.dtype != torch.uint8 or images_u8.dim() != 4 or tuple(images_u8.shape[1:]) != (self.H, self.W, 3):
            raise ValueError(f"expected uint8 [b,{self.H},{self.W},3], got {images_u8.dtype} {tuple(images_u8.shape)}")
        if not images_u8.is_cuda:
            raise RuntimeError("images must live on the GPU (there is no CPU path)")
        b = int(images_u8.shape[0])
        if not 1 <= b <= self.B:
            raise ValueError(f"batch {b} outside [1,{self.B}]")
        images_u8 = images_u8.contiguous()
        sl = self.slots[slot]
        key = (images_u8.data_ptr(), b)
        if sl.get("key") != key:               # same buffer as last time (a staging ring): the device descriptors still hold
            ev = sl.get("copied")
            if ev is not None:
                ev.synchronize()               # the previous upload FROM this pinned block must have left before it is rewritten
            sl["desc_h"].numpy()[:b, 0] = images_u8.data_ptr() + np.arange(b, dtype=np.int64) * (self.H * self.W * 3)
            sl["desc_d"][:b].copy_(sl["desc_h"][:b], non_blocking=True)
            sl["copied"] = torch.cuda.Event()
            sl["copied"].record()
            sl["key"] = key
        out = sl["out"][:b]
        L = _lib.lib()
        _lib.check(L.mmr_preprocess_batch_ex(sl["desc_d"].data_ptr(), b, self.S, int(self.rows), self.W, float(self.mean[0]),
                                             float(self.mean[1]), float(self.mean[2]), float(self.std[0]), float(self.std[1]),
                                             float(self.std[2]), out.data_ptr(), _lib.dtype_code(self.out_dtype),
                                             _lib.stream_ptr(self.device)))
        return out
