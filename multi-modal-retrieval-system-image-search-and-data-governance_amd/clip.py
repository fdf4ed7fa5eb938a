"""``clip``-package call surface over the HIP towers.

Mirrors what the reference scripts do with the third-party ``clip`` / ``transformers`` objects:
    model, preprocess = clip.load("ViT-B/32", device=device)       reference code/test_clip.py:6
    text = clip.tokenize([...]).to(device)                         reference code/test_clip.py:9
    model.encode_image(image) / model.encode_text(text)            reference code/test_clip.py:12-13
    logits_per_image, logits_per_text = model(image, text)         reference code/test_clip.py:15
    model.get_image_features(pixel_values=...) ; model.logit_scale reference code/test_taiyi.py:23,29
    model.dtype / .eval() / .cuda() / .float()                     reference code/main_custom.py:151
All arithmetic is in csrc/*.hip behind include/mmr.h; this file only owns tensors and handles.
No autograd: every reference call site wraps the encoder in ``torch.no_grad()``.
"""
import contextlib
import ctypes
import os
from typing import Dict, List, Optional, Union

import torch

from . import _lib
from .config import ClipConfig, TowerConfig, available_models, get_config  # noqa: F401
from .weights import make_clip_weights

_CONTEXT = 77
# preprocessing constants the reference carries in code/custom.py:28
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _fold_ln_default() -> bool:
    """LayerNorm folding (include/mmr.h, mmr_tower_cfg.fold_ln) is opt-in: ``fold_ln=True`` or MMR_FOLD_LN=1.
    Measured on MI355X it is within +-1 % of the separate-LayerNorm path on every tower, and it
    trades the bf16 rounding of LN(h) for a bf16 rounding of h itself, which is only as accurate when the
    per-token mean of the residual stream is small against its spread -- so the separate-LayerNorm path stays
    the default."""
    return os.environ.get("MMR_FOLD_LN", "0") == "1"


def _tower_cfg_struct(t: TowerConfig, fold_ln: bool = False) -> _lib.TowerCfg:
    return _lib.TowerCfg(kind=0 if t.kind == "vision" else 1, width=t.width, layers=t.layers, heads=t.heads,
                         mlp=t.mlp, tokens=t.tokens, embed_dim=t.embed_dim, image_size=t.image_size,
                         patch=t.patch, vocab=t.vocab, ln_eps=t.ln_eps, fold_ln=int(bool(fold_ln)))


def fold_layernorm(W: torch.Tensor, b: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor):
    """LN(h) @ W.T + b  ==  rstd * (h @ W'.T - mean * c) + b'   with
    W' = bf16(W * gamma), c = rowsum(W') (of the ROUNDED values, so the mean term cancels exactly against
    what the matrix cores accumulate) and b' = W @ beta + b.  Returns (W' as bf16, b', c), fp64 sums."""
    Wd, g, be = W.double(), gamma.double(), beta.double()
    Wf = (Wd * g[None, :]).to(torch.float32).to(torch.bfloat16)
    c = Wf.double().sum(dim=1).to(torch.float32)
    bf = (Wd @ be + b.double()).to(torch.float32)
    return Wf, bf, c


class _Tower:
    """One device-resident tower: weight blob + C handle + reusable workspace."""

    _LAYER_PARAMS = (("ln1.w", _lib.P_LN1_W, False), ("ln1.b", _lib.P_LN1_B, False),
                     ("qkv.w", _lib.P_QKV_W, True), ("qkv.b", _lib.P_QKV_B, False),
                     ("out.w", _lib.P_OUT_W, True), ("out.b", _lib.P_OUT_B, False),
                     ("ln2.w", _lib.P_LN2_W, False), ("ln2.b", _lib.P_LN2_B, False),
                     ("fc1.w", _lib.P_FC1_W, True), ("fc1.b", _lib.P_FC1_B, False),
                     ("fc2.w", _lib.P_FC2_W, True), ("fc2.b", _lib.P_FC2_B, False))

    def __init__(self, cfg: TowerConfig, w: Dict[str, torch.Tensor], device: torch.device, fold_ln: Optional[bool] = None):
        self.cfg, self.device = cfg, device
        self.L = _lib.lib()
        self.fold_ln = _fold_ln_default() if fold_ln is None else bool(fold_ln)
        self.c = _tower_cfg_struct(cfg, self.fold_ln)
        total = self.L.mmr_tower_weights_bytes(ctypes.byref(self.c))
        if total == 0:
            raise _lib.MMRError(-22, f"unsupported tower geometry {cfg}")
        blob = torch.zeros(total, dtype=torch.uint8)           # host staging, then one H2D copy
        pre = "v" if cfg.kind == "vision" else "t"

        def put(param, layer, tensor, as_bf16):
            off, nbytes = ctypes.c_size_t(), ctypes.c_size_t()
            _lib.check(self.L.mmr_tower_param_span(ctypes.byref(self.c), param, layer, ctypes.byref(off),
                                                   ctypes.byref(nbytes)))
            t = tensor.detach().to("cpu").contiguous()
            t = t.to(torch.bfloat16) if as_bf16 else t.to(torch.float32)
            raw = t.view(torch.uint8).reshape(-1)
            if raw.numel() != nbytes.value:
                raise ValueError(f"tensor for param {param} has {raw.numel()} bytes, layout wants {nbytes.value}")
            blob[off.value:off.value + nbytes.value] = raw

        if cfg.kind == "vision":
            pw = w["v.patch_w"].reshape(cfg.width, -1)
            pad = torch.zeros(cfg.width, cfg.patch_k_pad, dtype=pw.dtype)
            pad[:, :cfg.patch_k] = pw
            put(_lib.P_PATCH_W, 0, pad, True)
            put(_lib.P_CLS, 0, w["v.cls"], False)
            put(_lib.P_POS, 0, w["v.pos"], False)
            put(_lib.P_LN_PRE_W, 0, w["v.ln_pre.w"], False)
            put(_lib.P_LN_PRE_B, 0, w["v.ln_pre.b"], False)
            put(_lib.P_LN_FINAL_W, 0, w["v.ln_post.w"], False)
            put(_lib.P_LN_FINAL_B, 0, w["v.ln_post.b"], False)
            put(_lib.P_PROJ, 0, w["v.proj"], True)
        else:
            put(_lib.P_TOK_EMB, 0, w["t.tok"], True)
            put(_lib.P_POS, 0, w["t.pos"], False)
            put(_lib.P_LN_FINAL_W, 0, w["t.ln_final.w"], False)
            put(_lib.P_LN_FINAL_B, 0, w["t.ln_final.b"], False)
            put(_lib.P_PROJ, 0, w["t.proj"], True)
        for i in range(cfg.layers):
            lw = {name: w[f"{pre}.l{i}.{name}"].detach().to("cpu", torch.float32) for name, _, _ in self._LAYER_PARAMS}
            if self.fold_ln:
                lw["qkv.w"], lw["qkv.b"], qkv_c = fold_layernorm(lw["qkv.w"], lw["qkv.b"], lw["ln1.w"], lw["ln1.b"])
                lw["fc1.w"], lw["fc1.b"], fc1_c = fold_layernorm(lw["fc1.w"], lw["fc1.b"], lw["ln2.w"], lw["ln2.b"])
                put(_lib.P_QKV_C, i, qkv_c, False)
                put(_lib.P_FC1_C, i, fc1_c, False)
            for name, pid, bf in self._LAYER_PARAMS:
                put(pid, i, lw[name], bf)

        self.blob = blob.to(device)
        handle = ctypes.c_void_p()
        _lib.check(self.L.mmr_tower_create(ctypes.byref(self.c), self.blob.data_ptr(), self.blob.numel(),
                                           ctypes.byref(handle)))
        self.handle = handle
        self._lanes = {}           # lane -> workspace tensor; lane 0 is the default (`_ws` reads it)

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            try:
                self.L.mmr_tower_destroy(h)
            except Exception:
                pass

    @property
    def _ws(self):
        return self._lanes.get(0)

    def set_shared_chip(self, shared: bool) -> None:
        """Tile-policy hint (mmr_tower_set_shared_chip): forwards of this tower will run beside other concurrent work."""
        _lib.check(self.L.mmr_tower_set_shared_chip(self.handle, int(bool(shared))))

    def workspace(self, batch: int, lane: int = 0) -> torch.Tensor:
        """The caller-owned scratch of one forward pass (include/mmr.h).  The tower object itself is read-only during a
        forward, so calls that use DIFFERENT lanes may be in flight at once on different HIP streams (two batches of a
        gallery build: the second fills the CUs the first one's 150-200-tile GEMM launches leave idle)."""
        need = self.L.mmr_tower_workspace_bytes(self.handle, batch)
        ws = self._lanes.get(lane)
        if ws is None or ws.numel() < need:
            ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            self._lanes[lane] = ws
        return ws

    def status_word(self, lane: int = 0) -> int:
        """The forward pass's status word (include/mmr.h: first int32 of the workspace; bit 0 = a token id was out of
        range and clamped).  Reading it synchronises."""
        ws = self._lanes.get(lane)
        return 0 if ws is None else int(ws[:4].view(torch.int32)[0])

    def forward(self, inp: torch.Tensor, out_dtype: torch.dtype, normalize: bool, tap_after: int = -1,
                tap: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, lane: int = 0) -> torch.Tensor:
        B = inp.shape[0]
        if out is None:
            out = torch.empty(B, self.cfg.embed_dim, dtype=out_dtype, device=self.device)
        elif (tuple(out.shape) != (B, self.cfg.embed_dim) or out.dtype != out_dtype or not out.is_contiguous()
              or out.device != self.device):
            raise ValueError(f"out must be a contiguous {out_dtype} [{B},{self.cfg.embed_dim}] tensor on {self.device}")
        if B == 0:
            return out
        ws = self.workspace(B, lane)
        if inp.data_ptr() & 15:            # a slice of a larger batch can start off the 16-byte grid the kernels load on
            inp = inp.clone()
        in_code = _lib.dtype_code(inp.dtype) if self.cfg.kind == "vision" else _lib.MMR_F32
        _lib.check(self.L.mmr_tower_forward(self.handle, inp.data_ptr(), in_code, B, out.data_ptr(),
                                            _lib.dtype_code(out_dtype), int(bool(normalize)), int(tap_after),
                                            _lib.ptr(tap), ws.data_ptr(), ws.numel(), _lib.stream_ptr(self.device)))
        return out


class CLIP:
    """Drop-in for the object ``clip.load`` returns (and for HF ``CLIPModel`` where the reference
    uses ``get_image_features`` / ``logit_scale``)."""

    max_batch = 512   # images (or texts) per launch sequence; larger inputs are processed in slices

    def __init__(self, cfg: ClipConfig, weights: Dict[str, torch.Tensor], device="cuda", fold_ln: Optional[bool] = None):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("this CLIP runs on MI355X only (device must be 'cuda'); there is no CPU path")
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.cfg, self.device = cfg, device
        with torch.cuda.device(device):
            self.visual = _Tower(cfg.vision, weights, device, fold_ln)
            self.text = _Tower(cfg.text, weights, device, fold_ln)
        self.logit_scale = weights["logit_scale"].detach().clone().to(device=device, dtype=torch.float32)
        # exp(logit_scale) as a host number, computed once from the host copy of the weights: forward() must not
        # read a device scalar back (a host sync per call)
        self._logit_scale_exp = float(weights["logit_scale"].detach().double().cpu().exp())
        # Output dtype.  Arithmetic is always bf16-in / fp32-accumulate MFMA; the RETURNED tensors default
        # to float32 because the reference's callers do `.cpu().numpy()` on them
        # (code/search_image.py:109,158), which numpy cannot do for bf16.  `.bfloat16()` switches the
        # outputs to bf16 -- the gallery storage type of the search path.
        self._dtype = torch.float32
        self.training = False

    # ---- nn.Module-ish surface the reference touches
    @property
    def dtype(self) -> torch.dtype:
        return self._dtype

    @property
    def input_resolution(self) -> int:
        return self.cfg.vision.image_size

    @property
    def context_length(self) -> int:
        return self.cfg.text.tokens

    @property
    def vocab_size(self) -> int:
        return self.cfg.text.vocab

    @contextlib.contextmanager
    def shared_chip(self, shared: bool = True):
        """``with model.shared_chip():`` around code that keeps several batches in flight on different lanes / HIP streams
        (``gallery._Lanes`` does it): the towers' GEMMs then choose tiles for efficiency per FLOP rather than for filling the
        chip alone.  Same results; about +3 % images/s with two ViT-B/32 forwards in flight, -5 % with one."""
        for t in (self.visual, self.text):
            t.set_shared_chip(shared)
        try:
            yield self
        finally:
            for t in (self.visual, self.text):
                t.set_shared_chip(False)

    def eval(self):
        return self

    def train(self, mode: bool = False):
        if mode:
            raise RuntimeError("inference-only model: every reference call site runs it under torch.no_grad()")
        return self

    def requires_grad_(self, flag: bool = False):
        return self

    def cuda(self, device=None):
        return self

    def to(self, *args, **kwargs):
        for a in list(args) + list(kwargs.values()):
            if isinstance(a, torch.dtype):
                self._set_dtype(a)
            elif isinstance(a, (str, torch.device)) and torch.device(a).type != "cuda":
                raise RuntimeError("this CLIP cannot leave the GPU; there is no CPU path")
        return self

    def float(self):
        """Features come back as float32 (the default; arithmetic stays bf16-in / fp32-accumulate MFMA)."""
        return self._set_dtype(torch.float32)

    def bfloat16(self):
        return self._set_dtype(torch.bfloat16)

    def half(self):
        raise RuntimeError("fp16 is not built for gfx950 here; use bfloat16 (default) or float()")

    def _set_dtype(self, dt):
        if dt not in (torch.float32, torch.bfloat16):
            raise TypeError(f"model dtype must be float32 or bfloat16, got {dt}")
        self._dtype = dt
        return self

    # ---- encoders
    def _prep_pixels(self, image: torch.Tensor) -> torch.Tensor:
        if image.dim() != 4 or image.shape[1] != 3:
            raise ValueError(f"expected pixels [B,3,H,W], got {tuple(image.shape)}")
        S = self.cfg.vision.image_size
        if image.shape[2] != S or image.shape[3] != S:
            # same failure the HF tower raises (modeling_clip.py:204-207)
            raise ValueError(f"Input image size ({image.shape[2]}*{image.shape[3]}) doesn't match model ({S}*{S}).")
        if image.dtype not in (torch.float32, torch.bfloat16):
            image = image.to(torch.float32)
        return image.to(self.device).contiguous()

    @torch.no_grad()
    def encode_image(self, image: torch.Tensor, normalize: bool = False, out: Optional[torch.Tensor] = None,
                     lane: int = 0) -> torch.Tensor:
        """[B,3,S,S] -> [B,E] in ``self.dtype``; a fresh, writable tensor (callers do ``/=`` on it).  ``out``: write the
        rows into this preallocated [B,E] tensor instead (a slice of a gallery being built) and return it.  ``lane``: which
        of the model's independent workspaces the call uses -- calls on different lanes may run concurrently on different
        HIP streams (``gallery.encode_gallery(..., lanes=2)``); calls on one lane must be stream-ordered."""
        px = self._prep_pixels(image)
        if out is not None and tuple(out.shape) != (px.shape[0], self.cfg.embed_dim):
            raise ValueError(f"out must be [{px.shape[0]},{self.cfg.embed_dim}], got {tuple(out.shape)}")
        with torch.cuda.device(self.device):
            outs = [self.visual.forward(px[s:s + self.max_batch], self._dtype, normalize,
                                        out=None if out is None else out[s:s + self.max_batch], lane=lane)
                    for s in range(0, px.shape[0], self.max_batch)]
        if out is not None:
            return out
        return outs[0] if len(outs) == 1 else torch.cat(outs) if outs else \
            torch.empty(0, self.cfg.embed_dim, dtype=self._dtype, device=self.device)

    def _prep_ids(self, text: torch.Tensor) -> torch.Tensor:
        if text is None:
            raise ValueError("You have to specify input_ids")      # modeling_clip.py:535-536
        text = torch.as_tensor(text)
        if text.dim() == 1:
            text = text.unsqueeze(0)
        T = self.cfg.text.tokens
        if text.dim() != 2 or text.shape[1] != T:
            raise ValueError(f"expected token ids [N,{T}], got {tuple(text.shape)}")
        if text.dtype.is_floating_point:
            raise TypeError("token ids must be an integer tensor")
        # Range check without a device round trip: ids that are still on the host (what clip.tokenize returns) are
        # checked here; ids already on the GPU are checked by the embedding kernel itself, which clamps them and
        # raises the tower's status word (read it with `model.text_id_errors()` -- that read is the only sync).
        if not text.is_cuda and text.numel() and (int(text.min()) < 0 or int(text.max()) >= self.cfg.text.vocab):
            raise IndexError(f"token id outside [0,{self.cfg.text.vocab})")
        return text.to(device=self.device, dtype=torch.int32).contiguous()

    def text_id_errors(self, lane: int = 0) -> bool:
        """True if the LAST encode_text call on this lane -- all of its slices when N > max_batch -- saw a token id outside
        [0, vocab) (the kernel clamped it).
        Synchronises the device; meant for callers that feed ids produced on the GPU."""
        return self.text.status_word(lane) != 0

    @torch.no_grad()
    def encode_text(self, text: torch.Tensor, normalize: bool = False, lane: int = 0) -> torch.Tensor:
        """int [N,77] -> [N,E]; pooled at the EOT token = ``text.argmax(-1)``.  ``lane``: see ``encode_image``."""
        ids = self._prep_ids(text)
        with torch.cuda.device(self.device):
            starts = range(0, ids.shape[0], self.max_batch)
            # every C forward call zeroes the workspace's status word (include/mmr.h): over several slices the words are
            # OR-ed into one device int and written back after the last slice, so text_id_errors() covers the WHOLE call
            # (stream-ordered tensor ops, no host read; single-slice calls -- the common case -- launch nothing extra)
            acc = torch.zeros(1, dtype=torch.int32, device=self.device) if len(starts) > 1 else None
            outs = []
            for s in starts:
                outs.append(self.text.forward(ids[s:s + self.max_batch], self._dtype, normalize, lane=lane))
                if acc is not None:
                    acc |= self.text._lanes[lane][:4].view(torch.int32)
            if acc is not None:
                self.text._lanes[lane][:4].view(torch.int32).copy_(acc)
        return outs[0] if len(outs) == 1 else torch.cat(outs) if outs else \
            torch.empty(0, self.cfg.embed_dim, dtype=self._dtype, device=self.device)

    # HF spellings (reference code/test_taiyi.py:23, CLIP-Chinese/lab_chinese.py:114)
    def get_image_features(self, pixel_values: torch.Tensor = None, **_):
        return self.encode_image(pixel_values)

    def get_text_features(self, input_ids: torch.Tensor = None, **_):
        return self.encode_text(input_ids)

    @torch.no_grad()
    def forward(self, image: torch.Tensor, text: torch.Tensor):
        """-> (logits_per_image [B,N], logits_per_text [N,B]) = exp(logit_scale) * I_n @ T_n^T."""
        from .search import similarity

        img = self.encode_image(image, normalize=True).to(torch.float32)
        txt = self.encode_text(text, normalize=True).to(torch.float32)
        scale = self._logit_scale_exp
        logits_per_image = similarity(img, txt, scale)            # [B,N] = img @ txt.T * scale
        if logits_per_image.dim() == 1:
            logits_per_image = logits_per_image.unsqueeze(1)
        return logits_per_image.contiguous(), logits_per_image.t().contiguous()      # fp32, like CPU clip

    __call__ = forward


def _preprocess_factory(n_px: int, device: torch.device, model=None, pixel_dtype: Optional[torch.dtype] = None):
    """The ``preprocess`` callable ``clip.load`` returns: Resize(n_px, BICUBIC) -> CenterCrop(n_px) ->
    ToTensor -> Normalize(CLIP mean/std) (reference use sites code/search_image.py:127,155; constants
    code/custom.py:28).  The decoded uint8 pixels are uploaded once and everything else runs in
    csrc/preprocess.hip, bit-exact to Pillow's resize.  Accepts a PIL image, an HWC
    uint8 array or tensor; returns [3,n_px,n_px] ON THE MODEL'S DEVICE (the reference's following
    ``.unsqueeze(0).cuda()`` / ``torch.stack(...).cuda()`` are then no-ops).

    Output dtype: ``pixel_dtype`` if given (``load(..., pixel_dtype=torch.bfloat16)``), else the model's dtype at call
    time -- float32 for the default model (what the reference's ``preprocess`` returns), bfloat16 after
    ``model.bfloat16()``.  The encoder rounds pixels to bf16 on its way into the matrix cores either way, so the
    features are bit-identical; bf16 pixels let patch-32 towers gather patches inside the GEMM's A-tile loads
    (no im2col pass) and halve the pixel bytes."""
    from .preprocess import preprocess_image

    def preprocess(img, dtype: Optional[torch.dtype] = None) -> torch.Tensor:
        if isinstance(img, torch.Tensor):
            x = img
        else:
            import numpy as np

            if hasattr(img, "convert"):
                img = img.convert("RGB")
            x = torch.from_numpy(np.ascontiguousarray(np.asarray(img)))
        if x.dtype != torch.uint8 or x.dim() != 3 or x.shape[2] != 3:
            raise TypeError("preprocess expects a PIL image or a uint8 [H,W,3] array/tensor "
                            f"(got {x.dtype} {tuple(x.shape)})")
        dt = dtype or pixel_dtype or (model.dtype if model is not None else torch.float32)
        return preprocess_image(x.to(device, non_blocking=True), n_px, out_dtype=dt)

    return preprocess


class BatchFeature(dict):
    """What ``CLIPProcessor.__call__`` returns: a dict with attribute access and ``.to(device)``."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def to(self, *args, **kwargs):
        return BatchFeature({k: (v.to(*args, **kwargs) if isinstance(v, torch.Tensor) else v) for k, v in self.items()})


class ClipProcessor:
    """``CLIPProcessor`` spelling of the preprocess step, as the reference's HF flow calls it
    (code/test_taiyi.py:18-19 ``processor(images=img, return_tensors="pt")`` -> ``get_image_features(**image)``;
    CLIP-Chinese/lab_chinese.py:84).  images: one PIL image / uint8 HWC array or a list -> ``pixel_values``
    [B,3,S,S] on the model's device (resize-shorter-side bicubic, centre crop, /255, CLIP mean/std: the same
    csrc/preprocess.hip kernels as ``preprocess``).  text: strings -> ``input_ids`` / ``attention_mask`` through the
    CLIP BPE when its merge table is available (``tokenize``)."""

    def __init__(self, preprocess, context_length: int = _CONTEXT):
        self.preprocess, self.context_length = preprocess, context_length

    def __call__(self, text=None, images=None, return_tensors: str = "pt", padding=True, **_):
        if return_tensors != "pt":
            raise ValueError("only return_tensors='pt' is supported")
        out = BatchFeature()
        if images is not None:
            single = not isinstance(images, (list, tuple))
            imgs = [images] if single else list(images)
            out["pixel_values"] = torch.stack([self.preprocess(im) for im in imgs])
        if text is not None:
            ids = tokenize(text, self.context_length)
            out["input_ids"] = ids
            eot = ids.argmax(-1, keepdim=True)
            out["attention_mask"] = (torch.arange(ids.shape[1])[None, :] <= eot).to(torch.int64)
        if not out:
            raise ValueError("You have to specify either text or images")
        return out


def _find_checkpoint(name: str, download_root: Optional[str]) -> Optional[str]:
    """A checkpoint file for `name`: `name` itself if it is a file, else <download_root>/<ViT-B-32>.{safetensors,state.pt,pt,bin}
    (the OpenAI package caches "ViT-B/32" as the TorchScript archive ViT-B-32.pt under download_root, default ~/.cache/clip;
    checkpoint.read_state_dict reads it with an allow-listed unpickler, without torch.jit.load)."""
    if os.path.isfile(name):
        return name
    # no download_root: the OpenAI package's own default, where a `clip.load(name)` of the reference left its file
    root = download_root or os.path.join("~", ".cache", "clip")
    stem = name.replace("/", "-").replace("@", "-")
    for ext in (".safetensors", ".state.pt", ".pt", ".bin"):
        cand = os.path.join(os.path.expanduser(root), stem + ext)
        if os.path.isfile(cand):
            return cand
    return None


SYNTHETIC = "synthetic"
_SYNTHETIC_ONLY = ("tiny-test",)      # test geometries: no checkpoint of them exists anywhere


def load(name: str = "ViT-B/32", device: Union[str, torch.device] = "cuda", jit: bool = False,
         download_root: Optional[str] = None, weights: Union[None, str, Dict[str, torch.Tensor]] = None, seed: int = 0,
         fold_ln: Optional[bool] = None, pixel_dtype: Optional[torch.dtype] = None):
    """``clip.load`` signature.  Weights come from, in order: ``weights`` (a checkpoint path, or a state dict in
    OpenAI-CLIP, HF-CLIPModel or this package's naming -- checkpoint.py converts); a checkpoint file named by
    ``name`` or found under ``download_root`` (the OpenAI package's cache layout, ``ViT-B-32.pt``).
    Nothing can be downloaded here, so when none of these yields a file the call raises FileNotFoundError --
    it does NOT silently hand back a model that encodes noise.  ``weights="synthetic"`` asks for that on
    purpose: seeded random tensors of the named architecture (``seed``), the ones the golden fixtures, the
    tests and bench.py use."""
    from . import checkpoint

    synthetic = isinstance(weights, str) and weights == SYNTHETIC
    if synthetic:
        weights = None
    path = weights if isinstance(weights, (str, os.PathLike)) else \
        (_find_checkpoint(name, download_root) if weights is None and not synthetic else None)
    if path is not None:
        weights = checkpoint.read_state_dict(os.fspath(path))
    if weights is not None:
        kind, weights = checkpoint.convert_state_dict(weights)
        if kind != "clip":
            raise ValueError(f"checkpoint holds a {kind} model; use load_text_encoder for BERT text towers")
        inferred = checkpoint.infer_clip_config(weights, name)
        try:
            cfg = get_config(name)
        except RuntimeError:
            if not os.path.isfile(name):
                raise
            cfg = inferred
        if (cfg.vision, cfg.text) != (inferred.vision, inferred.text):
            raise ValueError(f"checkpoint geometry {inferred.vision} / {inferred.text} does not match {name}")
    else:
        cfg = get_config(name)            # unknown names raise here, as clip.load does
        if not synthetic and name not in _SYNTHETIC_ONLY:
            where = f" or under download_root={download_root!r}" if download_root else ""
            raise FileNotFoundError(
                f"no checkpoint for {name!r}: nothing at that path{where}, and this environment cannot download. "
                f"Pass weights=<file or state dict> (OpenAI-clip / HF CLIPModel / safetensors), put "
                f"{name.replace('/', '-').replace('@', '-')}.pt under download_root, or ask for random weights "
                f"explicitly with weights={SYNTHETIC!r}.")
        weights = make_clip_weights(cfg, seed=seed)
    model = CLIP(cfg, weights, device, fold_ln=fold_ln)
    model.synthetic_weights = path is None and synthetic or name in _SYNTHETIC_ONLY and path is None
    preprocess = _preprocess_factory(cfg.vision.image_size, model.device, model, pixel_dtype)
    model.processor = ClipProcessor(preprocess, cfg.text.tokens)
    return model, preprocess


_bpe_cache = {}


def tokenize(texts: Union[str, List[str], torch.Tensor, List[List[int]]], context_length: int = _CONTEXT,
             truncate: bool = False, bpe_path: Optional[str] = None) -> torch.Tensor:
    """``clip.tokenize`` surface -> IntTensor[N, context_length].

    Strings are encoded with the CLIP byte-level BPE (tokenizer.ClipBPETokenizer) when the merge table is
    supplied -- ``bpe_path`` or the ``MMR_CLIP_BPE`` environment variable pointing at the ``clip`` package's
    ``bpe_simple_vocab_16e6.txt.gz`` (or an HF ``merges.txt``).  That file is not reachable offline
    (SURVEY.md 8f row 4); without it strings raise RuntimeError and token ids are accepted instead (one list
    per text, SOT..EOT, unpadded or padded), zero-padded to ``context_length`` like the original.  Over-long
    inputs raise RuntimeError exactly as ``clip.tokenize`` does unless ``truncate``."""
    if isinstance(texts, str) or (isinstance(texts, (list, tuple)) and texts and isinstance(texts[0], str)):
        import os

        path = bpe_path or os.environ.get("MMR_CLIP_BPE")
        if not path:
            raise RuntimeError("BPE vocabulary is not available offline; pass token ids instead of strings, or give "
                               "bpe_path= / MMR_CLIP_BPE pointing at bpe_simple_vocab_16e6.txt.gz")
        if path not in _bpe_cache:
            from .tokenizer import ClipBPETokenizer

            _bpe_cache[path] = ClipBPETokenizer(path)
        return _bpe_cache[path](texts, context_length, truncate)
    if isinstance(texts, torch.Tensor):
        rows = texts.tolist() if texts.dim() == 2 else [texts.tolist()]
    else:
        rows = list(texts)
        if rows and not isinstance(rows[0], (list, tuple)):
            rows = [rows]
    out = torch.zeros(len(rows), context_length, dtype=torch.int32)
    for i, r in enumerate(rows):
        r = list(r)
        while r and r[-1] == 0:
            r.pop()
        if len(r) > context_length:
            if not truncate:
                raise RuntimeError(f"Input {i} is too long for context length {context_length}")
            r = r[:context_length - 1] + [r[-1]]
        out[i, :len(r)] = torch.tensor(r, dtype=torch.int32)
    return out
