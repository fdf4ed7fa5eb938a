"""BERT-style text classifier as a text tower (the Taiyi Chinese text encoder of the CN pipeline).

Mirrors how the reference uses it (code/test_taiyi.py:12,24; CLIP-Chinese/lab_chinese.py:81-93):
    text_encoder = BertForSequenceClassification.from_pretrained("IDEA-CCNL/Taiyi-CLIP-Roberta-large-326M-Chinese").eval()
    text_features = text_encoder(text).logits          # text = tokenizer(..., padding=True)['input_ids']
That call passes ids only, so all tokens (pads included) attend to each other with token type 0
(csrc/tower.hip::mmr_bert_forward).  The other spelling the reference uses,
    inputs = text_tokenizer([text], return_tensors="pt", padding=True).to(device); text_encoder(**inputs).logits
(CLIP/union_dataset.py:312-314, CLIP-Chinese/lab_chinese.py:90-92) hands over the tokenizer's whole output:
``attention_mask`` becomes a key-padding mask inside the attention kernels and ``token_type_ids`` select the
type-embedding row (mmr_bert_forward_masked), so a padded multi-text batch matches HF.
Weights / vocabulary are not reachable offline: pass ``weights=`` (a checkpoint, or names as in
weights.make_bert_weights); ``weights="synthetic"`` asks for the seeded random ones the golden fixtures use.
"""
import ctypes
from types import SimpleNamespace
from typing import Dict, Optional

import torch

from . import _lib
from .config import BertTextConfig, get_bert_config
from .weights import make_bert_weights


class BertTextEncoder:
    max_batch = 512

    def __init__(self, cfg: BertTextConfig, weights: Dict[str, torch.Tensor], device="cuda"):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("this encoder runs on MI355X only (device must be 'cuda'); there is no CPU path")
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.cfg, self.device = cfg, device
        self.L = _lib.lib()
        self.c = _lib.TowerCfg(kind=2, width=cfg.width, layers=cfg.layers, heads=cfg.heads, mlp=cfg.mlp,
                               tokens=cfg.max_positions, embed_dim=cfg.embed_dim, image_size=0, patch=0,
                               vocab=cfg.vocab, ln_eps=cfg.ln_eps)
        total = self.L.mmr_tower_weights_bytes(ctypes.byref(self.c))
        if total == 0:
            raise _lib.MMRError(-22, f"unsupported BERT geometry {cfg}")
        blob = torch.zeros(total, dtype=torch.uint8)

        def put(param, layer, tensor, as_bf16):
            off, nbytes = ctypes.c_size_t(), ctypes.c_size_t()
            _lib.check(self.L.mmr_tower_param_span(ctypes.byref(self.c), param, layer, ctypes.byref(off),
                                                   ctypes.byref(nbytes)))
            t = tensor.detach().to("cpu").contiguous()
            raw = (t.to(torch.bfloat16) if as_bf16 else t.to(torch.float32)).view(torch.uint8).reshape(-1)
            if raw.numel() != nbytes.value:
                raise ValueError(f"tensor for param {param} has {raw.numel()} bytes, layout wants {nbytes.value}")
            blob[off.value:off.value + nbytes.value] = raw

        w = weights
        put(_lib.P_TOK_EMB, 0, w["b.tok"], True)
        put(_lib.P_POS, 0, w["b.pos"], False)
        put(_lib.P_TYPE_EMB, 0, w["b.type"], False)
        put(_lib.P_LN_PRE_W, 0, w["b.ln_emb.w"], False)
        put(_lib.P_LN_PRE_B, 0, w["b.ln_emb.b"], False)
        put(_lib.P_POOL_W, 0, w["b.pool.w"], True)
        put(_lib.P_POOL_B, 0, w["b.pool.b"], False)
        put(_lib.P_PROJ, 0, w["b.cls.w"], True)
        put(_lib.P_PROJ_B, 0, w["b.cls.b"], False)
        for i in range(cfg.layers):
            for name, pid, bf in (("qkv.w", _lib.P_QKV_W, True), ("qkv.b", _lib.P_QKV_B, False),
                                  ("out.w", _lib.P_OUT_W, True), ("out.b", _lib.P_OUT_B, False),
                                  ("ln1.w", _lib.P_LN1_W, False), ("ln1.b", _lib.P_LN1_B, False),
                                  ("fc1.w", _lib.P_FC1_W, True), ("fc1.b", _lib.P_FC1_B, False),
                                  ("fc2.w", _lib.P_FC2_W, True), ("fc2.b", _lib.P_FC2_B, False),
                                  ("ln2.w", _lib.P_LN2_W, False), ("ln2.b", _lib.P_LN2_B, False)):
                put(pid, i, w[f"b.l{i}.{name}"], bf)
        self.blob = blob.to(device)
        handle = ctypes.c_void_p()
        _lib.check(self.L.mmr_tower_create(ctypes.byref(self.c), self.blob.data_ptr(), self.blob.numel(),
                                           ctypes.byref(handle)))
        self.handle = handle
        self._ws = None
        self._dtype = torch.float32

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            try:
                self.L.mmr_tower_destroy(h)
            except Exception:
                pass

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    def cuda(self, device=None):
        return self

    @property
    def dtype(self):
        return self._dtype

    def _ids_like(self, t, name, shape):
        t = torch.as_tensor(t)
        if t.dim() == 1:
            t = t.unsqueeze(0)
        if t.dtype.is_floating_point or t.dtype == torch.bool:
            t = t.to(torch.int32)
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f"{name} {tuple(t.shape)} does not match input_ids {tuple(shape)}")
        return t.to(device=self.device, dtype=torch.int32).contiguous()

    @torch.no_grad()
    def logits(self, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor] = None,
               token_type_ids: Optional[torch.Tensor] = None, normalize: bool = False, tap_after: int = -1,
               tap: Optional[torch.Tensor] = None) -> torch.Tensor:
        if input_ids is None:
            raise ValueError("You have to specify input_ids")
        ids = torch.as_tensor(input_ids)
        if ids.dim() == 1:
            ids = ids.unsqueeze(0)
        if ids.dim() != 2 or ids.dtype.is_floating_point:
            raise ValueError(f"expected integer token ids [N,T], got {ids.dtype} {tuple(ids.shape)}")
        N, T = ids.shape
        if not 1 <= T <= self.cfg.max_positions:
            raise ValueError(f"sequence length {T} outside [1,{self.cfg.max_positions}]")
        # host-resident ids are range-checked here; device-resident ones by the embedding kernel (status word), so a
        # call never reads device data back
        if not ids.is_cuda and ids.numel() and (int(ids.min()) < 0 or int(ids.max()) >= self.cfg.vocab):
            raise IndexError(f"token id outside [0,{self.cfg.vocab})")
        ids = ids.to(device=self.device, dtype=torch.int32).contiguous()
        mask = None if attention_mask is None else self._ids_like(attention_mask, "attention_mask", (N, T))
        types = None if token_type_ids is None else self._ids_like(token_type_ids, "token_type_ids", (N, T))
        out = torch.empty(N, self.cfg.embed_dim, dtype=self._dtype, device=self.device)
        with torch.cuda.device(self.device):
            need = self.L.mmr_bert_workspace_bytes(self.handle, min(self.max_batch, N), T) if N else 0
            if N and (self._ws is None or self._ws.numel() < need):
                self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            # several slices: OR the per-call status words into one device int (each C call zeroes the word), written
            # back after the last slice so id_errors() covers the whole call; stream-ordered, no host read
            acc = torch.zeros(1, dtype=torch.int32, device=self.device) if N > self.max_batch else None
            for s in range(0, N, self.max_batch):
                n = min(self.max_batch, N - s)
                _lib.check(self.L.mmr_bert_forward_masked(
                    self.handle, ids[s:s + n].data_ptr(), _lib.ptr(None if types is None else types[s:s + n]),
                    _lib.ptr(None if mask is None else mask[s:s + n]), n, T, out[s:s + n].data_ptr(),
                    _lib.dtype_code(self._dtype), int(bool(normalize)), int(tap_after), _lib.ptr(tap),
                    self._ws.data_ptr(), self._ws.numel(), _lib.stream_ptr(self.device)))
                if acc is not None:
                    acc |= self._ws[:4].view(torch.int32)
            if acc is not None:
                self._ws[:4].view(torch.int32).copy_(acc)
        return out

    def id_errors(self) -> bool:
        """True if the last call's kernels -- every slice of it -- saw a token / token-type id out of range (clamped).
        Synchronises; for callers whose ids were produced on the GPU."""
        return self._ws is not None and int(self._ws[:4].view(torch.int32)[0]) != 0

    def __call__(self, input_ids=None, attention_mask=None, token_type_ids=None, **_):
        """``text_encoder(text).logits`` / ``text_encoder(**tokenizer_output).logits`` surface."""
        return SimpleNamespace(logits=self.logits(input_ids, attention_mask, token_type_ids))


SYNTHETIC = "synthetic"
_SYNTHETIC_ONLY = ("tiny-bert-test",)


def load_text_encoder(name: str = "IDEA-CCNL/Taiyi-CLIP-Roberta-large-326M-Chinese", device="cuda",
                      weights=None, seed: int = 0) -> BertTextEncoder:
    """Counterpart of ``BertForSequenceClassification.from_pretrained(name)`` for the Taiyi text tower.
    ``weights``: a checkpoint path (safetensors / state-dict .bin) or a state dict in HF or this package's
    naming; ``name`` may also be a checkpoint file or a directory holding ``model.safetensors`` /
    ``pytorch_model.bin`` (the from_pretrained layout).  Nothing can be downloaded here: without a checkpoint the
    call raises FileNotFoundError unless random weights are asked for on purpose with ``weights="synthetic"``."""
    import os

    from . import checkpoint

    synthetic = isinstance(weights, str) and weights == SYNTHETIC
    if synthetic:
        weights = None
    path = weights if isinstance(weights, (str, os.PathLike)) else None
    if weights is None and not synthetic:
        if os.path.isfile(name):
            path = name
        elif os.path.isdir(name):
            for fn in ("model.safetensors", "pytorch_model.bin"):
                if os.path.isfile(os.path.join(name, fn)):
                    path = os.path.join(name, fn)
                    break
    if path is not None:
        weights = checkpoint.read_state_dict(os.fspath(path))
    if weights is not None:
        kind, weights = checkpoint.convert_state_dict(weights)
        if kind != "bert":
            raise ValueError(f"checkpoint holds a {kind} model, not a BERT text encoder")
        inferred = checkpoint.infer_bert_config(weights, os.path.basename(os.fspath(name)))
        try:
            cfg = get_bert_config(name)
        except RuntimeError:
            cfg = inferred
        geom = lambda c: (c.width, c.layers, c.heads, c.mlp, c.max_positions, c.vocab, c.embed_dim)  # noqa: E731
        if geom(cfg) != geom(inferred):
            raise ValueError(f"checkpoint geometry {geom(inferred)} does not match {name} {geom(cfg)}")
    else:
        cfg = get_bert_config(name)
        if not synthetic and name not in _SYNTHETIC_ONLY:
            raise FileNotFoundError(
                f"no checkpoint for {name!r}: not a file or a from_pretrained directory, and this environment cannot "
                f"download. Pass weights=<file or state dict>, or ask for random weights explicitly with "
                f"weights={SYNTHETIC!r}.")
        weights = make_bert_weights(cfg, seed=seed)
    return BertTextEncoder(cfg, weights, device)
