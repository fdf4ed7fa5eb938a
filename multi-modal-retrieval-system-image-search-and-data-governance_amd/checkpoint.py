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
import io
import math
import os
import pickle
import re
import zipfile
from collections import OrderedDict
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
# OpenAI publishes its CLIP checkpoints as TorchScript archives: ``clip.load("ViT-B/32")`` (reference
# code/search_image.py:327, code/test_clip.py:6) downloads ``ViT-B-32.pt`` into ``~/.cache/clip`` and that is the file a
# user of the reference has on disk.  ``torch.load(weights_only=True)`` refuses such an archive and ``torch.jit.load``
# would execute the code stored inside it.  The reader below does neither: a TorchScript archive is a zip holding
# ``<name>/data.pkl`` (the module tree: nested objects whose attributes are tensors, sub-modules and a few flags) and
# ``<name>/data/<key>`` (raw storages).  ``data.pkl`` is unpickled with an allow-list -- tensor rebuild calls, storage
# persistent ids, OrderedDict, and inert placeholder objects for the ``__torch__.*`` module classes; any other global
# raises -- the ``code/`` directory is never opened, and the state dict is the dotted attribute path of every tensor.
_STORAGE_DTYPES = {
    "FloatStorage": torch.float32, "HalfStorage": torch.float16, "BFloat16Storage": torch.bfloat16,
    "DoubleStorage": torch.float64, "LongStorage": torch.int64, "IntStorage": torch.int32, "ShortStorage": torch.int16,
    "CharStorage": torch.int8, "ByteStorage": torch.uint8, "BoolStorage": torch.bool,
}


class _ScriptObject:
    """Inert stand-in for a ``__torch__.*`` class instance: holds the attribute dict pickle's BUILD hands it."""

    def __setstate__(self, state):
        if not isinstance(state, dict):
            raise pickle.UnpicklingError("TorchScript module state is not an attribute dict")
        self.__dict__.update(state)


class _StorageType:
    def __init__(self, dtype):
        self.dtype = dtype


def _rebuild_tensor(storage, offset, size, stride, *_ignored):
    """torch._utils._rebuild_tensor_v2 restated: a strided view over the storage bytes read from the archive."""
    flat = storage
    size, stride = tuple(int(x) for x in size), tuple(int(x) for x in stride)
    need = 1 + sum((n - 1) * st for n, st in zip(size, stride)) if all(n > 0 for n in size) else 0
    if offset < 0 or any(n < 0 for n in size) or any(st < 0 for st in stride) or offset + need > flat.numel():
        raise pickle.UnpicklingError("tensor view outside its storage")
    return torch.as_strided(flat, size, stride, int(offset)).clone()


class _ScriptArchiveUnpickler(pickle.Unpickler):
    def __init__(self, data: bytes, zf, prefix: str):
        super().__init__(io.BytesIO(data))
        self._zf, self._prefix, self._storages = zf, prefix, {}

    def find_class(self, module, name):
        if module == "torch._utils" and name in ("_rebuild_tensor_v2", "_rebuild_tensor"):
            return _rebuild_tensor
        if module == "torch._utils" and name == "_rebuild_parameter":
            return lambda data, requires_grad=False, backward_hooks=None: data
        if module == "collections" and name == "OrderedDict":
            return OrderedDict
        if module == "torch.jit._pickle":
            # typed-container tags around plain lists / values (Conv2d.stride and the like): restated as identities
            if name in ("build_intlist", "build_doublelist", "build_boollist", "build_tensorlist"):
                return lambda data: data
            if name == "restore_type_tag":
                return lambda value, type_str=None: value
        if module == "torch" and name in _STORAGE_DTYPES:
            return _StorageType(_STORAGE_DTYPES[name])
        if module == "__torch__" or module.startswith("__torch__."):
            return type(name, (_ScriptObject,), {})           # a fresh inert class: nothing of the archive's code runs
        raise pickle.UnpicklingError(f"refusing global {module}.{name}: not part of a tensor-only TorchScript module tree")

    def persistent_load(self, pid):
        if not (isinstance(pid, tuple) and len(pid) >= 5 and pid[0] == "storage" and isinstance(pid[1], _StorageType)):
            raise pickle.UnpicklingError(f"unexpected persistent id {pid!r}")
        _, st, key, _location, numel = pid[:5]
        key = str(key)
        if not re.fullmatch(r"[0-9A-Za-z_]+", key):
            raise pickle.UnpicklingError(f"bad storage key {key!r}")
        if key not in self._storages:
            raw = self._zf.read(f"{self._prefix}/data/{key}")
            t = torch.frombuffer(bytearray(raw), dtype=st.dtype) if raw else torch.empty(0, dtype=st.dtype)
            if t.numel() < int(numel):
                raise pickle.UnpicklingError(f"storage {key} holds {t.numel()} elements, {numel} declared")
            self._storages[key] = t
        return self._storages[key]


def _is_torchscript_archive(path: str) -> bool:
    if not zipfile.is_zipfile(path):
        return False
    with zipfile.ZipFile(path) as zf:
        names = zf.namelist()
    return any(n.endswith("/constants.pkl") or "/code/" in n for n in names) and any(n.endswith("/data.pkl") for n in names)


def read_torchscript_state_dict(path: str):
    """State dict of a TorchScript module archive (``torch.jit.save`` / OpenAI's ``ViT-*.pt``) WITHOUT ``torch.jit.load``:
    nothing stored in the file is executed (see the comment above).  Returns {dotted attribute path: tensor}."""
    with zipfile.ZipFile(path) as zf:
        pk = [n for n in zf.namelist() if n.endswith("/data.pkl") and n.count("/") == 1]
        if len(pk) != 1:
            raise RuntimeError(f"{path}: expected exactly one <name>/data.pkl, found {pk}")
        prefix = pk[0].split("/")[0]
        root = _ScriptArchiveUnpickler(zf.read(pk[0]), zf, prefix).load()
    if not isinstance(root, _ScriptObject):
        raise RuntimeError(f"{path}: data.pkl does not hold a module object")
    sd: Dict[str, torch.Tensor] = {}
    seen = set()

    def walk(obj, dotted):
        if id(obj) in seen:                                   # shared sub-modules: first path wins, no cycles
            return
        seen.add(id(obj))
        for k, v in vars(obj).items():
            if isinstance(v, torch.Tensor):
                sd[dotted + k] = v
            elif isinstance(v, _ScriptObject):
                walk(v, dotted + k + ".")

    walk(root, "")
    if not sd:
        raise RuntimeError(f"{path}: the module tree holds no tensors")
    return sd


def read_state_dict(path: str):
    """Read a checkpoint file WITHOUT executing anything from it: ``.safetensors`` through safetensors, TorchScript
    archives (OpenAI's published ``ViT-*.pt``, what ``clip.load`` leaves in ``~/.cache/clip``) through
    :func:`read_torchscript_state_dict`, everything else through ``torch.load(weights_only=True)``."""
    if not os.path.isfile(path):
        raise FileNotFoundError(path)
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file

        return load_file(path, device="cpu")
    if _is_torchscript_archive(path):
        try:
            return read_torchscript_state_dict(path)
        except pickle.UnpicklingError as e:
            raise RuntimeError(f"{path}: TorchScript archive refused: {e}") from e
    try:
        obj = torch.load(path, map_location="cpu", weights_only=True)
    except Exception as e:  # pickled module / arbitrary objects
        raise RuntimeError(f"{path}: not a plain tensor state dict ({type(e).__name__}: {e}). "
                           "Only tensors are loaded (weights_only=True / the allow-listed TorchScript reader).") from e
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
