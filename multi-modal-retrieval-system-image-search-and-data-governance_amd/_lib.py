"""ctypes binding of csrc/libmmr_hip.so (the C ABI in include/mmr.h).

There is deliberately no fallback: if the HIP library is missing or a call fails, the
caller gets an exception -- never a silent CPU/torch path.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MMR_LIB") or os.path.join(_HERE, "csrc", "libmmr_hip.so")   # MMR_LIB: A/B builds

MMR_F32, MMR_BF16 = 0, 1
_ERRNAMES = {-5: "EIO", -22: "EINVAL", -28: "ENOSPC", -95: "ENOTSUP"}


class MMRError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libmmr_hip: {_ERRNAMES.get(code, code)}: {msg}")
        self.code = code


class TowerCfg(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in
                ("kind", "width", "layers", "heads", "mlp", "tokens", "embed_dim", "image_size", "patch", "vocab")]
    _fields_.append(("ln_eps", ctypes.c_float))
    _fields_.append(("fold_ln", ctypes.c_int))


# parameter ids, mirrored from include/mmr.h (mmr_param)
(P_PATCH_W, P_CLS, P_POS, P_LN_PRE_W, P_LN_PRE_B, P_LN1_W, P_LN1_B, P_QKV_W, P_QKV_B, P_OUT_W, P_OUT_B,
 P_LN2_W, P_LN2_B, P_FC1_W, P_FC1_B, P_FC2_W, P_FC2_B, P_LN_FINAL_W, P_LN_FINAL_B, P_PROJ, P_TOK_EMB,
 P_TYPE_EMB, P_POOL_W, P_POOL_B, P_PROJ_B, P_QKV_C, P_FC1_C, P_COUNT) = range(28)

_lib = None


def lib():
    """Load the HIP library once; raise loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for this path.")
    L = ctypes.CDLL(LIB_PATH)
    vp, i32, i64, f32, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_size_t
    L.mmr_last_error.restype = ctypes.c_char_p
    L.mmr_version.restype = i32
    L.mmr_search_workspace_bytes.restype = sz
    L.mmr_search_workspace_bytes.argtypes = [i64, i32, i32, i32]
    L.mmr_cosine_topk.restype = i32
    L.mmr_cosine_topk.argtypes = [vp, vp, i32, i32, i64, i32, i32, f32, f32, vp, vp, vp, vp, vp, sz, vp]
    L.mmr_cosine_topk_ex.restype = i32
    L.mmr_cosine_topk_ex.argtypes = [vp, vp, i32, i32, i64, i32, i32, f32, f32, vp, vp, vp, vp, vp, vp, sz, vp]
    L.mmr_gallery_split_bf16.restype = i32
    L.mmr_gallery_split_bf16.argtypes = [vp, i64, i32, vp, vp, vp, vp]
    L.mmr_cosine_topk_split.restype = i32
    L.mmr_cosine_topk_split.argtypes = [vp, vp, vp, vp, vp, i32, i64, i32, i32, f32, f32, vp, vp, vp, vp, vp, vp, sz, vp]
    L.mmr_gallery_norm_bound.restype = i32
    L.mmr_gallery_norm_bound.argtypes = [vp, i32, i64, i32, vp, vp]
    L.mmr_similarity.restype = i32
    L.mmr_similarity.argtypes = [vp, vp, i32, i32, i64, i32, f32, vp, vp]
    L.mmr_l2norm_rows.restype = i32
    L.mmr_l2norm_rows.argtypes = [vp, i32, i64, i32, vp]
    L.mmr_topk_merge.restype = i32
    L.mmr_topk_merge.argtypes = [vp, vp, i32, i32, i32, f32, vp, vp, vp, vp]
    L.mmr_topk_pack.restype = i32
    L.mmr_topk_pack.argtypes = [vp, vp, i32, i32, i64, vp, vp]
    L.mmr_topk_merge_packed.restype = i32
    L.mmr_topk_merge_packed.argtypes = [vp, i32, i32, i32, f32, vp, vp, vp, vp]
    L.mmr_comm_unique_id.restype = i32
    L.mmr_comm_unique_id.argtypes = [vp]
    L.mmr_comm_init.restype = i32
    L.mmr_comm_init.argtypes = [i32, i32, vp, ctypes.POINTER(vp)]
    L.mmr_comm_destroy.restype = None
    L.mmr_comm_destroy.argtypes = [vp]
    L.mmr_allgather_topk.restype = i32
    L.mmr_allgather_topk.argtypes = [vp, vp, vp, i32, i32, vp, vp, vp]
    L.mmr_allgather_topk_packed.restype = i32
    L.mmr_allgather_topk_packed.argtypes = [vp, vp, i32, i32, vp, vp]
    L.mmr_tip_adapter_logits.restype = i32
    L.mmr_tip_adapter_logits.argtypes = [vp, vp, vp, vp, i32, i64, i32, i32, i32, f32, f32, vp, vp, vp]
    L.mmr_preprocess_image.restype = i32
    L.mmr_preprocess_image.argtypes = [vp, i32, i32, i32, i32, i32, vp, vp, i32, vp, vp, i32, f32, f32, f32, f32, f32,
                                       f32, vp, vp, i32, vp, vp]
    L.mmr_preprocess_batch.restype = i32
    L.mmr_preprocess_batch.argtypes = [vp, i32, i32, i32, f32, f32, f32, f32, f32, f32, vp, i32, vp]
    L.mmr_preprocess_batch_ex.restype = i32
    L.mmr_preprocess_batch_ex.argtypes = [vp, i32, i32, i32, i32, f32, f32, f32, f32, f32, f32, vp, i32, vp]
    L.mmr_prof_enable.restype = i32
    L.mmr_prof_enable.argtypes = [i32, i32]
    L.mmr_prof_read.restype = i32
    L.mmr_prof_read.argtypes = [i32, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_longlong),
                                ctypes.POINTER(ctypes.c_longlong)]
    if hasattr(L, "mmr_tower_create"):
        cfgp = ctypes.POINTER(TowerCfg)
        L.mmr_tower_weights_bytes.restype = sz
        L.mmr_tower_weights_bytes.argtypes = [cfgp]
        L.mmr_tower_param_span.restype = i32
        L.mmr_tower_param_span.argtypes = [cfgp, i32, i32, ctypes.POINTER(sz), ctypes.POINTER(sz)]
        L.mmr_tower_create.restype = i32
        L.mmr_tower_create.argtypes = [cfgp, vp, sz, ctypes.POINTER(vp)]
        L.mmr_tower_set_shared_chip.restype = i32
        L.mmr_tower_set_shared_chip.argtypes = [vp, i32]
        L.mmr_tower_destroy.restype = None
        L.mmr_tower_destroy.argtypes = [vp]
        L.mmr_tower_workspace_bytes.restype = sz
        L.mmr_tower_workspace_bytes.argtypes = [vp, i32]
        L.mmr_vit_encode_image.restype = i32
        L.mmr_vit_encode_image.argtypes = [vp, vp, i32, i32, vp, i32, i32, vp, sz, vp]
        L.mmr_text_encode.restype = i32
        L.mmr_text_encode.argtypes = [vp, vp, i32, vp, i32, i32, vp, sz, vp]
        L.mmr_tower_forward.restype = i32
        L.mmr_tower_forward.argtypes = [vp, vp, i32, i32, vp, i32, i32, i32, vp, vp, sz, vp]
        L.mmr_bert_workspace_bytes.restype = sz
        L.mmr_bert_workspace_bytes.argtypes = [vp, i32, i32]
        L.mmr_bert_forward.restype = i32
        L.mmr_bert_forward.argtypes = [vp, vp, i32, i32, vp, i32, i32, i32, vp, vp, sz, vp]
        L.mmr_bert_forward_masked.restype = i32
        L.mmr_bert_forward_masked.argtypes = [vp, vp, vp, vp, i32, i32, vp, i32, i32, i32, vp, vp, sz, vp]
        L.mmr_debug_attention_masked.restype = i32
        L.mmr_debug_attention_masked.argtypes = [vp, vp, i32, i32, i32, vp, vp]
        L.mmr_debug_gemm.restype = i32
        L.mmr_debug_gemm.argtypes = [i32, vp, vp, i32, i32, i32, vp, vp, vp]
        L.mmr_debug_layernorm.restype = i32
        L.mmr_debug_layernorm.argtypes = [vp, vp, vp, vp, i64, i32, f32, vp]
        L.mmr_debug_attention.restype = i32
        L.mmr_debug_attention.argtypes = [vp, vp, i32, i32, i32, i32, vp]
    _lib = L
    return L


def check(rc: int):
    if rc != 0:
        raise MMRError(rc, lib().mmr_last_error().decode())


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return MMR_F32
    if dt == torch.bfloat16:
        return MMR_BF16
    raise TypeError(f"libmmr_hip handles float32 and bfloat16 tensors, got {dt}")


def stream_ptr(device=None) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def ptr(t):
    return 0 if t is None else t.data_ptr()


PROF_CLASSES = {"gemm": 0, "attention": 1, "rowwise": 2, "scan": 3, "finalize": 4, "exact": 5}


def prof_enable(on: bool, max_launches: int = 65536):
    check(lib().mmr_prof_enable(int(on), int(max_launches)))


def prof_read():
    """-> {class: (total_ms, launches)} for the launches recorded since prof_enable(True)."""
    out = {}
    for name, cls in PROF_CLASSES.items():
        ms, n, dr = ctypes.c_double(), ctypes.c_longlong(), ctypes.c_longlong()
        check(lib().mmr_prof_read(cls, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(dr)))
        out[name] = (ms.value, n.value)
    out["dropped"] = dr.value
    return out
