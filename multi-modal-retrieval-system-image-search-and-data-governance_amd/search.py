"""Similarity, L2-normalise and top-k over an embedding gallery -- the "search" half of the hot path.

Host-side mirror of the tensor expressions the reference writes inline:
    similarity = 100. * features.cuda() @ ref_feature.t()      reference code/search_image.py:107
    image_features /= image_features.norm(dim=-1, keepdim=True) reference code/search_image.py:133,157
    output.topk(topk, 1, True, True)                            reference code/utils.py:17
All arithmetic runs in csrc/search.hip through the C ABI (include/mmr.h); torch only owns the
device memory, the stream and (for the sharded index) the RCCL process group.
"""
from typing import Callable, Optional, Tuple

import torch

from . import _lib


def _as_2d(x: torch.Tensor) -> Tuple[torch.Tensor, bool]:
    if x.dim() == 1:
        return x.unsqueeze(0), True
    if x.dim() != 2:
        raise ValueError(f"expected a 1-D or 2-D tensor, got shape {tuple(x.shape)}")
    return x, False


def _prep_pair(q: torch.Tensor, gallery: torch.Tensor):
    if not gallery.is_cuda:
        raise RuntimeError("gallery must live on the GPU (there is no CPU path)")
    if gallery.dtype not in (torch.float32, torch.bfloat16):
        gallery = gallery.to(torch.float32)
    q = q.to(device=gallery.device, dtype=gallery.dtype)
    return q.contiguous(), gallery.contiguous()


def l2_normalize(x: torch.Tensor, inplace: bool = False) -> torch.Tensor:
    """Row-wise ``x / x.norm(dim=-1, keepdim=True)`` (no epsilon, as the reference)."""
    x2, squeezed = _as_2d(x)
    out = x2 if (inplace and x2.is_contiguous()) else x2.contiguous().clone()
    L = _lib.lib()
    _lib.check(L.mmr_l2norm_rows(out.data_ptr(), _lib.dtype_code(out.dtype), out.shape[0], out.shape[1],
                                 _lib.stream_ptr(out.device)))
    if inplace and out is not x2:
        x2.copy_(out)
        out = x2
    return out.squeeze(0) if squeezed else out


def similarity(features: torch.Tensor, ref_feature: torch.Tensor, scale: float = 100.0) -> torch.Tensor:
    """``scale * features @ ref_feature.t()`` -> fp32 [N] (1-D ref) or [N,Q] (2-D ref).

    Same shape convention as reference code/search_image.py:107; the query is NOT normalised
    here (the reference also scores with an un-normalised mean vector, search_image.py:315,387).
    Materialises the score matrix, so it is meant for the reference's small galleries;
    use ``cosine_topk`` for large ones.
    """
    r2, squeezed = _as_2d(ref_feature)
    if features.is_cuda and r2.dtype != features.dtype:
        # mixed precision (e.g. bf16 cache rows scored against an fp32 mean vector): this is the
        # small-gallery API, so widen both operands to fp32 rather than round the reference vector
        features = features.to(torch.float32)
        r2 = r2.to(torch.float32)
    q, g = _prep_pair(r2, features)
    Q, E = q.shape
    N = g.shape[0]
    out = torch.empty(Q, N, dtype=torch.float32, device=g.device)
    L = _lib.lib()
    _lib.check(L.mmr_similarity(q.data_ptr(), g.data_ptr(), _lib.dtype_code(g.dtype), Q, N, E, float(scale),
                                out.data_ptr(), _lib.stream_ptr(g.device)))
    return out[0] if squeezed else out.t()


def tip_adapter_logits(features: torch.Tensor, clip_weights: torch.Tensor, cache_keys: torch.Tensor,
                       cache_values: torch.Tensor, alpha: float, beta: float, return_clip_logits: bool = False):
    """Fused Tip-Adapter scoring (reference code/main_custom.py:111,124-127):

        clip_logits  = 100. * features @ clip_weights                      # clip_weights [E,C]
        affinity     = features @ cache_keys                               # cache_keys   [E,S]
        cache_logits = ((-1) * (beta - beta * affinity)).exp() @ cache_values * 10     # values [S,C]
        tip_logits   = clip_logits + cache_logits * alpha

    Returns fp32 ``tip_logits [N,C]`` (and ``clip_logits``).  The [N,S] affinity matrix is never written.
    """
    if not features.is_cuda:
        raise RuntimeError("features must live on the GPU (there is no CPU path)")
    dt = features.dtype if features.dtype in (torch.float32, torch.bfloat16) else torch.float32
    if clip_weights.dtype != dt or cache_keys.dtype != dt:
        dt = torch.float32
    dev = features.device
    f = features.to(dt).contiguous()
    wt = clip_weights.to(dev, dt).t().contiguous()                 # [C,E]
    kt = cache_keys.to(dev, dt).t().contiguous()                   # [S,E]
    v = cache_values.to(dev, torch.float32).contiguous()           # [S,C]
    N, E = f.shape
    C, S = wt.shape[0], kt.shape[0]
    if wt.shape[1] != E or kt.shape[1] != E or v.shape != (S, C):
        raise ValueError(f"shape mismatch: features {tuple(f.shape)}, clip_weights {tuple(clip_weights.shape)}, "
                         f"cache_keys {tuple(cache_keys.shape)}, cache_values {tuple(cache_values.shape)}")
    tip = torch.empty(N, C, dtype=torch.float32, device=dev)
    clip = torch.empty(N, C, dtype=torch.float32, device=dev) if return_clip_logits else None
    L = _lib.lib()
    _lib.check(L.mmr_tip_adapter_logits(f.data_ptr(), wt.data_ptr(), kt.data_ptr(), v.data_ptr(), _lib.dtype_code(dt),
                                        N, E, C, S, float(alpha), float(beta), tip.data_ptr(), _lib.ptr(clip),
                                        _lib.stream_ptr(dev)))
    return (tip, clip) if return_clip_logits else tip


def _local_topk(q, g, k, scale, norm_bound, want_dot64, want_status, workspace=None):
    Q, E = q.shape
    N = g.shape[0]
    dev = g.device
    L = _lib.lib()
    need = L.mmr_search_workspace_bytes(N, E, Q, k)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(max(need, 256), dtype=torch.uint8, device=dev)
    idx = torch.empty(Q, k, dtype=torch.int32, device=dev)
    score = torch.empty(Q, k, dtype=torch.float32, device=dev)
    dot64 = torch.empty(Q, k, dtype=torch.float64, device=dev) if want_dot64 else None
    status = torch.empty(Q, dtype=torch.int32, device=dev) if want_status else None
    _lib.check(L.mmr_cosine_topk(q.data_ptr(), g.data_ptr(), _lib.dtype_code(g.dtype), Q, N, E, k, float(scale),
                                 float(norm_bound), idx.data_ptr(), score.data_ptr(), _lib.ptr(dot64),
                                 _lib.ptr(status), workspace.data_ptr(), workspace.numel(), _lib.stream_ptr(dev)))
    return idx, score, dot64, status, workspace


def cosine_topk(queries: torch.Tensor, gallery: torch.Tensor, k: int = 10, scale: float = 1.0,
                gallery_norm_bound: float = 1.0, return_dot64: bool = False, return_status: bool = False):
    """Top-k gallery rows per query, like ``(scale * queries @ gallery.t()).topk(k, 1, True, True)``.

    Returns ``(values fp32 [Q,k], indices int64 [Q,k])`` -- torch.topk's order of results -- plus
    the exact fp64 dots and/or the per-query path status when asked.  Ranking is by
    (-dot, +row index) on fp64 dot products, so ties go to the lowest row id (torch's CPU tie
    order is unspecified) and indices are bit-reproducible against oracle/search_ref.c.
    Empty slots (k > N) hold index -1 / score -inf.
    """
    q2, squeezed = _as_2d(queries)
    q, g = _prep_pair(q2, gallery)
    if q.shape[1] != g.shape[1]:
        raise ValueError(f"query dim {q.shape[1]} != gallery dim {g.shape[1]}")
    idx, score, dot64, status, _ = _local_topk(q, g, int(k), scale, gallery_norm_bound, return_dot64, return_status)
    idx = idx.to(torch.int64)
    if squeezed:
        idx, score = idx[0], score[0]
        dot64 = dot64[0] if dot64 is not None else None
    out = (score, idx)
    if return_dot64:
        out = out + (dot64,)
    if return_status:
        out = out + (status,)
    return out


def merge_topk(idx_parts: torch.Tensor, dot_parts: torch.Tensor, scale: float = 1.0):
    """Merge per-shard lists [parts,Q,k] (global int64 ids, fp64 dots) -> (values, indices, dot64)."""
    idx_parts = idx_parts.contiguous()
    dot_parts = dot_parts.contiguous()
    parts, Q, k = idx_parts.shape
    dev = idx_parts.device
    idx = torch.empty(Q, k, dtype=torch.int64, device=dev)
    score = torch.empty(Q, k, dtype=torch.float32, device=dev)
    dot64 = torch.empty(Q, k, dtype=torch.float64, device=dev)
    L = _lib.lib()
    _lib.check(L.mmr_topk_merge(idx_parts.data_ptr(), dot_parts.data_ptr(), parts, Q, k, float(scale),
                                idx.data_ptr(), score.data_ptr(), dot64.data_ptr(), _lib.stream_ptr(dev)))
    return score, idx, dot64


class GalleryIndex:
    """A device-resident embedding matrix [N,E] with a reusable search workspace.

    Stands where the reference keeps ``test_features`` (code/search_image.py:167-182) and
    scores it against reference vectors; rows are whatever ``encode_image`` produced.
    """

    def __init__(self, gallery: torch.Tensor, norm_bound: float = 1.0):
        if not gallery.is_cuda:
            raise RuntimeError("GalleryIndex needs a CUDA/HIP tensor")
        if gallery.dtype not in (torch.float32, torch.bfloat16):
            gallery = gallery.float()
        self.gallery = gallery.contiguous()
        self.norm_bound = float(norm_bound)
        self._ws = None

    @property
    def num_rows(self) -> int:
        return self.gallery.shape[0]

    def search(self, queries: torch.Tensor, k: int = 10, scale: float = 1.0, return_dot64: bool = False):
        q2, squeezed = _as_2d(queries)
        q = q2.to(device=self.gallery.device, dtype=self.gallery.dtype).contiguous()
        idx, score, dot64, _, self._ws = _local_topk(q, self.gallery, int(k), scale, self.norm_bound,
                                                     return_dot64, False, self._ws)
        idx = idx.to(torch.int64)
        if squeezed:
            idx, score = idx[0], score[0]
        return (score, idx, dot64) if return_dot64 else (score, idx)

    def scores(self, ref_feature: torch.Tensor, scale: float = 100.0) -> torch.Tensor:
        """``get_similarity``'s first line for this gallery (reference code/search_image.py:107)."""
        return similarity(self.gallery, ref_feature, scale)


class ShardedGalleryIndex:
    """Row-sharded gallery: rank r holds rows [offset_r, offset_r + n_r) and searches them locally;
    ONE all-gather (RCCL over xGMI with the nccl backend) of the packed per-shard top-k, then an
    exact merge.  Global top-k is a subset of the union of local top-k lists, so the merged result
    equals the single-GPU result bit for bit.

    ``local_search`` / ``merge`` exist so the collective and offset logic can be exercised on a
    CPU ``gloo`` group in tests; the defaults are the HIP kernels.
    """

    def __init__(self, local_gallery: torch.Tensor, group=None, norm_bound: float = 1.0,
                 local_search: Optional[Callable] = None, merge: Optional[Callable] = None):
        import torch.distributed as dist

        self.dist = dist
        self.group = group
        self.use_dist = dist.is_available() and dist.is_initialized()   # collectives run even for a 1-rank group
        self.world = dist.get_world_size(group) if self.use_dist else 1
        self.rank = dist.get_rank(group) if self.use_dist else 0
        self.local = local_gallery
        self._index = GalleryIndex(local_gallery, norm_bound) if local_search is None else None
        self._local_search = local_search
        self._merge = merge or merge_topk
        n_local = torch.tensor([local_gallery.shape[0]], dtype=torch.int64, device=local_gallery.device)
        if self.use_dist:
            counts = [torch.zeros_like(n_local) for _ in range(self.world)]
            dist.all_gather(counts, n_local, group=group)
            counts = torch.cat(counts)
        else:
            counts = n_local
        self.counts = counts.cpu()
        self.offsets = torch.cumsum(self.counts, 0) - self.counts
        self.offset = int(self.offsets[self.rank])
        self.total_rows = int(self.counts.sum())

    def search(self, queries: torch.Tensor, k: int = 10, scale: float = 1.0):
        """Every rank passes the same queries; every rank gets the same global (values, int64 ids)."""
        if self._local_search is None:
            _, lidx, ldot = self._index.search(queries, k, scale, return_dot64=True)
        else:
            lidx, ldot = self._local_search(queries, self.local, k)
        gidx = torch.where(lidx >= 0, lidx.to(torch.int64) + self.offset, lidx.to(torch.int64))
        # one packed message per rank: [Q,k,2] int64 = (global id, fp64 dot bits)
        packed = torch.stack([gidx, ldot.view(torch.int64)], dim=-1).contiguous()
        if self.use_dist:
            parts = [torch.empty_like(packed) for _ in range(self.world)]
            self.dist.all_gather(parts, packed, group=self.group)     # the one collective of the search path
            gathered = torch.stack(parts)
        else:
            gathered = packed.unsqueeze(0)
        idx_parts = gathered[..., 0].contiguous()
        dot_parts = gathered[..., 1].contiguous().view(torch.float64)
        score, idx, _ = self._merge(idx_parts, dot_parts, scale)
        return score, idx
