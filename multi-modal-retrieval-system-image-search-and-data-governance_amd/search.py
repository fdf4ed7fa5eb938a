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


def gallery_norm_bound(gallery: torch.Tensor) -> torch.Tensor:
    """Largest row L2 norm of ``gallery`` as a 1-element fp32 DEVICE tensor (no host sync): what sizes the
    margin of the fast path's exactness certificate.  One HBM-bound pass."""
    if not gallery.is_cuda:
        raise RuntimeError("gallery must live on the GPU (there is no CPU path)")
    g = gallery.contiguous()
    out = torch.empty(1, dtype=torch.float32, device=g.device)
    L = _lib.lib()
    _lib.check(L.mmr_gallery_norm_bound(g.data_ptr(), _lib.dtype_code(g.dtype), g.shape[0], g.shape[1], out.data_ptr(),
                                        _lib.stream_ptr(g.device)))
    return out


def _local_topk(q, g, k, scale, norm_bound, want_dot64, want_status, workspace=None, norm_bound_dev=None, split=None):
    """norm_bound: caller's bound (None / <= 0: none); norm_bound_dev: measured device scalar (None: none).
    Neither given -> the C call measures the gallery itself.  split: (hi, lo, resid_bound) of an fp32 gallery
    (mmr_gallery_split_bf16: two bf16 arrays and the device scalar max_row |g - hi|) -> the tiered split search
    (same results)."""
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
    nb = 0.0 if norm_bound is None else float(norm_bound)
    if nb != nb or nb == float("inf"):
        raise ValueError("gallery_norm_bound must be finite")
    if split is not None and g.dtype == torch.float32:
        _lib.check(L.mmr_cosine_topk_split(q.data_ptr(), g.data_ptr(), split[0].data_ptr(), split[1].data_ptr(),
                                           _lib.ptr(split[2]), Q, N, E, k, float(scale), nb, _lib.ptr(norm_bound_dev), idx.data_ptr(), score.data_ptr(),
                                           _lib.ptr(dot64), _lib.ptr(status), workspace.data_ptr(), workspace.numel(),
                                           _lib.stream_ptr(dev)))
        return idx, score, dot64, status, workspace
    _lib.check(L.mmr_cosine_topk_ex(q.data_ptr(), g.data_ptr(), _lib.dtype_code(g.dtype), Q, N, E, k, float(scale),
                                    nb, _lib.ptr(norm_bound_dev), idx.data_ptr(), score.data_ptr(), _lib.ptr(dot64),
                                    _lib.ptr(status), workspace.data_ptr(), workspace.numel(), _lib.stream_ptr(dev)))
    return idx, score, dot64, status, workspace


def cosine_topk(queries: torch.Tensor, gallery: torch.Tensor, k: int = 10, scale: float = 1.0,
                gallery_norm_bound: Optional[float] = None, return_dot64: bool = False, return_status: bool = False):
    """Top-k gallery rows per query, like ``(scale * queries @ gallery.t()).topk(k, 1, True, True)``.

    Returns ``(values fp32 [Q,k], indices int64 [Q,k])`` -- torch.topk's order of results -- plus
    the exact fp64 dots and/or the per-query path status when asked.  Ranking is by
    (-dot, +row index) on fp64 dot products, so ties go to the lowest row id (torch's CPU tie
    order is unspecified) and indices are bit-reproducible against oracle/search_ref.c.
    Empty slots (k > N) hold index -1 / score -inf.

    ``gallery_norm_bound`` (an upper bound on any row's L2 norm) sizes the certificate of the fast path.
    Left at None it is MEASURED from the gallery in the same call (one extra streaming pass; a
    ``GalleryIndex`` measures once); a caller who passes a number promises it holds (include/mmr.h).
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


def merge_topk_packed(gathered: torch.Tensor, scale: float = 1.0):
    """Merge the gathered per-shard messages [parts,Q,k,2] int64 (mmr_topk_pack layout) -> (values, indices, dot64)."""
    gathered = gathered.contiguous()
    parts, Q, k, _ = gathered.shape
    dev = gathered.device
    idx = torch.empty(Q, k, dtype=torch.int64, device=dev)
    score = torch.empty(Q, k, dtype=torch.float32, device=dev)
    dot64 = torch.empty(Q, k, dtype=torch.float64, device=dev)
    L = _lib.lib()
    _lib.check(L.mmr_topk_merge_packed(gathered.data_ptr(), parts, Q, k, float(scale), idx.data_ptr(), score.data_ptr(),
                                       dot64.data_ptr(), _lib.stream_ptr(dev)))
    return score, idx, dot64


class GalleryIndex:
    """A device-resident embedding matrix [N,E] with a reusable search workspace.

    Stands where the reference keeps ``test_features`` (code/search_image.py:167-182) and
    scores it against reference vectors; rows are whatever ``encode_image`` produced -- normalised or
    not: the largest row norm is measured once here (device scalar, no host sync) and sizes the
    certificate's margin; a caller-supplied ``norm_bound`` can only widen it.

    The exactness certificate rests on that measured bound, so ``self.gallery`` is treated as FROZEN after
    construction: change rows through ``update_rows`` (writes them and re-measures) or call
    ``refresh_norm_bound()`` after writing into ``self.gallery`` yourself -- an in-place update that raises a
    row norm without a re-measure would understate the bound and could let the certificate pass wrongly.
    """

    def __init__(self, gallery: torch.Tensor, norm_bound: Optional[float] = None, presplit: Optional[bool] = None):
        """``presplit`` (fp32 galleries only; default: on from 4096 rows): keep the gallery's hi / lo bf16 split next to it
        (as many bytes again as the gallery): searches then scan the bf16 ``hi`` half alone first and fall back to the
        three-product scan over both halves only for queries that pass cannot certify -- identical results, about half the
        time of the per-call path on galleries whose top scores are not crowded."""
        if not gallery.is_cuda:
            raise RuntimeError("GalleryIndex needs a CUDA/HIP tensor")
        if gallery.dtype not in (torch.float32, torch.bfloat16):
            gallery = gallery.float()
        self.gallery = gallery.contiguous()
        self.norm_bound = None if norm_bound is None else float(norm_bound)
        self.norm_bound_dev = gallery_norm_bound(self.gallery)
        self._ws_lanes = {}        # lane -> search workspace: searches on different lanes may overlap on different streams
        self._split = None
        if presplit is None:
            presplit = self.gallery.shape[0] >= 4096
        self._presplit = bool(presplit) and self.gallery.dtype == torch.float32 and self.gallery.shape[0] > 0
        self._refresh_split()

    def _refresh_split(self) -> None:
        if not self._presplit:
            return
        g = self.gallery
        if self._split is None:
            self._split = (torch.empty(g.shape, dtype=torch.bfloat16, device=g.device),
                           torch.empty(g.shape, dtype=torch.bfloat16, device=g.device),
                           torch.zeros(1, dtype=torch.float32, device=g.device))       # max_row |g - hi|, same scalar for life
        L = _lib.lib()
        _lib.check(L.mmr_gallery_split_bf16(g.data_ptr(), g.shape[0], g.shape[1], self._split[0].data_ptr(),
                                            self._split[1].data_ptr(), self._split[2].data_ptr(), _lib.stream_ptr(g.device)))

    @property
    def num_rows(self) -> int:
        return self.gallery.shape[0]

    def refresh_norm_bound(self) -> None:
        """Re-measure the largest row norm into the SAME device scalar (one HBM-bound pass, no host sync), so searches
        already captured in a hipGraph see the new bound too."""
        g = self.gallery
        L = _lib.lib()
        _lib.check(L.mmr_gallery_norm_bound(g.data_ptr(), _lib.dtype_code(g.dtype), g.shape[0], g.shape[1],
                                            self.norm_bound_dev.data_ptr(), _lib.stream_ptr(g.device)))
        self._refresh_split()

    def update_rows(self, rows: torch.Tensor, values: torch.Tensor) -> None:
        """``gallery[rows] = values`` followed by a re-measure of the norm bound (keeps the certificate sound)."""
        self.gallery[rows.to(self.gallery.device)] = values.to(device=self.gallery.device, dtype=self.gallery.dtype)
        self.refresh_norm_bound()

    def search(self, queries: torch.Tensor, k: int = 10, scale: float = 1.0, return_dot64: bool = False,
               return_status: bool = False, lane: int = 0):
        """``lane``: which of the index's workspaces the call uses; searches on different lanes may be in flight at once on
        different HIP streams (they only read the gallery), searches on one lane must be stream-ordered."""
        q2, squeezed = _as_2d(queries)
        q = q2.to(device=self.gallery.device, dtype=self.gallery.dtype).contiguous()
        idx, score, dot64, status, self._ws_lanes[lane] = _local_topk(q, self.gallery, int(k), scale, self.norm_bound,
                                                                      return_dot64, return_status, self._ws_lanes.get(lane),
                                                                      self.norm_bound_dev, self._split)
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

    def search_packed(self, queries2d: torch.Tensor, k: int, scale: float, row_offset: int, lane: int = 0) -> torch.Tensor:
        """This shard's all-gather message for [Q,E] queries: [Q,k,2] int64 = (global row id or -1, fp64 dot bits),
        written by one kernel straight from the search outputs (mmr_topk_pack)."""
        q = queries2d.to(device=self.gallery.device, dtype=self.gallery.dtype).contiguous()
        idx, _, dot64, _, self._ws_lanes[lane] = _local_topk(q, self.gallery, int(k), scale, self.norm_bound, True, False,
                                                             self._ws_lanes.get(lane), self.norm_bound_dev, self._split)
        packed = torch.empty(q.shape[0], int(k), 2, dtype=torch.int64, device=q.device)
        L = _lib.lib()
        _lib.check(L.mmr_topk_pack(idx.data_ptr(), dot64.data_ptr(), q.shape[0], int(k), int(row_offset), packed.data_ptr(),
                                   _lib.stream_ptr(q.device)))
        return packed

    def scores(self, ref_feature: torch.Tensor, scale: float = 100.0) -> torch.Tensor:
        """``get_similarity``'s first line for this gallery (reference code/search_image.py:107)."""
        return similarity(self.gallery, ref_feature, scale)


class _PendingSearch:
    """One query batch of a ShardedGalleryIndex whose all-gather is in flight."""

    def __init__(self, owner, gathered, work, scale, squeezed):
        self.owner, self.gathered, self.work, self.scale, self.squeezed = owner, gathered, work, scale, squeezed

    def result(self, return_dot64: bool = False):
        """(values fp32 [Q,k], global int64 ids [Q,k] [, exact fp64 dots]); stream-ordered behind the collective."""
        if self.work is not None:
            self.work.wait()          # nccl: the CURRENT STREAM waits for the collective, the host does not
            self.work = None
        if self.owner._merge is None:                                    # HIP path: rank the gathered messages as they are
            out = merge_topk_packed(self.gathered, self.scale)
        else:
            idx_parts = self.gathered[..., 0].contiguous()
            dot_parts = self.gathered[..., 1].contiguous().view(torch.float64)
            out = self.owner._merge(idx_parts, dot_parts, self.scale)   # (score, idx, dot64)
        if self.squeezed:
            out = tuple(t[0] for t in out)
        return out if return_dot64 else out[:2]


class ShardedGalleryIndex:
    """Row-sharded gallery: rank r holds rows [offset_r, offset_r + n_r) and searches them locally;
    ONE all-gather (RCCL over xGMI with the nccl backend) of the packed per-shard top-k, then an
    exact merge.  Global top-k is a subset of the union of local top-k lists, so the merged result
    equals the single-GPU result bit for bit.

    ``search`` is one batch, start to finish.  ``search_async`` returns as soon as the batch's collective
    is issued (``all_gather_into_tensor(async_op=True)``: it runs on the backend's own stream), so the next
    batch's gallery scan -- or the caller's next encode -- overlaps it; ``.result()`` makes the current
    stream wait for the collective and merges.  ``search_pipelined`` does that over a list of batches.

    ``local_search`` / ``merge`` exist so the collective and offset logic can be exercised on a
    CPU ``gloo`` group in tests; the defaults are the HIP kernels.
    """

    def __init__(self, local_gallery: torch.Tensor, group=None, norm_bound: Optional[float] = None,
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
        self._merge = merge if merge is not None else (None if local_search is None else merge_topk)
        n_local = torch.tensor([local_gallery.shape[0]], dtype=torch.int64, device=local_gallery.device)
        if self.use_dist:
            counts = torch.empty(self.world, dtype=torch.int64, device=local_gallery.device)
            dist.all_gather_into_tensor(counts, n_local, group=group)
        else:
            counts = n_local
        self.counts = counts.cpu()
        self.offsets = torch.cumsum(self.counts, 0) - self.counts
        self.offset = int(self.offsets[self.rank])
        self.total_rows = int(self.counts.sum())

    def local_topk(self, queries: torch.Tensor, k: int = 10, scale: float = 1.0):
        """This rank's (global int64 ids [Q,k], fp64 dots [Q,k]) -- the payload of the collective."""
        q2, _ = _as_2d(queries)
        if self._local_search is None:
            _, lidx, ldot = self._index.search(q2, k, scale, return_dot64=True)
        else:
            lidx, ldot = self._local_search(q2, self.local, k)
        gidx = torch.where(lidx >= 0, lidx.to(torch.int64) + self.offset, lidx.to(torch.int64))
        return gidx, ldot

    def search_async(self, queries: torch.Tensor, k: int = 10, scale: float = 1.0, lane: int = 0) -> _PendingSearch:
        squeezed = queries.dim() == 1
        if self._local_search is None:
            q2, _ = _as_2d(queries)
            packed = self._index.search_packed(q2, k, scale, self.offset, lane)     # one kernel after the search
        else:
            gidx, ldot = self.local_topk(queries, k, scale)
            # one packed message per rank: [Q,k,2] int64 = (global id, fp64 dot bits)
            packed = torch.stack([gidx, ldot.view(torch.int64)], dim=-1).contiguous()
        if self.use_dist:
            # output = the ranks' messages concatenated along dim 0 (the layout both nccl and gloo accept)
            flat = torch.empty((self.world * packed.shape[0],) + tuple(packed.shape[1:]), dtype=packed.dtype,
                               device=packed.device)
            work = self.dist.all_gather_into_tensor(flat, packed, group=self.group, async_op=True)   # THE collective
            gathered = flat.view((self.world,) + tuple(packed.shape))
        else:
            gathered, work = packed.unsqueeze(0), None
        return _PendingSearch(self, gathered, work, scale, squeezed)

    def search(self, queries: torch.Tensor, k: int = 10, scale: float = 1.0):
        """Every rank passes the same queries; every rank gets the same global (values, int64 ids)."""
        return self.search_async(queries, k, scale).result()

    def search_pipelined(self, batches, k: int = 10, scale: float = 1.0):
        """[(values, ids)] for a sequence of query batches: batch i's all-gather overlaps batch i+1's scan."""
        out, pending = [], None
        for qb in batches:
            nxt = self.search_async(qb, k, scale)
            if pending is not None:
                out.append(pending.result())
            pending = nxt
        if pending is not None:
            out.append(pending.result())
        return out


def verify_exact_topk(index, gallery: torch.Tensor, queries: torch.Tensor, k: int = 10):
    """Proof that ``index`` (a ``GalleryIndex`` or a ``ShardedGalleryIndex`` over this rank's ``gallery`` rows) returns
    the exact global top-k for ``queries`` -- four checks that together leave no room for a wrong list:

      (a) every returned global id is re-scored in fp64 by the ONE rank that owns the row and must reproduce the merged
          dot product (sum over ranks of the owners' re-scores == merged dot64, owner count == 1);
      (b) the list is in (-dot, +id) order (so ids are unique and ties go to the lowest id);
      (c) exactly k-1 rows precede the merged k-th entry: every rank counts the entries of its local top-(k+1) list
          that rank before it, the counts are summed (all-reduce) and must equal k-1 -- a row that beats the merged
          k-th but is missing from the list (dropped by the merge, or outside a shard's candidates) breaks the count;
      (d) all ranks hold the same answer (MIN / MAX all-reduce of position-weighted checksums).

    Collectives run on the index's process group (gloo on CPU in tests, RCCL on the GPUs); with no group it checks the
    single-shard result the same way.  Returns ``("ok" | "FAILED: ...", detail dict)``.  Diagnostic: outside any timed
    region (bench.py calls it after the clock stops).
    """
    sharded = isinstance(index, ShardedGalleryIndex)
    use_dist = sharded and index.use_dist
    dist = index.dist if sharded else None
    group = index.group if sharded else None
    q2, _ = _as_2d(queries)
    qd = q2.double()
    if sharded:
        _, gidx, d64 = index.search_async(q2, k, 1.0).result(return_dot64=True)
        offset = index.offset
        lidx, ld = index.local_topk(q2, k + 1, 1.0)
    else:
        _, gidx, d64 = index.search(q2, k, 1.0, return_dot64=True)
        offset = 0
        _, lidx, ld = index.search(q2, k + 1, 1.0, return_dot64=True)
    n_local = gallery.shape[0]
    Q = q2.shape[0]
    problems = []
    # (a) owner re-score
    mine = (gidx >= offset) & (gidx < offset + n_local)
    rows = gallery[(gidx - offset).clamp(0, max(n_local - 1, 0)).reshape(-1)].double().reshape(Q, k, -1)
    re = (rows * qd.unsqueeze(1)).sum(-1)
    re = torch.where(mine, re, torch.zeros_like(re))
    owners = mine.to(torch.int32)
    if use_dist:
        dist.all_reduce(re, group=group)
        dist.all_reduce(owners, group=group)
    if not bool((owners == 1).all()):
        problems.append("an id is owned by no rank or by several")
    err = float((re - d64).abs().max()) if re.numel() else 0.0
    if not err < 1e-12:
        problems.append(f"re-scored dot differs from the merged dot64 by {err:.3e}")
    # (b) order and uniqueness
    if k > 1 and not bool(((d64[:, :-1] > d64[:, 1:]) | ((d64[:, :-1] == d64[:, 1:]) & (gidx[:, :-1] < gidx[:, 1:]))).all()):
        problems.append("merged list is not in (-dot, +id) order")
    # (c) exactly k-1 candidates rank before the merged k-th.  Each rank counts the entries of its local top-(k+1) list
    # that precede the merged k-th entry in (-dot, +id) order; the counts are summed over ranks.  If any row of any
    # shard beat the merged k-th without being in the merged list, its shard would count it (or, were it outside that
    # shard's top-(k+1), count k+1 better ones): the sum equals k-1 only for the exact global top-k.
    lval = lidx >= 0
    kd, ki = d64[:, k - 1:k], gidx[:, k - 1:k]
    before = lval & ((ld > kd) | ((ld == kd) & (lidx < ki)))
    full = gidx[:, k - 1] >= 0                                  # fewer than k rows in total: every candidate is in the list
    cnt = torch.where(full, before.sum(1), lval.sum(1)).to(torch.int64)
    if use_dist:
        dist.all_reduce(cnt, group=group)
    want = torch.where(full, torch.full_like(cnt, k - 1), (gidx >= 0).sum(1).to(torch.int64))
    if not bool((cnt == want).all()):
        problems.append("a candidate that beats the merged k-th is missing from the merged list")
    # (d) all ranks hold the same answer (position-weighted checksums: a permuted list differs)
    if use_dist:
        wq = torch.arange(1, Q + 1, dtype=torch.float64, device=gidx.device).unsqueeze(1)
        wk = torch.arange(1, k + 1, dtype=torch.float64, device=gidx.device).unsqueeze(0)
        chk = torch.stack([(gidx.double() * wq * wk).sum(), (d64 * wq * wk).sum()])
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
        if not bool((lo == hi).all()):
            problems.append("ranks disagree on the merged result")
    detail = {"rescored_max_abs_err": err, "queries": Q, "k": k, "ranks": index.world if sharded else 1,
              "checks": "owner re-score == merged dot64; (-dot,+id) order; exactly k-1 candidates precede the merged k-th; ranks agree"}
    return ("ok" if not problems else "FAILED: " + "; ".join(problems)), detail
