"""Gallery build: drive the image tower over a dataset in batches and keep / persist the [N,E] matrix.

Counterpart of the reference's per-image loops:
    build_cache          reference code/search_image.py:142-165  (batch-1 loop, two PCIe hops per image,
                                                                  pickle of {relpath: float32[512]})
    pre_load_features    reference code/utils.py:135-157         (DataLoader batches -> {split}_f.pt / _l.pt)
    build_cache_model    reference code/utils.py:99-132          (keys_{shots}shots.pt / values_...)
Here the loop body is one batched ``encode_image`` per <= batch_size images; features stay on the
GPU (bf16 by default, the search kernel's gallery dtype) and are only copied to the host when a
cache file in the reference's format is asked for.
"""
import contextlib
import os
import pickle
from typing import Callable, Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch


def _images_of(batch):
    """DataLoader-style batches are (images, labels[, paths]); bare tensors are accepted too."""
    if isinstance(batch, torch.Tensor):
        return batch, None
    if isinstance(batch, (tuple, list)) and len(batch) >= 1 and isinstance(batch[0], torch.Tensor):
        return batch[0], (batch[1] if len(batch) > 1 else None)
    raise TypeError(f"cannot find an image tensor in a batch of type {type(batch)}")


class _Lanes:
    """``n`` HIP streams for work items that are independent of each other (gallery batches): item i runs on stream i % n
    with the model / index lane of the same number.  Entering the context makes the lane streams wait for what the caller's
    stream holds; leaving it makes the caller's stream wait for them.  n = 1 is the caller's stream itself."""

    def __init__(self, device, n: int, model=None):
        self.n = max(1, int(n))
        self.main = torch.cuda.current_stream(device)
        self.streams = [self.main] if self.n == 1 else [torch.cuda.Stream(device) for _ in range(self.n)]
        self._hint = model.shared_chip() if (model is not None and self.n > 1) else None      # tile policy for a shared chip

    def __enter__(self):
        if self.n > 1:
            for st in self.streams:
                st.wait_stream(self.main)
        if self._hint is not None:
            self._hint.__enter__()
        return self

    def __exit__(self, *exc):
        if self._hint is not None:
            self._hint.__exit__(*exc)
        if self.n > 1:
            for st in self.streams:
                self.main.wait_stream(st)
        return False

    @contextlib.contextmanager
    def run(self, i: int, *inputs: torch.Tensor):
        """Stream context of work item i; yields its lane number.  The lane's stream first waits for what the caller's
        stream holds NOW (a batch a generator produced there a moment ago), and ``inputs`` are marked as in use on it so the
        caching allocator does not hand their memory out while the lane still reads them."""
        k = i % self.n
        st = self.streams[k]
        if self.n > 1:
            st.wait_stream(self.main)
            for t in inputs:
                if isinstance(t, torch.Tensor) and t.is_cuda:
                    t.record_stream(st)
        with torch.cuda.stream(st):
            yield k


@torch.no_grad()
def encode_gallery(model, batches: Iterable, normalize: bool = True, out_dtype: torch.dtype = torch.bfloat16,
                   return_labels: bool = False, lanes: int = 2):
    """Encode every batch of ``batches`` -> device tensor [N,E] (rows in arrival order).

    ``model`` is a ``mmr_amd.CLIP``; each batch is a float tensor [b,3,S,S] (or a DataLoader tuple whose
    first item is one).  ``normalize`` fuses the reference's ``f /= f.norm(dim=-1, keepdim=True)``
    into the encoder's last kernel.  Host batches are uploaded with non-blocking copies so the upload
    of batch i+1 overlaps the encode of batch i when the loader yields pinned memory.

    ``lanes`` batches are in flight at once, each on its own HIP stream and model workspace (``encode_image(lane=)``):
    a forward at batch 256 is mostly GEMM launches of 150-200 tiles on 256 CUs followed by bandwidth-bound row kernels, and
    a second, independent forward fills what the first leaves idle -- measured +9.7 % images/s on ViT-B/32 at batch 256
    (2.97 -> 2.70 ms per forward; a third lane adds nothing), bit-identical rows.  ``lanes=1`` is the plain loop.
    """
    prev_dtype = model.dtype
    model.to(out_dtype)
    feats: List[torch.Tensor] = []
    labels: List[torch.Tensor] = []
    try:
        with _Lanes(model.device, lanes, model) as L:
            i = 0
            for batch in batches:
                images, lab = _images_of(batch)
                if images.numel() == 0:
                    continue
                with L.run(i, images) as lane:
                    feats.append(model.encode_image(images.to(model.device, non_blocking=True), normalize=normalize, lane=lane))
                if lab is not None:
                    labels.append(torch.as_tensor(lab))
                i += 1
    finally:
        model.to(prev_dtype)
    if lanes > 1:
        main = torch.cuda.current_stream(model.device)
        for f in feats:                        # allocated on a lane's stream, read by the concatenation on the caller's
            f.record_stream(main)
    out = torch.cat(feats) if feats else torch.empty(0, model.cfg.embed_dim, dtype=out_dtype, device=model.device)
    if return_labels:
        return out, (torch.cat(labels) if labels else None)
    return out


def _lowest_stream_priority() -> int:
    """Numerically largest (= lowest) stream priority of the current device; 0 when the query is not available."""
    pr = os.environ.get("MMR_BUILD_SIDE_PRIORITY")
    if pr is not None:
        return int(pr)
    try:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        least, greatest = ctypes.c_int(0), ctypes.c_int(0)
        if hip.hipDeviceGetStreamPriorityRange(ctypes.byref(least), ctypes.byref(greatest)) == 0:
            return int(least.value)
    except Exception:
        pass
    return 0


@torch.no_grad()
def build_gallery_overlapped(model, raw_batches: Iterable[torch.Tensor], total: Optional[int] = None,
                             out_dtype: torch.dtype = torch.bfloat16, overlap: bool = False,
                             gallery: Optional[torch.Tensor] = None, lanes: int = 2) -> torch.Tensor:
    """The reference's gallery loop ``preprocess -> encode_image -> normalise -> keep the row``
    (code/search_image.py:153-158) over batches of raw uint8 images [b,H,W,3] that already sit on the GPU: Pillow-exact
    bicubic resize + crop + normalise (csrc/preprocess.hip), the towers, and the L2-normalised rows written directly into a
    preallocated gallery [N,E] (``normalize=1`` fused into the encoder's last kernel; no torch.cat, no host round trip).

    Schedules, bit-identical results (tested):
      * ``lanes=2`` (default): two batches in flight, each preprocess -> encode on its own HIP stream with its own pixel
        buffer and model workspace (``encode_image(lane=)``); the second batch fills the CUs the first one's 150-200-tile
        GEMM launches and row kernels leave idle.  ``lanes=1``: everything back to back on the caller's stream (measured on
        MI355X, ViT-B/32, 256 VGA images per batch: 0.94 of the one-lane encode-only rate -- the preprocess costs 0.15 ms of
        the 3.1 ms step).
      * ``overlap=True`` (one lane): batch i+1 is preprocessed on a low-priority side stream while batch i is encoded.
        Measured SLOWER (0.77 of encode-only, whatever the stream priority): the encoder's GEMMs are one 160 KiB-LDS
        workgroup per CU, and a CU that holds preprocess workgroups cannot take one, so the persistent tile schedules start
        ragged.  Kept because it is the schedule to use with an encoder that leaves CUs free (small batches).
    All batches must share one image size (``UniformBatchPreprocessor``).  Returns the gallery (rows in arrival order)."""
    from .preprocess import UniformBatchPreprocessor

    it = iter(raw_batches)
    first = next(it, None)
    dev = model.device
    E = model.cfg.embed_dim
    if first is None:
        return torch.empty(0, E, dtype=out_dtype, device=dev) if gallery is None else gallery[:0]
    B, H, W = int(first.shape[0]), int(first.shape[1]), int(first.shape[2])
    B = max(B, 1)
    if gallery is None:
        if total is None:
            raise ValueError("pass total= (rows to allocate) or a preallocated gallery")
        gallery = torch.empty(int(total), E, dtype=out_dtype, device=dev)
    prev_dtype = model.dtype
    model.to(gallery.dtype)
    S = model.input_resolution
    lanes = 1 if overlap else max(1, int(lanes))
    pre = UniformBatchPreprocessor(B, H, W, S, out_dtype=torch.bfloat16, device=dev, slots=max(2, lanes))
    if lanes > 1:
        row = 0
        try:
            with _Lanes(dev, lanes, model) as L:
                cur, i = first, 0
                while cur is not None:
                    b = int(cur.shape[0])
                    if row + b > gallery.shape[0]:
                        raise ValueError(f"gallery of {gallery.shape[0]} rows is too small")
                    with L.run(i, cur) as lane:
                        model.encode_image(pre(cur, lane), normalize=True, out=gallery[row:row + b], lane=lane)
                    row += b
                    cur, i = next(it, None), i + 1
        finally:
            model.to(prev_dtype)
        return gallery[:row]
    main = torch.cuda.current_stream(dev)
    # the preprocess stream runs at the LOWEST priority the device offers: its short workgroups (8 image rows each) fill
    # CUs the encoder's launches leave idle, and the encoder's one-workgroup-per-CU GEMMs are dispatched ahead of them
    side = torch.cuda.Stream(dev, priority=_lowest_stream_priority()) if overlap else main
    ready = [torch.cuda.Event(), torch.cuda.Event()]          # slot's pixels written (side -> main)
    row = 0
    try:
        def launch_pre(raw, slot):
            # Called BEFORE the current batch's encode is enqueued: waiting for everything the main stream holds at this
            # point means (a) the raw images, if the caller produced them on the main stream, are complete and (b) the
            # encode two batches back -- the last reader of this slot's pixel buffer -- has finished; the current batch's
            # encode is enqueued afterwards and runs concurrently with this preprocess.
            if overlap:
                side.wait_stream(main)
                raw.record_stream(side)
            with torch.cuda.stream(side):
                px = pre(raw, slot)
                ready[slot].record(side)
            return px

        cur, i = first, 0
        px = launch_pre(cur, 0)
        while cur is not None:
            nxt = next(it, None)
            px_next = launch_pre(nxt, (i + 1) & 1) if nxt is not None else None       # flies under this batch's encode
            if overlap:
                main.wait_event(ready[i & 1])
            b = int(cur.shape[0])
            if row + b > gallery.shape[0]:
                raise ValueError(f"gallery of {gallery.shape[0]} rows is too small")
            model.encode_image(px, normalize=True, out=gallery[row:row + b])
            row += b
            cur, px, i = nxt, px_next, i + 1
    finally:
        model.to(prev_dtype)
    return gallery[:row]


def batched(items: Sequence, load: Callable, batch_size: int = 256):
    """Yield stacked [b,3,S,S] tensors from ``load(item)`` results (one preprocessed image each)."""
    buf = []
    for it in items:
        buf.append(load(it))
        if len(buf) == batch_size:
            yield torch.stack(buf)
            buf = []
    if buf:
        yield torch.stack(buf)


# ------------------------------------------------------------------ reference cache formats
def save_feature_cache(path: str, keys: Sequence[str], features: torch.Tensor) -> None:
    """Write ``./caches/search/features.pkl``'s format: a pickled dict {relpath: np.float32[E]}
    (reference code/search_image.py:158-160)."""
    if len(keys) != features.shape[0]:
        raise ValueError(f"{len(keys)} keys for {features.shape[0]} feature rows")
    arr = features.detach().to("cpu", torch.float32).numpy()
    d = {k: np.ascontiguousarray(arr[i]) for i, k in enumerate(keys)}
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "wb") as f:
        pickle.dump(d, f)


class _FeatureCacheUnpickler(pickle.Unpickler):
    """features.pkl is ``{str: np.float32[E]}`` and nothing else: the only globals such a pickle names are numpy's array
    reconstruction helpers.  Everything else raises -- a cache file cannot run code here."""

    _ALLOWED = {("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
                ("numpy", "ndarray"), ("numpy", "dtype"),
                ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar")}

    def find_class(self, module, name):
        if (module, name) in self._ALLOWED:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"feature cache names {module}.{name}: only str keys and numpy arrays are accepted")


def load_feature_cache(path: str) -> Tuple[List[str], torch.Tensor]:
    """Read a features.pkl written by ``save_feature_cache`` or by the reference's ``build_cache``
    (reference code/search_image.py:162-164) through an allow-listed unpickler: a dict of str -> numeric numpy array,
    anything else in the file raises before it can execute."""
    with open(path, "rb") as f:
        d = _FeatureCacheUnpickler(f).load()
    if not isinstance(d, dict):
        raise pickle.UnpicklingError(f"{path}: expected a dict of features, got {type(d).__name__}")
    for k, v in d.items():
        if not isinstance(k, str) or not isinstance(v, np.ndarray) or v.dtype.kind not in "fiu":
            raise pickle.UnpicklingError(f"{path}: entry {k!r} is not a str -> numeric array pair")
    keys = list(d.keys())
    feats = torch.from_numpy(np.stack([np.asarray(d[k], dtype=np.float32).reshape(-1) for k in keys])) if keys \
        else torch.empty(0, 0)
    return keys, feats


def build_cache(model, keys: Sequence[str], load: Callable, cache_path: Optional[str] = None,
                batch_size: int = 256, out_dtype: torch.dtype = torch.bfloat16):
    """``build_cache`` of the reference (code/search_image.py:142-165) with a batched loop body.

    ``load(key)`` returns the preprocessed [3,S,S] tensor of one image.  Returns (keys, features [N,E]
    on the GPU, L2-normalised).  If ``cache_path`` exists it is loaded instead of re-encoding, and if
    given it is (re)written in the reference's pickle format.
    """
    if cache_path and os.path.exists(cache_path):
        k, f = load_feature_cache(cache_path)
        return k, f.to(model.device, out_dtype)
    feats = encode_gallery(model, batched(keys, load, batch_size), normalize=True, out_dtype=out_dtype)
    if cache_path:
        save_feature_cache(cache_path, keys, feats)
    return list(keys), feats


def save_split_features(cache_dir: str, split: str, features: torch.Tensor, labels: torch.Tensor) -> None:
    """``{split}_f.pt`` / ``{split}_l.pt`` of ``pre_load_features`` (reference code/utils.py:150-151)."""
    os.makedirs(cache_dir, exist_ok=True)
    torch.save(features.detach().cpu(), os.path.join(cache_dir, f"{split}_f.pt"))
    torch.save(labels.detach().cpu(), os.path.join(cache_dir, f"{split}_l.pt"))


def load_split_features(cache_dir: str, split: str):
    """Counterpart of reference code/utils.py:154-155 (tensors only: ``weights_only=True``)."""
    f = torch.load(os.path.join(cache_dir, f"{split}_f.pt"), weights_only=True)
    l = torch.load(os.path.join(cache_dir, f"{split}_l.pt"), weights_only=True)
    return f, l


@torch.no_grad()
def build_cache_model(model, loader_factory: Callable[[], Iterable], augment_epoch: int, num_classes: int,
                      cache_dir: Optional[str] = None, shots: int = 0):
    """Tip-Adapter cache keys / values (reference code/utils.py:99-132): keys = mean over
    ``augment_epoch`` passes of the encoded few-shot set, L2-normalised, transposed to [E, N];
    values = one-hot labels [N, C].  ``loader_factory()`` returns a fresh iterable of (images, target)."""
    from .search import l2_normalize

    acc, values = None, None
    for ep in range(augment_epoch):
        f, lab = encode_gallery(model, loader_factory(), normalize=False, out_dtype=torch.float32, return_labels=True)
        acc = f if acc is None else acc + f
        if ep == 0:
            values = lab
    keys = l2_normalize(acc / float(augment_epoch)).t().contiguous()
    vals = torch.nn.functional.one_hot(values.long().to(model.device), num_classes).to(torch.float32)
    if cache_dir:
        os.makedirs(cache_dir, exist_ok=True)
        torch.save(keys.cpu(), os.path.join(cache_dir, f"keys_{shots}shots.pt"))
        torch.save(vals.cpu(), os.path.join(cache_dir, f"values_{shots}shots.pt"))
    return keys, vals
