"""Device input pipeline with the call surface the reference uses (ref src/dataset.py:69-71):

    pixel_values = image_processor(images=image, return_tensors="pt")["pixel_values"]

``DeviceImageProcessor`` is the HF ``OwlViTImageProcessor`` (PIL backend: bicubic resize to size x size, no crop,
x(1/255), CLIP mean/std) re-built for the GPU: the u8 image goes to HBM once (0.9 MB for a COCO image instead of
7 MB of f32 pixel_values), two HIP kernels reproduce Pillow's fixed-point separable bicubic bit-for-bit and the
normalisation is a 768-entry table of the reference's own float results.  Output is [B,3,S,S] f32 (the reference's
contract) or bf16 (what the patch-embed loader consumes) on the device.  No CPU fallback: the resize runs in
libowlhip.so or not at all; only Pillow's f64 tap tables are computed on the host (``owl_bicubic_coeffs``), cached
per image size.
"""
import math

import numpy as np
import torch

from . import _lib, ops

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)     # HF OPENAI_CLIP_MEAN / OPENAI_CLIP_STD (OwlViTImageProcessor defaults)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def normalize_lut(mean, std, rescale_factor) -> np.ndarray:
    """[3,256] f32 = what transformers' rescale (f64 multiply, f32 cast) + normalize (f32) give for each u8 level."""
    v = (np.arange(256, dtype=np.uint8).astype(np.float64) * rescale_factor).astype(np.float32)
    m = np.array(mean, dtype=np.float32)[:, None]
    s = np.array(std, dtype=np.float32)[:, None]
    return ((v[None, :] - m) / s).astype(np.float32)


class DeviceImageProcessor:
    def __init__(self, size=768, image_mean=CLIP_MEAN, image_std=CLIP_STD, rescale_factor=1 / 255, device="cuda",
                 dtype=torch.float32):
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("DeviceImageProcessor: dtype must be torch.float32 or torch.bfloat16")
        self.size = int(size)
        self.device = torch.device(device)
        self.dtype = dtype
        self.lut = torch.from_numpy(normalize_lut(image_mean, image_std, rescale_factor)).to(self.device)
        self._tables = {}
        self._tmp = None

    def _axis_tables(self, in_size: int):
        key = in_size
        if key not in self._tables:
            out = self.size
            ksize = int(math.ceil(2.0 * max(in_size / out, 1.0))) * 2 + 1
            bounds = torch.zeros(out * 2, dtype=torch.int32)
            kk = torch.zeros(out * ksize, dtype=torch.int32)
            ks = torch.zeros(1, dtype=torch.int32)
            _lib.call("owl_bicubic_coeffs", in_size, out, bounds, kk, kk.numel(), ks)
            assert int(ks.item()) == ksize
            self._tables[key] = (bounds.to(self.device), kk.to(self.device), ksize)
        return self._tables[key]

    def _to_device(self, img):
        if isinstance(img, np.ndarray):
            img = torch.from_numpy(np.ascontiguousarray(img))
        elif not torch.is_tensor(img):                       # PIL.Image without importing PIL here
            img = torch.from_numpy(np.ascontiguousarray(np.asarray(img.convert("RGB") if hasattr(img, "convert") else img)))
        if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] != 3:
            raise ValueError(f"DeviceImageProcessor: expected an RGB uint8 [H,W,3] image, got {img.dtype} {tuple(img.shape)}")
        return img.to(self.device, non_blocking=True).contiguous()

    def __call__(self, images, return_tensors="pt", **_):
        single = not isinstance(images, (list, tuple))
        imgs = [self._to_device(im) for im in ([images] if single else list(images))]
        n = len(imgs)
        out = torch.empty(n, 3, self.size, self.size, dtype=self.dtype, device=self.device)
        # one launch pair for the whole (ragged) batch: a 10-int64 descriptor per image
        rows, off, max_h = [], 0, 0
        for im in imgs:
            H, W = int(im.shape[0]), int(im.shape[1])
            bx, kx, ksx = self._axis_tables(W)
            by, ky, ksy = self._axis_tables(H)
            rows.append((im.data_ptr(), H, W, bx.data_ptr(), kx.data_ptr(), ksx, by.data_ptr(), ky.data_ptr(), ksy, off))
            off += (H * self.size * 3 + 255) // 256 * 256
            max_h = max(max_h, H)
        if self._tmp is None or self._tmp.numel() < off:
            self._tmp = torch.empty(off, dtype=torch.uint8, device=self.device)
        desc_d = torch.from_numpy(np.asarray(rows, dtype=np.int64)).to(self.device, non_blocking=True)
        _lib.call("owl_preprocess_u8_batch", ops.stream(), desc_d, n, max_h, self._tmp, self.lut, out,
                  1 if self.dtype == torch.bfloat16 else 0, self.size, self.size)
        self._keep = (imgs, desc_d)          # keep the sources alive until the stream has consumed them
        return {"pixel_values": out}

    def normalize_sized(self, src_u8: torch.Tensor, chw: bool) -> torch.Tensor:
        """Images that already have the model's size: u8 [B,S,S,3] (chw=False) or [B,3,S,S] (chw=True) ON THE DEVICE -> [B,3,S,S] self.dtype.  Pillow's resize to
        the size an image already has is a copy, so only the table step of the reference pipeline is left (owl_normalize_u8)."""
        if src_u8.dtype != torch.uint8 or src_u8.dim() != 4 or not src_u8.is_cuda:
            raise ValueError(f"normalize_sized: expected a device uint8 [B,S,S,3] / [B,3,S,S] tensor, got {src_u8.dtype} {tuple(src_u8.shape)} on {src_u8.device}")
        n = int(src_u8.shape[0])
        H, W_ = (int(src_u8.shape[2]), int(src_u8.shape[3])) if chw else (int(src_u8.shape[1]), int(src_u8.shape[2]))
        if (int(src_u8.shape[1]) if chw else int(src_u8.shape[3])) != 3 or (H, W_) != (self.size, self.size):
            raise ValueError(f"normalize_sized: expected {self.size} x {self.size} RGB images, got {tuple(src_u8.shape)} (chw={chw})")
        src_u8 = src_u8.contiguous()
        out = torch.empty(n, 3, self.size, self.size, dtype=self.dtype, device=self.device)
        _lib.call("owl_normalize_u8", ops.stream(), src_u8, 1 if chw else 0, self.lut, out, 1 if self.dtype == torch.bfloat16 else 0, n, H, W_)
        return out


# ---------------------------------------------------------------------------------------------------------------------------------------
# The reference's loop does `image = image.to(device)` in front of every step (ref main.py:77-79): 7.1 MB of f32 pixels per B/16 image over
# PCIe, serialised with the step -- the HBM-resident kernels then wait for the bus (measured: 1003 against 1217 img/s at batch 32).  What
# the hot path needs from its caller is the next batch already in HBM when the step starts, and as few bytes over the bus as the data has:
# the u8 pixels (1.8 MB per 768^2 image; 0.9 MB for a COCO-size one that is resized on the device).
# ---------------------------------------------------------------------------------------------------------------------------------------
def _as_u8_tensor(img):
    """torch u8 tensor view of one host image (torch / numpy / PIL), no copy where the source allows it."""
    if torch.is_tensor(img):
        return img
    if isinstance(img, np.ndarray):
        return torch.from_numpy(np.ascontiguousarray(img))
    return torch.from_numpy(np.ascontiguousarray(np.asarray(img.convert("RGB") if hasattr(img, "convert") else img)))


def classify_images(images, size: int):
    """What a host batch's image part is -> (kind, items).  kind: 'dense' (f32 / bf16 [B,3,S,S] pixel_values, the reference DataLoader's output: copied, cast to
    the compute type on the device), 'u8_chw' / 'u8_hwc' (uint8 [B,3,S,S] / [B,S,S,3] already at the model's size: table step only), 'u8_ragged' (a list of -- or
    a [B,H,W,3] tensor of -- uint8 HWC images of any sizes: Pillow-exact bicubic resize + table on the device).  Pure host logic (CPU-tested)."""
    if isinstance(images, (list, tuple)):
        items = [_as_u8_tensor(im) for im in images]
        if not items:
            raise ValueError("DevicePrefetcher: empty image list")
        if all(t.dtype == torch.uint8 and t.dim() == 3 and t.shape[2] == 3 for t in items):
            return "u8_ragged", items
        if all(t.dtype in (torch.float32, torch.bfloat16) and tuple(t.shape) == (3, size, size) for t in items):
            return "dense", [torch.stack(items)]
        raise ValueError("DevicePrefetcher: a list of images must hold uint8 [H,W,3] images (or [3,S,S] pixel_values)")
    t = images if torch.is_tensor(images) else _as_u8_tensor(images)
    if t.dtype in (torch.float32, torch.bfloat16):
        if t.dim() == 3:
            t = t.unsqueeze(0)
        if t.dim() != 4 or tuple(t.shape[1:]) != (3, size, size):
            raise ValueError(f"DevicePrefetcher: pixel_values must be [B,3,{size},{size}], got {tuple(t.shape)}")
        return "dense", [t]
    if t.dtype != torch.uint8:
        raise TypeError(f"DevicePrefetcher: images must be uint8 (raw pixels) or float32 / bfloat16 (pixel_values), got {t.dtype}")
    if t.dim() == 3:
        t = t.unsqueeze(0)
    if t.dim() != 4:
        raise ValueError(f"DevicePrefetcher: uint8 images must be [B,H,W,3] or [B,3,S,S], got {tuple(t.shape)}")
    if tuple(t.shape[1:]) == (3, size, size):
        return "u8_chw", [t]
    if t.shape[3] == 3:
        if tuple(t.shape[1:3]) == (size, size):
            return "u8_hwc", [t]
        return "u8_ragged", [t[i] for i in range(t.shape[0])]
    raise ValueError(f"DevicePrefetcher: cannot interpret a uint8 tensor of shape {tuple(t.shape)}")


def pack_plan(items, align: int = 256):
    """Byte offsets of `items` (host tensors) in one staging slab, each aligned -> (offsets, total bytes)."""
    offs, off = [], 0
    for t in items:
        offs.append(off)
        off += (t.numel() * t.element_size() + align - 1) // align * align
    return offs, off


class _PinnedRing:
    """`n` pinned host slabs (grown on demand), each with the event of the last H2D copy that read it: a slab is rewritten only after that copy has finished."""

    def __init__(self, n):
        self.slabs = [None] * n
        self.events = [None] * n
        self.k = 0

    def next(self, nbytes):
        i = self.k
        self.k = (self.k + 1) % len(self.slabs)
        if self.events[i] is not None:
            self.events[i].synchronize()                    # (blocks the staging thread only)
        if self.slabs[i] is None or self.slabs[i].numel() < nbytes:
            self.slabs[i] = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, pin_memory=True)
        return i, self.slabs[i]


class DevicePrefetcher:
    """Wraps any iterable of host batches `(images, *targets)` -- the reference's `train_dataloader` (ref main.py:70-73) -- and yields the same tuples with the
    images ALREADY in HBM as the model's bf16 `[B,3,S,S]` input, one batch ahead of the step that consumes them:

        host batch -> pinned ring slab -> copy stream (H2D) -> DeviceImageProcessor (resize / table / cast, same stream) -> event -> handed to the caller's stream

    `images` may be uint8 raw pixels (a list of `[H,W,3]` images of any sizes, a `[B,H,W,3]` tensor, or `[B,S,S,3]` / `[B,3,S,S]` at the model's size) --
    a quarter (or an eighth, for COCO-size images) of the PCIe bytes of f32 pixel_values, processed bit-exactly like the reference's HF processor (fixture F7) --
    or the reference's own f32 `[B,3,S,S]` pixel_values (copied and cast only).  Targets: `target_transform(*targets)` runs on the HOST first (e.g. the
    reference's `coco_to_model_input`, ref main.py:79), then tensors and lists of tensors are moved to the device on the copy stream (`move_targets`); dicts
    (the reference's `metadata`) and everything else pass through untouched.  The reference loop's own `.to(device)` calls become no-ops.

    Ordering: the consumer's stream waits for the batch's event and the tensors are `record_stream`-ed on it, so the caching allocator does not recycle them
    while the step still reads them; a pinned slab is reused only after its own copy has completed.  `threaded=True` pulls from the loader, stages and
    enqueues from a background thread (`depth` batches ahead); `threaded=False` does the same work inside `__next__`, one batch ahead.
    No CPU fallback: images are processed by libowlhip.so on the device or not at all."""

    _END = object()

    def __init__(self, loader, device="cuda", size=768, dtype=torch.bfloat16, depth=2, processor=None, target_transform=None,
                 move_targets=True, threaded=True):
        self.loader = loader
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ValueError("DevicePrefetcher: the device must be a GPU (there is no CPU path)")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.size = int(size)
        self.dtype = dtype
        self.depth = max(1, int(depth))
        self.processor = processor if processor is not None else DeviceImageProcessor(size=size, device=self.device, dtype=dtype)
        if self.processor.size != self.size or self.processor.dtype != dtype:
            raise ValueError("DevicePrefetcher: the processor's size / dtype must match")
        self.target_transform = target_transform
        self.move_targets = bool(move_targets)
        self.threaded = bool(threaded)
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self._ring = _PinnedRing(self.depth + 2)
        from collections import deque
        self._inflight = deque()
        self._thread = None
        self._stop = None
        self._q = None
        self.bytes_h2d = 0                 # bytes sent over the bus so far (statistics: bench.py reports bytes per image)
        self.batches = 0
        self.stage_seconds = 0.0           # host time spent staging (pull from the loader excluded): must stay below the step time for the stage to hide

    def __len__(self):
        return len(self.loader)

    # -- staging (runs in the background thread when threaded) ------------------------------------------------------------------------------
    def _h2d(self, items):
        """Host tensors (the batch's images AND its target tensors) -> ONE pinned ring slab -> ONE H2D copy -> device views (never one small copy per tensor: a
        pageable source makes every `.to(device)` a blocking round trip).  A large tensor the caller already pinned (DataLoader(pin_memory=True)) goes as it is.
        The host-side copy into the slab is a plain memmove (ctypes: releases the GIL): torch's own CPU copy fans out over the intra-op thread pool, which --
        called from a second thread on a many-core host -- costs several times the copy itself (measured: 40 ms per 57 MB batch on the 256-thread GPU box)."""
        import ctypes
        while self._inflight and self._inflight[0][0].query():          # caller-pinned sources whose copies have completed
            self._inflight.popleft()
        out = [None] * len(items)
        packed = []
        for k, t in enumerate(items):
            if t.is_cuda:
                out[k] = t if t.device == self.device else t.to(self.device, non_blocking=True)
            elif t.is_pinned() and t.is_contiguous() and t.numel() * t.element_size() >= (1 << 20):
                dst = torch.empty(t.shape, dtype=t.dtype, device=self.device)
                dst.copy_(t, non_blocking=True)
                ev = torch.cuda.Event(); ev.record(self.copy_stream)
                self._inflight.append((ev, t))                           # keeps the pinned source alive until its copy has run
                self.bytes_h2d += t.numel() * t.element_size()
                out[k] = dst
            else:
                packed.append(k)
        if packed:
            src = [items[k].contiguous() for k in packed]
            offs, total = pack_plan(src)
            i, slab = self._ring.next(total)
            base = slab.data_ptr()
            for t, o in zip(src, offs):
                n = t.numel() * t.element_size()
                if n:
                    ctypes.memmove(base + o, t.data_ptr(), n)
            dslab = torch.empty(total, dtype=torch.uint8, device=self.device)
            dslab.copy_(slab[:total], non_blocking=True)
            ev = torch.cuda.Event(); ev.record(self.copy_stream)
            self._ring.events[i] = ev
            self.bytes_h2d += total
            for k, t, o in zip(packed, src, offs):
                out[k] = dslab[o:o + t.numel() * t.element_size()].view(t.dtype).view(t.shape)
        return out

    def _images_on_device(self, kind, dev):
        if kind == "dense":
            x = dev[0]
            if x.dtype == self.dtype:
                return x
            if x.dtype == torch.float32 and self.dtype == torch.bfloat16:
                return ops.cast_bf16(x.contiguous())         # the model's own first step (same kernel, same rounding), off the critical path
            return x.to(self.dtype)
        if kind in ("u8_chw", "u8_hwc"):
            return self.processor.normalize_sized(dev[0], chw=(kind == "u8_chw"))
        return self.processor(dev)["pixel_values"]

    def _stage(self, batch):
        if not isinstance(batch, (list, tuple)) or len(batch) < 1:
            raise TypeError("DevicePrefetcher: the loader must yield (images, *targets) tuples")
        images, rest = batch[0], tuple(batch[1:])
        if self.target_transform is not None:
            rest = self.target_transform(*rest)
            if not isinstance(rest, tuple):
                rest = (rest,)
        import time
        t0 = time.perf_counter()
        kind, items = classify_images(images, self.size)
        # target tensors ride in the same slab: (position in `rest`, index inside a list or None)
        where, tensors = [], []
        if self.move_targets:
            for r, obj in enumerate(rest):
                if torch.is_tensor(obj):
                    where.append((r, None)); tensors.append(obj)
                elif isinstance(obj, (list, tuple)) and obj and all(torch.is_tensor(o) for o in obj):
                    for j, o in enumerate(obj):
                        where.append((r, j)); tensors.append(o)
        with torch.cuda.stream(self.copy_stream):
            dev = self._h2d(list(items) + tensors)
            img = self._images_on_device(kind, dev[:len(items)])
            if tensors:
                seq = {r: type(rest[r]) for r, j in where if j is not None}       # lists / tuples of tensors keep their type
                rest = [list(o) if r in seq else o for r, o in enumerate(rest)]
                for (r, j), d in zip(where, dev[len(items):]):
                    if j is None:
                        rest[r] = d
                    else:
                        rest[r][j] = d
                rest = tuple(seq[r](o) if r in seq else o for r, o in enumerate(rest))
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        self.batches += 1
        self.stage_seconds += time.perf_counter() - t0
        return (img,) + tuple(rest), ev

    # -- hand-over (the caller's thread and stream) ------------------------------------------------------------------------------------------------
    def _hand_over(self, staged):
        out, ev = staged
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ev)
        for o in out:
            for t in (o if isinstance(o, (list, tuple)) else (o,)):
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(cur)
        return out

    def _worker(self, it, q, stop):
        import queue
        try:
            torch.cuda.set_device(self.device)
            for batch in it:
                item = self._stage(batch)
                while not stop.is_set():
                    try:
                        q.put(item, timeout=0.1)
                        break
                    except queue.Full:
                        continue
                if stop.is_set():
                    return
            item = self._END
        except BaseException as e:          # re-raised in the consumer's thread
            item = e
        while not stop.is_set():
            try:
                q.put(item, timeout=0.1)
                return
            except queue.Full:
                continue

    def close(self):
        """Stop the background thread (also called when the iterator is exhausted, abandoned by a new __iter__, or collected)."""
        if self._stop is not None:
            self._stop.set()
        if self._thread is not None and self._thread.is_alive():
            self._thread.join(timeout=5.0)
        self._thread = self._stop = self._q = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __iter__(self):
        self.close()
        it = iter(self.loader)
        if not self.threaded:
            from collections import deque
            pending = deque()
            done = False
            while True:
                while not done and len(pending) < 1 + 1:          # the batch handed over now + one in flight behind it
                    try:
                        pending.append(self._stage(next(it)))
                    except StopIteration:
                        done = True
                if not pending:
                    return
                yield self._hand_over(pending.popleft())
        import queue
        import threading
        q, stop = queue.Queue(maxsize=self.depth), threading.Event()
        th = threading.Thread(target=self._worker, args=(it, q, stop), daemon=True, name="owl-prefetch")
        self._thread, self._stop, self._q = th, stop, q
        th.start()
        try:
            while True:
                item = q.get()
                if item is self._END:
                    return
                if isinstance(item, BaseException):
                    raise item
                yield self._hand_over(item)
        finally:
            stop.set()
