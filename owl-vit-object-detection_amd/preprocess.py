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
