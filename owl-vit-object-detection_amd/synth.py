"""Synthetic COCO-shaped batches with the tensor contract of the reference
data path (reference src/dataset.py:60-73 + src/train_util.py:4-13):
``image [B,3,S,S] f32`` CLIP-normalised, ``labels`` int64, ``boxes`` f32 normalised xyxy.
"""
import numpy as np

from . import rng
from .config import OwlConfig

CLIP_MEAN = np.array([0.48145466, 0.4578275, 0.40821073], dtype=np.float64)
CLIP_STD = np.array([0.26862954, 0.26130258, 0.27577711], dtype=np.float64)


def make_images(cfg: OwlConfig, batch: int, seed: int = 1234, first: int = 0) -> np.ndarray:
    """[B,3,S,S] float32: u8 ~ U{0..255} -> /255 -> (x - mean) / std."""
    S = cfg.image_size
    out = np.empty((batch, 3, S, S), dtype=np.float32)
    for b in range(batch):
        u8 = rng.randint(seed, f"image/{first + b}", 3 * S * S, 256).reshape(3, S, S)
        x = u8.astype(np.float64) / 255.0
        out[b] = ((x - CLIP_MEAN[:, None, None]) / CLIP_STD[:, None, None]).astype(np.float32)
    return out


def make_images_u8(cfg: OwlConfig, batch: int, seed: int = 1234, first: int = 0, layout: str = "hwc") -> np.ndarray:
    """The uint8 levels make_images() normalises (the same draws), as a camera / decoder hands them over: [B,S,S,3] (`hwc`) or [B,3,S,S] (`chw`).
    What the u8 input stage (preprocess.DevicePrefetcher) is fed with; its table step gives make_images()'s values to within one f32 ulp (the HF
    processor rounds x/255 to f32 before normalising, make_images() normalises in f64)."""
    S = cfg.image_size
    out = np.empty((batch, 3, S, S), dtype=np.uint8)
    for b in range(batch):
        out[b] = rng.randint(seed, f"image/{first + b}", 3 * S * S, 256).reshape(3, S, S).astype(np.uint8)
    return out if layout == "chw" else np.ascontiguousarray(out.transpose(0, 2, 3, 1))


def make_targets(cfg: OwlConfig, batch: int, seed: int = 1234, first: int = 0, max_boxes: int = 16):
    """Per image: n_i = 1 + h(seed, i) mod max_boxes; boxes valid, inside the image."""
    labels, boxes = [], []
    for b in range(batch):
        i = first + b
        n = 1 + int(rng.randint(seed, f"nbox/{i}", 1, max_boxes)[0])
        x0 = rng.uniform(seed, f"box/{i}", n, 0) * 0.6
        y0 = rng.uniform(seed, f"box/{i}", n, 1) * 0.6
        w = 0.02 + rng.uniform(seed, f"box/{i}", n, 2) * 0.35
        h = 0.02 + rng.uniform(seed, f"box/{i}", n, 3) * 0.35
        boxes.append(np.stack([x0, y0, x0 + w, y0 + h], axis=1).astype(np.float32))
        labels.append(rng.randint(seed, f"label/{i}", n, cfg.n_classes))
    return labels, boxes


def class_scales(cfg: OwlConfig, labels) -> np.ndarray:
    """``round(ln(max_count / count_c) + 3, 1)`` (reference src/dataset.py:88-98)."""
    cnt = np.bincount(np.concatenate(labels), minlength=cfg.n_classes).astype(np.float64)
    cnt = np.maximum(cnt, 1.0)
    return np.round(np.log(cnt.max() / cnt) + 3.0, 1).astype(np.float32)
