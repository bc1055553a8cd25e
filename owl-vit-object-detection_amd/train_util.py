"""Input-contract glue -- mirror of reference src/train_util.py:4-13 (+ src/util.py:83-93,123-129).

`coco_to_model_input(boxes, metadata)`: absolute COCO `xywh` pixels -> `xyxy` normalised by the image
(width, height); same argument order and metadata keys ("width", "height") as the reference, batch-aware
(boxes [B,n,4], metadata values scalars or [B]).  Host-side elementwise glue on a few dozen numbers per
image (the reference runs it on the CPU before `.to(device)`, main.py:79) -- not a kernel.
"""
import torch


def box_convert(boxes: torch.Tensor, in_fmt: str, out_fmt: str) -> torch.Tensor:
    """ref src/util.py:123-129 (torchvision.ops.box_convert semantics for xywh <-> xyxy)."""
    if in_fmt == out_fmt:
        return boxes.clone()
    if (in_fmt, out_fmt) == ("xywh", "xyxy"):
        x, y, w, h = boxes.unbind(-1)
        return torch.stack([x, y, x + w, y + h], dim=-1)
    if (in_fmt, out_fmt) == ("xyxy", "xywh"):
        x0, y0, x1, y1 = boxes.unbind(-1)
        return torch.stack([x0, y0, x1 - x0, y1 - y0], dim=-1)
    raise ValueError(f"unsupported conversion {in_fmt} -> {out_fmt}")


def scale_bounding_box(boxes: torch.Tensor, imwidth, imheight, mode: str) -> torch.Tensor:
    """ref src/util.py:83-93: mode "down" divides x by width and y by height, "up" multiplies."""
    assert mode in ("down", "up")
    shape = (-1,) + (1,) * (boxes.dim() - 1)          # per-image scalars broadcast over [n, coord]
    w = torch.as_tensor(imwidth, dtype=boxes.dtype, device=boxes.device).reshape(shape)
    h = torch.as_tensor(imheight, dtype=boxes.dtype, device=boxes.device).reshape(shape)
    out = boxes.clone()
    if mode == "down":
        out[..., 0::2] = out[..., 0::2] / w
        out[..., 1::2] = out[..., 1::2] / h
    else:
        out[..., 0::2] = out[..., 0::2] * w
        out[..., 1::2] = out[..., 1::2] * h
    return out


def coco_to_model_input(boxes: torch.Tensor, metadata) -> torch.Tensor:
    """absolute xywh -> relative xyxy (ref src/train_util.py:4-13)."""
    boxes = box_convert(boxes, "xywh", "xyxy")
    return scale_bounding_box(boxes, metadata["width"], metadata["height"], mode="down")


def model_output_to_image(boxes: torch.Tensor, metadata) -> torch.Tensor:
    """ref src/train_util.py:16-24: normalised xyxy -> pixels."""
    return scale_bounding_box(boxes, metadata["width"], metadata["height"], mode="up")
