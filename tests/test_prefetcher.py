"""The product-level input stage (ref main.py:70-79 `image.to(device)` in front of every step): preprocess.DevicePrefetcher.
CPU part: how a host batch's image part is classified and laid out in the staging slab.  GPU part: the pixels it delivers are the reference
processor's, bit for bit (fixture F7 = PIL + HF OwlViTImageProcessor), a train step fed by it is bit-identical to the HBM-resident step, batches
arrive in order from both the threaded and the in-line mode, and a loader exception surfaces in the consumer."""
import os

import numpy as np
import pytest
import torch
import torch.utils.data

from owl_vit_object_detection_amd.preprocess import classify_images, pack_plan


def test_classify_images_forms():
    S = 96
    u8 = lambda *s: torch.zeros(*s, dtype=torch.uint8)
    assert classify_images(u8(4, S, S, 3), S)[0] == "u8_hwc"
    assert classify_images(u8(4, 3, S, S), S)[0] == "u8_chw"
    kind, items = classify_images(u8(4, 60, 80, 3), S)
    assert kind == "u8_ragged" and len(items) == 4 and tuple(items[0].shape) == (60, 80, 3)
    kind, items = classify_images([np.zeros((50, 70, 3), np.uint8), u8(33, 44, 3)], S)
    assert kind == "u8_ragged" and all(t.dtype == torch.uint8 for t in items)
    assert classify_images(torch.zeros(2, 3, S, S), S)[0] == "dense"
    assert classify_images(torch.zeros(3, S, S), S)[1][0].shape == (1, 3, S, S)
    kind, items = classify_images([torch.zeros(3, S, S), torch.zeros(3, S, S)], S)
    assert kind == "dense" and items[0].shape == (2, 3, S, S)
    kind, items = classify_images(u8(S, S, 3), S)               # one image
    assert kind == "u8_hwc" and items[0].shape == (1, S, S, 3)
    with pytest.raises(ValueError):
        classify_images(torch.zeros(2, 3, S + 1, S), S)
    with pytest.raises(TypeError):
        classify_images(torch.zeros(2, 3, S, S, dtype=torch.float64), S)
    with pytest.raises(ValueError):
        classify_images(u8(2, 5, 5, 4), S)
    with pytest.raises(ValueError):
        classify_images([], S)


def test_pack_plan_alignment():
    items = [torch.zeros(5, 7, 3, dtype=torch.uint8), torch.zeros(100, 3, dtype=torch.uint8), torch.zeros(3, 4, 4)]
    offs, total = pack_plan(items)
    assert offs == [0, 256, 768] and total == 768 + 256
    assert all(o % 256 == 0 for o in offs)


# ---- GPU -------------------------------------------------------------------------------------------------------------------------------------
def _f7(golden_dir):
    return np.load(os.path.join(golden_dir, "f7_preprocess.npz"))


def _img(z, k):
    if f"img_{k}" in z:
        return z[f"img_{k}"]
    from owl_vit_object_detection_amd import rng as crng
    H, W = (int(v) for v in z[f"shape_{k}"])
    return crng.randint(77, f"f7/{k}", H * W * 3, 256).reshape(H, W, 3).astype(np.uint8)


@pytest.mark.gpu
@pytest.mark.parametrize("chw", [False, True])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_normalize_u8_is_the_table(chw, dtype):
    """owl_normalize_u8 against indexing the reference's [3,256] table (exact), both source layouts, and against the resize path run on images that already
    have the model's size (Pillow's taps at scale 1 are (0, 1, 0, 0): the same bits)."""
    from owl_vit_object_detection_amd.preprocess import DeviceImageProcessor
    S, B = 96, 5
    g = torch.Generator().manual_seed(3)
    hwc = torch.randint(0, 256, (B, S, S, 3), generator=g, dtype=torch.uint8)
    ip = DeviceImageProcessor(size=S, dtype=dtype)
    src = (hwc.permute(0, 3, 1, 2).contiguous() if chw else hwc).cuda()
    got = ip.normalize_sized(src, chw=chw)
    lut = ip.lut.cpu()
    exp = torch.stack([lut[c][hwc[..., c].long()] for c in range(3)], dim=1).to(dtype)
    assert got.dtype == dtype and torch.equal(got.cpu(), exp)
    via_resize = ip(images=[hwc[i] for i in range(B)])["pixel_values"]
    assert torch.equal(via_resize, got)
    with pytest.raises(ValueError):
        ip.normalize_sized(src[:, :, :50].contiguous() if chw else src[:, :50].contiguous(), chw=chw)


@pytest.mark.gpu
@pytest.mark.parametrize("threaded", [True, False])
def test_prefetcher_delivers_the_reference_pixels(golden_dir, threaded):
    """Ragged uint8 batches through the prefetcher: F7-exact pixel_values (PIL bicubic + HF rescale / normalize), batches in order, targets on the device,
    metadata untouched, target_transform applied on the host."""
    from owl_vit_object_detection_amd.preprocess import DevicePrefetcher
    z = _f7(golden_dir)
    by_size = {}
    for k in range(int(z["n_cases"])):
        by_size.setdefault(int(z[f"size_{k}"]), []).append(k)
    S, ks = max(by_size.items(), key=lambda kv: len(kv[1]))
    seen = []

    def loader():
        for rep in range(4):               # the same ragged batch four times, distinguishable by its targets
            labels = [torch.tensor([rep, k]) for k in ks]
            boxes = torch.full((len(ks), 2, 4), float(rep))
            yield [_img(z, k) for k in ks], labels, boxes, {"rep": rep, "width": torch.tensor([7])}

    def tt(labels, boxes, meta):
        seen.append(meta["rep"])
        return labels, boxes * 2.0, meta

    pf = DevicePrefetcher(loader(), "cuda", size=S, dtype=torch.float32, depth=2, target_transform=tt, threaded=threaded)
    n = 0
    for rep, (img, labels, boxes, meta) in enumerate(pf):
        assert img.is_cuda and img.dtype == torch.float32 and tuple(img.shape) == (len(ks), 3, S, S)
        out = img.cpu().numpy()
        for i, k in enumerate(ks):
            st = int(z[f"stride_{k}"])
            assert np.array_equal(out[i][:, ::st, ::st], z[f"pixel_values_{k}"])
        assert all(l.is_cuda for l in labels) and [int(l[0]) for l in labels] == [rep] * len(ks)
        assert boxes.is_cuda and float(boxes[0, 0, 0]) == 2.0 * rep
        assert meta["rep"] == rep and not meta["width"].is_cuda
        n += 1
    assert n == 4 and seen == [0, 1, 2, 3]
    assert pf.bytes_h2d > 0 and pf.batches == 4


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["u8_hwc", "u8_chw", "f32", "f32_pinned", "ragged"])
def test_train_step_fed_by_the_prefetcher_is_bit_identical(form):
    """Two train steps on the tiny config: images handed over by the prefetcher (every host form) against the same pixel_values resident in HBM as f32
    (the HBM-resident path of bench.py): same losses, same parameters after AdamW, bit for bit."""
    from owl_vit_object_detection_amd import synth, weights
    from owl_vit_object_detection_amd.config import get_config
    from owl_vit_object_detection_amd.losses import PushPullLoss
    from owl_vit_object_detection_amd.models import OwlViT
    from owl_vit_object_detection_amd.optim import FusedAdamW
    from owl_vit_object_detection_amd.preprocess import DeviceImageProcessor, DevicePrefetcher
    cfg = get_config("tiny")
    S, B, steps = cfg.image_size, 4, 2
    g = torch.Generator().manual_seed(11)
    if form == "ragged":
        raw = [[torch.randint(0, 256, (50 + 9 * i + 3 * k, 130 - 11 * i, 3), generator=g, dtype=torch.uint8) for i in range(B)] for k in range(steps)]
    else:
        raw = [torch.randint(0, 256, (B, S, S, 3), generator=g, dtype=torch.uint8) for _ in range(steps)]
    ip32 = DeviceImageProcessor(size=S, dtype=torch.float32)
    resident = [ip32(images=[r[i] for i in range(B)])["pixel_values"].clone() for r in raw]          # f32 [B,3,S,S] in HBM: what the reference's loop holds after .to(device)
    labels, boxes = synth.make_targets(cfg, B * steps, max_boxes=5)
    lab = [[torch.from_numpy(l) for l in labels[k * B:(k + 1) * B]] for k in range(steps)]
    box = [[torch.from_numpy(b) for b in boxes[k * B:(k + 1) * B]] for k in range(steps)]

    def host_batches():
        for k in range(steps):
            if form == "u8_hwc" or form == "ragged":
                im = raw[k]
            elif form == "u8_chw":
                im = raw[k].permute(0, 3, 1, 2).contiguous()
            else:
                im = resident[k].cpu()
                if form == "f32_pinned":
                    im = im.pin_memory()
            yield im, lab[k], box[k]

    def run(feed):
        model = OwlViT(cfg, weights.make_weights(cfg), "cuda")
        crit = PushPullLoss(cfg.n_classes, None)
        opt = FusedAdamW(model, lr=1e-3, weight_decay=0.1)
        out = []
        for img, l, b in feed:
            opt.zero_grad()
            pb, _, ps, _ = model(img)
            losses = crit(ps, l, pb, b)
            (losses["loss_ce"] + losses["loss_bg"] + losses["loss_bbox"] + losses["loss_giou"]).backward()
            opt.step()
            out.append(torch.stack([losses[k].detach() for k in ("loss_ce", "loss_bg", "loss_bbox", "loss_giou")]).cpu())
        torch.cuda.synchronize()
        return torch.stack(out), model.flat_param.detach().clone().cpu()

    ref_l, ref_p = run((resident[k], [t.cuda() for t in lab[k]], [t.cuda() for t in box[k]]) for k in range(steps))
    got_l, got_p = run(DevicePrefetcher(host_batches(), "cuda", size=S))
    assert torch.equal(ref_l, got_l) and torch.equal(ref_p, got_p)


@pytest.mark.gpu
def test_prefetcher_surfaces_loader_errors_and_stops():
    from owl_vit_object_detection_amd.preprocess import DevicePrefetcher
    S = 96

    def bad():
        yield torch.zeros(2, S, S, 3, dtype=torch.uint8), torch.zeros(2)
        raise RuntimeError("loader died")

    it = iter(DevicePrefetcher(bad(), "cuda", size=S))
    next(it)
    with pytest.raises(RuntimeError, match="loader died"):
        next(it)
    # abandoning an iterator mid-way does not leave a thread blocked on the queue
    pf = DevicePrefetcher(((torch.zeros(1, S, S, 3, dtype=torch.uint8),) for _ in range(50)), "cuda", size=S, depth=1)
    it = iter(pf)
    next(it)
    th = pf._thread
    it.close()
    pf.close()
    assert th is None or not th.is_alive()
    with pytest.raises(ValueError):
        DevicePrefetcher([], "cpu")


class _RawImages(torch.utils.data.Dataset):
    """Module level (picklable): the DataLoader workers of the test below are SPAWNED -- a forked child of a process that has initialised the HIP runtime
    (and, under pytest, a few hundred tests' worth of streams and threads) is not safe to run in; first attempt: worker killed by SIGSEGV."""

    def __len__(self):
        return 6

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(100 + i)
        H, W = 60 + 7 * i, 90 - 5 * i
        img = torch.randint(0, 256, (H, W, 3), generator=g, dtype=torch.uint8)
        n = 1 + i % 3
        boxes = torch.rand(n, 4, generator=g) * 20.0 + 1.0                       # xywh pixels
        return img, torch.arange(n), boxes, {"width": W, "height": H, "impath": f"img{i}"}


def _collate_lists(items):                                                        # batch of 2, ragged: lists (the reference's loader is batch 1)
    imgs, labels, boxes, meta = zip(*items)
    return list(imgs), list(labels), list(boxes), list(meta)


@pytest.mark.gpu
def test_prefetcher_around_a_torch_dataloader_with_workers():
    """The reference's own shape of the input side (ref src/dataset.py:60-106: a Dataset returning (image, labels, boxes, metadata), DataLoader with worker
    processes) with the HF processor taken OUT of the dataset: workers hand over raw uint8 images of different sizes, the prefetcher resizes + normalises on the
    device -- pixel_values equal to DeviceImageProcessor's own (which F7 pins to PIL + HF), targets converted on the host by the reference's coco_to_model_input."""
    from torch.utils.data import DataLoader
    from owl_vit_object_detection_amd.preprocess import DeviceImageProcessor, DevicePrefetcher
    from owl_vit_object_detection_amd.train_util import coco_to_model_input
    S = 96

    def tt(labels, boxes, meta):
        return labels, [coco_to_model_input(b[None], m)[0] for b, m in zip(boxes, meta)], meta

    ds = _RawImages()
    loader = DataLoader(ds, batch_size=2, shuffle=False, num_workers=2, collate_fn=_collate_lists, multiprocessing_context="spawn")
    ip = DeviceImageProcessor(size=S, dtype=torch.bfloat16)
    k = 0
    for img, labels, boxes, meta in DevicePrefetcher(loader, "cuda", size=S, target_transform=tt):
        assert img.dtype == torch.bfloat16 and tuple(img.shape) == (2, 3, S, S)
        exp = ip(images=[ds[2 * k + j][0] for j in range(2)])["pixel_values"]
        assert torch.equal(img, exp)
        for j in range(2):
            raw = ds[2 * k + j]
            assert labels[j].is_cuda and torch.equal(labels[j].cpu(), raw[1])
            want = coco_to_model_input(raw[2][None], raw[3])[0]
            assert boxes[j].is_cuda and torch.equal(boxes[j].cpu(), want)
            assert meta[j]["impath"] == f"img{2 * k + j}"
        k += 1
    assert k == 3
