"""Time the device input pipeline on COCO-sized u8 images already resident in HBM, and from host."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from owl_vit_object_detection_amd.preprocess import DeviceImageProcessor
from oracle import owl_oracle as O

rng = np.random.default_rng(0)
for (S, dtype) in [(768, torch.bfloat16), (768, torch.float32), (840, torch.bfloat16)]:
    imgs = [rng.integers(0, 256, (480, 640, 3), dtype=np.uint8) for _ in range(32)]
    dev = [torch.from_numpy(i).cuda() for i in imgs]
    pin = [torch.from_numpy(i).pin_memory() for i in imgs]
    ip = DeviceImageProcessor(size=S, dtype=dtype)
    for src, tag in ((dev, "HBM-resident"), (pin, "pinned host (PCIe incl.)")):
        for _ in range(5): ip(images=src)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): ip(images=src)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        alg = 32 * (480 * 640 * 3 + 3 * S * S * (2 if dtype == torch.bfloat16 else 4))
        print(f"S={S} {dtype}: {tag}: {dt*1e3:.2f} ms / 32 images = {32/dt:.0f} img/s; algorithmic {alg/dt/1e9:.1f} GB/s", flush=True)
t0 = time.perf_counter(); O.preprocess_image(imgs[0], 768); print(f"CPU oracle (numpy): {(time.perf_counter()-t0)*1e3:.0f} ms/img")
try:
    from PIL import Image
    im = Image.fromarray(imgs[0]); t0 = time.perf_counter()
    for _ in range(10): np.asarray(im.resize((768, 768), resample=Image.BICUBIC))
    print(f"PIL resize alone (1 core): {(time.perf_counter()-t0)*100:.1f} ms/img")
except Exception as e:
    print("PIL not timed:", e)
