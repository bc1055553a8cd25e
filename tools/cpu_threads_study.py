"""CPU-baseline thread count (bench.py `cpu_baseline`): the oracle's batch-1 train step on the GPU box's host at several torch thread
counts.  SURVEY.md section 8(d)(ii) names os.cpu_count(); this records what that costs against fewer threads on the box's 256 logical
cores (profiles/r03_cpu_threads.md), and bench.py uses the fastest setting and reports it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import owl_oracle as O
from owl_vit_object_detection_amd import synth, weights
from owl_vit_object_detection_amd.config import get_config

cfg = get_config("owlvit-base-patch16")
w = {k: torch.from_numpy(v) for k, v in weights.make_weights(cfg).items()}
img = torch.from_numpy(synth.make_images(cfg, 1))
labels, boxes = synth.make_targets(cfg, 1, max_boxes=16)
lab = [torch.from_numpy(l) for l in labels]; tb = [torch.from_numpy(b) for b in boxes]
scales = torch.from_numpy(synth.class_scales(cfg, labels))
total = os.cpu_count()
print(f"| torch threads (of {total} logical cores) | s / step (median of 3 after 1 warm-up) | images / s |\n|---|---|---|")
for n in [t for t in (16, 32, 64, 128, 256) if t <= total] + ([total] if total not in (16, 32, 64, 128, 256) else []):
    torch.set_num_threads(n)
    O.train_step(cfg, w, img, lab, tb, scales)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); O.train_step(cfg, w, img, lab, tb, scales); ts.append(time.perf_counter() - t0)
    m = float(np.median(ts))
    print(f"| {n} | {m:.2f} | {1 / m:.3f} |", flush=True)
