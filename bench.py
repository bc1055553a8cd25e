#!/usr/bin/env python
"""Headline benchmark: OWL-ViT-B/16 768x768 TRAIN images/sec on N MI355X (BASELINE.json `metric`).

One step = what the reference's train loop does per batch (ref main.py:74-91), through the same call
surface: zero_grad -> model(image) -> PushPullLoss -> sum of 4 losses -> backward -> [one RCCL all-reduce
of the flat gradient bucket] -> AdamW.  Workload = BASELINE configs[2] (batch 32 per GPU, bf16 compute,
full train step); N > 1 is configs[3] (global batch 32*N, weak scaling, data parallel).
Synthetic COCO-shaped inputs resident in HBM, random-init weights (no network on the box).

Prints ONE JSON line (rank 0) with the driver's contract plus:
  roofline     -- for the dominant kernel (fused attention forward): algorithmic FLOPs per launch / mean
                  launch duration measured with HIP events on the launch stream inside the timed region;
                  peak = 2.5 PFLOP/s dense bf16 MFMA (MI355X_MICROARCH.md)
  cpu_baseline -- the CPU oracle (parity-checked restatement of the reference path) timed on this box's
                  host cores on a bounded sample (batch-1 train steps), rank 0 at N = 1 only
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
_alias = os.path.join(ROOT, "owl_vit_object_detection_amd")
if not os.path.exists(_alias):
    try:
        os.symlink("owl-vit-object-detection_amd", _alias)
    except FileExistsError:
        pass

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA, MI355X_MICROARCH.md "Peak BF16/FP16 MFMA ~2.5 PF dense"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU")
    ap.add_argument("--arch", default="owlvit-base-patch16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--forward-only", action="store_true", help="BASELINE configs[1]-style forward throughput (diagnostic)")
    ap.add_argument("--gemm-tile", type=int, default=0, help="owl_gemm_set_tile override (tuning A/B only; 0 = automatic)")
    return ap.parse_args()


def synth_batches(cfg, B, device, rank, n_batches=2, seed=1234):
    """CLIP-normalised uniform-u8 pixels (f32, as the reference's DataLoader yields) + 1..16 boxes/image."""
    from owl_vit_object_detection_amd import synth
    from owl_vit_object_detection_amd.matcher import PackedTargets
    g = torch.Generator(device=device).manual_seed(seed + rank)
    mean = torch.tensor(synth.CLIP_MEAN, dtype=torch.float32, device=device).view(1, 3, 1, 1)
    std = torch.tensor(synth.CLIP_STD, dtype=torch.float32, device=device).view(1, 3, 1, 1)
    out = []
    for k in range(n_batches):
        u8 = torch.randint(0, 256, (B, 3, cfg.image_size, cfg.image_size), generator=g, device=device, dtype=torch.int32)
        img = ((u8.float() / 255.0) - mean) / std
        labels, boxes = synth.make_targets(cfg, B, seed, first=(rank * n_batches + k) * B, max_boxes=16)
        tg = PackedTargets([torch.from_numpy(l) for l in labels], [torch.from_numpy(b) for b in boxes], device)
        out.append((img.contiguous(), tg, labels))
    return out


def cpu_baseline(cfg, steps):
    """Time the CPU oracle (restated reference path, fp32, all host cores) on batch-1 train steps."""
    from oracle import owl_oracle as O
    from owl_vit_object_detection_amd import synth, weights
    # a few hundred host threads on these op sizes is slower than a few dozen (oversubscription): use
    # at most 32 and report the number actually used
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    w = {k: torch.from_numpy(v) for k, v in weights.make_weights(cfg).items()}
    img = torch.from_numpy(synth.make_images(cfg, 1))
    labels, boxes = synth.make_targets(cfg, 1, max_boxes=16)
    lab = [torch.from_numpy(l) for l in labels]; tb = [torch.from_numpy(b) for b in boxes]
    scales = torch.from_numpy(synth.class_scales(cfg, labels))
    O.train_step(cfg, w, img, lab, tb, scales)          # warm-up
    ts = []
    for _ in range(steps):
        t0 = time.perf_counter()
        O.train_step(cfg, w, img, lab, tb, scales)
        ts.append(time.perf_counter() - t0)
    med = float(np.median(ts))
    return {"value": round(1.0 / med, 4), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"{steps} fp32 batch-1 train steps (fwd+matcher+loss+bwd) of {cfg.name} on the CPU oracle, median {med:.2f} s/step"}


def main():
    args = parse()
    from owl_vit_object_detection_amd import ddp, ops, weights
    from owl_vit_object_detection_amd.config import get_config
    from owl_vit_object_detection_amd.losses import PushPullLoss
    from owl_vit_object_detection_amd.models import OwlViT
    from owl_vit_object_detection_amd.optim import FusedAdamW
    if args.gemm_tile:
        from owl_vit_object_detection_amd import _lib
        _lib.call("owl_gemm_set_tile", args.gemm_tile)
    from owl_vit_object_detection_amd import synth

    rank, world, local = ddp.init_from_env("nccl")
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cfg = get_config(args.arch)
    B = args.batch

    model = OwlViT(cfg, weights.make_weights(cfg), dev)           # identical weights on every rank (seeded)
    batches = synth_batches(cfg, B, dev, rank)
    scales = synth.class_scales(cfg, [l for l in batches[0][2]])
    crit = PushPullLoss(cfg.n_classes, scales)
    opt = FusedAdamW(model, lr=3e-6, weight_decay=0.1)            # ref config.yaml:10,12
    dp = ddp.DataParallel(model, opt)

    # ---- dominant-kernel timing: HIP events around every fused-attention-forward launch --------------
    attn_events = []
    record = {"on": False}

    def timed(orig):
        def f(*a, **k):
            if not record["on"]:
                return orig(*a, **k)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            r = orig(*a, **k)
            e1.record()
            attn_events.append((e0, e1))
            return r
        return f

    # the model calls the V-row-major entry (one QKV GEMM, no V^T copy); the V^T entry is hooked too for completeness
    ops.attention_fwd_vrow = timed(ops.attention_fwd_vrow)
    ops.attention_fwd = timed(ops.attention_fwd)

    def step(i):
        img, tg, _ = batches[i % len(batches)]
        if args.forward_only:
            with torch.no_grad():
                model(img)
            return
        opt.zero_grad()
        pred_boxes, _, pred_sims, _ = model(img)
        losses = crit(pred_sims, tg, pred_boxes)
        loss = losses["loss_ce"] + losses["loss_bg"] + losses["loss_bbox"] + losses["loss_giou"]
        loss.backward()
        dp.sync_and_step()

    for i in range(args.warmup):
        step(i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    record["on"] = True
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    record["on"] = False
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = B * world * args.steps / dt
        flops_img = cfg.flops_forward() if args.forward_only else cfg.flops_train_step()
        attn_ms = float(np.mean([a.elapsed_time(b) for a, b in attn_events])) if attn_events else float("nan")
        attn_flops = 4.0 * B * cfg.heads * cfg.tokens * cfg.tokens * cfg.head_dim          # QK^T + PV per launch
        achieved = attn_flops / (attn_ms * 1e-3) / 1e12
        out = {
            "metric": ("forward" if args.forward_only else "train") + " images/sec, "
                      + ("OWL-ViT-B/16 768x768" if cfg.name == "owlvit-base-patch16" else f"{cfg.name} {cfg.image_size}x{cfg.image_size}"),
            "value": round(value, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic (CLIP-normalised uniform-u8 pixels, 1-16 boxes/image, 10 classes; random-init weights)",
            "config": {"workload": f"{cfg.name} bf16, batch={B}/GPU, {cfg.image_size}x{cfg.image_size}, "
                                   + ("forward only" if args.forward_only else "full train step (matcher+loss+backward+AdamW)")
                                   + (f", DDP over {world} GPUs, one RCCL all-reduce of the flat grad bucket/step" if world > 1 else ""),
                       "global_batch": B * world, "tokens": cfg.tokens, "parallelism": f"dp{world}",
                       "gflop_per_image": round(flops_img / 1e9, 1),
                       "step_mfma_frac": round(flops_img * value / world / 1e12 / PEAK_BF16_TFLOPS, 4)},
            "roofline": {"bound": "mfma", "kernel": "attn_fwd_kernel<VROW>", "achieved": round(achieved, 1), "peak": PEAK_BF16_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4),
                         # HBM bytes per launch from PMC (FETCH_SIZE x2 + WRITE_SIZE, profiles/r01_pmc_attn_fwd.md): measured for
                         # the default workload only (algorithmic = 4*M*D*2 B = 454 MB; r01_pmc_final.md)
                         "traffic": 462.0e6 if (cfg.name == "owlvit-base-patch16" and B == 32) else None,
                         "launches_timed": len(attn_events), "ms_per_launch": round(attn_ms, 4)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, args.cpu_steps)
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
