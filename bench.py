#!/usr/bin/env python
"""Headline benchmark: OWL-ViT-B/16 768x768 TRAIN images/sec on N MI355X (BASELINE.json `metric`).

One step = what the reference's train loop does per batch (ref main.py:74-91), through the same call
surface: zero_grad -> model(image) -> PushPullLoss(pred_sims, labels, pred_boxes, boxes) with the
per-image label / box LISTS the reference's loop hands over (already moved to the device, ref
main.py:77-79) -> sum of 4 losses -> backward -> [one RCCL all-reduce of the flat gradient bucket] ->
AdamW.  Workload = BASELINE configs[2] (batch 32 per GPU, bf16 compute, full train step); N > 1 is
configs[3] (global batch 32*N, weak scaling, data parallel).  Synthetic COCO-shaped inputs resident in
HBM, random-init weights (no network on the box).

`python bench.py --gpus N` with N > 1 and no torchrun environment launches the N ranks itself
(torch.distributed.run, one process per GPU, RCCL); it exits non-zero if fewer than N GPUs are visible.

Prints ONE JSON line (rank 0) with the driver's contract plus:
  roofline       -- the dominant op by time (the bias-epilogue GEMM: QKV / out-proj / fc2 / dX, ~30 % of the step; one op =
                    `gemm_pp2_kernel<bias>` + its half-height remainder launch `gemm_pph_kernel<bias>` where the dispatcher
                    splits): algorithmic FLOPs per op / mean op duration, HIP events on the launch stream inside the timed
                    region; peak = 2.5 PFLOP/s dense bf16 MFMA (MI355X_MICROARCH.md)
  roofline_other -- the same measurement for the fc1 GEMM (`gemm_pp2_kernel<qgelu>`) and the fused attention forward
  cpu_baseline   -- the CPU oracle (parity-checked restatement of the reference path) timed on this box's host cores
                    (batch 1 -- the reference's own batch size --, median of >= 5 steps after 2 warm-ups), rank 0 at N = 1 only
With more than one rank the line also carries `replicas_equal` (MIN / MAX all-reduce of an exact checksum of the trainable bucket) and the
result of the schedule pre-flight (`config.schedule_check`): the deferred-tail schedule is only reported after it reproduced the in-line
schedule's losses and parameters bitwise on this very node; an in-line measurement is taken FIRST and is what the line falls back to.
"""
import argparse
import json
import os
import socket
import subprocess
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
# HBM bytes per launch of the timed kernels for the default workload (B/16, batch 32): PMC passes over this very command
# (rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE in separate passes, FETCH x2 per MI355X_MICROARCH.md "HBM"); None elsewhere
# The JSON carries the digest of the kernel sources it was measured on (tools/pmc_traffic.py); a figure taken on other sources is
# refused (traffic = null) rather than quoted: a kernel edit must not silently keep an old number.
TRAFFIC_FILE = "profiles/r06_traffic.json"
TRAFFIC_SOURCE = "profiles/r06_hbm_traffic.md"
LABEL_BIAS = "gemm op <bias> (gemm_pp2_kernel + gemm_pph_kernel remainder)"
LABEL_QGELU = "gemm_pp2_kernel<qgelu>"
LABEL_ATTN = "attn_fwd_kernel<VROW>"
LABEL_DQGELU = "gemm_pp2_kernel<dX through quick-GELU'>"
LABEL_ATTN_BWD = "attention backward (attn_dvec + attn_bwd_dkdv + attn_bwd_dq kernels)"
TRAFFIC_ALL = {}            # workload key ("<arch>/<batch per GPU>") -> {label: bytes per op}
TRAFFIC_NOTE = None


def kernel_source_digest():
    """sha256 over the HIP / C++ sources the SHIPPED libowlhip.so is built from (names + contents, sorted).  Files that are tuning-build experiments as a
    whole -- their first preprocessor line is `#ifdef OWL_TUNING` -- compile to nothing in the shipped library and are left out: editing an experiment
    does not invalidate counter passes taken on the product's kernels."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "owl-vit-object-detection_amd", "csrc")
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".hip", ".h", ".cpp")):
            data = open(os.path.join(d, fn), "rb").read()
            first = next((l.strip() for l in data.decode("utf-8", "replace").splitlines() if l.lstrip().startswith("#")), "")
            if first.startswith("#ifdef OWL_TUNING"):
                continue
            h.update(fn.encode()); h.update(data)
    return h.hexdigest()[:16]


try:
    with open(os.path.join(ROOT, TRAFFIC_FILE)) as _f:
        _t = json.load(_f)
    if _t.get("kernel_source_digest") == kernel_source_digest():
        TRAFFIC_ALL = _t.get("workloads", {})
    else:
        TRAFFIC_NOTE = f"{TRAFFIC_FILE} was measured on other kernel sources (digest {_t.get('kernel_source_digest')}): not quoted"
except (OSError, ValueError):
    TRAFFIC_NOTE = f"{TRAFFIC_FILE} missing"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU")
    ap.add_argument("--arch", default="owlvit-base-patch16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=5)
    ap.add_argument("--cpu-batch8", action="store_true", help="also time the CPU oracle at batch 8 (~20 s per step: 2 warm-ups + --cpu-steps timed steps)")
    ap.add_argument("--weights", choices=("init", "trained_like", "trained_like_hard"), default="init",
                    help="weights.make_weights profile: init = HF initialisation scales (headline); trained_like = massive residual channels, wide LayerNorm "
                         "gains, attention logits of std ~ 8 with sink keys, |sims| > 0.9 (fixture F10): the attention forward then takes its slow path")
    ap.add_argument("--forward-only", action="store_true", help="BASELINE configs[1]-style forward throughput (diagnostic)")
    ap.add_argument("--targets", choices=("lists", "packed"), default="lists",
                    help="lists = per-image label/box lists through PushPullLoss.__call__ as ref main.py:77-83 (headline); "
                         "packed = targets padded once outside the timed region (diagnostic)")
    ap.add_argument("--no-compare", action="store_true", help="skip the second (other --targets mode) measurement")
    ap.add_argument("--overlap", action="store_true", help="backward + all-reduce + AdamW on the model's tail stream under the next step's frozen prefix "
                                                           "(models.OwlViT.overlap_tail; bitwise the in-line result)")
    ap.add_argument("--no-overlap", action="store_true", help="in-line all-reduce + AdamW even with more than one rank")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="nccl = RCCL (the product path).  gloo: test-only -- lets several ranks share one visible GPU (tests/test_ddp_rccl_gpu.py runs the "
                         "multi-rank flow of this file on a 1-GPU box); never a benchmark result")
    ap.add_argument("--inject-fault", choices=("mismatch", "raise", "hang"), default=None,
                    help="test-only: make the deferred-tail phase of the multi-rank flow fail that way (tests/test_ddp_rccl_gpu.py checks that the in-line line survives)")
    ap.add_argument("--watchdog-seconds", type=float, default=None, help="limit for the deferred-tail phase (default: max(120, 40 x the in-line measurement))")
    ap.add_argument("--no-kernel-events", action="store_true", help="no HIP events around the GEMM / attention launches (overhead A/B)")
    ap.add_argument("--windows", type=int, default=5,
                    help="timed windows of --steps steps each, back to back after ONE warm-up (each bracketed by barrier + synchronize, max over ranks); "
                         "`value` / `ms_per_step` are the MEDIAN window's, config.window_values lists them all (VERDICT r05 #4: one 0.5 s window cannot resolve 2 %%)")
    ap.add_argument("--pretranspose", type=int, choices=(0, 1), default=0,
                    help="0 (default, = the product): the backward makes its weight transposes itself; 1: the forward launches them on their own stream "
                         "(models.OwlViT.pretranspose; measured no faster: A/B, profiles/r06_tail.md)")
    ap.add_argument("--dw-items", type=int, default=256, help="work items per weight-gradient GEMM (autograd.DW_ITEMS; A/B)")
    ap.add_argument("--tn-small-n", type=int, choices=(0, 1), default=1, help="1 (default): the class head's 32 x Dt prompt-gradient product on the TN kernel (autograd.TN_SMALL_N); 0: transposes + NT split-K (A/B)")
    ap.add_argument("--fold-bias", type=int, choices=(0, 1), default=1,
                    help="1 (default, = the product): bias gradients out of the dW GEMM's own pass (autograd.FOLD_BIAS_COLSUM); 0: the separate column-sum kernel (A/B)")
    ap.add_argument("--ablate", default="", help="TIMING ONLY (results are wrong, the line says so): comma list of work to skip -- colsum (bias-gradient column-sum launches; only with --fold-bias 0), "
                                                "slab_reduce (split-K reductions), transposes (the backward's weight transposes), dqhat (the class head's prompt-gradient product), loss (matcher + "
                                                "loss chain replaced by two means), castimg (images resident as bf16): upper bounds on what removing each could buy (profiles/r06_tail.md)")
    ap.add_argument("--encoder-streams", type=int, default=2,
                    help="sub-batches of the encoder forward / dX-only backward, one HIP stream each (OwlViT(encoder_streams=...)); 1 = one stream "
                         "(what the rocprofv3 profiles are taken with: kernel durations are then exclusive)")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) outside a torchrun environment: start the N ranks ourselves."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    have = torch.cuda.device_count()
    if have < args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} requested but only {have} GPU(s) are visible on this node\n")
        raise SystemExit(2)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.call(cmd, env=env))


def synth_batches(cfg, B, device, rank, n_batches=2, seed=1234):
    """CLIP-normalised uniform-u8 pixels (f32, as the reference's DataLoader yields) + 1..16 boxes/image, already on the device.
    Images, targets and weights all come from the repo's counter-based RNG (bit-identical on every box; seed + rank
    for the data, identical weights on every rank)."""
    from owl_vit_object_detection_amd import synth
    from owl_vit_object_detection_amd.matcher import PackedTargets
    out = []
    for k in range(n_batches):
        img = torch.from_numpy(synth.make_images(cfg, B, seed + rank, first=k * B)).to(device)
        labels, boxes = synth.make_targets(cfg, B, seed, first=(rank * n_batches + k) * B, max_boxes=16)
        lab_l = [torch.from_numpy(l).to(device) for l in labels]          # ref main.py:78-79: labels.to(device), boxes.to(device)
        box_l = [torch.from_numpy(b).to(device) for b in boxes]
        tg = PackedTargets(lab_l, box_l, device, cfg.n_classes)
        img_host = torch.from_numpy(synth.make_images(cfg, B, seed + rank, first=k * B)).pin_memory()      # what a DataLoader(pin_memory=True) hands over
        # the same pixels as the uint8 levels they were normalised from (host, NOT pinned: what a DataLoader of raw images yields) + host targets:
        # the feed of the product-level input stage (preprocess.DevicePrefetcher)
        u8 = torch.from_numpy(synth.make_images_u8(cfg, B, seed + rank, first=k * B, layout="hwc"))
        out.append(dict(img=img.contiguous(), img_host=img_host, packed=tg, labels=lab_l, boxes=box_l, labels_np=labels,
                        u8_host=u8, labels_host=[torch.from_numpy(l) for l in labels], boxes_host=[torch.from_numpy(b) for b in boxes]))
    return out


def _cpu_model_string():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def cpu_baseline(cfg, steps, batch8=False):
    """Time the CPU oracle (restated reference path, fp32) on this box's host cores: full train steps (fwd + matcher + loss +
    bwd + AdamW on the 29 trainable tensors) at batch 1 (the reference's own batch size; SURVEY.md 8(d)(ii) protocol:
    median of >= 5 steps after 2 warm-ups).  Batch 8 costs ~20 s per step and the oracle gains nothing from batching (0.41 against 0.48 img/s,
    profiles/r03_cpu_threads.md): that leg runs only with --cpu-batch8, at the same protocol."""
    from oracle import owl_oracle as O
    from owl_vit_object_detection_amd import synth, weights
    # SURVEY.md section 8(d)(ii) names os.cpu_count() threads; measured on the GPU box's 256 logical cores (profiles/r03_cpu_threads.md) the
    # step takes 2.23 s with 32 threads, 2.93 s with 64, 5.5 s with 128 and 58.7 s with 256: the FASTEST setting is used (the most
    # favourable one for the CPU) and the number actually used is reported
    total = os.cpu_count() or 1
    cores = min(total, 32)
    torch.set_num_threads(cores)
    w = {k: torch.from_numpy(v) for k, v in weights.make_weights(cfg).items()}
    res = {}
    steps = max(5, steps)
    for B, n_warm, n_steps in ((1, 2, steps),) + (((8, 2, steps),) if batch8 else ()):
        img = torch.from_numpy(synth.make_images(cfg, B))
        labels, boxes = synth.make_targets(cfg, B, max_boxes=16)
        lab = [torch.from_numpy(l) for l in labels]; tb = [torch.from_numpy(b) for b in boxes]
        scales = torch.from_numpy(synth.class_scales(cfg, labels))
        names = O.trainable_names(w)
        state = {n: (torch.zeros_like(w[n]), torch.zeros_like(w[n])) for n in names}

        def cpu_step(k):          # ref main.py:74-91: forward + matcher + loss + backward + AdamW (lr / wd of ref config.yaml)
            _, _, grads = O.train_step(cfg, w, img, lab, tb, scales)
            with torch.no_grad():
                for n in names:
                    m, v = state[n]
                    w[n], m, v = O.adamw_step(w[n], grads[n], m, v, k + 1, lr=3e-6, wd=0.1)
                    state[n] = (m, v)
        for k in range(n_warm):
            cpu_step(k)                                          # warm-ups
        ts = []
        for k in range(n_steps):
            t0 = time.perf_counter()
            cpu_step(n_warm + k)
            ts.append(time.perf_counter() - t0)
        res[B] = (float(np.median(ts)), len(ts), n_warm)
    m1, n1, w1 = res[1]
    out = {"value": round(1.0 / m1, 4), "unit": "images/sec", "cores": cores, "kind": "port",
           "batch1_images_per_sec": round(1.0 / m1, 4),
           "cpu": f"{_cpu_model_string()} ({total} logical cores, {cores} torch threads used)",
           "sample": f"fp32 train steps (fwd+matcher+loss+bwd+AdamW) of {cfg.name} on the CPU oracle at the reference's batch size of 1: median {m1:.2f} s/step "
                     f"over {n1} steps after {w1} warm-ups; {cores} of {total} threads = the fastest setting on this CPU (thread study: profiles/r03_cpu_threads.md); "
                     f"the reference itself: 0.32 img/s on 8 vCPUs (BASELINE.md section 2)"}
    if 8 in res:
        m8, n8, w8 = res[8]
        out["batch8_images_per_sec"] = round(8.0 / m8, 4)
        out["value"] = round(max(1.0 / m1, 8.0 / m8), 4)
        out["sample"] += f"; batch 8: median {m8:.2f} s/step over {n8} steps after {w8} warm-ups (value = the better of the two)"
    return out


EVENT_EVERY = 10


class KernelTimer:
    """HIP events around selected launches on the launch stream (= torch's current stream, which the ops enqueue on)."""

    def __init__(self):
        self.on = False
        self.rec = {}           # kernel label -> list of (event0, event1, flops)
        self.windows_recorded = 1       # timed windows whose event steps fed `rec`
        self.traffic = {}               # label -> counter bytes per op for THIS workload (profiles/r05_traffic.json), if measured on these kernel sources

    def wrap(self, orig, classify):
        def f(*a, **k):
            if not self.on:
                return orig(*a, **k)
            tag = classify(*a, **k)
            if tag is None:
                return orig(*a, **k)
            label, flops = tag
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            r = orig(*a, **k)
            e1.record()
            self.rec.setdefault(label, []).append((e0, e1, flops))
            return r
        return f

    def summary(self, label):
        ev = self.rec.get(label, [])
        if not ev:
            return None
        ms = [a.elapsed_time(b) for a, b, _ in ev]
        flops = float(np.mean([f for _, _, f in ev]))
        mean_ms = float(np.mean(ms))
        achieved = flops / (mean_ms * 1e-3) / 1e12
        return {"bound": "mfma", "kernel": label, "achieved": round(achieved, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": self.traffic.get(label),
                "traffic_source": TRAFFIC_SOURCE if self.traffic.get(label) else (TRAFFIC_NOTE or f"{TRAFFIC_FILE} holds no counter pass for this workload"),
                "launches_timed": len(ev), "ms_per_launch": round(mean_ms, 4), "gflop_per_launch": round(flops / 1e9, 2),
                "ms_total_per_step": None}


def main():
    args = parse()
    self_launch(args)
    from owl_vit_object_detection_amd import ddp, ops, weights
    from owl_vit_object_detection_amd.config import get_config
    from owl_vit_object_detection_amd.losses import PushPullLoss
    from owl_vit_object_detection_amd.models import OwlViT
    from owl_vit_object_detection_amd.optim import FusedAdamW
    from owl_vit_object_detection_amd import synth

    rank, world, local = ddp.init_from_env(args.backend)
    if args.backend == "gloo":
        local = local % max(1, torch.cuda.device_count())
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run --nproc-per-node {args.gpus}, "
                         f"or plain `python bench.py --gpus {args.gpus}`)")
    if torch.cuda.device_count() <= local:
        raise SystemExit(f"bench.py: LOCAL_RANK={local} but only {torch.cuda.device_count()} GPU(s) are visible")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cfg = get_config(args.arch)
    B = args.batch

    model = OwlViT(cfg, weights.make_weights(cfg, profile=args.weights), dev, encoder_streams=args.encoder_streams)    # identical weights on every rank (seeded)
    model.pretranspose = bool(args.pretranspose)
    from owl_vit_object_detection_amd import autograd as _autograd
    _autograd.FOLD_BIAS_COLSUM = bool(args.fold_bias)
    _autograd.TN_SMALL_N = bool(args.tn_small_n)
    _autograd.DW_ITEMS = int(args.dw_items)
    ablate = [a for a in args.ablate.split(",") if a]
    ablate_names = set(ablate)
    if ablate:
        from owl_vit_object_detection_amd import _lib as _L, autograd as _AG
        if "colsum" in ablate:
            ops.colsum_bf16 = lambda *a, **k: None
        if "transposes" in ablate:                      # the backward's weight transposes
            ops.transpose_bf16 = lambda src, dst, *a, **k: dst
        if "dqhat" in ablate:                           # the prompt-gradient product of the class head (explicit transposes + NT split-K GEMM): no-ops
            ops.transpose_colsum = lambda *a, **k: None
            _g0 = ops.gemm
            ops.gemm = lambda epi, *a, **k: (None if epi == ops.EPI_SLAB_F32 else _g0(epi, *a, **k))
        if "slab_reduce" in ablate:
            _orig_call = _L.call
            _AG._lib = type("LibNoSlabReduce", (), {"call": staticmethod(lambda name, *a: None if name == "owl_slab_reduce" else _orig_call(name, *a)),
                                                    "load": staticmethod(_L.load)})
    slow_tiles = torch.zeros(1, dtype=torch.int32, device=dev)      # attention forward: (wave, key tile) pairs that left the fast path (csrc/attention_fwd.hip)
    ops.ATTN_SLOW_TILES = slow_tiles
    batches = synth_batches(cfg, B, dev, rank)
    if "castimg" in ablate_names:
        for bt in batches:
            bt["img"] = bt["img"].to(torch.bfloat16)
    scales = synth.class_scales(cfg, [l for l in batches[0]["labels_np"]])
    crit = PushPullLoss(cfg.n_classes, scales)
    opt = FusedAdamW(model, lr=3e-6, weight_decay=0.1)            # ref config.yaml:10,12
    dist_active = world > 1 or os.environ.get("OWL_FORCE_DIST", "0") == "1"
    want_overlap = (args.overlap or world > 1) and not args.no_overlap
    dp = ddp.DataParallel(model, opt, overlap=want_overlap)
    dp.check_equal_batches(B)

    # ---- kernel timing: HIP events around the GEMM (bf16-output epilogues) and fused-attention-forward launches -------
    kt = KernelTimer()
    kt.traffic = {} if (args.forward_only or args.weights != "init") else TRAFFIC_ALL.get(f"{cfg.name}/{B}", {})

    def classify_gemm(epi, A, W, out, bias=None, resid=None, aux=None, M=None, N=None, K=None, **kw):
        if epi not in (ops.EPI_BIAS_BF16, ops.EPI_QGELU_BF16, ops.EPI_DQGELU_BF16):
            return None
        K = K if K is not None else A.shape[-1]; N = N if N is not None else W.shape[0]; M = M if M is not None else A.shape[0]
        if not (K % 128 == 0 and M >= 512 and N >= 256):
            return None                                            # not the ping-pong kernel (csrc/gemm.hip dispatch)
        label = {ops.EPI_BIAS_BF16: LABEL_BIAS, ops.EPI_QGELU_BF16: LABEL_QGELU, ops.EPI_DQGELU_BF16: LABEL_DQGELU}[epi]
        conc = kw.get("concurrency")
        if conc is not None and 2 * int(conc) * ((M + 255) // 256) * ((N + 255) // 256) <= ops.CHIP_CUS:
            label += " [small problem: half-height tiles, gemm_pph_kernel (tile 6)]"       # ops.gemm's small-problem rule (batch 1-2 alone on the chip)
        return (label, 2.0 * M * N * K)

    def classify_attn(q, k, v, ld, out, ld_out, lse, B_, H, T, Tp, scale):
        return (LABEL_ATTN, 4.0 * B_ * H * T * T * 64)   # QK^T + PV per launch

    def classify_attn_bwd(qkv, dO, O, lse, dvec, dqkv, B_, H, T, Tp, scale):
        return (LABEL_ATTN_BWD, 10.0 * B_ * H * T * T * 64)   # the algorithmic five matmuls (S, dP, dV, dK, dQ); the kernels execute seven (P recomputed in both)

    if not args.no_kernel_events:
        ops.gemm = kt.wrap(ops.gemm, classify_gemm)
        ops.attention_fwd_vrow = kt.wrap(ops.attention_fwd_vrow, classify_attn)
        ops.attention_bwd = kt.wrap(ops.attention_bwd, classify_attn_bwd)

    ar_events = []
    if world > 1 or os.environ.get("OWL_FORCE_DIST", "0") == "1":
        orig_ar = ddp.allreduce_flat

        def timed_ar(flat_grad, group=None):
            if not kt.on:
                return orig_ar(flat_grad, group)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); r = orig_ar(flat_grad, group); e1.record()
            ar_events.append((e0, e1))
            return r
        ddp.allreduce_flat = timed_ar

    h2d = [False]          # True: the step starts from f32 images in pinned HOST memory and copies them itself, as ref main.py:77 `image.to(device)` does
    feed = [None]          # an iterator of preprocess.DevicePrefetcher: the step takes images AND targets from it (host u8 -> HBM bf16, one batch ahead)

    def step(i, mode):
        bt = batches[i % len(batches)]
        if feed[0] is not None:
            img, lab_f, box_f = next(feed[0])
            opt.zero_grad()
            pred_boxes, _, pred_sims, _ = model(img)
            losses = crit(pred_sims, lab_f, pred_boxes, box_f)
            loss = losses["loss_ce"] + losses["loss_bg"] + losses["loss_bbox"] + losses["loss_giou"]
            loss.backward()
            dp.sync_and_step()
            return loss.detach()
        img = bt["img_host"].to(dev, non_blocking=True) if h2d[0] else bt["img"]
        if args.forward_only:
            with torch.no_grad():
                model(img)
            return None
        opt.zero_grad()
        pred_boxes, _, pred_sims, _ = model(img)
        if "loss" in ablate_names:
            loss = pred_boxes.mean() + pred_sims.mean()
            loss.backward()
            dp.sync_and_step()
            return loss.detach()
        if mode == "lists":
            losses = crit(pred_sims, bt["labels"], pred_boxes, bt["boxes"])       # ref main.py:83
        else:
            losses = crit(pred_sims, bt["packed"], pred_boxes)
        loss = losses["loss_ce"] + losses["loss_bg"] + losses["loss_bbox"] + losses["loss_giou"]
        loss.backward()
        dp.sync_and_step()
        return loss.detach()

    def set_schedule(overlap):
        dp.finish(); torch.cuda.synchronize()
        dp.overlap = bool(overlap)
        model.overlap_tail = bool(overlap)

    gpu_seconds = [0.0]    # sum of all timed windows of this process (the driver's smi sampling sees a multi-second busy interval)

    def timed_run(mode, steps, warmup, record, windows=1):
        """`windows` back-to-back windows of exactly `steps` steps each after ONE warm-up; every window is bracketed by barrier + synchronize on both sides
        and timed as the MAX over ranks.  Returns (median window seconds, slow-path tiles of the median-sized average window, all window seconds)."""
        for i in range(warmup):
            step(i, mode)
        dts, lohi, slows = [], [], []
        base = warmup
        for w in range(max(1, windows)):
            if world > 1:
                dist.barrier()
            dp.finish()
            torch.cuda.synchronize()
            slow_tiles.zero_()
            t0 = time.perf_counter()
            for i in range(steps):
                # Kernel events on every EVENT_EVERY-th timed step only: an event pair around each of a step's ~70 GEMM / attention launches
                # costs ~1.2 % of the step (measured A/B, --no-kernel-events).  Those steps also run the ONE-stream schedule (same kernels,
                # same bits, ~3 % slower): with two sub-batches in flight a launch shares the chip with the other stream's kernel and its
                # event-to-event time is not the kernel's own duration.  (config.one_stream_steps says how many of a window's steps these are.)
                kt.on = record and (i % EVENT_EVERY == 0)
                model.encoder_streams = 1 if kt.on else args.encoder_streams
                if dp.overlap:                   # ... and in-line, behind the previous step's deferred tail, for the same reason
                    model.overlap_tail = not kt.on
                    if kt.on:
                        model.finish()
                step(base + i, mode)
            model.encoder_streams = args.encoder_streams
            model.overlap_tail = dp.overlap
            dp.finish()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            dt = time.perf_counter() - t0
            kt.on = False
            base += steps
            lo_hi = (dt, dt)                                   # (fastest, slowest) rank of this window; the reported time is the slowest's
            if dist_active and dist.is_initialized():
                t = torch.tensor([dt, -dt], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t[0].item())
                lo_hi = (-float(t[1].item()), dt)
            dts.append(dt); lohi.append(lo_hi); slows.append(int(slow_tiles.item()))
            gpu_seconds[0] += dt
        k = int(np.argsort(dts)[len(dts) // 2])                # the median window (an actual window, not an average)
        timed_run.rank_seconds = lohi[k]
        timed_run.window_seconds = list(dts)
        timed_run.one_stream_steps = len(range(0, steps, EVENT_EVERY)) if record else 0
        return dts[k], slows[k]

    # ---- more than one rank (or one forced rank): make the run unloseable ---------------------------------------------
    # The deferred-tail schedule (backward + RCCL + AdamW on the model's tail stream under the next forward's frozen prefix) is the faster one with
    # peers, but an 8-GPU node is the first place it meets RCCL with more than one rank.  So: (1) measure the IN-LINE schedule first -- RCCL on the
    # compute stream, the plain usage -- and keep its line; (2) pre-flight: the same two steps from the same state in-line and deferred must give the
    # same losses and parameters BITWISE (tests/test_ddp_rccl_gpu.py does this over gloo); (3) only then measure the deferred schedule, under a
    # watchdog that prints the in-line line and exits if the deferred phase does not come back.  `replicas_equal` is checked after each phase.
    def replicas_equal():
        if not dist_active:
            return None
        # two integer checksums of the parameter BITS (exact arithmetic in int64: no rounding) -- the plain sum, which a pair of compensating differences
        # could leave unchanged, and a position-weighted sum, which the same pair cannot also leave unchanged; MIN and MAX over ranks must agree
        bits = model.flat_param.view(torch.int32).to(torch.int64)
        wgt = (torch.arange(bits.numel(), device=bits.device, dtype=torch.int64) % 1000003) + 1
        c = torch.stack([bits.sum(), (bits * wgt).sum()])
        t = torch.cat([c, -c])
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(int(t[0]) == -int(t[2]) and int(t[1]) == -int(t[3]))

    def snapshot():
        dp.finish(); torch.cuda.synchronize()
        return (model.flat_param.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(), opt.step_count)

    def restore(st):
        dp.finish(); torch.cuda.synchronize()
        model.flat_param.copy_(st[0]); opt.exp_avg.copy_(st[1]); opt.exp_avg_sq.copy_(st[2]); opt.step_count = st[3]
        model.flat_grad.zero_(); model._grad_clean = False
        model.refresh_compute_weights(force=True)
        torch.cuda.synchronize()

    def preflight(mode):
        st = snapshot()
        res = {}
        for overlap in (False, True):
            set_schedule(overlap)
            losses = [step(k, mode) for k in range(2)]
            dp.finish(); torch.cuda.synchronize()
            res[overlap] = (torch.stack(losses).cpu(), model.flat_param.clone())
            restore(st)
        same = torch.equal(res[False][0], res[True][0]) and torch.equal(res[False][1], res[True][1])
        if dist_active:                                       # every rank must reach the same verdict
            t = torch.tensor([1 if same else 0], device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            same = bool(int(t[0]))
        return same

    def report(dt, slow, schedule, extra_cfg):
        ms = dt / args.steps * 1e3
        value = B * world * args.steps / dt
        flops_img = cfg.flops_forward() if args.forward_only else cfg.flops_train_step()
        out = {
            "metric": ("forward" if args.forward_only else "train") + " images/sec, "
                      + ("OWL-ViT-B/16 768x768" if cfg.name == "owlvit-base-patch16" else f"{cfg.name} {cfg.image_size}x{cfg.image_size}"),
            "value": round(value, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic (CLIP-normalised uniform-u8 pixels, 1-16 boxes/image, 10 classes; "
                                     + ("random-init weights)" if args.weights == "init" else f"random weights reshaped to trained-like statistics, weights.make_weights(profile='{args.weights}'))"),
            "config": {"workload": f"{cfg.name} bf16, batch={B}/GPU, {cfg.image_size}x{cfg.image_size}, "
                                   + ("forward only" if args.forward_only else "full train step (matcher+loss+backward+AdamW)")
                                   + (f", DDP over {world} GPUs, one RCCL all-reduce of the flat grad bucket/step" if world > 1 else ""),
                       "global_batch": B * world, "tokens": cfg.tokens, "parallelism": f"dp{world}",
                       "targets": args.targets + (" (per-image label/box lists through PushPullLoss.__call__, ref main.py:77-83)" if args.targets == "lists" else " (pre-padded)"),
                       "encoder_streams": args.encoder_streams,
                       "weights": args.weights,
                       "attention_slow_tiles_per_step": round(slow / max(1, args.steps), 1),
                       "optimizer_schedule": schedule,
                       **({"ABLATION_timing_only_results_wrong": args.ablate} if args.ablate else {}),
                       "gflop_per_image": round(flops_img / 1e9, 1),
                       "step_mfma_frac": round(flops_img * value / world / 1e12 / PEAK_BF16_TFLOPS, 4),
                       # `value` / `ms_per_step` = the MEDIAN of `windows` back-to-back windows of exactly `steps` steps (one warm-up in front of the first)
                       "windows": len(timed_run.window_seconds),
                       "window_values": [round(B * world * args.steps / d, 2) for d in timed_run.window_seconds],
                       "window_spread_pct": round(100.0 * (max(timed_run.window_seconds) - min(timed_run.window_seconds)) / dt, 2),
                       "one_stream_steps": f"{timed_run.one_stream_steps} of each window's {args.steps} steps run the one-stream schedule with HIP-event pairs around the "
                                           f"GEMM / attention launches (the roofline figures come from them; ~3 % slower than the other steps: `value` is conservative by ~"
                                           f"{round(3.0 * timed_run.one_stream_steps / max(1, args.steps), 2)} %)"},
            "rccl_ranks": dist.get_world_size() if dist.is_initialized() else 1,
            "backend": (dist.get_backend() + (" (RCCL over xGMI)" if dist.get_backend() == "nccl" else " (TEST ONLY: not a benchmark)")) if dist.is_initialized() else "none (single process)",
        }
        out["config"].update(extra_cfg)
        if dist_active:
            # read back from the communicator itself, so that the first real multi-GPU run documents what it ran on
            ones = torch.ones(1, device=dev, dtype=torch.int32)
            dist.all_reduce(ones)
            out["nranks_seen"] = int(ones.item())          # ranks that actually took part in a collective on this group
            try:
                out["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version()) if dist.get_backend() == "nccl" else None
            except Exception as e:                          # (never lose the line to a version query)
                out["rccl_version"] = f"unavailable ({type(e).__name__})"
            out["devices_visible"] = torch.cuda.device_count()
            lo, hi = timed_run.rank_seconds                # per-rank throughput of THIS run: the slowest rank sets `value`
            out["images_per_sec_per_rank_min"] = round(B * args.steps / hi, 2)
            out["images_per_sec_per_rank_max"] = round(B * args.steps / lo, 2)
            out["schedule"] = schedule
        # HIP-event fields (roofline*, allreduce_ms) are recorded during ONE phase only -- the in-line one when both schedules are measured (an event
        # pair around a launch on the tail stream would time the other stream's kernels too): the line says which
        events_phase = "in-line schedule" if (schedule == "in-line" or events_from_inline[0]) else schedule
        if ar_events:
            out["allreduce_ms"] = round(float(np.mean([a.elapsed_time(b) for a, b in ar_events])), 4)
            out["allreduce_bytes"] = int(model.flat_numel * 4)
            out["allreduce_measured_under"] = events_phase
        main_r, others = None, []
        for label in (LABEL_BIAS, LABEL_QGELU, LABEL_ATTN, LABEL_DQGELU, LABEL_ATTN_BWD):     # (the last two: the kernels furthest below their roofline)
            r = kt.summary(label)
            if r is None:                                   # batch 1-2: the event steps' GEMMs take the small-problem rule and carry its suffix (classify_gemm)
                r = next((kt.summary(l) for l in kt.rec if l.startswith(label + " [")), None)
            if r is None:
                continue
            r["ms_total_per_step"] = round(r["ms_per_launch"] * r["launches_timed"] / (len(range(0, args.steps, EVENT_EVERY)) * kt.windows_recorded), 3)
            r["measured_on"] = (f"every {EVENT_EVERY}th timed step, which runs the one-stream schedule (whole-batch launches, the kernel alone on the chip); "
                                f"the other steps run {args.encoder_streams} sub-batch streams")
            if dist_active:
                r["measured_under"] = events_phase
            if main_r is None:
                main_r = r
            else:
                others.append(r)
        if main_r is not None:
            out["roofline"] = main_r
            out["roofline_other"] = others
        return out

    events_from_inline = [False]      # set once the in-line phase of the two-schedule flow has recorded the HIP events the deferred line re-uses
    SCHED_TAIL = "backward + all-reduce + AdamW on the tail stream under the next step's frozen prefix (bitwise the in-line schedule)"
    extra = {}
    line = None
    if dist_active and want_overlap and not args.forward_only:
        import threading
        set_schedule(False)
        kt.windows_recorded = args.windows
        dt_in, slow_in = timed_run(args.targets, args.steps, args.warmup, True, windows=args.windows)
        eq_in = replicas_equal()
        extra["images_per_sec_inline_schedule"] = round(B * world * args.steps / dt_in, 2)
        line_inline = report(dt_in, slow_in, "in-line", dict(extra))
        line_inline["replicas_equal"] = eq_in
        line_inline["deferred_phase"] = "not run"
        events_from_inline[0] = True

        def bail():            # the deferred phase did not come back: the in-line measurement is the result
            if rank == 0:
                line_inline["config"]["schedule_check"] = f"deferred-tail phase did not finish within {int(limit)} s: in-line schedule reported"
                line_inline["deferred_phase"] = "watchdog fired"
                print(json.dumps(line_inline), flush=True)
            os._exit(0)
        limit = args.watchdog_seconds if args.watchdog_seconds else max(120.0, 40.0 * dt_in)
        dog = threading.Timer(limit, bail); dog.daemon = True; dog.start()
        try:
            ok = preflight(args.targets)
            if args.inject_fault == "mismatch":
                ok = False
            elif args.inject_fault == "raise":
                raise RuntimeError("injected fault (test)")
            elif args.inject_fault == "hang":
                time.sleep(limit + 30.0)
            if ok:
                set_schedule(True)
                dt, slow = timed_run(args.targets, args.steps, args.warmup, False, windows=args.windows)
                eq = replicas_equal()
                if eq is False or eq_in is False:
                    ok = False
            dog.cancel()
        except Exception as e:                      # (a collective that failed on the side stream: the process group may be unusable -- report and leave)
            dog.cancel()
            if rank == 0:
                line_inline["config"]["schedule_check"] = f"deferred-tail phase raised {type(e).__name__}: {str(e)[:200]} -- in-line schedule reported"
                line_inline["deferred_phase"] = "raised"
                print(json.dumps(line_inline), flush=True)
            os._exit(0)
        if ok:
            extra["schedule_check"] = "pre-flight: 2 steps from one state, deferred tail == in-line bitwise (losses and parameters); in-line measured first"
            extra["images_per_sec_deferred_tail_schedule"] = round(B * world * args.steps / dt, 2)
            line = report(dt, slow, SCHED_TAIL, extra)
            line["replicas_equal"] = bool(eq and eq_in)
            line["deferred_phase"] = "ok"
        else:
            set_schedule(False)
            line_inline["config"]["schedule_check"] = "pre-flight MISMATCH between the deferred-tail and the in-line schedule (or replicas diverged): in-line schedule reported"
            line_inline["deferred_phase"] = "mismatch"
            line = line_inline
    else:
        set_schedule(want_overlap and not args.forward_only and model.flat_grad.is_cuda)
        kt.windows_recorded = args.windows
        dt, slow = timed_run(args.targets, args.steps, args.warmup, True, windows=args.windows)
        line = report(dt, slow, SCHED_TAIL if dp.overlap else "in-line", extra)
        if dist_active:
            line["replicas_equal"] = replicas_equal()
    if not args.forward_only and not args.no_compare:
        other_mode = "packed" if args.targets == "lists" else "lists"
        dt_o, _ = timed_run(other_mode, args.steps, 1, False)
        line["config"]["images_per_sec_" + other_mode + "_targets"] = round(B * world * args.steps / dt_o, 2)

    if not args.no_compare and world == 1:      # (a one-GPU diagnostic: the multi-rank flow stays exactly what the gloo / RCCL tests exercise)
        # the same step with the host-to-device copy of the images INSIDE it (ref main.py:77; 7.08 MB f32 per B/16 image over PCIe): reported beside the
        # HBM-resident headline, never as `value`
        h2d[0] = True
        dt_h, _ = timed_run(args.targets, args.steps, 1, False)
        h2d[0] = False
        line["config"]["images_per_sec_incl_h2d"] = round(B * world * args.steps / dt_h, 2)
        line["config"]["h2d_note"] = ("images_per_sec_incl_h2d: f32 images in pinned host memory, image.to(device, non_blocking=True) inside the timed step in front of the "
                                      "forward -- the reference's own loop (ref main.py:77)")
        if not args.forward_only:
            # The PRODUCT's input stage: preprocess.DevicePrefetcher around a host loader -- uint8 pixels and host targets in, bf16 [B,3,S,S]
            # and device targets out, one batch ahead on a copy stream.  Same pixels as the HBM-resident headline (every one of the 768 table values rounds to
            # the same bf16 as synth.make_images' f64 formula), so the losses must agree bitwise with the resident step's -- checked below.
            from owl_vit_object_detection_amd.preprocess import DevicePrefetcher

            def host_loader(form):
                k = 0
                while True:
                    bt = batches[k % len(batches)]; k += 1
                    if form == "u8":
                        yield bt["u8_host"], bt["labels_host"], bt["boxes_host"]                       # [B,S,S,3] uint8, pageable
                    elif form == "u8_coco":
                        yield bt["u8_coco"], bt["labels_host"], bt["boxes_host"]                       # [B,480,640,3] uint8: resized on the device (Pillow-exact)
                    else:
                        yield bt["img_host"], bt["labels_host"], bt["boxes_host"]                      # f32 pixel_values, pinned (the reference DataLoader's form)

            def product_run(form, windows):
                pf = DevicePrefetcher(host_loader(form), dev, size=cfg.image_size, dtype=torch.bfloat16, depth=2)
                feed[0] = iter(pf)
                try:
                    dt_f, _ = timed_run(args.targets, args.steps, 2, False, windows=windows)
                    vals = [round(B * world * args.steps / d, 2) for d in timed_run.window_seconds]
                finally:
                    feed[0] = None
                    pf.close()
                return dt_f, vals, pf

            # bitwise check first (one step from one state, resident vs fed), then the measurements
            st = snapshot()
            l_res = step(0, args.targets); torch.cuda.synchronize(); p_res = model.flat_param.clone()
            restore(st)
            pf = DevicePrefetcher(host_loader("u8"), dev, size=cfg.image_size, dtype=torch.bfloat16, depth=1, threaded=False)
            feed[0] = iter(pf)
            l_fed = step(0, args.targets); torch.cuda.synchronize(); p_fed = model.flat_param.clone()
            feed[0] = None; pf.close()
            restore(st)
            same = bool(torch.equal(l_res, l_fed) and torch.equal(p_res, p_fed))
            gcoco = torch.Generator().manual_seed(7)
            for bt in batches:
                bt["u8_coco"] = torch.randint(0, 256, (B, 480, 640, 3), generator=gcoco, dtype=torch.uint8)
            dt_u, vals_u, pf_u = product_run("u8", 3)
            dt_c, vals_c, pf_c = product_run("u8_coco", 1)
            dt_f, vals_f, pf_f = product_run("f32", 1)
            resident = line["value"]
            c = line["config"]
            c["images_per_sec_from_host_u8_product"] = round(B * world * args.steps / dt_u, 2)
            c["from_host_u8_product_vs_resident"] = round(B * world * args.steps / dt_u / resident, 4)
            c["from_host_u8_product_windows"] = vals_u
            c["from_host_u8_product_bitwise_equals_resident_step"] = same
            c["from_host_u8_product_bytes_per_image"] = int(pf_u.bytes_h2d / max(1, pf_u.batches) / B)
            c["from_host_u8_product_host_ms_per_batch"] = round(1e3 * pf_u.stage_seconds / max(1, pf_u.batches), 2)
            c["from_host_u8_coco_size_host_ms_per_batch"] = round(1e3 * pf_c.stage_seconds / max(1, pf_c.batches), 2)
            c["images_per_sec_from_host_u8_coco_size_product"] = round(B * world * args.steps / dt_c, 2)
            c["from_host_u8_coco_size_bytes_per_image"] = int(pf_c.bytes_h2d / max(1, pf_c.batches) / B)
            c["images_per_sec_from_host_f32_product"] = round(B * world * args.steps / dt_f, 2)
            c["from_host_product_note"] = ("preprocess.DevicePrefetcher(host loader) feeding the SAME train step: uint8 [B,S,S,3] pageable host pixels + host label / box lists -> "
                                           "pinned ring -> copy stream -> owl_normalize_u8 (reference table) -> bf16, one batch ahead (3 windows, median); `_coco_size_`: 480x640 "
                                           "uint8 images resized on the device (Pillow-exact bicubic, owl_preprocess_u8_batch); `_f32_`: the reference DataLoader's pinned f32 "
                                           "pixel_values, copied + cast one batch ahead.  Never `value`: the headline stays HBM-resident")
    line["config"]["gpu_seconds"] = round(gpu_seconds[0], 2)

    ops.ATTN_SLOW_TILES = None          # (process-global statistic hook: not left armed behind the measurement)
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg, args.cpu_steps, args.cpu_batch8)
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
