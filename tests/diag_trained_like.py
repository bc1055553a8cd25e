"""Diagnostic (run by hand on the GPU box, not collected by pytest): where does the forward error of the HIP path come from on trained-like weights?
Residual stream after every encoder layer, feats, box-head stages and outputs against the CPU oracle (checker), split by token class and channel class.
    python tests/diag_trained_like.py [arch] [profile]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if not os.path.exists(os.path.join(ROOT, "owl_vit_object_detection_amd")):
    os.symlink("owl-vit-object-detection_amd", os.path.join(ROOT, "owl_vit_object_detection_amd"))

from oracle import owl_oracle as O  # noqa: E402  (checker)
from owl_vit_object_detection_amd import synth, weights  # noqa: E402
from owl_vit_object_detection_amd.config import get_config  # noqa: E402
from owl_vit_object_detection_amd.models import OwlViT  # noqa: E402


def main():
    arch = sys.argv[1] if len(sys.argv) > 1 else "owlvit-base-patch16"
    profile = sys.argv[2] if len(sys.argv) > 2 else "trained_like"
    cfg = get_config(arch)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    Wnp = weights.make_weights(cfg, profile=profile)
    img = synth.make_images(cfg, 1)
    w = {k: torch.from_numpy(v) for k, v in Wnp.items()}
    taps = {}
    rb, rs = O.model_forward(cfg, w, torch.from_numpy(img), taps)
    model = OwlViT(cfg, Wnp, "cuda", encoder_streams=1)
    T, D, P = cfg.tokens, cfg.hidden, cfg.patches
    caps = []
    orig = OwlViT._encoder_layer

    def hooked(self, i, ws, B, save, st):
        orig(self, i, ws, B, save, st)
        x = st["xs"][:T].float().clone()
        if st["pending1"] is not None:
            x = x + st["pending1"][:T].float()
        x = x + st["pending"][:T].float()
        caps.append(x.cpu())
    OwlViT._encoder_layer = hooked
    with torch.no_grad():
        pb, _, ps, _ = model(torch.from_numpy(img).cuda())
    torch.cuda.synchronize()
    OwlViT._encoder_layer = orig
    t = weights.TRAINED_LIKE
    ch = [int(f * D) for f in t["massive_channels"]]
    sinks = [1 + min(P - 1, int(f * P)) for f in t["sink_tokens"]]
    normal_ch = np.setdiff1d(np.arange(D), ch)
    normal_tok = np.setdiff1d(np.arange(T), sinks)
    print(f"{arch} / {profile}: massive channels {ch}, sink tokens {sinks}")
    for i, x in enumerate(caps):
        ref = taps[f"backbone.encoder.layers.{i}.out"][0]
        d = (x - ref).abs()
        print(f"layer {i:2d} out: |ref| rms {float(ref.pow(2).mean().sqrt()):7.3f} | err max {float(d.max()):.3e} rms {float(d.pow(2).mean().sqrt()):.3e}"
              f" | normal ch/tok: max {float(d[normal_tok][:, normal_ch].max()):.3e} rms {float(d[normal_tok][:, normal_ch].pow(2).mean().sqrt()):.3e}"
              f" | massive ch: max {float(d[:, ch].max()):.3e} | sink tok: max {float(d[sinks].max()):.3e}")
    ws = model._ws[("eval", 1)]
    feats = ws["feats"][:P].float().cpu()
    rf = taps["feats"][0]
    d = (feats - rf).abs()
    print(f"feats: |ref| rms {float(rf.pow(2).mean().sqrt()):.3f} max {float(rf.abs().max()):.2f} | err max {float(d.max()):.3e} rms {float(d.pow(2).mean().sqrt()):.3e};"
          f" per-patch rms err: median {float(d.pow(2).mean(1).sqrt().median()):.3e} max {float(d.pow(2).mean(1).sqrt().max()):.3e} at patch {int(d.pow(2).mean(1).argmax())}")
    # box head stages on the oracle side
    F = torch.nn.functional
    h0 = F.gelu(F.linear(rf, w["box_head.dense0.weight"], w["box_head.dense0.bias"]))
    h1 = F.gelu(F.linear(h0, w["box_head.dense1.weight"], w["box_head.dense1.bias"]))
    pre = F.linear(h1, w["box_head.dense2.weight"], w["box_head.dense2.bias"])
    for name, ours, ref in (("hb0", ws["hb0"][:P].float().cpu(), h0), ("hb1", ws["hb1"][:P].float().cpu(), h1)):
        d = (ours - ref).abs()
        print(f"{name}: |ref| rms {float(ref.pow(2).mean().sqrt()):.3f} max {float(ref.abs().max()):.2f} | err max {float(d.max()):.3e} rms {float(d.pow(2).mean().sqrt()):.3e}")
    # what the box head alone does to the ORACLE's feats when its inputs / intermediates are rounded to bf16 (no HIP involved)
    bf = lambda x: x.bfloat16().float()
    W0, W1 = bf(w["box_head.dense0.weight"]), bf(w["box_head.dense1.weight"])
    g0 = bf(F.gelu(F.linear(bf(rf), W0, w["box_head.dense0.bias"])))
    g1 = bf(F.gelu(F.linear(g0, W1, w["box_head.dense1.bias"])))
    pre_bf = F.linear(g1, w["box_head.dense2.weight"], w["box_head.dense2.bias"])
    print(f"pre-sigmoid box logits: |ref| rms {float(pre.pow(2).mean().sqrt()):.3f} max {float(pre.abs().max()):.2f}; bf16-rounded box head on EXACT feats: err max {float((pre_bf - pre).abs().max()):.3e}"
          f" rms {float((pre_bf - pre).pow(2).mean().sqrt()):.3e}")
    db = (pb[0].cpu() - rb[0]).abs()
    worst = int(db.max(1).values.argmax())
    print(f"boxes: err max {float(db.max()):.3e} rms {float(db.pow(2).mean().sqrt()):.3e}; worst patch {worst} (token {worst + 1}): ours {pb[0, worst].cpu().numpy().round(4)} ref {rb[0, worst].numpy().round(4)}")
    ds = (ps[0].cpu() - rs[0]).abs()
    print(f"sims: err max {float(ds.max()):.3e} rms {float(ds.pow(2).mean().sqrt()):.3e}")
    # bf16 STORAGE alone (tests/bf16_emulation.py): the same oracle with the HIP path's rounding points
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from bf16_emulation import model_forward_bf16_storage
    etaps = {}
    eb, es = model_forward_bf16_storage(cfg, w, torch.from_numpy(img), etaps)
    for i in (0, 5, 11):
        ref = taps[f"backbone.encoder.layers.{i}.out"][0]
        print(f"layer {i:2d} out: bf16-storage emulation vs fp32 reference rms {float((etaps[f'backbone.encoder.layers.{i}.out'][0] - ref).pow(2).mean().sqrt()):.3e};"
              f" HIP vs emulation rms {float((caps[i] - etaps[f'backbone.encoder.layers.{i}.out'][0]).pow(2).mean().sqrt()):.3e}")
    de = (eb[0] - rb[0]).abs(); dh = (pb[0].cpu() - eb[0]).abs()
    print(f"boxes: emulation vs fp32 reference max {float(de.max()):.3e} rms {float(de.pow(2).mean().sqrt()):.3e} | HIP vs emulation max {float(dh.max()):.3e} rms {float(dh.pow(2).mean().sqrt()):.3e}")
    de = (es[0] - rs[0]).abs(); dh = (ps[0].cpu() - es[0]).abs()
    print(f"sims:  emulation vs fp32 reference max {float(de.max()):.3e} rms {float(de.pow(2).mean().sqrt()):.3e} | HIP vs emulation max {float(dh.max()):.3e} rms {float(dh.pow(2).mean().sqrt()):.3e}")
    q = torch.quantile(db.max(1).values, torch.tensor([0.5, 0.9, 0.99, 0.999]))
    print("per-patch max box error quantiles 50/90/99/99.9 %:", q.numpy().round(5))


if __name__ == "__main__":
    main()
