"""Round 6, VERDICT r05 #6a: attention backward's dK/dV kernel and dQ kernel only share inputs (and write disjoint column thirds of dqkv).  Does running them
CONCURRENTLY (two streams behind the dvec pass) beat running them back to back?  Same kernels either way (bitwise the same dqkv, checked).
  serial     : dvec -> dK/dV -> dQ on one stream                           (what ships)
  concurrent : dvec ; fork ; dK/dV on the stream, dQ on a side stream ; join
  two images : two sub-batches as the L/14 backward runs them (two streams, each its own serial chain) against the same two sub-batches with the dQ / dK/dV
               launches of the pair crossed (stream 1: dvec, dK/dV(a), dQ(b) ... ) -- the verdict's "different sub-batches" form
Alternating timed repetitions, HIP events on the launch stream around the whole group, median."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from owl_vit_object_detection_amd import ops
DEV = "cuda"


def make(B, H, T, seed):
    torch.manual_seed(seed)
    Tp = (T + 7) // 8 * 8; D = H * 64; M = B * Tp
    qkv = torch.zeros(ops.pad_rows(M), 3 * D, device=DEV, dtype=torch.bfloat16); qkv[:M] = torch.randn(M, 3 * D, device=DEV).bfloat16()
    o = torch.zeros(ops.pad_rows(M), D, device=DEV, dtype=torch.bfloat16); lse = torch.zeros(B, H, Tp, device=DEV)
    ops.attention_fwd_vrow(qkv, qkv[:, D:], qkv[:, 2 * D:], 3 * D, o, D, lse, B, H, T, Tp, 0.125)
    do = torch.zeros_like(o); do[:M] = (torch.randn(M, D, device=DEV) * 0.1).bfloat16()
    return dict(qkv=qkv, o=o, lse=lse, do=do, dvec=torch.zeros(B, H, Tp, device=DEV), dqkv=torch.zeros_like(qkv), B=B, H=H, T=T, Tp=Tp)


def call(a, phases):
    ops.attention_bwd(a["qkv"], a["do"], a["o"], a["lse"], a["dvec"], a["dqkv"], a["B"], a["H"], a["T"], a["Tp"], 0.125, phases=phases)


def timed(fn, reps=15):
    ms = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    return float(np.median(ms))


def run(B, H, T, tag):
    side = torch.cuda.Stream(); main = torch.cuda.current_stream()
    ev, ev2 = torch.cuda.Event(), torch.cuda.Event()
    a = make(B, H, T, 1)

    def serial():
        call(a, 0)

    def concurrent():
        call(a, 1)
        ev.record(main); side.wait_event(ev)
        with torch.cuda.stream(side):
            call(a, 4)
            ev2.record(side)
        call(a, 2)
        main.wait_event(ev2)

    serial(); torch.cuda.synchronize(); ref = a["dqkv"].clone(); a["dqkv"].zero_()
    concurrent(); torch.cuda.synchronize(); same = torch.equal(ref, a["dqkv"])
    res = {"serial": [], "concurrent": []}
    for rnd in range(3):
        res["serial"].append(timed(serial)); res["concurrent"].append(timed(concurrent))
    s, c = float(np.median(res["serial"])), float(np.median(res["concurrent"]))
    print(f"{tag} B={B} H={H} T={T}: serial {s:.3f} ms, dK/dV || dQ {c:.3f} ms ({100 * (c / s - 1):+.1f} %), bitwise equal: {same}   rounds {res}", flush=True)
    # two sub-batches on two streams (the L/14 backward's schedule): plain, and with the second stream's order crossed (dQ first) so that the pair in flight is mixed
    b1, b2 = make(B // 2, H, T, 2), make(B - B // 2, H, T, 3)

    def two_plain():
        ev.record(main); side.wait_event(ev)
        with torch.cuda.stream(side):
            call(b2, 0); ev2.record(side)
        call(b1, 0)
        main.wait_event(ev2)

    def two_crossed():
        ev.record(main); side.wait_event(ev)
        with torch.cuda.stream(side):
            call(b2, 1); call(b2, 4); call(b2, 2); ev2.record(side)        # dQ before dK/dV on this stream: the other stream runs dK/dV first
        call(b1, 0)
        main.wait_event(ev2)

    r2 = {"plain": [], "crossed": []}
    for rnd in range(3):
        r2["plain"].append(timed(two_plain)); r2["crossed"].append(timed(two_crossed))
    p_, x_ = float(np.median(r2["plain"])), float(np.median(r2["crossed"]))
    print(f"{tag} two sub-batches on two streams: same order {p_:.3f} ms, crossed order {x_:.3f} ms ({100 * (x_ / p_ - 1):+.1f} %); whole batch serial {s:.3f} ms   rounds {r2}", flush=True)


run(32, 12, 2305, "B/16 batch 32")
run(16, 16, 3601, "L/14 batch 16")
