"""HBM traffic per launch and kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over the same command.
Usage: pmc_traffic.py <dir with FETCH_SIZE pass> <dir with WRITE_SIZE pass> [--json out.json]
Units and corrections as MI355X_MICROARCH.md "HBM" prescribes: both counters are in KiB; on gfx950 FETCH_SIZE reports half the bytes of
a wide coalesced streaming read (128-B requests tallied at 64 B) -> x2 on the read side (an upper bound for kernels with narrow reads)."""
import csv, glob, json, os, re, sys
from collections import defaultdict


def load(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            acc[re.sub(r"\(.*", "", row["Kernel_Name"]).strip()].append(float(row["Counter_Value"]))
    return acc


fetch = load(sys.argv[1], "FETCH_SIZE")
write = load(sys.argv[2], "WRITE_SIZE")
rows = []
for k in fetch:
    n = len(fetch[k])
    rd = sum(fetch[k]) / n * 1024 * 2
    wr = sum(write.get(k, [0])) / max(1, len(write.get(k, [0]))) * 1024
    rows.append((k, n, rd, wr))
rows.sort(key=lambda r: -(r[2] + r[3]) * r[1])
print("| kernel | launches | read MB / launch (FETCH_SIZE x2) | written MB / launch | total MB / launch |\n|---|---|---|---|---|")
for k, n, rd, wr in rows[:40]:
    print(f"| {k[:80]} | {n} | {rd/1e6:.1f} | {wr/1e6:.1f} | {(rd+wr)/1e6:.1f} |")
if "--json" in sys.argv:
    # bench.py's `roofline.traffic`: bytes per OP of the three timed ops for one workload ("<arch>/<batch per GPU>", --workload), merged into the JSON if it
    # was written on the same kernel sources (else the file starts afresh: a figure measured on other sources is never kept)
    def per_op(pat, extra_pat=None):
        n = sum(n_ for k, n_, rd, wr in rows if re.match(pat, k))
        tot = sum((rd + wr) * n_ for k, n_, rd, wr in rows if re.match(pat, k) or (extra_pat and re.match(extra_pat, k)))
        return round(tot / n) if n else None
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import kernel_source_digest, LABEL_BIAS, LABEL_QGELU, LABEL_ATTN
    w = {LABEL_BIAS: per_op(r"void gemm_pp2_kernel<0[,>]", r"void gemm_pph_kernel<0[,>]"),        # the remainder launch belongs to the op
         LABEL_QGELU: per_op(r"void gemm_pp2_kernel<1[,>]", r"void gemm_pph_kernel<1[,>]"),
         LABEL_ATTN: per_op(r"void attn_fwd_kernel<true, (true|false), 4>")}
    path = sys.argv[sys.argv.index("--json") + 1]
    key = sys.argv[sys.argv.index("--workload") + 1] if "--workload" in sys.argv else "owlvit-base-patch16/32"
    out = {"kernel_source_digest": kernel_source_digest(), "workloads": {}}
    try:
        old = json.load(open(path))
        if old.get("kernel_source_digest") == out["kernel_source_digest"]:
            out["workloads"] = old.get("workloads", {})
    except (OSError, ValueError):
        pass
    out["workloads"][key] = {k: v for k, v in w.items() if v}
    json.dump(out, open(path, "w"), indent=1)
