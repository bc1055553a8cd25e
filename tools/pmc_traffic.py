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
    out = {}
    for k, n, rd, wr in rows:
        if re.match(r"void gemm_pp2_kernel<0[,>]", k) and "gemm_pp2_kernel<bias>" not in out: out["gemm_pp2_kernel<bias>"] = round(rd + wr)   # (rows are sorted by total traffic: the shipped instantiation first)
        elif re.match(r"void gemm_pp2_kernel<1[,>]", k) and "gemm_pp2_kernel<qgelu>" not in out: out["gemm_pp2_kernel<qgelu>"] = round(rd + wr)
        elif re.match(r"void attn_fwd_kernel<true, true[,>]", k) and "12>" not in k: out["attn_fwd_kernel<VROW>"] = round(rd + wr)        # (class token peeled: what T = 1 + 64 n runs)
        elif re.match(r"void attn_fwd_kernel<true, false[,>]", k) and "attn_fwd_kernel<VROW>" not in out: out["attn_fwd_kernel<VROW>"] = round(rd + wr)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import kernel_source_digest            # bench.py quotes these figures only for the sources they were measured on
    out["kernel_source_digest"] = kernel_source_digest()
    json.dump(out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
