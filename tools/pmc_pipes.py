"""Matrix-pipe / VALU busy share and effective clock per kernel from rocprofv3 --pmc passes over bench.py (profiles/r03_pmc.md).
Usage: pmc_pipes.py <pass dir> [<pass dir> ...]
Every pass must carry GRBM_GUI_ACTIVE (the cycle base of that pass) next to its SQ counters; the counter csv has the dispatch's own
start / end timestamps, so the effective clock = GRBM_GUI_ACTIVE / 8 XCDs / (end - start) comes from the SAME profiled launch
(MI355X_MICROARCH.md "DVFS give-back": never mix a profiled and an un-profiled arm).
Units (same guide, per-instruction table): SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_*
count quad-cycles; SQ_BUSY_CYCLES is per shader engine (x32)."""
import csv, glob, os, re, sys
from collections import defaultdict

KEEP = ("gemm_", "attn_", "ln_", "merge_ln", "colsum", "slab_reduce", "class_sims", "box_final", "adamw", "hungarian")
XCDS, SIMDS = 8, 1024


def short(name):
    n = re.sub(r"^void ", "", name)
    n = re.sub(r"\(.*", "", n)
    return n.strip()


per = {}   # kernel -> counter -> list of (value, dur_ns, gui)
for d in sys.argv[1:]:
    rows = defaultdict(dict)    # dispatch id -> {counter: value, "_k": name, "_t": dur}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if not k.startswith(KEEP):
                continue
            e = rows[(f, r["Dispatch_Id"])]
            e["_k"] = k
            e["_t"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    for e in rows.values():
        if "GRBM_GUI_ACTIVE" not in e:
            continue
        acc = per.setdefault(e["_k"], defaultdict(list))
        cyc = e["GRBM_GUI_ACTIVE"] / XCDS
        acc["_cyc"].append(cyc)
        acc["_ns"].append(e["_t"])
        for c, v in e.items():
            if c[0] != "_" and c != "GRBM_GUI_ACTIVE":
                acc[c].append(v / cyc)      # counter per chip cycle of ITS OWN launch


def mean(v):
    return sum(v) / len(v) if v else float("nan")


order = sorted(per, key=lambda k: -sum(per[k]["_ns"]))
print("| kernel | launches (all passes) | mean us (profiled) | eff. clock GHz | matrix pipe busy % | VALU busy % | wave-cycles waiting % (WAIT_ANY) | issue stall % (WAIT_INST_ANY) | waves resident / SIMD |")
print("|---|---|---|---|---|---|---|---|---|")
for k in order[:28]:
    a = per[k]
    ghz = mean([c / t for c, t in zip(a["_cyc"], a["_ns"])])
    mfma = 100 * mean(a["SQ_VALU_MFMA_BUSY_CYCLES"]) / SIMDS if a["SQ_VALU_MFMA_BUSY_CYCLES"] else float("nan")
    valu = 100 * mean(a["SQ_ACTIVE_INST_VALU"]) * 4 / SIMDS if a["SQ_ACTIVE_INST_VALU"] else float("nan")
    wc = mean(a["SQ_WAVE_CYCLES"]) if a["SQ_WAVE_CYCLES"] else float("nan")
    wait = 100 * mean(a["SQ_WAIT_ANY"]) / wc if a["SQ_WAIT_ANY"] else float("nan")
    stall = 100 * mean(a["SQ_WAIT_INST_ANY"]) / wc if a["SQ_WAIT_INST_ANY"] else float("nan")
    occ = wc * 4 / SIMDS
    print(f"| `{k[:70]}` | {len(a['_ns'])} | {mean(a['_ns'])/1e3:.1f} | {ghz:.2f} | {mfma:.1f} | {valu:.1f} | {wait:.1f} | {stall:.1f} | {occ:.2f} |")
