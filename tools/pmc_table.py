"""Aggregate rocprofv3 --pmc csv outputs (one directory per pass) into a per-kernel table of mean counter values per launch."""
import csv, glob, os, re, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            name = re.sub(r"\(.*", "", row.get("Kernel_Name", ""))
            if not name.startswith(("void attn", "attn", "void gemm", "gemm", "void ln_", "ln_")):
                continue
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name, cs in acc.items():
    print(f"## {name}")
    for c, v in sorted(cs.items()):
        print(f"  {c:32s} {sum(v)/len(v):.4e}  (n={len(v)})")
