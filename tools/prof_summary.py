"""Summarise a rocprofv3 rocpd results DB (kernel-trace) into a per-kernel table (markdown)."""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|")
for name, n, s, a, mn, mx in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    short = re.sub(r"\(.*", "", name)[:90]
    print(f"| {short} | {n} | {s/1e6:.3f} | {a/1e3:.1f} | {mn/1e3:.1f} | {mx/1e3:.1f} | {100*s/tot:.1f} |")
print(f"\ntotal kernel time {tot/1e6:.2f} ms")
