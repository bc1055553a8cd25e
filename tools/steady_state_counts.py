"""Per-step launch counts in the steady state: difference of two rocprofv3 kernel-trace DBs taken with different --steps (the set-up and
warm-up launches cancel).  Usage: steady_state_counts.py <db short run> <steps short> <db long run> <steps long>"""
import re, sqlite3, sys
def counts(path):
    cur = sqlite3.connect(path).cursor()
    return {re.sub(r"\(.*", "", n)[:100]: (c, t) for n, c, t in cur.execute("select name, count(*), sum(end-start) from kernels group by name")}
a, sa, b, sb = counts(sys.argv[1]), int(sys.argv[2]), counts(sys.argv[3]), int(sys.argv[4])
rows = []
for k in b:
    dc = (b[k][0] - a.get(k, (0, 0))[0]) / (sb - sa)
    dt = (b[k][1] - a.get(k, (0, 0))[1]) / (sb - sa) / 1e3
    if dc > 0: rows.append((dt, dc, k))
rows.sort(reverse=True)
print("| kernel | launches / step | us / step |\n|---|---|---|")
for dt, dc, k in rows: print(f"| {k} | {dc:.2f} | {dt:.1f} |")
print(f"\ntotal {sum(r[1] for r in rows):.1f} launches, {sum(r[0] for r in rows)/1e3:.3f} ms of kernel time per step")
