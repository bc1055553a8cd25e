"""One train step of a rocprofv3 --kernel-trace DB (one-stream schedule), launch by launch: start offset, duration, idle gap in front of it -- and per kernel
name the launches / step, mean duration, mean gap in front.  Round 6: what the loss chain, the column sums and the remainder launches cost as
time on the critical path, gaps included.  usage: step_listing.py <db> [steps=5] [--list]"""
import collections, re, sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = list(cur.execute("select name, start, end from kernels order by start"))
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 5
idx = [i for i, r in enumerate(rows) if r[0].startswith("adamw_kernel")]
short = lambda n: re.sub(r"\[clone.*", "", re.sub(r"^void ", "", n)).replace("(anonymous namespace)::", "")[:70]
agg = collections.OrderedDict()
walls = []
for s in range(n_steps):
    lo, hi = idx[-n_steps - 1 + s], idx[-n_steps + s]
    seg = rows[lo + 1: hi + 1]
    walls.append((seg[-1][2] - seg[0][1]) / 1e3)
    prev_end = seg[0][1]
    for nm, st, en in seg:
        a = agg.setdefault(short(nm), [0, 0.0, 0.0])
        a[0] += 1; a[1] += (en - st) / 1e3; a[2] += max(0, st - prev_end) / 1e3
        prev_end = max(prev_end, en)
print(f"{n_steps} steps, {sum(walls)/n_steps:.1f} us wall per step (first kernel start -> adamw end); per kernel name: launches/step, us per launch, idle us in front per launch, total us/step (duration + gap)")
tot_d = tot_g = 0.0
for k, (n, d, g) in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
    print(f"  {n/n_steps:6.1f}  {d/n:9.1f}  {g/n:7.1f}  {(d+g)/n_steps:9.1f}  {k}")
    tot_d += d; tot_g += g
print(f"sum of durations {tot_d/n_steps:.1f} us/step, sum of idle gaps {tot_g/n_steps:.1f} us/step")
if "--list" in sys.argv:
    lo, hi = idx[-2], idx[-1]
    seg = rows[lo + 1: hi + 1]
    t0, prev_end = seg[0][1], seg[0][1]
    print("last step, launch by launch: offset us, duration us, gap us, kernel")
    for nm, st, en in seg:
        print(f"  {(st-t0)/1e3:9.1f} {(en-st)/1e3:8.1f} {max(0, st-prev_end)/1e3:6.1f}  {short(nm)}")
        prev_end = max(prev_end, en)
