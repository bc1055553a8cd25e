"""Union of kernel intervals of a rocprofv3 --kernel-trace DB: GPU-busy time, idle gaps and what surrounds the largest ones (default two-stream schedule)."""
import re, sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
rows = list(cur.execute("select name, start, end from kernels order by start"))
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
# take the last n_steps * (launches per step) kernels: find step boundaries by the adamw kernel
idx = [i for i, r in enumerate(rows) if r[0].startswith("adamw_kernel")]
lo, hi = idx[-n_steps - 1], idx[-1]
seg = rows[lo + 1: hi + 1]
t0, t1 = seg[0][1], max(r[2] for r in seg)
busy, gaps, end = 0, [], seg[0][1]
for i, (nm, s, e) in enumerate(seg):
    if s > end:
        gaps.append((s - end, i)); 
    busy += max(0, e - max(s, end)); end = max(end, e)
wall = t1 - t0
print(f"{n_steps} steps: wall {wall/1e6/n_steps:.3f} ms/step, GPU busy (union of kernel intervals) {busy/1e6/n_steps:.3f} ms/step, idle {(wall-busy)/1e6/n_steps:.3f} ms/step, sum of kernel durations {sum(e-s for _,s,e in seg)/1e6/n_steps:.3f} ms/step")
gaps.sort(reverse=True)
print("largest idle gaps (us): kernel before -> kernel after")
short = lambda n: re.sub(r"\(.*", "", n)[:60]
for g, i in gaps[:12]:
    print(f"  {g/1e3:7.1f}  {short(seg[i-1][0])} -> {short(seg[i][0])}")
import collections
c = collections.Counter()
for g, i in gaps: c[short(seg[i][0])] += g
print("idle time by the kernel that follows the gap (us/step):")
for k, v in c.most_common(12): print(f"  {v/1e3/n_steps:7.1f}  {k}")
