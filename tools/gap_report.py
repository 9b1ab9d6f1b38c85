"""Device idle time inside the training step, from a rocprofv3 kernel trace:  python tools/gap_report.py <kernel_trace.csv>
Steps are delimited by the optimizer kernel (adamw_kernel); for the steps between the first and the last one the union of all kernels' [start, end)
intervals (all streams) is taken: idle = wall - union.  Gaps are attributed to the pair (kernel that ended last before the gap -> kernel that starts
after it) and summed over the steps."""
import csv, sys, collections
rows = []
with open(sys.argv[1]) as f:
    rd = csv.DictReader(f)
    for r in rd:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
short = lambda n: n.replace("void ", "").replace("alpro::(anonymous namespace)::", "").split("(")[0][:70]
opt = [i for i, r in enumerate(rows) if "adamw_kernel" in r[2]]
assert len(opt) >= 3, "need at least three optimizer launches in the trace"
lo, hi = opt[1], opt[-1]           # from the end of the 2nd optimizer launch to the end of the last one: len(opt) - 2 whole steps
steps = len(opt) - 2
t0, t1 = rows[lo][1], rows[hi][1]
win = [r for r in rows[lo + 1:hi + 1]]
busy, gaps, cur_end, last = 0, collections.Counter(), t0, short(rows[lo][2])
gapn = collections.Counter()
overlap = 0
for s, e, n in win:
    if s > cur_end:
        gaps[(last, short(n))] += s - cur_end
        gapn[(last, short(n))] += 1
        busy += e - s
        cur_end, last = e, short(n)
    else:
        overlap += min(e, cur_end) - s
        if e > cur_end:
            busy += e - cur_end
            cur_end, last = e, short(n)
wall = t1 - t0
print("steps %d: wall %.2f ms per step, some kernel running %.2f ms, idle %.2f ms per step (%.1f %%); kernel time running beside another kernel %.2f ms per step"
      % (steps, wall / steps / 1e6, busy / steps / 1e6, (wall - busy) / steps / 1e6, 100.0 * (wall - busy) / wall, overlap / steps / 1e6))
print("largest idle gaps by (kernel before -> kernel after), us per step / count per step / mean us:")
for (a, b), t in gaps.most_common(25):
    print("  %8.1f  %5.1f  %6.1f   %s -> %s" % (t / steps / 1e3, gapn[(a, b)] / steps, t / gapn[(a, b)] / 1e3, a, b))
hist = collections.Counter()
for (a, b), t in gaps.items():
    pass
