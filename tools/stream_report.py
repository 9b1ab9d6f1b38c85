"""Which stream ran what, where and when inside one training step, from a rocprofv3 kernel trace:  python tools/stream_report.py <kernel_trace.csv>
One steady step (the last timed one of a bench.py run): per (Stream_Id, Queue_Id) the number of kernels, their summed duration, the span from the
first start to the last end relative to the step's start, and the three kernels that took most of the time."""
import csv, sys, collections
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Stream_Id"], r["Queue_Id"]))
rows.sort()
short = lambda n: n.replace("void ", "").replace("alpro::(anonymous namespace)::", "").replace("alpro::", "").split("(")[0][:44]
opt = [i for i, r in enumerate(rows) if "adamw_kernel" in r[2]]
lo, hi = opt[-3], opt[-2]   # (the last step of a bench.py run is its one-stream accounting pass)
t0, t1 = rows[lo][1], rows[hi][1]
print("step: %.2f ms" % ((t1 - t0) / 1e6))
by = collections.defaultdict(list)
for s, e, n, st, q in rows[lo + 1:hi + 1]:
    by[(st, q)].append((s, e, n))
for (st, q), ks in sorted(by.items(), key=lambda kv: kv[1][0][0]):
    dur = sum(e - s for s, e, _ in ks)
    top = collections.Counter()
    for s, e, n in ks:
        top[short(n)] += e - s
    print("stream %3s queue %2s: %5d kernels, %8.2f ms of kernel time, from %7.2f to %7.2f ms | %s"
          % (st, q, len(ks), dur / 1e6, (ks[0][0] - t0) / 1e6, (max(e for _, e, _ in ks) - t0) / 1e6,
             ", ".join("%s %.1f" % (k, v / 1e6) for k, v in top.most_common(3))))
