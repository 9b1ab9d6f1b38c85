"""Dump a rocprofv3 rocpd (.db) kernel-trace into a per-kernel stats CSV (same columns as --stats)."""
import sqlite3
import sys


def main(db, out):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    with open(out, "w") as f:
        f.write("Name,Calls,TotalDurationUs,AverageUs,Percentage\n")
        for n, calls, tot, avg, pct in rows:
            f.write('"%s",%d,%.3f,%.3f,%.4f\n' % (n.replace('"', "'"), calls, tot, avg, pct))
    print(open(out).read()[:3000])
    print("-- GEMM by grid size (us): grid_x/256, calls, avg")
    for r in c.execute("select grid_x/workgroup_x, count(*), avg(duration)/1000.0 from kernels where name like '%gemm_nt_kernel%' group by grid_x order by 3 desc"):
        print(r)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
