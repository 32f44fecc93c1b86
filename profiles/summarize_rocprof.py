#!/usr/bin/env python
"""Turn a rocprofv3 rocpd database (…_results.db) into the small text summaries kept under profiles/.

  python profiles/summarize_rocprof.py gpurun_out/prof_r1/r1_results.db > profiles/r01_kernel_stats.md
  python profiles/summarize_rocprof.py --pmc gpurun_out/pmc/x_results.db   (per-kernel counter means)
"""
import sqlite3
import sys


def kernel_stats(db):
    con = sqlite3.connect(db)
    rows = con.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    print("| kernel | calls | total us | avg us | % |")
    print("|---|---|---|---|---|")
    for name, calls, tot, avg, pct in rows:
        print("| `%s` | %d | %.1f | %.3f | %.2f |" % (name, calls, tot, avg, pct))


def pmc_stats(db):
    con = sqlite3.connect(db)
    cols = [d[0] for d in con.execute("select * from counters_collection limit 1").description]
    name_c = "kernel_name" if "kernel_name" in cols else "name"
    q = ("select %s, counter_name, count(*), avg(value), sum(value) from counters_collection "
         "group by %s, counter_name order by sum(value) desc" % (name_c, name_c))
    print("| kernel | counter | dispatches | mean per dispatch | total |")
    print("|---|---|---|---|---|")
    for name, ctr, n, mean, tot in con.execute(q):
        print("| `%s` | %s | %d | %.4g | %.6g |" % (name, ctr, n, mean, tot))


if __name__ == "__main__":
    if sys.argv[1] == "--pmc":
        pmc_stats(sys.argv[2])
    else:
        kernel_stats(sys.argv[1])
