#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (--kernel-trace --stats) into a text table:
    python tools/prof_summary.py gpurun_out/prof_X/X_results.db [> profiles/rNN_x.txt]
Durations are in microseconds (rocpd stores ns)."""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    rows = list(c.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(grid_y), max(workgroup_x) "
        "from kernels group by name order by sum(duration) desc"))
    tot = sum(r[2] for r in rows) or 1
    print("# rocprofv3 --kernel-trace --stats summary of %s" % path)
    print("%-62s %7s %12s %10s %10s %10s %6s %5s %5s %5s %7s" % (
        "kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct", "vgpr", "agpr", "sgpr", "lds"))
    for r in rows:
        print("%-62s %7d %12.1f %10.2f %10.2f %10.2f %6.2f %5d %5d %5d %7d" % (
            r[0][:62], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / tot, r[6], r[7], r[8], r[9]))
    print("# total kernel time %.1f us over %d dispatches" % (tot / 1e3, sum(r[1] for r in rows)))


if __name__ == "__main__":
    main(sys.argv[1])
