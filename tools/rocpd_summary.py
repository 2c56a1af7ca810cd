#!/usr/bin/env python3
"""Dump the per-kernel summary (the `--stats` table) of a rocprofv3 rocpd .db as CSV.

rocprofv3 7.x writes a SQLite "rocpd" database by default; its `top_kernels`
view is the kernel-trace stats table.  Usage: rocpd_summary.py results.db > out.csv
"""
import csv
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    rows = con.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    w = csv.writer(sys.stdout)
    w.writerow(["Name", "Calls", "TotalDuration(us)", "AverageDuration(us)", "Percentage"])
    for r in rows:
        w.writerow([r[0], r[1], f"{r[2]:.3f}", f"{r[3]:.3f}", f"{r[4]:.2f}"])
    # min / max / launch geometry for our own kernels
    q = ("select name, count(*), min(duration), max(duration), avg(duration), grid_x, workgroup_x, vgpr_count, "
         "sgpr_count, lds_size, scratch_size from kernels where name like '%k_%' group by name, grid_x")
    w.writerow([])
    w.writerow(["Name", "Calls", "Min(ns)", "Max(ns)", "Avg(ns)", "grid_x", "workgroup_x", "vgpr", "sgpr", "lds", "scratch"])
    for r in con.execute(q):
        w.writerow(list(r))


if __name__ == "__main__":
    main()
