#!/usr/bin/env python3
"""Per-kernel totals of a rocprofv3 (rocpd .db) run as CSV, the same table `--stats` prints:
    python tools/rocpd_top_kernels.py gpurun_out/prof_x/x_results.db > profiles/r01_x_kernel_stats.csv
(durations in microseconds)."""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
w = csv.writer(sys.stdout)
w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
for row in db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
    w.writerow(row)
