#!/usr/bin/env python3
"""Dump the per-kernel summary (rocprofv3 --kernel-trace --stats, rocpd sqlite output) as CSV.
usage: rocpd_stats.py results.db > kernel_stats.csv"""
import csv
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
w = csv.writer(sys.stdout)
w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
for name, calls, total, avg, pct in c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
    w.writerow([name[:160], calls, f"{total:.3f}", f"{avg:.3f}", f"{pct:.3f}"])
