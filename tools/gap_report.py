#!/usr/bin/env python3
"""Where the step's idle time sits: gaps between consecutive kernels of the timed loop, from a rocprofv3 --kernel-trace CSV.
usage: gap_report.py <kernel_trace.csv>   (prints the mean gap in front of every kernel of a steady-state step)"""
import collections, csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].replace("void ", "").split("(")[0][:40]
# steady state: the last 60 % of the trace
rows = rows[int(len(rows) * 0.4):]
gaps, durs = collections.defaultdict(list), collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    gaps[(name(a), name(b))].append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
    durs[name(b)].append(int(b["End_Timestamp"]) - int(b["Start_Timestamp"]))
tot_gap = sum(sum(v) for v in gaps.values()); tot_busy = sum(sum(v) for v in durs.values())
print(f"busy {tot_busy/1e6:.3f} ms, idle between kernels {tot_gap/1e6:.3f} ms ({100*tot_gap/(tot_gap+tot_busy):.1f} %)")
for (a, b), v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:24]:
    if len(v) >= 5:
        print(f"{a:40s} -> {b:40s} n {len(v):4d} mean gap {sum(v)/len(v)/1e3:7.2f} us   ({b} runs {sum(durs[b])/len(durs[b])/1e3:7.1f} us)")
