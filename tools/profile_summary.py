#!/usr/bin/env python3
"""One JSON per profiled bench command: per-kernel launch time (rocprofv3 --kernel-trace --stats), HBM traffic (FETCH_SIZE and
WRITE_SIZE from SEPARATE --pmc passes, MI355X guide) and SQ counters, stamped with the identity of the library build
(lg_build_id: sha1 of the kernel sources) so that bench.py can refuse numbers taken from another build.

usage: profile_summary.py --mode fwdbwd --stats <kernel_stats.csv> [--fetch <counter csv>] [--write <counter csv>] [--sq <csv> ...]
Units: time ns; FETCH_SIZE / WRITE_SIZE are KiB per dispatch in rocprofv3's output -> bytes here.  gfx950: FETCH_SIZE reports
1/2 of the bytes of wide (16 B/lane) coalesced reads: `hbm_bytes_low` = FETCH + WRITE, `hbm_bytes_high` = 2*FETCH + WRITE."""
import argparse
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def kname(k):
    return k.replace("void ", "").split("(")[0].strip()


def ours(k):
    return kname(k).startswith("lg_")


def load_counters(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); launches = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if not ours(r["Kernel_Name"]):
            continue
        n = kname(r["Kernel_Name"])
        agg[n][r["Counter_Name"]] += float(r["Counter_Value"]); launches[n].add(r["Dispatch_Id"])
    return {k: {c: v / len(launches[k]) for c, v in d.items()} for k, d in agg.items()}, {k: len(v) for k, v in launches.items()}


ap = argparse.ArgumentParser()
ap.add_argument("--mode", required=True); ap.add_argument("--stats"); ap.add_argument("--fetch"); ap.add_argument("--write")
ap.add_argument("--sq", nargs="*", default=[]); ap.add_argument("--command", default="")
a = ap.parse_args()
from lightgaussian_amd import _lib  # noqa: E402
out = {"_meta": {"build_id": _lib.build_id(), "git_sha": os.environ.get("LG_GIT_SHA", "n/a (the GPU box receives a snapshot without .git; build_id identifies the sources)"),
                 "mode": a.mode, "command": a.command,
                 "workload": "C3: 3M synthetic Gaussians (seed 20250103), 1920x1080, SH degree 3, bench.py default path",
                 "notes": "avg_ns from rocprofv3 --kernel-trace --stats; fetch/write from separate --pmc FETCH_SIZE / WRITE_SIZE passes; "
                          "SQ_* per launch summed over the device, *_INSTS_* are wave64 instructions"},
       "kernels": {}}
K = out["kernels"]
if a.stats:
    for r in csv.DictReader(open(a.stats)):
        name = r.get("Name") or r.get("Kernel_Name") or ""
        n = kname(name)
        ent = K.setdefault(n, {})
        ent.update({"calls": int(float(r.get("Calls", 0))), "avg_ns": float(r.get("AverageNs", r.get("Average", 0))),
                    "total_ns": float(r.get("TotalDurationNs", r.get("TotalDuration", 0))), "pct": float(r.get("Percentage", 0))})
if a.fetch:
    f, nf = load_counters(a.fetch)
    for n, d in f.items():
        K.setdefault(n, {})["fetch_bytes_raw"] = round(d.get("FETCH_SIZE", 0.0) * 1024.0); K[n]["pmc_launches"] = nf[n]
if a.write:
    w, _ = load_counters(a.write)
    for n, d in w.items():
        K.setdefault(n, {})["write_bytes_raw"] = round(d.get("WRITE_SIZE", 0.0) * 1024.0)
for n, e in K.items():
    if "fetch_bytes_raw" in e and "write_bytes_raw" in e:
        e["hbm_bytes_low"] = e["fetch_bytes_raw"] + e["write_bytes_raw"]
        e["hbm_bytes_high"] = 2 * e["fetch_bytes_raw"] + e["write_bytes_raw"]
for p in a.sq:
    s, _ = load_counters(p)
    for n, d in s.items():
        K.setdefault(n, {}).update({c: round(v) for c, v in d.items()})
out["_meta"]["kernel_symbols"] = sorted(k for k in K if k.startswith("lg_"))
print(json.dumps(out, indent=1, sort_keys=True))
