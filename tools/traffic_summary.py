#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE).
Units: the counters are in KiB (rocprofv3 derived metrics FETCH_SIZE/WRITE_SIZE = requests * 64 B / 1024).
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports exactly 1/2 of the bytes of a wide
coalesced streaming read -> we report both the raw value and the x2-corrected upper estimate; WRITE_SIZE is
uncalibrated and reported raw."""
import collections
import csv
import json
import sys


def load(path):
    agg = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if not (k.startswith("lg_") or k.startswith("void lg_")):
            continue
        name = k.replace("void ", "").split("(")[0].split("<")[0]
        agg[name] += float(r["Counter_Value"]); n[name] += 1
    return {k: (agg[k] / n[k]) for k in agg}, n


fetch, nf = load(sys.argv[1]); write, _ = load(sys.argv[2])
out = {}
for k in sorted(fetch):
    f_raw = fetch[k] * 1024.0; w_raw = write.get(k, 0.0) * 1024.0
    out[k] = {"launches_sampled": nf[k], "fetch_bytes_raw": round(f_raw), "fetch_bytes_x2_wide_read_correction": round(2 * f_raw),
              "write_bytes_raw": round(w_raw), "hbm_bytes_per_launch_low": round(f_raw + w_raw), "hbm_bytes_per_launch_high": round(2 * f_raw + w_raw)}
print(json.dumps(out, indent=1))
