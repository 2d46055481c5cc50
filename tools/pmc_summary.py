#!/usr/bin/env python3
"""Per-launch SQ / GRBM counters of our kernels from tools/gpu_pmc.sh passes -> JSON (profiles/r01_final_pmc_valu.json).
usage: pmc_summary.py gpurun_out/pmc1/p_counter_collection.csv gpurun_out/pmc2/... > out.json"""
import collections
import csv
import json
import sys


def load(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); launches = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if not k.startswith(("lg_", "void lg_")):
            continue
        name = k.replace("void ", "").split("(")[0]
        agg[name][r["Counter_Name"]] += float(r["Counter_Value"]); launches[name].add(r["Dispatch_Id"])
    return {k: {c: v / len(launches[k]) for c, v in d.items()} for k, d in agg.items()}


out = collections.defaultdict(dict)
for p in sys.argv[1:]:
    for k, d in load(p).items():
        out[k].update({c: round(v) for c, v in d.items()})
meta = {"command": "rocprofv3 --pmc <set> --kernel-trace --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline "
                   "--no-roofline --no-literal (tools/gpu_pmc.sh; one pass per counter set: {SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES}, "
                   "{SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES}, {GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_LDS})",
        "units": "per launch, summed over the device (all XCDs / SIMDs); *_INSTS_* are wave64 instructions",
        "workload": "C3: 3M Gaussians, 1920x1080, default bench path"}
print(json.dumps({"_meta": meta, **out}, indent=1))
