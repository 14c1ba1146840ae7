#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files for this repo's kernels:
per kernel (short name) and counter, the mean over dispatches, plus per-frame-iteration figures when
--frame-iters is given.  usage: pmc_summary.py [--frame-iters N] file.csv [file.csv ...]"""
import collections
import csv
import sys

args = sys.argv[1:]
fi = None
if args and args[0] == "--frame-iters":
    fi = float(args[1])
    args = args[2:]
rows = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for path in args:
    with open(path) as f:
        for r in csv.DictReader(f):
            if "ccamd" not in r["Kernel_Name"]:
                continue
            # (the whole template argument list: SINGLE / CHAIN sit at its end and name different kernels)
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("ccamd::", "").split("(")[0][:150]
            rows[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
            rows[name]["duration_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
            meta[name] = "grid=%s wg=%s vgpr=%s+%s sgpr=%s lds=%s" % (r["Grid_Size"], r["Workgroup_Size"], r["VGPR_Count"],
                                                                   r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"])
for name, ctr in rows.items():
    print(name, "|", meta[name])
    for c, v in sorted(ctr.items()):
        mean = sum(v) / len(v)
        extra = ("   per frame-iteration: %.1f" % (mean / fi)) if fi and c != "duration_us" else ""
        print("   %-24s n=%-3d mean=%.6g%s" % (c, len(v), mean, extra))
