#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel (name truncated), per counter, mean over dispatches."""
import csv, sys, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"].split("(")[0][-60:]
            rows[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, ctr in rows.items():
    if "minsum" not in name and "alg" not in name and "encode" not in name and "mc_" not in name:
        continue
    print(name)
    for c, v in sorted(ctr.items()):
        print("   %-28s n=%d mean=%.4g" % (c, len(v), sum(v) / len(v)))
