#!/usr/bin/env python3
"""Trim a rocprofv3 *_kernel_stats.csv to readable width (kernel names cut to 90 chars)."""
import csv, sys
w = csv.writer(sys.stdout)
for i, row in enumerate(csv.reader(open(sys.argv[1]))):
    row[0] = row[0][:90]
    w.writerow(row)
