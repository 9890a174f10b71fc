#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean per dispatch of each counter for
kernels whose name contains a pattern.  usage: pmc_summary.py PATTERN dir [dir ...]"""
import csv, glob, sys, collections
pat = sys.argv[1]
for d in sys.argv[2:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(acc.items()):
            print(f"{d.rstrip('/').split('/')[-1]:12s} {k:24s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
