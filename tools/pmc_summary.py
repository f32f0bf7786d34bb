#!/usr/bin/env python3
"""Summarise gpurun_out/pmc_<tag>/p*/**/*_counter_collection.csv: per-launch median of every counter, per kernel whose name
contains the given substring (default k_step).  `python tools/pmc_summary.py <tag> [substring]`"""
import csv
import glob
import re
import sys
from collections import defaultdict

tag = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "k_step"
vals = defaultdict(list)
for f in glob.glob("gpurun_out/pmc_%s/p*/**/*_counter_collection.csv" % tag, recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            m = re.search(r"(k_\w+(?:<[^>]*>)?)", r["Kernel_Name"])
            vals[(m.group(1) if m else r["Kernel_Name"][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k in sorted(vals):
    v = sorted(vals[k])
    print("%-34s %-30s n=%3d median %16.1f" % (k[0], k[1], len(v), v[len(v) // 2]))
