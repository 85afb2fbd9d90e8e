#!/usr/bin/env python3
"""Mean per-dispatch PMC counter values per kernel from rocprofv3 counter_collection CSVs:
   python tools/pmc_summary.py <dir> [kernel substring]"""
import collections
import csv
import glob
import re
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_\w+)", r["Kernel_Name"])
        if m:
            agg[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
want = sys.argv[2] if len(sys.argv) > 2 else ""
for k, d in sorted(agg.items()):
    if want in k:
        print(k, {c: round(sum(v) / len(v), 1) for c, v in sorted(d.items())})
