#!/usr/bin/env python3
"""Mean per-launch value of every counter rocprofv3 --pmc wrote for the step
kernel under a directory:  python tools/pmc_summary.py gpurun_out/pmc_x [...]"""
import collections
import csv
import glob
import sys

for d in sys.argv[1:]:
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "afe_step_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(d)
    for k in sorted(agg):
        v = agg[k]
        print("  %-28s n=%4d mean=%14.1f min=%14.1f max=%14.1f" % (k, len(v), sum(v) / len(v), min(v), max(v)))
