"""Summarise rocprofv3 --pmc CSV output per kernel: sum of every counter over the dispatches of the
kernels whose name contains the given substring.  usage: pmc_summary.py <dir> [substr]"""
import csv, glob, json, sys
from collections import defaultdict
d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "k_expand"
out = {}
import os
files = sorted(glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True), key=os.path.getmtime)
for f in files[-1:]:  # the newest run only (gpurun merges every call's output into the same local directory)
    sums, disp = defaultdict(float), defaultdict(set)
    for row in csv.DictReader(open(f)):
        if sub in row["Kernel_Name"]:
            sums[row["Counter_Name"]] += float(row["Counter_Value"])
            disp[row["Counter_Name"]].add(row["Dispatch_Id"])
    for k, v in sums.items():
        out[k] = {"sum": v, "dispatches": len(disp[k])}
print(json.dumps(out, indent=1))
