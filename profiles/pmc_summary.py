#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files per kernel (sum over dispatches)."""
import collections
import csv
import glob
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(int)
dur = collections.defaultdict(float)
first = sorted(glob.glob(f"{root}/p*/*/*_counter_collection.csv"))
for f in first:
    seen = set()
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        if "mirt" not in k:
            continue
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        if f == first[0] and row["Dispatch_Id"] not in seen:
            seen.add(row["Dispatch_Id"]); n[k] += 1
            dur[k] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
for k in agg:
    print("==", k, "launches", n[k], "total us", round(dur[k], 1))
    for c, v in sorted(agg[k].items()):
        print("   %-28s %.5g" % (c, v))
