#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV passes per kernel: mean counter value per dispatch."""
import csv, glob, os, sys, collections, json
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        short = ("integrate" if "jur_integrate" in k else "trace" if "jur_trace" in k else
                 "ega" if "jur_ega" in k else "combine" if "jur_combine" in k else None)
        if short is None:
            continue
        agg[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}
for k, d in out.items():
    d["dispatches"] = max(len(v) for v in agg[k].values())
print(json.dumps(out, indent=1))
