#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV passes per kernel (mean counter value per dispatch) and file the result as
profiles/pmc_current.json, the counter summary bench.py's roofline block reads.

usage: pmc_summary.py <dir with the passes> <workload> <rays per step> [out.json]

The summary records the sha256 of the kernel sources (bench.kernel_source_sha) it was measured on; bench.py
refuses a summary whose sha differs from the sources it runs.  An existing out.json measured on the same sources
keeps its other workloads; one measured on other sources is replaced."""
import collections
import csv
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

root, workload, rays = sys.argv[1], sys.argv[2], float(sys.argv[3])
out_path = sys.argv[4] if len(sys.argv) > 4 else bench.PMC_SUMMARY
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        short = ("trace" if "jur_trace_kernel" in k else "ega" if "jur_ega_kernel" in k else
                 "combine" if "jur_combine" in k else "pencil" if "jur_pencil_kernel" in k else None)
        if short is None:
            continue
        agg[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
kernels = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}
for k, d in kernels.items():
    d["dispatches"] = max(len(v) for v in agg[k].values())
# a kernel launched several times per step (chunked workloads: the look-up and the radiance update of the many-channel
# set) covers fewer rays per launch than one launched once: rays per launch per kernel, from the dispatch counts
steps = min(d["dispatches"] for d in kernels.values())
for k, d in kernels.items():
    d["rays_per_launch"] = rays * steps / d["dispatches"]
sha = bench.kernel_source_sha()
doc = {"kernel_source_sha256": sha, "workloads": {}}
if os.path.exists(out_path):
    try:
        old = json.load(open(out_path))
        if old.get("kernel_source_sha256") == sha:
            doc = old
    except ValueError:
        pass
doc["workloads"][workload] = {
    "rays_per_launch": rays, "kernels": kernels, "measured": time.strftime("%Y-%m-%d %H:%M:%S"),
    "how": "tools/pmc_profile.sh: separate rocprofv3 --pmc passes (SQ x2, FETCH_SIZE, WRITE_SIZE, TCC, TCP), no trace "
           "domains beside counters; mean per dispatch.  HBM bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB (gfx950 FETCH_SIZE "
           "half-count, MI355X_MICROARCH.md)"}
json.dump(doc, open(out_path, "w"), indent=1, sort_keys=True)
print(json.dumps({k: {c: d.get(c) for c in ("SQ_INSTS_VALU", "FETCH_SIZE", "WRITE_SIZE", "dispatches")} for k, d in kernels.items()}))
