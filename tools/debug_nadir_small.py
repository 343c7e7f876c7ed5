#!/usr/bin/env python3
"""Package latency in different process contexts (why does bench.py's package_api see ~1 ms more per call than
tools/bench_small.py?): plain, with torch imported, after a large batched call, with slices of a large array."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "jurassic-gpu_amd"), os.path.join(ROOT, "tests")]
import numpy as np
WITH_TORCH = len(sys.argv) > 1 and sys.argv[1] == "torch"
if WITH_TORCH:                      # torch's HIP runtime has to be the first one in the process
    import torch
    torch.cuda.is_available(); _x = torch.zeros(10, device="cuda"); torch.cuda.synchronize()
import common
from jurassic_hip import lib, synth
big = synth.nadir_geometry(100_000, seed=1000)
case = common.nadir_case(geom=big[:8])
m = lib.Model(case.ctl, case.lib_tables()); m.set_atm(case.atm)
out = {}
def timed(geoms, n=40):
    m.formod_host(geoms[0])
    t0 = time.perf_counter()
    for i in range(n): m.formod_host(geoms[i % len(geoms)])
    return round(1e3 * (time.perf_counter() - t0) / n, 3)
same = [np.ascontiguousarray(big[:1088])]
out["plain_same_array"] = timed(same)
slices = [big[i * 1088:(i + 1) * 1088] for i in range(40)]
out["plain_slices"] = timed(slices)
m.formod_host(big)
out["after_big_call_slices"] = timed(slices)
out["with_torch"] = WITH_TORCH
m.enable_timing(True); m.formod_host(slices[3]); out["kernel_ms"] = m.kernel_ms()["pencil_ms"]
print(json.dumps(out))
