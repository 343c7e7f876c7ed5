#!/usr/bin/env python3
"""jur_formod_host (host arrays in, host arrays out) against jur_formod_device for the limb_1e6 workload."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "jurassic-gpu_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import bench
from jurassic_hip import lib
case = bench.build_case("limb_1e6", bench.global_geometry("limb_1e6", 1_000_000, 1000))
m = lib.Model(case.ctl, case.lib_tables())
m.set_atm(case.atm)
m.formod_host(case.geom[:1000])
m.formod_host(case.geom)
t0 = time.perf_counter()
n = 3
for _ in range(n):
    m.formod_host(case.geom)
dt = (time.perf_counter() - t0) / n
print(json.dumps({"rays": len(case.geom), "host_entry_ms": 1e3 * dt, "host_entry_rays_per_s": len(case.geom) / dt}))
