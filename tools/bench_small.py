#!/usr/bin/env python3
"""Latency of one drop-in sized call (reference package: <= 1088 rays) through jur_formod_host."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "jurassic-gpu_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import common
from jurassic_hip import lib, synth
out = {}
for name, case in (("limb_66", common.limb_case()), ("nadir_90", common.nadir_case()),
                   ("limb_1088", common.limb_case(geom=synth.limb_geometry(1088, seed=1)))):
    m = lib.Model(case.ctl, case.lib_tables())
    m.set_atm(case.atm)
    m.formod_host(case.geom)
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        m.formod_host(case.geom)
    dt = (time.perf_counter() - t0) / n
    m.enable_timing(True)
    m.formod_host(case.geom)
    k = m.kernel_ms()
    out[name] = dict(ms_per_call=1e3 * dt, rays=len(case.geom), trace_ms=k["trace_ms"], ega_ms=k["ega_ms"], combine_ms=k["combine_ms"])
    m.close()
print(json.dumps(out))
