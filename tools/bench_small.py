#!/usr/bin/env python3
"""Latency of one drop-in sized call (reference package: <= 1088 rays) through jur_formod_host: the fused kernel
(jur_pencil_kernel, default for such sizes) against the three batched kernels, and the fused kernel's sweep over
call size and rays per workgroup."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "jurassic-gpu_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import common
from jurassic_hip import lib, synth


def timed(m, geom, n=20):
    m.formod_host(geom)
    t0 = time.perf_counter()
    for _ in range(n):
        m.formod_host(geom)
    dt = (time.perf_counter() - t0) / n
    m.enable_timing(True)
    m.formod_host(geom)
    k = m.kernel_ms()
    m.enable_timing(False)
    return dict(ms_per_call=1e3 * dt, rays=len(geom), rays_per_s=len(geom) / dt,
                kernel_ms={a: round(b, 4) for a, b in k.items() if a.endswith("_ms") and b > 0})


out = {}
for name, case in (("limb_66", common.limb_case()), ("nadir_90", common.nadir_case()),
                   ("limb_1088", common.limb_case(geom=synth.limb_geometry(1088, seed=1))),
                   ("limb_1088_4ch_64prof", common.limb_case(geom=synth.limb_geometry(1088, seed=1, nprofiles=64), nu=common.CTM4_NU,
                                                             nprofiles=64))):
    m = lib.Model(case.ctl, case.lib_tables())
    m.set_atm(case.atm)
    m.set_pencil(0)
    res = {"batched": timed(m, case.geom)}
    m.set_pencil(1 << 20, 0)
    res["fused"] = timed(m, case.geom)
    if name in ("limb_1088", "limb_1088_4ch_64prof"):
        res["fused_rays_per_group"] = {}
        for rb in (1, 2, 4, 8, 16):
            m.set_pencil(1 << 20, rb)
            res["fused_rays_per_group"][rb] = timed(m, case.geom, n=10)["ms_per_call"]
        big = {}
        for nr in (272, 544, 2176, 4352, 8704, 17408):
            g = synth.limb_geometry(nr, seed=2, nprofiles=64 if "64prof" in name else 1)
            row = {}
            m.set_pencil(0)
            row["batched_ms"] = timed(m, g, n=5)["ms_per_call"]
            for rb in (1, 2, 4, 8, 16):
                m.set_pencil(1 << 20, rb)
                row["fused_rb%d_ms" % rb] = timed(m, g, n=5)["ms_per_call"]
            big[nr] = row
        res["size_sweep"] = big
    out[name] = res
    m.close()
print(json.dumps(out))
