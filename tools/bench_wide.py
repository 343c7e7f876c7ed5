#!/usr/bin/env python3
"""AIRS-like many-channel nadir case (BASELINE configs[4], SURVEY 8d "C5") on one GPU: 2378 channels
650..2665 cm^-1, CO2/H2O/O3, full-size synthetic tables (33 p x 10 T x ~203 u per pair = 7134 tables,
~4.8e8 entries, 3.8 GB: far beyond L2 + Infinity Cache).  Needs the ND=2378 build:
    JUR_ND=2378 JUR_NG=3 JUR_SUFFIX=_nd2378 python3 tools/bench_wide.py [nrays]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "jurassic-gpu_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import common
from jurassic_hip import abi, lib, synth
assert (abi.ND, abi.NG) == (2378, 3), "export JUR_ND=2378 JUR_NG=3 JUR_SUFFIX=_nd2378"
nr = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nu = [650.0 + i * (2665.0 - 650.0) / 2377 for i in range(2378)]
geom = synth.nadir_geometry(nr, seed=7)
ctl = abi.make_ctl(["CO2", "H2O", "O3"], nu)
from jurassic_hip import textio
atm = textio.read_atm(os.path.join(common.GOLD, "limb", "atm.tab"), ctl)
t0 = time.perf_counter()
tb = lib.Tables(3, 2378)
for g, em in enumerate(["CO2", "H2O", "O3"]):
    for d, v in enumerate(nu):
        tb.feed_rows(g, d, synth.table_rows(em, v, id_=d % 7))
        if g == 0:
            tb.set_filter(d, *synth.boxcar_filter(v))
t_tab = time.perf_counter() - t0
model = lib.Model(ctl, tb)
model.set_atm(atm)
model.formod_host(geom)                  # warm-up at full size: workspace allocation happens here
model.enable_timing(True)
t0 = time.perf_counter()
res = model.formod_host(geom)
dt = time.perf_counter() - t0
k = model.kernel_ms()
assert np.isfinite(res["rad"]).all()
print(json.dumps({"what": "nadir, 2378 channels x 3 emitters, 1 MI355X", "rays": nr, "table_entries": tb.entries(),
                  "table_build_s": t_tab, "seconds": dt, "rays_per_s": nr / dt, "spectra_channels_per_s": nr * 2378 / dt,
                  "ega_calls_per_s": nr * 182 * 7134 / dt, "kernel_ms": k, "workspace_bytes": model.workspace_bytes()}))
