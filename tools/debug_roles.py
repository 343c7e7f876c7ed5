#!/usr/bin/env python3
"""Which role of the fused kernel sets the latency of a package?  Variants that lighten one role at a time."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "jurassic-gpu_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import common
from jurassic_hip import lib, synth
geom = synth.limb_geometry(1088, seed=1)
out = {}
for name, kw in (("all", {}), ("no_refraction", dict(refrac=0)), ("no_continua", dict(ctm_co2=0, ctm_h2o=0, ctm_n2=0, ctm_o2=0, ctm_auto=1)),
                 ("no_tables", dict(table_kw=dict(nlev=1))), ("no_tables_no_continua", dict(table_kw=dict(nlev=1), ctm_co2=0, ctm_h2o=0, ctm_n2=0, ctm_o2=0, ctm_auto=1)),
                 ("no_tables_no_continua_no_refraction", dict(table_kw=dict(nlev=1), refrac=0, ctm_co2=0, ctm_h2o=0, ctm_n2=0, ctm_o2=0, ctm_auto=1))):
    case = common.limb_case(geom=geom, **kw)
    m = lib.Model(case.ctl, case.lib_tables())
    m.set_atm(case.atm)
    m.formod_host(geom)
    m.enable_timing(True)
    for _ in range(5):
        m.formod_host(geom)
    k = m.kernel_ms()
    out[name] = round(k["pencil_ms"] / max(1, k["pencil_launches"]), 4)
    m.close()
print(json.dumps(out))
