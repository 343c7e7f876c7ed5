import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "jurassic-gpu_amd"), os.path.join(ROOT, "tests")]
import numpy as np, common
from jurassic_hip import lib, synth
n = 500000
out = {}
for name, kw in (("refrac1", {}), ("refrac0", dict(refrac=0))):
    geom = synth.limb_geometry(n, seed=1000, nprofiles=64)
    case = common.limb_case(geom=geom, nu=common.CTM4_NU, nprofiles=64, **kw)
    m = lib.Model(case.ctl, case.lib_tables()); m.set_atm(case.atm)
    m.formod_host(case.geom); m.enable_timing(True); m.formod_host(case.geom)
    out[name] = m.kernel_ms(); m.close()
print(json.dumps(out))
