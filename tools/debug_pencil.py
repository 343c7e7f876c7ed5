#!/usr/bin/env python3
"""Where do the fused and the batched path differ on the edge-geometry case?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "jurassic-gpu_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import common
from jurassic_hip import lib, synth
g = synth.limb_geometry(8, scan=True, zmin=-20.0, zmax=2.0)
extra = np.array([[0, 30.0, 0, 0, 5.0, 0, 3.0], [0, 10.0, 0, 0, 60.0, 0, 2.0], [0, 20.0, 0, 0, 80.0, 0, 0.0],
                  [0, 780.0, 0, 0, 95.0, 0, 20.0], [0, -1.0, 0, 0, 10.0, 0, 1.0]])
geom = np.vstack([g, extra, synth.limb_geometry(40, seed=1)])
for missing in (set(), {(3, 0), (4, 1), (0, 1)}):
    case = common.limb_case(geom=geom, missing=missing)
    m = lib.Model(case.ctl, case.lib_tables())
    m.set_atm(case.atm)
    m.set_pencil(0)
    a = m.formod_host(geom)
    for rb in (1, 5, 64):
        m.set_pencil(1 << 20, rb)
        b = m.formod_host(geom)
        for k in ("rad", "tau", "tp"):
            bad = np.argwhere(a[k] != b[k])
            print("missing", sorted(missing), "rb", rb, k, "differing", len(bad), [(int(i), int(j), a["np"][i], float(a[k][i, j]), float(b[k][i, j])) for i, j in bad[:8]])
    m.close()
