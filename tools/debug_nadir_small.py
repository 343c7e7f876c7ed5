import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "jurassic-gpu_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import common
from jurassic_hip import lib, synth
case = common.nadir_case(geom=synth.nadir_geometry(1088, seed=3))
m = lib.Model(case.ctl, case.lib_tables()); m.set_atm(case.atm)
out = {}
def timed(n=20):
    m.formod_host(case.geom)
    t0 = time.perf_counter()
    for _ in range(n): m.formod_host(case.geom)
    dt = (time.perf_counter() - t0) / n
    m.enable_timing(True); m.formod_host(case.geom); k = m.kernel_ms(); m.enable_timing(False)
    return round(1e3 * dt, 3), {a: round(b, 3) for a, b in k.items() if a.endswith("_ms") and b > 0}
m.set_pencil(0); out["batched"] = timed()
for rb in (0, 1, 2, 4, 8, 16):
    m.set_pencil(1 << 20, rb); out["rb%d" % rb] = timed()
print(json.dumps(out))
