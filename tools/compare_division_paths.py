#!/usr/bin/env python3
"""Bit-for-bit check of the look-up's two division paths on the bench workload.

jur_ega_kernel divides through reciprocal bracket widths / the bare division sequence when the tables
are strictly increasing; JUR_EGA_NO_RCP=1 selects the compiler's fp64 division
instead.  Both must return the same doubles: this script runs limb and nadir batches through both and
compares radiance and transmittance bitwise."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "jurassic-gpu_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import bench
from jurassic_hip import lib

out = {}
for workload, n in (("limb_1e6", int(sys.argv[1]) if len(sys.argv) > 1 else 300_000), ("nadir_1e5", 100_000)):
    case = bench.build_case(workload, bench.global_geometry(workload, n, 1000))
    m = lib.Model(case.ctl, case.lib_tables())
    m.set_atm(case.atm)
    os.environ.pop("JUR_EGA_NO_RCP", None)
    a = m.formod_host(case.geom)
    os.environ["JUR_EGA_NO_RCP"] = "1"
    b = m.formod_host(case.geom)
    os.environ.pop("JUR_EGA_NO_RCP", None)
    m.close()
    same = {k: bool(np.array_equal(a[k].view(np.uint64), b[k].view(np.uint64))) for k in ("rad", "tau")}
    ndiff = {k: int(np.count_nonzero(a[k].view(np.uint64) != b[k].view(np.uint64))) for k in ("rad", "tau")}
    out[workload] = dict(rays=n, values=int(a["rad"].size), bit_identical=same, differing_values=ndiff)
print(json.dumps(out))
