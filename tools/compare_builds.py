#!/usr/bin/env python3
"""Bit-for-bit comparison of two builds of the library on the bench workloads.

usage: compare_builds.py A.so B.so [limb rays]   -- each build runs in its own process
       (JURASSIC_HIP_SO), radiance / transmittance / tangent points are compared bitwise."""
import os, subprocess, sys, json, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == "--dump":
    sys.path[:0] = [ROOT, os.path.join(ROOT, "jurassic-gpu_amd"), os.path.join(ROOT, "tests")]
    import numpy as np
    import bench
    from jurassic_hip import lib
    out = {}
    for workload, n in (("limb_1e6", int(sys.argv[3])), ("nadir_1e5", 100_000)):
        case = bench.build_case(workload, bench.global_geometry(workload, n, 1000))
        m = lib.Model(case.ctl, case.lib_tables())
        m.set_atm(case.atm)
        r = m.formod_host(case.geom)
        m.close()
        for k in ("rad", "tau", "tp"):
            out[f"{workload}_{k}"] = r[k]
    np.savez(sys.argv[2], **out)
    sys.exit(0)
import numpy as np
a_so, b_so = sys.argv[1], sys.argv[2]
n = sys.argv[3] if len(sys.argv) > 3 else "200000"
files = []
for so in (a_so, b_so):
    f = tempfile.mktemp(suffix=".npz")
    subprocess.check_call([sys.executable, __file__, "--dump", f, n], env=dict(os.environ, JURASSIC_HIP_SO=os.path.abspath(so)))
    files.append(f)
A, B = np.load(files[0]), np.load(files[1])
res = {}
for k in A.files:
    x, y = A[k].view(np.uint64), B[k].view(np.uint64)
    d = A[k] - B[k]
    res[k] = dict(values=int(x.size), differing=int(np.count_nonzero(x != y)),
                  max_rel=float(np.nanmax(np.abs(d) / np.maximum(np.abs(A[k]), 1e-300))) if x.size else 0.0)
print(json.dumps(res))
for f in files:
    os.remove(f)
