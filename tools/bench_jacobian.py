#!/usr/bin/env python3
"""Jacobian throughput: reference-style kernel() = n+1 forward models
(oracle, OpenMP over rays, as jurassic.c:812-857 runs them one after the other) against
jur_kernel = one batched call on the GPU.  Limb example geometry (66 rays, 2 channels, 5
emitters), state vector = p, T, all 5 mixing ratios and extinction on all 91 levels."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "jurassic-gpu_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import common
from jurassic_hip import abi, lib
from oracle import orc

case = common.limb_case()
c = case.ctl
c.retp_zmin, c.retp_zmax, c.rett_zmin, c.rett_zmax = 0.0, 90.0, 0.0, 90.0
for g in range(c.ng):
    c.retq_zmin[g], c.retq_zmax[g] = 0.0, 90.0
c.retk_zmin[0], c.retk_zmax[0] = 0.0, 90.0


def obs_of(geom):
    o = abi.obs_t()
    o.nr = len(geom)
    for k, name in enumerate(("time", "obsz", "obslon", "obslat", "vpz", "vplon", "vplat")):
        np.ctypeslib.as_array(getattr(o, name))[:o.nr] = geom[:, k]
    return o


model = lib.Model(c, case.lib_tables())
model.set_atm(case.atm)
obs = obs_of(case.geom)
model.kernel(case.atm, obs)                    # warm-up (workspace allocation)
t0 = time.perf_counter()
K = model.kernel(case.atm, obs_of(case.geom))
t_gpu = time.perf_counter() - t0
out = {"what": "forward-difference Jacobian, limb example", "rows": K.shape[0], "columns": K.shape[1],
       "gpu_s": t_gpu, "gpu_columns_per_s": K.shape[1] / t_gpu}
if "--no-cpu" not in sys.argv:
    orc.build()
    import bench
    orc.set_threads(bench.usable_cores())   # the cores the box grants (cgroup quota), not the affinity mask
    tb = case.oracle_tables(orc)
    t0 = time.perf_counter()
    K_ref = orc.kernel(c, case.atm, obs_of(case.geom), tb)
    t_cpu = time.perf_counter() - t0
    # columns are (y1 - y0)/h: last-bit differences of y between the two implementations enter
    # as ~1e-12 |y| / h (h is as small as 1e-15 for trace-gas mixing ratios near zero), so each
    # column is judged against max(|K_ref|) plus that floor
    n0 = case.atm.np
    x0 = np.concatenate([np.ctypeslib.as_array(case.atm.p)[:n0], np.ctypeslib.as_array(case.atm.t)[:n0]] +
                        [np.ctypeslib.as_array(case.atm.q)[g, :n0] for g in range(c.ng)] +
                        [np.ctypeslib.as_array(case.atm.k)[0, :n0]])
    h = np.concatenate([np.maximum(np.abs(0.01 * x0[:n0]), 1e-7), np.ones(n0),
                        np.maximum(np.abs(0.01 * x0[2 * n0:(2 + c.ng) * n0]), 1e-15), np.full(n0, 1e-4)])
    y = np.ctypeslib.as_array(obs.rad)[:obs.nr, :c.nd].ravel()
    tol = 1e-6 * np.abs(K_ref).max(axis=0)[None, :] + 1e-12 * np.abs(y)[:, None] / h[None, :]
    out.update(cpu_s=t_cpu, cpu_columns_per_s=K.shape[1] / t_cpu, cpu_threads=bench.usable_cores(),
               max_dev_over_tolerance=float(np.max(np.abs(K - K_ref) / tol)),
               tolerance="1e-6 * max|K[:, j]| + 1e-12 * |y_i| / h_j")
print(json.dumps(out))
