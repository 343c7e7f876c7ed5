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
K_mode, t_mode = {}, {}
for name, mode in (("fast", lib.ARITH_FAST), ("exact", lib.ARITH_EXACT)):
    model.set_arithmetic(mode)
    model.kernel(case.atm, obs_of(case.geom))
    t0 = time.perf_counter()
    K_mode[name] = model.kernel(case.atm, obs_of(case.geom))
    t_mode[name] = time.perf_counter() - t0
K = K_mode["fast"]
out = {"what": "forward-difference Jacobian, limb example, jur_kernel under both arithmetics of the look-up (jur_model_set_arithmetic)",
       "rows": K.shape[0], "columns": K.shape[1], "gpu_s": t_mode["fast"], "gpu_columns_per_s": K.shape[1] / t_mode["fast"],
       "gpu_s_exact": t_mode["exact"], "entries_that_differ_between_the_modes": int(np.count_nonzero(K_mode["fast"] != K_mode["exact"]))}
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
    kinds = ["p"] * n0 + ["T"] * n0 + sum([["q%d" % g] * n0 for g in range(c.ng)], []) + ["k"] * n0
    out.update(cpu_s=t_cpu, cpu_columns_per_s=K.shape[1] / t_cpu, cpu_threads=bench.usable_cores(),
               tolerance="1e-6 * max|K[:, j]| + 1e-12 * |y_i| / h_j")
    for name in ("fast", "exact"):
        r = np.abs(K_mode[name] - K_ref) / tol
        i, j = np.unravel_index(np.argmax(r), r.shape)
        # the same comparison with the floor term alone removed: deviation relative to the column's largest entry
        rel_col = np.abs(K_mode[name] - K_ref).max(axis=0) / np.maximum(np.abs(K_ref).max(axis=0), 1e-300)
        out["max_dev_over_tolerance_" + name] = float(r.max())
        out["worst_entry_" + name] = {"row": int(i), "column": int(j), "state": kinds[j], "level": int(j % n0), "h": float(h[j]),
                                      "K": float(K_mode[name][i, j]), "K_ref": float(K_ref[i, j]), "y": float(y[i])}
        out["worst_column_deviation_over_its_largest_entry_" + name] = float(rel_col.max())
    out["max_dev_over_tolerance"] = out["max_dev_over_tolerance_fast"]
    # Who is nearer the derivative?  A forward difference with the reference's steps carries the rounding noise of y
    # divided by h.  For optically thin rays the ALGORITHM forms a segment's emissivity as 1 - (1 - 1e-9): y itself is
    # good to ~1e-12 relative only, in the reference as much as here, and two implementations that round differently
    # (since round 3 the radiance update divides consecutive path products instead of multiplying per-gas quotients)
    # disagree by that much whatever the look-up's arithmetic is -- which is what max_dev_over_tolerance > 1 shows.
    # The yardstick: central differences of the oracle with 4 x the step (noise / 5.7; truncation error of third order,
    # (4h)^2 / 6 times the third derivative, instead of h / 2 times the second).  Per column, the distance of the
    # oracle's and of the GPU's forward differences from it.
    import copy

    def perturbed(j, sign):
        a = copy.deepcopy(case.atm)
        kind, lev = j // n0, j % n0
        arr = (np.ctypeslib.as_array(a.p) if kind == 0 else np.ctypeslib.as_array(a.t) if kind == 1 else
               np.ctypeslib.as_array(a.q)[kind - 2] if kind < 2 + c.ng else np.ctypeslib.as_array(a.k)[0])
        arr[lev] = x0[j] + sign * 4 * h[j]
        return a
    fin = np.isfinite(y)
    K_c = np.zeros_like(K_ref)
    t0 = time.perf_counter()
    for j in range(K_ref.shape[1]):
        yp = orc.formod_rays(c, perturbed(j, +1), tb, case.geom)["rad"].ravel()[fin]
        ym = orc.formod_rays(c, perturbed(j, -1), tb, case.geom)["rad"].ravel()[fin]
        K_c[:, j] = (yp - ym) / (8 * h[j])
    out["central_difference_seconds"] = time.perf_counter() - t0
    dist = {name: np.linalg.norm(Kx - K_c, axis=0) for name, Kx in (("oracle", K_ref), ("fast", K_mode["fast"]), ("exact", K_mode["exact"]))}
    scale = np.linalg.norm(K_c, axis=0) + 1e-300
    for name in ("fast", "exact"):
        worse = (dist[name] - dist["oracle"]) / scale          # > 0: the GPU column lies further from the yardstick
        out["distance_from_central_differences_" + name] = {
            "columns": int(len(scale)), "columns_further_than_the_oracle": int(np.count_nonzero(worse > 0)),
            "worst_excess_over_column_norm": float(worse.max()),
            "median_ratio_gpu_over_oracle": float(np.median(dist[name] / np.maximum(dist["oracle"], 1e-300))),
            "max_relative_distance_gpu": float((dist[name] / scale).max()),
            "max_relative_distance_oracle": float((dist["oracle"] / scale).max())}
print(json.dumps(out))
