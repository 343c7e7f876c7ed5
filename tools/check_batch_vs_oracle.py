#!/usr/bin/env python3
"""One-off: a large slice of the bench workloads through the batched kernels against the oracle, EVERY ray (the test
suite compares 3 000 sampled rays of the 1e6-ray batch).  usage: check_batch_vs_oracle.py [limb rays=300000]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "jurassic-gpu_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch  # noqa: F401  (first)
import bench, common
from oracle import orc
from jurassic_hip import lib

out = {}
for workload, n in (("limb_1e6", int(sys.argv[1]) if len(sys.argv) > 1 else 300_000), ("nadir_1e5", 100_000)):
    geom = bench.global_geometry(workload, 1_000_000 if workload.startswith("limb") else n, 1000)[:n]
    case = bench.build_case(workload, geom)
    m = lib.Model(case.ctl, case.lib_tables())
    m.set_atm(case.atm)
    lib.tune_combine(4, 8, 0)                     # the grouped radiance update, whatever the launch size
    t0 = time.time()
    got = m.formod_host(case.geom)
    t1 = time.time()
    orc.set_threads(0)
    ref = orc.formod_rays(case.ctl, case.atm, case.oracle_tables(orc), case.geom, serial_trace=2)
    t2 = time.time()
    m.close()
    fin = np.isfinite(ref["rad"])
    out[workload] = dict(rays=n, hip_s=round(t1 - t0, 3), oracle_s=round(t2 - t1, 1),
                         np_equal=bool(np.array_equal(got["np"], ref["np"])),
                         rad_max_rel=float(common.rel_err(got["rad"][fin], ref["rad"][fin]).max()),
                         tau_max_abs=float(np.abs(got["tau"] - ref["tau"]).max()),
                         tp_max_abs=float(np.abs(got["tp"] - ref["tp"]).max()),
                         # the suite's transmittance criterion (tests/test_parity_gpu.py): worst |dtau| / allowance, and the
                         # worst |dtau| by decade of tau
                         tau_worst_over_allowance=float((np.abs(got["tau"] - ref["tau"]) /
                                                         (1e-9 * np.abs(ref["tau"]) + common.tau_atol(ref["tau"]))).max()),
                         tau_abs_by_decade={"1e%d" % k: float(np.abs(got["tau"] - ref["tau"])[(ref["tau"] >= 10.0 ** k) & (ref["tau"] < 10.0 ** (k + 1))].max(initial=0))
                                            for k in range(-10, 0)},
                         nonfinite=int((~np.isfinite(got["rad"])).sum()))
    print(workload, out[workload], flush=True)
ok = all(v["np_equal"] and v["rad_max_rel"] < 1e-9 and v["tau_max_abs"] < 1e-9 and v["nonfinite"] == 0 for v in out.values())
print(json.dumps(dict(result="PASS" if ok else "FAIL", **out)))
sys.exit(0 if ok else 1)
