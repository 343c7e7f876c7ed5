#!/usr/bin/env python3
"""Which seeds of a range fail the randomised parity check, and by how much (HIP path against the oracle).
usage: debug_fuzz_seed.py first count"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "jurassic-gpu_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import common, test_parity_gpu as T
from oracle import orc
from jurassic_hip import lib
first, count = int(sys.argv[1]), int(sys.argv[2])
T._state["arr"] = os.environ.get("JUR_FUZZ_ARR", "fused")            # arrangement of the device calls, as tools/fuzz_parity.py
if T._state["arr"] in ("batched_grouped",):
    lib.tune_combine(4, 8, 0)
for seed in range(first, first + count):
    case = T._random_case(seed)
    try:
        out, ref = T.run_both(lib, orc, case)
    except lib.JurassicError:
        continue
    T._oracle_cache.clear()
    try:
        T.assert_parity(out, ref)
    except AssertionError:
        dtp = np.abs(out["tp"] - ref["tp"])
        kt = np.unravel_index(np.argmax(dtp), dtp.shape)
        print("seed", seed, "worst tangent-point deviation at ray", kt[0], "column", kt[1], "%.3e" % dtp[kt], "hip", out["tp"][kt[0]], "oracle", ref["tp"][kt[0]],
              "geometry", case.geom[kt[0]], flush=True)
        d = np.abs(out["tau"] - ref["tau"])
        k = np.unravel_index(np.argmax(d / (1e-9 * np.abs(ref["tau"]) + 1e-13)), d.shape)
        fin = np.isfinite(ref["rad"])
        print("seed", seed, "ng", case.ctl.ng, "nd", case.ctl.nd, "rays", len(case.geom), "worst tau at", k, "hip %.17g" % out["tau"][k],
              "oracle %.17g" % ref["tau"][k], "abs %.3e" % d[k], "rad max rel %.3e" % common.rel_err(out["rad"][fin], ref["rad"][fin]).max(),
              "np", int(out["np"][k[0]]), "hex", float(out["tau"][k]).hex(), flush=True)
print("done", first, count)
