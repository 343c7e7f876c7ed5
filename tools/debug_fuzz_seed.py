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
for seed in range(first, first + count):
    case = T._random_case(seed)
    try:
        out, ref = T.run_both(lib, orc, case)
    except lib.JurassicError:
        continue
    try:
        T.assert_parity(out, ref)
    except AssertionError:
        d = np.abs(out["tau"] - ref["tau"])
        k = np.unravel_index(np.argmax(d / (1e-9 * np.abs(ref["tau"]) + 1e-13)), d.shape)
        fin = np.isfinite(ref["rad"])
        print("seed", seed, "ng", case.ctl.ng, "nd", case.ctl.nd, "rays", len(case.geom), "worst tau at", k, "hip %.17g" % out["tau"][k],
              "oracle %.17g" % ref["tau"][k], "abs %.3e" % d[k], "rad max rel %.3e" % common.rel_err(out["rad"][fin], ref["rad"][fin]).max(),
              "np", int(out["np"][k[0]]), "hex", float(out["tau"][k]).hex(), flush=True)
print("done", first, count)
