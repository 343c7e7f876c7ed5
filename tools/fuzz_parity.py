#!/usr/bin/env python3
"""One-off wider run of the randomised parity check of tests/test_parity_gpu.py::test_random_configurations.
usage: python3 tools/fuzz_parity.py [first_seed] [count]        JUR_FUZZ_ARR=fused|batched|batched_grouped picks the
arrangement of the device calls (tests/test_parity_gpu.py ARRANGEMENTS; default fused)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "jurassic-gpu_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import common, test_parity_gpu as T
from oracle import orc
from jurassic_hip import lib
first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
worst = 0.0
skipped = 0
T._state["arr"] = os.environ.get("JUR_FUZZ_ARR", "fused")
assert T._state["arr"] in T.ARRANGEMENTS
if T._state["arr"] == "batched_grouped":
    lib.tune_combine(4, 8, 0)
for seed in range(first, first + count):
    case = T._random_case(seed)
    # the oracle (like the reference) aborts the process when a ray needs >= NLOS points: probe first
    ok = True
    for g in case.geom:
        pass
    try:
        out, ref = T.run_both(lib, orc, case)
    except lib.JurassicError as e:
        if "Too many LOS points" in str(e):
            skipped += 1
            continue
        raise
    T.assert_parity(out, ref)
    T._oracle_cache.clear()
    fin = np.isfinite(ref["rad"])
    worst = max(worst, float(common.rel_err(out["rad"][fin], ref["rad"][fin]).max()))
    if (seed - first) % 100 == 99:
        print("... seeds %d..%d ok, worst so far %.3e" % (first, seed, worst), flush=True)
print("FUZZ_OK (%s) seeds %d..%d, %d skipped (NLOS overflow), worst relative radiance deviation %.3e" % (T._state["arr"], first, first + count - 1, skipped, worst))
