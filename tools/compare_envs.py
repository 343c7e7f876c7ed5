#!/usr/bin/env python3
"""Bit-for-bit comparison of the bench workloads under two settings of the library's environment switches.

usage: compare_envs.py "JUR_X=0" "JUR_X=1 JUR_Y=2" [limb rays]   -- each setting runs in its own process
       (the switches are read once per process); radiance / transmittance / tangent points compared bitwise."""
import os, subprocess, sys, json, tempfile
HERE = os.path.dirname(os.path.abspath(__file__))
a_env, b_env = sys.argv[1], sys.argv[2]
n = sys.argv[3] if len(sys.argv) > 3 else "200000"
import numpy as np
files = []
for spec in (a_env, b_env):
    f = tempfile.mktemp(suffix=".npz")
    env = dict(os.environ)
    env.update(kv.split("=", 1) for kv in spec.split())
    subprocess.check_call([sys.executable, os.path.join(HERE, "compare_builds.py"), "--dump", f, n], env=env)
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
