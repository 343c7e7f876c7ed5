#!/usr/bin/env python3
"""Device memory in use after each formod() / jur_dropin_finalize() cycle of one process (does the drop-in state come
back in full?).  usage: debug_finalize_cycles.py [cycles]"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "jurassic-gpu_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch, common
from jurassic_hip import abi, lib, synth, textio
tmp = tempfile.mkdtemp()
os.chdir(tmp)
geom = synth.limb_geometry(300, seed=8)
case = common.limb_case(geom=geom, useGPU=1)
case.write_files(tmp, base="fin")
torch.cuda.init()
used = lambda: torch.cuda.mem_get_info()[1] - torch.cuda.mem_get_info()[0]
print("start", used())
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    obs = abi.obs_t()
    obs.nr = len(geom)
    for c, name in enumerate(textio.OBS_COLS[:7]):
        np.ctypeslib.as_array(getattr(obs, name))[:obs.nr] = geom[:, c]
    lib.formod(case.ctl, case.atm, obs)
    running = used()
    lib.dropin_finalize()
    print("cycle", k, "running", running, "after finalize", used(), flush=True)
