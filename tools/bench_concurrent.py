#!/usr/bin/env python3
"""Do package-sized calls on different streams overlap on the GPU?  One model, N streams, N sets of device
buffers: enqueue N calls of jur_formod_device (fused kernel) without a host wait in between, then wait once."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "jurassic-gpu_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch
import common
from jurassic_hip import lib, synth

nr = 1088
case = common.limb_case(geom=synth.limb_geometry(nr, seed=1))
m = lib.Model(case.ctl, case.lib_tables())
m.set_atm(case.atm)
dev = torch.device("cuda", 0)
nd = case.ctl.nd
out = {}
for nstream in (1, 2, 4, 8, 16):
    streams = [torch.cuda.Stream() for _ in range(nstream)]
    bufs = []
    for s in range(nstream):
        g = torch.from_numpy(np.ascontiguousarray(synth.limb_geometry(nr, seed=10 + s).T)).to(dev)
        bufs.append((g, torch.zeros((nr, nd), dtype=torch.float64, device=dev), torch.zeros((nr, nd), dtype=torch.float64, device=dev),
                     torch.zeros((3, nr), dtype=torch.float64, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)))
    def round_():
        for s, (g, rad, tau, tp, st) in zip(streams, bufs):
            m.formod_device(nr, g.data_ptr(), rad.data_ptr(), tau.data_ptr(), tp.data_ptr(), 0, st.data_ptr(), s.cuda_stream)
        torch.cuda.synchronize()
    round_()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        round_()
    dt = (time.perf_counter() - t0) / n
    out[nstream] = dict(ms_per_round=1e3 * dt, rays_per_s=nstream * nr / dt)
print(json.dumps(out))
