#!/usr/bin/env python3
"""Where the 1e4 .. 1e6-ray regime stands: device-resident time per call over the call size, for
the nadir shape of configs[1] (1 emitter, 3 channels) and the limb shape of configs[2] (5 emitters, 4 channels, 64
profiles): the fused kernel (where its LDS rings admit the shape), the batched kernels with one lane per ray in the
tracer, and with the lanes per ray chosen from the launch size (2 / 4 lanes below 131 072 / 65 536 rays)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "jurassic-gpu_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch
import bench
from jurassic_hip import lib

dev = torch.device("cuda", 0)
sizes = [int(x) for x in (sys.argv[1:] or "10000 20000 50000 100000 200000 500000 1000000".split())]
out = {"what": __doc__.split("\n\n")[0], "unit": "ms per call, device-resident (inputs in HBM), best of the timed steps", "rows": []}
for workload in ("nadir_1e5", "limb_1e6"):
    case = bench.build_case(workload, bench.workload_rays(workload, np.arange(max(sizes))))
    m = lib.Model(case.ctl, case.lib_tables())
    m.set_atm(case.atm)
    nd = case.ctl.nd
    for n in sizes:
        d_geom = torch.from_numpy(np.ascontiguousarray(case.geom[:n].T)).to(dev)
        d_rad = torch.zeros((n, nd), dtype=torch.float64, device=dev)
        d_tau, d_tp = torch.zeros_like(d_rad), torch.zeros((3, n), dtype=torch.float64, device=dev)
        d_st = torch.zeros(1, dtype=torch.int32, device=dev)
        row = {"shape": workload.split("_")[0], "rays": n}
        ref = None
        for name, pencil, lanes in (("batched_1_lane", 0, 1), ("batched_auto_lanes", 0, 0), ("fused", n, 0)):
            if name == "fused" and n > 100000:
                continue
            m.set_pencil(pencil)
            lib.tune_trace(lanes)
            m.reserve(n)

            def step():
                d_rad.zero_()
                m.formod_device(n, d_geom.data_ptr(), d_rad.data_ptr(), d_tau.data_ptr(), d_tp.data_ptr(), 0, d_st.data_ptr(),
                                torch.cuda.current_stream().cuda_stream)
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            m.enable_timing(True)
            best = 1e9
            for _ in range(10):
                t0 = time.perf_counter()
                step()
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            k = m.kernel_ms()
            m.enable_timing(False)
            r = d_rad.clone()
            if ref is None:
                ref = r
            row[name] = {"ms": round(1e3 * best, 4), "M_rays_per_s": round(n / best / 1e6, 2), "same_bits_as_first": bool(torch.equal(r, ref)),
                         "kernel_ms": {a: round(k[a + "_ms"] / max(1, k[a + "_launches"]), 4) for a in ("trace", "ega", "combine") if k[a + "_launches"]},
                         "fused_kernel_ms": round(k["pencil_ms"] / max(1, k["pencil_launches"]), 4) if k["pencil_launches"] else None}
        lib.tune_trace(0)
        out["rows"].append(row)
        print(json.dumps(row), flush=True)
    m.close()
print(json.dumps(out))
