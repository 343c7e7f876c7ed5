import os, sys, time, json
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "jurassic-gpu_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import common
from jurassic_hip import lib, synth
res = {}
for nu in ([792.0, 832.0], [792.0, 832.0, 1450.0], common.CTM4_NU, [792.0, 832.0, 1450.0, 2150.0, 700.0, 960.0]):
    nd = len(nu)
    n = 500_000
    geom = synth.limb_geometry(n, seed=1000, nprofiles=64)
    case = common.limb_case(geom=geom, nu=nu, nprofiles=64)
    m = lib.Model(case.ctl, case.lib_tables()); m.set_atm(case.atm)
    dev = torch.device("cuda", 0)
    d_geom = torch.from_numpy(np.ascontiguousarray(case.geom.T)).to(dev)
    d_rad = torch.zeros((n, nd), dtype=torch.float64, device=dev); d_tau = torch.zeros_like(d_rad)
    d_tp = torch.zeros((3, n), dtype=torch.float64, device=dev); d_np = torch.zeros(n, dtype=torch.int32, device=dev)
    d_st = torch.zeros(1, dtype=torch.int32, device=dev)
    m.reserve(n)
    out = {}
    for name, args in (("single", (0, 8, 0)), ("grouped", (4, 8, 0)), ("grouped6", (6, 8, 0))):
        lib.tune_combine(*args)
        m.enable_timing(True)
        for _ in range(4):
            d_rad.zero_()
            m.formod_device(n, d_geom.data_ptr(), d_rad.data_ptr(), d_tau.data_ptr(), d_tp.data_ptr(), d_np.data_ptr(), d_st.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        k = m.kernel_ms()
        out[name] = round(k["combine_ms"] / k["combine_launches"], 3)
        m.enable_timing(False)
    res[nd] = out
    m.close()
print(json.dumps(res))
