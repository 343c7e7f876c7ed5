import sys, time
sys.path[:0] = ['/root/repo', '/root/repo/jurassic-gpu_amd', '/root/repo/tests']
import numpy as np
from oracle import orc
from jurassic_hip import lib
import common
orc.build()
for name, case in (("limb", common.limb_case()), ("nadir", common.nadir_case()), ("limb4", common.limb_case(nu=common.CTM4_NU))):
    ot = case.oracle_tables(orc)
    t0=time.time(); ref = orc.formod_rays(case.ctl, case.atm, ot, case.geom); t1=time.time()
    m = lib.Model(case.ctl, case.lib_tables())
    m.set_atm(case.atm)
    out = m.formod_host(case.geom); t2=time.time()
    out = m.formod_host(case.geom); t3=time.time()
    print(name, "oracle %.3fs gpu(first) %.3fs gpu %.4fs"%(t1-t0,t2-t1,t3-t2))
    print("  np equal:", np.array_equal(ref['np'], out['np']), ref['np'][:5], out['np'][:5])
    print("  rad max rel", common.rel_err(out['rad'], ref['rad']).max(), " tau max rel", common.rel_err(out['tau'], ref['tau']).max())
    print("  tp max abs", np.abs(out['tp']-ref['tp']).max(axis=0))
    print("  rad sample", ref['rad'][0], out['rad'][0], ref['tau'][0], out['tau'][0])
