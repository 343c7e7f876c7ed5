"""ctypes mirrors of the structs that cross the formod() boundary.

Field order and sizes follow include/jurassic_abi.h, which restates the
reference's src/jurassic.h:215-226 (atm_t), :229-347 (ctl_t), :371-385 (obs_t).
"""
import ctypes as C

import os

# ND / NG are compile-time dimensions of the C side (reference src/jurassic.h:138-145); a library built with
# other values (make JUR_ND=... JUR_NG=... SUFFIX=...) is used by exporting the same numbers before import.
ND = int(os.environ.get("JUR_ND", "100"))
NG = int(os.environ.get("JUR_NG", "30"))
NP, NR, NW, LEN, NLOS = 9600, 1088, 1, 5000, 400
NSHAPE, NFOV = 2048, 5
TBLNP, TBLNT, TBLNU, TBLNS = 40, 30, 304, 1201

d = C.c_double
i = C.c_int


class atm_t(C.Structure):
    _fields_ = [("time", d * NP), ("z", d * NP), ("lon", d * NP), ("lat", d * NP),
                ("p", d * NP), ("t", d * NP), ("q", (d * NP) * NG), ("k", (d * NP) * NW),
                ("np", i), ("init", i)]


class ctl_t(C.Structure):
    _fields_ = [("ng", i), ("emitter", (C.c_char * LEN) * NG), ("nd", i), ("nw", i),
                ("nu", d * ND), ("window", i * ND), ("tblbase", C.c_char * LEN),
                ("hydz", d), ("ctm_co2", i), ("ctm_h2o", i), ("ctm_n2", i), ("ctm_o2", i),
                ("ip", i), ("cz", d), ("cx", d), ("refrac", i), ("rayds", d), ("raydz", d),
                ("fov", C.c_char * LEN),
                ("retp_zmin", d), ("retp_zmax", d), ("rett_zmin", d), ("rett_zmax", d),
                ("retq_zmin", d * NG), ("retq_zmax", d * NG), ("retk_zmin", d * NW), ("retk_zmax", d * NW),
                ("write_bbt", i), ("write_matrix", i), ("formod", i),
                ("rfmbin", C.c_char * LEN), ("rfmhit", C.c_char * LEN), ("rfmxsc", (C.c_char * LEN) * NG),
                ("useGPU", i), ("checkmode", i), ("MPIglobrank", i), ("MPIlocalrank", i),
                ("read_binary", i), ("write_binary", i), ("gpu_nbytes_shared_memory", i)]


class obs_t(C.Structure):
    _fields_ = [("time", d * NR), ("obsz", d * NR), ("obslon", d * NR), ("obslat", d * NR),
                ("vpz", d * NR), ("vplon", d * NR), ("vplat", d * NR),
                ("tpz", d * NR), ("tplon", d * NR), ("tplat", d * NR),
                ("tau", (d * ND) * NR), ("rad", (d * ND) * NR), ("nr", i)]


assert (ND, NG) != (100, 30) or (C.sizeof(ctl_t), C.sizeof(atm_t), C.sizeof(obs_t)) == (321856, 2841608, 1827848)


def make_ctl(emitters, nu, tblbase="-", **kw):
    """Control block with read_ctl's defaults (reference src/jurassic.c:920-1021),
    including the automatic continuum switch-off at :954-968."""
    ctl = ctl_t()
    ctl.ng = len(emitters)
    for g, name in enumerate(emitters):
        ctl.emitter[g].value = name.encode()
    ctl.nd = len(nu)
    ctl.nw = 1
    for k, v in enumerate(nu):
        ctl.nu[k] = float(v)
        ctl.window[k] = 0
    ctl.tblbase = tblbase.encode()
    ctl.hydz = -999.0
    ctl.ctm_co2 = ctl.ctm_h2o = ctl.ctm_n2 = ctl.ctm_o2 = 1
    ctl.ip = 1
    ctl.refrac = 1
    ctl.rayds = 10.0
    ctl.raydz = 0.5
    ctl.fov = b"-"
    ctl.retp_zmin = ctl.retp_zmax = ctl.rett_zmin = ctl.rett_zmax = -999.0
    for g in range(NG):
        ctl.retq_zmin[g] = ctl.retq_zmax[g] = -999.0
    for w in range(NW):
        ctl.retk_zmin[w] = ctl.retk_zmax[w] = -999.0
    ctl.formod = 2
    ctl.read_binary = 0      # as the example control files set them (limb.ctl:23-24); read_ctl's own
    ctl.write_binary = 0     # defaults are -1 / 1
    for k, v in kw.items():
        setattr(ctl, k, v)
    if "ctm_auto" not in kw:
        if not any(x < 4000 for x in nu): ctl.ctm_co2 = 0
        if not any(x < 20000 for x in nu): ctl.ctm_h2o = 0
        if not any(2120 <= x <= 2605 for x in nu): ctl.ctm_n2 = 0
        if not any(1360 <= x <= 1805 for x in nu): ctl.ctm_o2 = 0
    return ctl
