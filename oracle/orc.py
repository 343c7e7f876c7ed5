"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product package never does.
"""
import ctypes as C
import os
import subprocess
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "jurassic-gpu_amd"))
from jurassic_hip import abi  # noqa: E402

_lib = None
dp = C.POINTER(C.c_double)


SUFFIX = os.environ.get("JUR_SUFFIX", "")      # dimension variant, e.g. _nd2378 with JUR_ND / JUR_NG exported


def build():
    subprocess.check_call(["make", "-s", "-C", HERE, "JUR_ND=%d" % abi.ND, "JUR_NG=%d" % abi.NG, "SUFFIX=" + SUFFIX])


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(HERE, "liboracle%s.so" % SUFFIX)
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_tbl_new.restype = C.c_void_p
        L.orc_tbl_new.argtypes = [C.c_int] * 5
        L.orc_tbl_free.argtypes = [C.c_void_p]
        L.orc_tbl_read_ascii.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_tbl_planck_filt.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_tbl_feed_rows.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_long, dp, dp, dp, dp]
        L.orc_tbl_planck_shape.argtypes = [C.c_void_p, C.c_int, C.c_int, dp, dp]
        L.orc_formod.argtypes = [C.c_void_p] * 4
        L.orc_formod_fov.argtypes = [C.c_void_p, C.c_void_p, C.c_int, dp, dp]
        L.orc_formod_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_int] + [dp] * 12 + \
            [C.POINTER(C.c_int), dp, C.c_int]
        L.orc_algorithmic_bytes.restype = C.c_double
        L.orc_algorithmic_bytes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long] + [dp] * 7 + \
            [C.POINTER(C.c_long), dp, dp]
        for name, n in (("orc_ega_eps", 0), ("orc_ctmco2", 4), ("orc_ctmh2o", 5), ("orc_ctmn2", 3),
                        ("orc_ctmo2", 3), ("orc_planck", 2), ("orc_brightness", 2)):
            f = getattr(L, name)
            f.restype = C.c_double
            if n:
                f.argtypes = [C.c_double] * n
        L.orc_ega_eps.argtypes = [C.c_void_p] + [C.c_double] * 4 + [C.c_int, C.c_int]
        L.orc_src_planck.restype = C.c_double
        L.orc_src_planck.argtypes = [C.c_void_p, C.c_double, C.c_int]
        L.orc_new_obs.restype = None
        L.orc_new_obs.argtypes = [C.c_double] * 3 + [dp, dp]
        L.orc_add_surface.restype = None
        L.orc_add_surface.argtypes = [C.c_void_p, C.c_double, C.c_int, dp, C.c_double]
        L.orc_traceray.restype = C.c_int
        L.orc_traceray.argtypes = [C.c_void_p, C.c_void_p] + [dp] * 12
        L.orc_hydrostatic.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_curtis_godson.restype = C.c_int
        L.orc_curtis_godson.argtypes = [C.c_void_p, C.c_void_p, dp, dp, dp, dp]
        L.orc_atm2x.restype = C.c_size_t
        L.orc_atm2x.argtypes = [C.c_void_p, C.c_void_p, dp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_obs2y.restype = C.c_size_t
        L.orc_obs2y.argtypes = [C.c_void_p, C.c_void_p, dp]
        L.orc_intpol_atm.argtypes = [C.c_void_p] * 3
        L.orc_kernel.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, dp, C.c_size_t, C.c_size_t]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(dp)


class Tables:
    """Owner of one orc_tbl_t."""

    def __init__(self, ng, nd, mp=0, mt=0, mu=0):
        self.h = lib().orc_tbl_new(ng, nd, mp, mt, mu)
        self.ng, self.nd = ng, nd

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_tbl_free(self.h)
            self.h = None

    def read_ascii(self, ctl):
        return lib().orc_tbl_read_ascii(self.h, C.byref(ctl))

    def planck_filt(self, ctl):
        return lib().orc_tbl_planck_filt(self.h, C.byref(ctl))

    def feed_rows(self, ig, id_, rows):
        r = np.ascontiguousarray(rows, dtype=np.float64)
        cols = [np.ascontiguousarray(r[:, k]) for k in range(4)]
        lib().orc_tbl_feed_rows(self.h, ig, id_, len(r), *[_p(c) for c in cols])

    def planck_shape(self, id_, nu, f):
        nu = np.ascontiguousarray(nu, dtype=np.float64)
        f = np.ascontiguousarray(f, dtype=np.float64)
        lib().orc_tbl_planck_shape(self.h, id_, len(nu), _p(nu), _p(f))


def formod(ctl, atm, obs, tables):
    lib().orc_formod(C.byref(ctl), C.byref(atm), C.byref(obs), tables.h)


def formod_rays(ctl, atm, tables, geom, rad_in=None, serial_trace=False):
    """geom: (nr, 7) array [time, obsz, obslon, obslat, vpz, vplon, vplat].
    -> dict(rad, tau, tp (nr,3), np, tsurf).
    serial_trace: False/0 packages of 1088 rays with OpenMP inside each (tracing too), True/1 the reference's
    arrangement (tracing serial), 2 every thread traces and integrates its own rays (no packages)."""
    g = np.ascontiguousarray(np.asarray(geom, dtype=np.float64).T)
    nr, nd = g.shape[1], ctl.nd
    rad = np.zeros((nr, nd)) if rad_in is None else np.ascontiguousarray(rad_in, dtype=np.float64).copy()
    tau = np.zeros((nr, nd))
    tp = np.zeros((3, nr))
    npts = np.zeros(nr, dtype=np.int32)
    tsurf = np.zeros(nr)
    lib().orc_formod_rays(C.byref(ctl), C.byref(atm), tables.h, nr, nd, *[_p(g[k]) for k in range(7)],
                          _p(tp[0]), _p(tp[1]), _p(tp[2]), _p(rad), _p(tau),
                          npts.ctypes.data_as(C.POINTER(C.c_int)), _p(tsurf), int(serial_trace))
    return dict(rad=rad, tau=tau, tp=np.ascontiguousarray(tp.T), np=npts, tsurf=tsurf)


def algorithmic_bytes(ctl, atm, tables, geom):
    g = np.ascontiguousarray(np.asarray(geom, dtype=np.float64).T)
    nseg = C.c_long(0)
    tr = C.c_double(0)
    eg = C.c_double(0)
    b = lib().orc_algorithmic_bytes(C.byref(ctl), C.byref(atm), tables.h, g.shape[1],
                                    *[_p(g[k]) for k in range(7)], C.byref(nseg), C.byref(tr), C.byref(eg))
    return dict(total=b, trace=tr.value, ega=eg.value, combine=b - tr.value - eg.value,
                integrate=b - tr.value, segments=nseg.value, rays=g.shape[1])


def ega_eps(tables, ig, id_, tau, t, u, p):
    """ega_eps (jr_common.h:237-268) element by element."""
    f = lib().orc_ega_eps
    return np.array([f(tables.h, a, b, c, d, ig, id_) for a, b, c, d in zip(tau, t, u, p)])


def continua(nu, p, t, q, u_co2, u_h2o):
    """(4, n): continua_ctmco2 / ctmh2o / ctmn2 / ctmo2 (jr_common.h:315-390) at wavenumber nu."""
    L = lib()
    return np.array([[L.orc_ctmco2(nu, a, b, d) for a, b, d in zip(p, t, u_co2)],
                     [L.orc_ctmh2o(nu, a, b, c, e) for a, b, c, e in zip(p, t, q, u_h2o)],
                     [L.orc_ctmn2(nu, a, b) for a, b in zip(p, t)],
                     [L.orc_ctmo2(nu, a, b) for a, b in zip(p, t)]])


def new_obs(tables, id_, t, tau_gas, beta_ds, rad, tau):
    """src_planck_core + new_obs_core per element -> (rad, tau, src)."""
    L = lib()
    rad, tau = np.array(rad, dtype=np.float64), np.array(tau, dtype=np.float64)
    src = np.array([L.orc_src_planck(tables.h, x, id_) for x in t])
    for i in range(len(t)):
        r, tt = C.c_double(rad[i]), C.c_double(tau[i])
        L.orc_new_obs(tau_gas[i], beta_ds[i], src[i], C.byref(r), C.byref(tt))
        rad[i], tau[i] = r.value, tt.value
    return rad, tau, src


def epilogue(tables, id_, nu, tsurf, bbt, rad, tau):
    """add_surface_core, then brightness_core where bbt != 0 -> rad."""
    L = lib()
    rad = np.array(rad, dtype=np.float64)
    for i in range(len(rad)):
        r = C.c_double(rad[i])
        L.orc_add_surface(tables.h, tsurf[i], id_, C.byref(r), tau[i])
        rad[i] = L.orc_brightness(r.value, nu) if bbt[i] else r.value
    return rad


def traceray(ctl, atm, geom7):
    n = abi.NLOS
    g = np.ascontiguousarray(geom7, dtype=np.float64)
    out = {k: np.zeros(n) for k in ("z", "lon", "lat", "p", "t", "ds", "k")}
    q = np.zeros((abi.NG, n))
    u = np.zeros((abi.NG, n))
    tsurf = np.zeros(1)
    tp = np.zeros(3)
    npts = lib().orc_traceray(C.byref(ctl), C.byref(atm), _p(g), _p(out["z"]), _p(out["lon"]), _p(out["lat"]),
                              _p(out["p"]), _p(out["t"]), _p(out["ds"]), _p(out["k"]), _p(q), _p(u),
                              _p(tsurf), _p(tp))
    res = {k: v[:npts] for k, v in out.items()}
    res.update(q=q[:ctl.ng, :npts], u=u[:ctl.ng, :npts], np=npts, tsurf=tsurf[0], tp=tp)
    return res


def state_size(ctl, atm):
    return lib().orc_atm2x(C.byref(ctl), C.byref(atm), None, None, None)


def kernel(ctl, atm, obs, tables):
    """Forward-difference Jacobian (m, n) of the reference's kernel(); obs gets the base result."""
    n = state_size(ctl, atm)
    m = sum(1 for ir in range(obs.nr) for d in range(ctl.nd) if np.isfinite(obs.rad[ir][d]))
    k = np.zeros((m, n))
    lib().orc_kernel(C.byref(ctl), C.byref(atm), C.byref(obs), tables.h, _p(k), m, n)
    return k


def intpol_atm(ctl, dest, src):
    """intpol_atm (jurassic.c:675-804) -> 0 or the negative number of the upstream error."""
    return lib().orc_intpol_atm(C.byref(ctl), C.byref(dest), C.byref(src))


def set_threads(n=0):
    """OpenMP threads of the following calls (0: unchanged) -> current maximum."""
    return lib().orc_set_threads(int(n))


def formod_fov(ctl, obs, dz, w):
    """formod_fov (jurassic.c:214-258) on obs in place, with the shape already read; -> 0 or -1 (fewer than
    two rays share a time stamp: the reference aborts)."""
    dz = np.ascontiguousarray(dz, dtype=np.float64)
    w = np.ascontiguousarray(w, dtype=np.float64)
    return lib().orc_formod_fov(C.byref(ctl), C.byref(obs), len(dz), _p(dz), _p(w))


def curtis_godson(ctl, atm, geom7):
    g = np.ascontiguousarray(geom7, dtype=np.float64)
    out = [np.zeros((max(ctl.ng, 1), abi.NLOS)) for _ in range(3)]
    npts = lib().orc_curtis_godson(C.byref(ctl), C.byref(atm), _p(g), _p(out[0]), _p(out[1]), _p(out[2]))
    return dict(cgp=out[0], cgt=out[1], cgu=out[2], np=npts)
