"""Readers/writers for the reference's whitespace tables (host logic).

Formats: reference src/jurassic.c:882-917 (read_atm), :1041-1068 (read_obs),
:1426-1470 (write_obs).  Lines that do not parse are skipped, as the TOK macro
does (jurassic.h:95-99).
"""
import numpy as np
from . import abi


def _rows(path, ncol):
    out = []
    with open(path) as fh:
        for line in fh:
            tok = line.split()
            if len(tok) < ncol:
                continue
            try:
                out.append([float(t) for t in tok[:ncol]])
            except ValueError:
                continue
    return np.array(out, dtype=np.float64).reshape(-1, ncol)


def read_atm(path, ctl):
    a = _rows(path, 6 + ctl.ng + ctl.nw)
    atm = abi.atm_t()
    n = len(a)
    assert 0 < n <= abi.NP
    atm.np = n
    for name, col in (("time", 0), ("z", 1), ("lon", 2), ("lat", 3), ("p", 4), ("t", 5)):
        np.ctypeslib.as_array(getattr(atm, name))[:n] = a[:, col]
    q = np.ctypeslib.as_array(atm.q)
    for g in range(ctl.ng):
        q[g, :n] = a[:, 6 + g]
    k = np.ctypeslib.as_array(atm.k)
    for w in range(ctl.nw):
        k[w, :n] = a[:, 6 + ctl.ng + w]
    return atm


OBS_COLS = ("time", "obsz", "obslon", "obslat", "vpz", "vplon", "vplat", "tpz", "tplon", "tplat")


def read_obs_array(path, nd):
    """-> (n, 10 + 2 nd) array: geometry, rad[nd], tau[nd]."""
    return _rows(path, 10 + 2 * nd)


def read_obs(path, ctl, max_rays=None):
    a = read_obs_array(path, ctl.nd)
    if max_rays:
        a = a[:max_rays]
    obs = abi.obs_t()
    n = len(a)
    assert 0 < n <= abi.NR
    obs.nr = n
    for c, name in enumerate(OBS_COLS):
        np.ctypeslib.as_array(getattr(obs, name))[:n] = a[:, c]
    np.ctypeslib.as_array(obs.rad)[:n, :ctl.nd] = a[:, 10:10 + ctl.nd]
    np.ctypeslib.as_array(obs.tau)[:n, :ctl.nd] = a[:, 10 + ctl.nd:10 + 2 * ctl.nd]
    return obs


def fmt_g(x):
    """C's %g for one double."""
    return "%g" % x


def obs_line(time, geom9, rad, tau):
    """One data line exactly as write_obs prints it (jurassic.c:1455-1466)."""
    s = "%.2f" % time + "".join(" %g" % v for v in geom9)
    s += "".join(" %g" % v for v in rad) + "".join(" %g" % v for v in tau)
    return s
