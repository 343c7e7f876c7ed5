"""Atmosphere regridding, the step in front of the path for atmospheres that are not one profile (SURVEY 8f row
f-4, second half): intpol_atm with IP = 1 / 2 / 3 (reference jurassic.c:675-804).  The hot path itself asserts
IP == 1 upstream (jr_common.h:573), so these functions are exercised on their own: the oracle's restatement against
closed-form expectations (CPU), the HIP kernel against the oracle (GPU)."""
import ctypes as C
import os
import numpy as np
import pytest
import common
from jurassic_hip import abi, textio

NG = 3


def _ctl(ip, cx=0.0, cz=0.0):
    ctl = abi.make_ctl(["CO2", "H2O", "O3"], [792.0], ip=ip)
    ctl.cx, ctl.cz = cx, cz
    return ctl


def _fill(atm, z, lon, lat, p, t, q, k):
    n = len(z)
    atm.np = n
    for name, a in (("z", z), ("lon", lon), ("lat", lat), ("p", p), ("t", t)):
        np.ctypeslib.as_array(getattr(atm, name))[:n] = a
    np.ctypeslib.as_array(atm.q)[:NG, :n] = q
    np.ctypeslib.as_array(atm.k)[:1, :n] = k
    return atm


def _profile():
    base = textio.read_atm(os.path.join(common.GOLD, "limb", "atm.tab"), abi.make_ctl(common.LIMB_EMITTERS, [792.0]))
    n = base.np
    g = lambda name: np.ctypeslib.as_array(getattr(base, name))[:n].copy()
    return g("z"), g("p"), g("t"), np.ctypeslib.as_array(base.q)[:NG, :n].copy(), np.ctypeslib.as_array(base.k)[:1, :n].copy()


def track(nprof=12, seed=0):
    """A satellite track: nprof profiles 3 degrees apart, each the limb profile scaled in p and shifted in T."""
    rng = np.random.default_rng(seed)
    z, p, t, q, k = _profile()
    n = len(z)
    lat = np.repeat(-15.0 + 3.0 * np.arange(nprof), n)
    lon = np.repeat(10.0 + rng.uniform(-2, 2, nprof), n)
    fp, ft = 1 + rng.uniform(-0.05, 0.05, nprof), rng.uniform(-20, 20, nprof)
    return _fill(abi.atm_t(), np.tile(z, nprof), lon, lat, (p[None] * fp[:, None]).ravel(), (t[None] + ft[:, None]).ravel(),
                 np.tile(q, (1, nprof)) * np.repeat(1 + 0.1 * np.arange(nprof), n)[None], np.tile(k, (1, nprof)) + 1e-4)


def cloud(n=3000, seed=1):
    rng = np.random.default_rng(seed)
    z, lon, lat = rng.uniform(0, 60, n), rng.uniform(-5, 5, n), rng.uniform(40, 50, n)
    p = 1013.25 * np.exp(-z / 7.0) * (1 + 0.01 * (lat - 45))
    t = 288.0 - 6.5 * np.minimum(z, 11) + 0.3 * lon
    q = np.stack([3.7e-4 + 0 * z, 1e-3 * np.exp(-z / 2.0), 1e-6 * (1 + z / 10)])
    return _fill(abi.atm_t(), z, lon, lat, p, t, q, 1e-4 * np.exp(-z / 5.0)[None])


def targets(n, seed, lat=(-20, 20), lon=(5, 15), z=(-2, 95)):
    rng = np.random.default_rng(seed)
    return _fill(abi.atm_t(), rng.uniform(*z, n), rng.uniform(*lon, n), rng.uniform(*lat, n), np.zeros(n), np.zeros(n),
                 np.zeros((NG, n)), np.zeros((1, n)))


def values(atm):
    n = atm.np
    return {"p": np.ctypeslib.as_array(atm.p)[:n].copy(), "t": np.ctypeslib.as_array(atm.t)[:n].copy(),
            "q": np.ctypeslib.as_array(atm.q)[:NG, :n].copy(), "k": np.ctypeslib.as_array(atm.k)[:1, :n].copy()}


def test_oracle_regridding_against_closed_forms(oracle):
    z, p, t, q, k = _profile()
    one = _fill(abi.atm_t(), z, 0 * z, 0 * z, p, t, q, k)
    dest = targets(400, 3, z=(0.5, 89.5))
    assert oracle.intpol_atm(_ctl(1), dest, one) == 0
    v = values(dest)
    zd = np.ctypeslib.as_array(dest.z)[:400]
    assert np.allclose(v["t"], np.interp(zd, z, t), rtol=1e-13) and np.allclose(v["q"][1], np.interp(zd, z, q[1]), rtol=1e-12)
    assert np.allclose(np.log(v["p"]), np.interp(zd, z, np.log(p)), rtol=1e-12)                  # exponential in p
    # IP = 2: at the position of a profile of the track that profile comes back (weight 0 for the neighbour)
    tr = track()
    n = len(z)
    dest = _fill(abi.atm_t(), z[5:80], np.full(75, np.ctypeslib.as_array(tr.lon)[4 * n]), np.full(75, -3.0), 0 * z[5:80], 0 * z[5:80],
                 np.zeros((NG, 75)), np.zeros((1, 75)))
    assert oracle.intpol_atm(_ctl(2), dest, tr) == 0
    assert np.allclose(values(dest)["t"], np.ctypeslib.as_array(tr.t)[4 * n + 5:4 * n + 80], rtol=1e-13)
    # ... and half way between two profiles their mean
    mid = targets(50, 4, lat=(-1.5, -1.5), lon=(10, 10), z=(5, 60))
    np.ctypeslib.as_array(tr.lon)[:] = 10.0
    tr.init = 0
    assert oracle.intpol_atm(_ctl(2), mid, tr) == 0
    zm = np.ctypeslib.as_array(mid.z)[:50]
    ta, tb = (np.interp(zm, z, np.ctypeslib.as_array(tr.t)[i * n:(i + 1) * n]) for i in (4, 5))
    assert np.allclose(values(mid)["t"], 0.5 * (ta + tb), rtol=1e-9)
    # IP = 3: a lone source point inside the radius of influence is returned, nothing in reach gives NaN
    src = _fill(abi.atm_t(), np.array([10.0, 50.0]), np.array([0.0, 0.0]), np.array([45.0, 45.0]), np.array([250.0, 1.0]),
                np.array([220.0, 270.0]), np.ones((NG, 2)) * [[1.0, 2.0]], np.array([[0.1, 0.2]]))
    dest = _fill(abi.atm_t(), np.array([11.0, 30.0]), np.array([0.5, 0.0]), np.array([45.2, 45.0]), np.zeros(2), np.zeros(2),
                 np.zeros((NG, 2)), np.zeros((1, 2)))
    assert oracle.intpol_atm(_ctl(3, cx=200.0, cz=3.0), dest, src) == 0
    v = values(dest)
    assert abs(v["t"][0] - 220.0) < 1e-12 and abs(v["p"][0] - 250.0) < 1e-12        # (w * x) / w
    assert np.isnan(v["t"][1]) and np.isnan(v["q"][:, 1]).all()
    # what upstream aborts on
    bad = _fill(abi.atm_t(), np.array([1.0, 2.0, 3.0]), np.array([0.0, 0.0, 1.0]), np.array([0.0, 0.0, 1.0]), np.ones(3), np.ones(3),
                np.ones((NG, 3)), np.ones((1, 3)))
    assert oracle.intpol_atm(_ctl(2), dest, bad) == -2                  # a "profile" of one point
    far = track(3)
    np.ctypeslib.as_array(far.lat)[len(z):2 * len(z)] = 40.0
    assert oracle.intpol_atm(_ctl(2), dest, far) == -3                  # profiles more than 10 degrees apart
    assert oracle.intpol_atm(_ctl(4), dest, far) == -4


@pytest.mark.gpu
def test_hip_regridding_matches_the_oracle(oracle):
    from jurassic_hip import lib
    z, p, t, q, k = _profile()
    one = _fill(abi.atm_t(), z, 0 * z, 0 * z, p, t, q, k)
    cases = [(_ctl(1), one, targets(5000, 5)), (_ctl(2), track(), targets(6000, 6)),
             (_ctl(3, cx=300.0, cz=4.0), cloud(), targets(4000, 7, lat=(38, 52), lon=(-7, 7), z=(-3, 66)))]
    for ctl, src, dest in cases:
        ref = abi.atm_t()
        C.memmove(C.byref(ref), C.byref(dest), C.sizeof(abi.atm_t))
        assert oracle.intpol_atm(ctl, ref, src) == 0
        lib.intpol_atm(ctl, dest, src)
        a, b = values(dest), values(ref)
        nan = np.isnan(b["t"])
        assert np.array_equal(np.isnan(a["t"]), nan) and (ctl.ip != 3 or 0 < nan.sum() < len(nan))
        ok = ~nan
        for key in ("t", "q", "k"):                       # pure IEEE arithmetic in the reference's order: same doubles
            x, y = a[key][..., ok], b[key][..., ok]
            assert np.array_equal(x.view(np.uint64), y.view(np.uint64)), (ctl.ip, key)
        if ctl.ip == 3:
            assert np.array_equal(a["p"][ok].view(np.uint64), b["p"][ok].view(np.uint64))
        else:                                             # exp / log of the device library
            assert np.max(np.abs(a["p"][ok] / b["p"][ok] - 1)) < 1e-13
    bad = _fill(abi.atm_t(), np.array([1.0, 2.0, 3.0]), np.array([0.0, 0.0, 1.0]), np.array([0.0, 0.0, 1.0]), np.ones(3), np.ones(3),
                np.ones((NG, 3)), np.ones((1, 3)))
    with pytest.raises(lib.JurassicError, match="Cannot identify profiles"):
        lib.intpol_atm(_ctl(2), targets(4, 1), bad)
    far = track(3)
    np.ctypeslib.as_array(far.lat)[len(z):2 * len(z)] = 40.0
    with pytest.raises(lib.JurassicError, match="Distance of profiles"):
        lib.intpol_atm(_ctl(2), targets(4, 1), far)
    with pytest.raises(lib.JurassicError, match="Unknown interpolation"):
        lib.intpol_atm(_ctl(4), targets(4, 1), far)
