"""CPU tests: the oracle against the reference's own golden files and against
physics identities that need no reference run."""
import os
import numpy as np
import pytest
import common
from jurassic_hip import abi, synth, textio

GOLD = common.GOLD


def g(x):
    return "%g" % x


def test_limb_tangent_points_match_rad_org(oracle):
    """example/limb/rad.org columns 8-10 (reference run.sh `diff rad.tab rad.org`):
    tpz and tplat text-identical at %g, tplon (atan2 noise ~1e-8 deg) to 1e-11."""
    case = common.limb_case()
    ref = textio.read_obs_array(os.path.join(GOLD, "limb", "rad.org"), 2)
    assert len(ref) == 66
    res = oracle.formod_rays(case.ctl, case.atm, case.oracle_tables(oracle), ref[:, :7])
    for i, row in enumerate(ref):
        assert g(res["tp"][i, 0]) == g(row[7]), (i, res["tp"][i], row[7:10])
        assert g(res["tp"][i, 2]) == g(row[9])
        assert abs(res["tp"][i, 1] - row[8]) < 1e-11
    assert (res["np"].min(), res["np"].max()) == (122, 393)   # SURVEY.md section 6 probe of the reference


def test_nadir_tangent_points_match_rad_org(oracle):
    case = common.nadir_case()
    ref = textio.read_obs_array(os.path.join(GOLD, "nadir", "rad.org"), 3)
    assert len(ref) == 90
    res = oracle.formod_rays(case.ctl, case.atm, case.oracle_tables(oracle), ref[:, :7])
    for i, row in enumerate(ref):
        assert g(res["tp"][i, 2]) == g(row[9])
        assert abs(res["tp"][i, 0] - row[7]) < 1e-11      # ground hit: |tpz| ~ 1e-7 km of round-off
        assert abs(res["tp"][i, 1] - row[8]) < 1e-11
    assert set(res["np"]) <= {181, 182}
    assert np.all(res["tsurf"] > 200)


def test_limb_generator_reproduces_obs_tab():
    """limb.c:49-59: vplat = acos((RE+z)/(RE+obsz)), tangent heights 3..68 km."""
    ctl = abi.make_ctl(common.LIMB_EMITTERS, common.LIMB_NU)
    shipped = textio.read_obs_array(os.path.join(GOLD, "limb", "obs.tab"), ctl.nd)
    mine = synth.limb_geometry(66, scan=True)
    for a, b in zip(mine, shipped):
        assert [g(v) for v in a] == [g(v) for v in b[:7]]


def _isothermal(case, temp=250.0):
    n = case.atm.np
    np.ctypeslib.as_array(case.atm.t)[:n] = temp
    return temp


def test_isothermal_radiance_is_planck_times_absorptance(oracle):
    """With T constant along the path the update rad += B*eps*tau, tau *= 1-eps
    (jr_common.h:296-298) telescopes to rad = B(T) (1 - tau)."""
    case = common.limb_case(nu=common.CTM4_NU)
    temp = _isothermal(case)
    ot = case.oracle_tables(oracle)
    res = oracle.formod_rays(case.ctl, case.atm, ot, case.geom)
    for d, (x, f) in enumerate(case.filters):
        b = sum(fi * oracle.lib().orc_planck(temp, xi) for xi, fi in zip(x, f)) / f.sum()
        assert np.allclose(res["rad"][:, d], b * (1 - res["tau"][:, d]), rtol=1e-12, atol=0)


def test_extinction_only_is_beer_lambert(oracle):
    case = common.limb_case(missing={(g_, d) for g_ in range(5) for d in range(2)},
                            ctm_co2=0, ctm_h2o=0, ctm_n2=0, ctm_o2=0)
    n = case.atm.np
    np.ctypeslib.as_array(case.atm.k)[0, :n] = 2e-4
    ot = case.oracle_tables(oracle)
    res = oracle.formod_rays(case.ctl, case.atm, ot, case.geom)
    for i, geom in enumerate(case.geom[::8]):
        los = oracle.traceray(case.ctl, case.atm, geom)
        assert np.allclose(res["tau"][i * 8], np.exp(-np.sum(los["k"] * los["ds"])), rtol=1e-12)


def test_tau_in_unit_interval_and_radiance_positive(oracle):
    case = common.limb_case(nu=common.CTM4_NU, nprofiles=4,
                            geom=synth.limb_geometry(400, seed=3, nprofiles=4))
    res = oracle.formod_rays(case.ctl, case.atm, case.oracle_tables(oracle), case.geom)
    assert np.all((res["tau"] >= 0) & (res["tau"] <= 1))
    assert np.all(res["rad"] > 0)
    # warmer/denser low tangent heights emit more than the highest ones
    order = np.argsort(case.geom[:, 4])
    assert res["rad"][order[:20], 0].mean() > 10 * res["rad"][order[-20:], 0].mean()


def test_nan_mask_round_trip(oracle):
    """Channels whose input radiance is non-finite come back NaN (jr_common.h:193-210)."""
    case = common.limb_case()
    rad_in = np.zeros((len(case.geom), 2))
    rad_in[3, 1] = np.nan
    rad_in[10, 0] = np.inf
    res = oracle.formod_rays(case.ctl, case.atm, case.oracle_tables(oracle), case.geom, rad_in=rad_in)
    assert np.isnan(res["rad"][3, 1]) and np.isnan(res["rad"][10, 0])
    assert np.isfinite(res["rad"]).sum() == res["rad"].size - 2
    assert np.isfinite(res["tau"]).all()


def test_ascii_loader_equals_row_feed_and_keeps_reference_row_rules(oracle, tmp_path):
    """orc_tbl_read_ascii (jurassic.c:329-400) and the in-memory feed give the same
    tables; rows whose u or eps does not grow overwrite the last entry."""
    case = common.limb_case()
    r = case.rows[(0, 0)]
    k = 40
    dup = r[k].copy()
    dup[3] *= 0.5                              # smaller eps: must not extend the curve
    case.rows[(0, 0)] = np.vstack([r[:k + 1], dup, r[k + 1:]])
    case.write_files(str(tmp_path))
    t_file = oracle.Tables(case.ctl.ng, case.ctl.nd)
    assert t_file.read_ascii(case.ctl) == 0
    assert t_file.planck_filt(case.ctl) == 0
    a = oracle.formod_rays(case.ctl, case.atm, t_file, case.geom)
    b = oracle.formod_rays(case.ctl, case.atm, case.oracle_tables(oracle), case.geom)
    # the ASCII round trip goes through %.9g text, tables are fp32: identical after rounding
    assert np.allclose(a["rad"], b["rad"], rtol=1e-7) and np.allclose(a["tau"], b["tau"], rtol=1e-7)


def test_hydrostatic_rebuilds_pressure(oracle):
    """hydrostatic_1d_h2o (jr_common.h:728-761): pressure at the reference level
    stays, the rest follows from T; the climatological profile is already close
    to hydrostatic, so the change is small but not zero."""
    case = common.limb_case(hydz=10.0)
    n = case.atm.np
    p0 = np.ctypeslib.as_array(case.atm.p)[:n].copy()
    oracle.lib().orc_hydrostatic(__import__("ctypes").byref(case.ctl), __import__("ctypes").byref(case.atm))
    p1 = np.ctypeslib.as_array(case.atm.p)[:n]
    assert p1[10] == p0[10]
    rel = np.abs(p1 / p0 - 1)
    assert 0 < rel.max() < 0.2


def test_algorithmic_bytes_per_ray_matches_survey_estimate(oracle):
    """SURVEY.md 8d works out ~1.7 MB/ray for the nd=2, ng=5 limb shape."""
    case = common.limb_case()
    ab = oracle.algorithmic_bytes(case.ctl, case.atm, case.oracle_tables(oracle), case.geom)
    assert ab["segments"] == 16708
    assert 1.5e6 < ab["total"] / ab["rays"] < 1.9e6
    assert 0 < ab["trace"] < 0.1 * ab["total"]


def test_oracle_jacobian_is_consistent_with_two_forward_runs(oracle):
    """kernel() column j is (F(x + h e_j) - F(x)) / h (jurassic.c:830-849): rebuild two columns by hand."""
    case = common.limb_case(geom=common.golden_geometry("limb")[::6])
    c = case.ctl
    c.rett_zmin, c.rett_zmax = 10.0, 12.0                 # 3 temperature elements
    c.retq_zmin[1], c.retq_zmax[1] = 5.0, 5.0             # 1 H2O element
    tb = case.oracle_tables(oracle)
    obs = abi.obs_t()
    obs.nr = len(case.geom)
    for k, name in enumerate(("time", "obsz", "obslon", "obslat", "vpz", "vplon", "vplat")):
        np.ctypeslib.as_array(getattr(obs, name))[:obs.nr] = case.geom[:, k]
    K = oracle.kernel(c, case.atm, obs, tb)
    assert K.shape == (obs.nr * 2, 4)
    base = oracle.formod_rays(c, case.atm, tb, case.geom)["rad"].ravel()
    t = np.ctypeslib.as_array(case.atm.t)
    t[11] += 1.0                                          # element 1 of the state vector: T at 11 km, h = 1
    col = (oracle.formod_rays(c, case.atm, tb, case.geom)["rad"].ravel() - base) / 1.0
    t[11] -= 1.0
    assert np.allclose(K[:, 1], col, rtol=1e-12, atol=0)
    q = np.ctypeslib.as_array(case.atm.q)
    h = max(abs(0.01 * q[1, 5]), 1e-15)
    q0 = q[1, 5]
    q[1, 5] = q0 + h
    col = (oracle.formod_rays(c, case.atm, tb, case.geom)["rad"].ravel() - base) / h
    q[1, 5] = q0
    assert np.allclose(K[:, 3], col, rtol=1e-12, atol=0)
    assert np.all(K[:, :3].max(axis=0) > 0)               # warmer air radiates more


def test_oracle_reproduces_its_committed_example_results(oracle):
    """tests/golden/oracle_examples.json pins the oracle itself (not reference-produced: DESIGN.md section 2).
    1e-13 relative leaves room for a libm that rounds exp/pow/tanh differently; point counts are exact."""
    for name, (case, gold) in common.oracle_goldens().items():
        ref = oracle.formod_rays(case.ctl, case.atm, case.oracle_tables(oracle), case.geom)
        assert np.array_equal(ref["np"], gold["np"]), name
        for k in ("rad", "tau"):
            assert np.max(common.rel_err(ref[k], gold[k])) < 1e-13, (name, k)
        assert np.max(np.abs(ref["tp"] - gold["tp"])) < 1e-11, name
