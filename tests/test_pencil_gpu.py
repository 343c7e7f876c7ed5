"""The fused kernel for package-sized calls (jur_pencil_kernel: tracer, emissivity-growth and radiance-update
wavefronts of one workgroup handing the line of sight on through LDS) against the batched kernels: the two paths
run the same device functions in the same order, so every output must agree BIT FOR BIT -- for every way the
workgroup can be shaped (rays per workgroup, chains per lane) and for the edge cases of the path.  (Against the
oracle the fused path is checked by all the small cases of test_parity_gpu.py, which take it by default.)
"""
import os
import numpy as np
import pytest
import common
from jurassic_hip import abi, synth

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]


@pytest.fixture(scope="module")
def hip():
    from jurassic_hip import lib
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return lib


def same(a, b):
    for k in ("rad", "tau", "tp", "np"):
        x, y = np.asarray(a[k]), np.asarray(b[k])
        if x.dtype.kind == "f":
            assert np.array_equal(np.isnan(x), np.isnan(y)), k
            m = ~np.isnan(x)
            assert np.array_equal(x[m].view(np.uint64), y[m].view(np.uint64)), (k, int(np.count_nonzero(x[m] != y[m])))
        else:
            assert np.array_equal(x, y), k


def both(hip, case, rbs=(0,), rad_in=None):
    m = hip.Model(case.ctl, case.lib_tables())
    m.set_atm(case.atm)
    m.set_pencil(0)
    batched = m.formod_host(case.geom, rad_in=rad_in)
    for rb in rbs:
        m.set_pencil(1 << 20, rb)
        m.enable_timing(True)
        fused = m.formod_host(case.geom, rad_in=rad_in)
        k = m.kernel_ms()
        assert k["pencil_launches"] == 1 and k["ega_launches"] == 0 and k["trace_launches"] == 0, (rb, k)
        same(fused, batched)
    m.close()
    return batched


def test_limb_package_every_workgroup_shape(hip):
    """BASELINE configs[2]'s shape at the reference's package size: 1088 rays, 4 channels x 5 emitters (20 chains
    per ray), 64 profiles; 1, 2, 3, 8, 16 and 64 rays per workgroup (1 .. 8 ega waves, chains looped over lanes)."""
    geom = synth.limb_geometry(1088, seed=41, nprofiles=64)
    case = common.limb_case(geom=geom, nu=common.CTM4_NU, nprofiles=64)
    out = both(hip, case, rbs=(0, 1, 2, 3, 8, 16, 64))
    assert np.isfinite(out["rad"]).all() and out["np"].min() > 100


def test_nadir_package_with_surface_and_brightness(hip):
    both(hip, common.nadir_case(), rbs=(1, 7))
    both(hip, common.nadir_case(geom=synth.nadir_geometry(1088, seed=3)), rbs=(0, 4))


def test_more_rays_than_the_chip_holds_at_once(hip):
    """20 000 one-ray workgroups: several rounds over the CUs, ragged last workgroup for 3 rays per group."""
    geom = synth.limb_geometry(20_000, seed=42, nprofiles=4)
    both(hip, common.limb_case(geom=geom, nprofiles=4), rbs=(1, 3))


def test_edge_geometries_mask_and_missing_tables(hip):
    g = synth.limb_geometry(8, scan=True, zmin=-20.0, zmax=2.0)          # tangent below ground
    extra = np.array([[0, 30.0, 0, 0, 5.0, 0, 3.0], [0, 10.0, 0, 0, 60.0, 0, 2.0], [0, 20.0, 0, 0, 80.0, 0, 0.0],
                      [0, 780.0, 0, 0, 95.0, 0, 20.0], [0, -1.0, 0, 0, 10.0, 0, 1.0]])
    geom = np.vstack([g, extra, synth.limb_geometry(40, seed=1)])
    rad_in = np.zeros((len(geom), 2))
    rad_in[0, 0] = np.nan
    rad_in[20, 1] = -np.inf
    out = both(hip, common.limb_case(geom=geom, missing={(3, 0), (4, 1), (0, 1)}), rbs=(1, 5, 64), rad_in=rad_in)
    assert list(out["np"][11:13]) == [0, 0] and np.isnan(out["rad"][0, 0]) and np.isnan(out["rad"][20, 1])


@pytest.mark.parametrize("kw", [dict(table_kw=dict(descending=True)),           # unsorted tables: the reference's bisections
                                dict(table_kw=dict(nlev=1)),                    # no pair has a table
                                dict(table_kw=dict(dup_every=5)),
                                dict(refrac=0), dict(hydz=10.0), dict(rayds=20.0, raydz=1.0),
                                dict(ctm_co2=0, ctm_h2o=0, ctm_n2=0, ctm_o2=0, ctm_auto=1)])
def test_control_switches_and_table_shapes(hip, kw):
    both(hip, common.limb_case(geom=synth.limb_geometry(130, seed=2), nu=common.CTM4_NU, **kw), rbs=(1, 6))


def test_many_chains_per_ray(hip):
    """100 channels x 3 emitters = 300 chains per ray: several ega waves per ray, and with 8 rays per workgroup
    2400 chains looped over 512 lanes; channels across all four continuum windows."""
    nu = list(np.round(np.linspace(650.0, 2665.0, 100), 4))
    geom = np.vstack([synth.nadir_geometry(40, seed=3), synth.limb_geometry(40, seed=4)])
    case = common.Case(["CO2", "H2O", "O3"], nu, os.path.join(common.GOLD, "limb", "atm.tab"), geom)
    both(hip, case, rbs=(1, 8))


def test_no_emitters_and_one_channel(hip):
    geom = synth.limb_geometry(70, seed=6)
    case = common.Case([], [792.0], os.path.join(common.GOLD, "limb", "atm.tab"), geom)
    both(hip, case, rbs=(1, 64))
    case = common.Case(["CO2"], [792.0], os.path.join(common.GOLD, "limb", "atm.tab"), geom)
    both(hip, case, rbs=(1, 2))


def test_nlos_overflow_is_reported_by_the_fused_kernel(hip):
    case = common.limb_case(geom=synth.limb_geometry(64, scan=True), raydz=0.2)
    m = hip.Model(case.ctl, case.lib_tables())
    m.set_atm(case.atm)
    m.set_pencil(1 << 20, 1)
    with pytest.raises(hip.JurassicError, match="Too many LOS points"):
        m.formod_host(case.geom)
    m.close()


def test_configurations_beyond_the_lds_rings_fall_back(hip, tmp_path):
    """100 channels x 30 emitters: 3000 chains per ray do not fit the rings; the call runs batched."""
    nu = list(np.round(np.linspace(700.0, 900.0, 100), 4))
    em = ["G%02d" % i for i in range(30)]
    rows = np.loadtxt(os.path.join(common.GOLD, "limb", "atm.tab"), comments="#")      # time z lon lat p T q[5] k
    wide = np.hstack([rows[:, :6], np.tile(rows[:, 6:11], (1, 6)), rows[:, 11:12]])
    np.savetxt(tmp_path / "atm30.tab", wide, fmt="%.10g")
    case = common.Case(em, nu, str(tmp_path / "atm30.tab"), synth.limb_geometry(16, seed=8),
                       table_kw=dict(nlev=3, ntemp=2), missing={(g, d) for g in range(30) for d in range(100) if (g + d) % 9})
    m = hip.Model(case.ctl, case.lib_tables())
    m.set_atm(case.atm)
    m.set_pencil(1 << 20, 1)
    m.enable_timing(True)
    out = m.formod_host(case.geom)
    k = m.kernel_ms()
    assert k["pencil_launches"] == 0 and k["ega_launches"] >= 1 and np.isfinite(out["rad"]).all()
    m.close()
