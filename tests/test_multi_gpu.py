"""GPU tests of the round-4 entry points: several models in one process (jur_formod_host_multi,
jur_formod_device_multi -- on a one-GPU box the same device is listed 1, 2 and 3 times, as the lanes of the drop-in
entry rehearse concurrency on one device), the per-model arithmetic switch, and the channel-group look-up kernel.

Everything here is an ARRANGEMENT of the same arithmetic: the assertions are bit for bit against the single-model,
one-pair-per-workgroup results, which tests/test_parity_gpu.py holds against the oracle."""
import os
import subprocess
import numpy as np
import pytest
import common
from jurassic_hip import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from jurassic_hip import lib
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    assert os.path.exists(lib.SO), "libjurassic_hip.so missing: the HIP path must be built"
    return lib


def same_bits(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return a.shape == b.shape and np.array_equal(np.nan_to_num(a, nan=-1.0).view(np.uint8), np.nan_to_num(b, nan=-1.0).view(np.uint8))


def scan_case(nr, **kw):
    """A tangent-height scan in its natural order (3 .. 68 km ascending: 393 .. 122 LOS points per ray), with a nadir
    sweep and two rays that never enter the atmosphere appended: what contiguous equal-count shares serve badly."""
    g = synth.limb_geometry(nr, scan=True, nprofiles=kw.get("nprofiles", 1))
    extra = np.array([[0, 780.0, 0, 0, 95.0, 0, 20.0], [0, 780.0, 0, 0, 99.0, 0, 10.0]])
    geom = np.vstack([g, synth.nadir_geometry(200, seed=3, nprofiles=kw.get("nprofiles", 1)), extra])
    return common.limb_case(geom=geom, nu=common.CTM4_NU, **kw)


@pytest.mark.parametrize("nmodel", [1, 2, 3])
def test_host_multi_equals_single_model(hip, nmodel):
    """jur_formod_host_multi with the same device listed nmodel times: every output equals the single-model call bit
    for bit (masked input radiances included), and the shares carry equal LOS points within 5 % although the scan is
    sorted -- equal ray counts would differ by a factor of two between the first and the last share."""
    case = scan_case(30000, nprofiles=2)
    rad_in = np.zeros((len(case.geom), case.ctl.nd))
    rad_in[7, 1] = np.nan
    rad_in[29000, 0] = np.inf
    single = hip.Model(case.ctl, case.lib_tables())
    single.set_atm(case.atm)
    ref = single.formod_host(case.geom, rad_in=rad_in)
    models = [hip.Model(case.ctl, case.lib_tables()) for _ in range(nmodel)]
    hip.models_set_atm(models, case.atm)
    out = hip.formod_host_multi(models, case.geom, rad_in=rad_in)
    for k in ("rad", "tau", "tp", "np"):
        assert same_bits(out[k], ref[k]), k
    assert np.isnan(out["rad"][7, 1]) and np.isnan(out["rad"][29000, 0])
    bounds = hip.multi_balance(models[0], case.geom, nmodel)
    assert bounds[0] == 0 and bounds[-1] == len(case.geom) and all(b1 >= b0 for b0, b1 in zip(bounds, bounds[1:]))
    if nmodel > 1:
        pts = [int(ref["np"][lo:hi].sum()) for lo, hi in zip(bounds, bounds[1:])]
        eq = [int(ref["np"][len(case.geom) * k // nmodel:len(case.geom) * (k + 1) // nmodel].sum()) for k in range(nmodel)]
        print("LOS points per share: balanced", pts, "equal ray counts", eq)
        assert max(pts) <= 1.05 * min(pts), pts
        assert max(eq) > 1.3 * min(eq), eq               # (the scan really is the unbalanced case)
    for m in models + [single]:
        m.close()


def test_host_multi_small_and_ragged_calls(hip):
    """Fewer rays than models, one ray, package-sized calls (the shares then take the fused kernel): still the
    single-model bits; a model listed twice is refused."""
    case = common.limb_case(geom=synth.limb_geometry(2500, seed=5, nprofiles=3), nprofiles=3)
    models = [hip.Model(case.ctl, case.lib_tables()) for _ in range(3)]
    hip.models_set_atm(models, case.atm)
    ref = models[0].formod_host(case.geom)
    for n in (1, 2, 3, 64, 1088, 2500):
        out = hip.formod_host_multi(models, case.geom[:n])
        for k in ("rad", "tau", "tp", "np"):
            assert same_bits(out[k], ref[k][:n]), (k, n)
    with pytest.raises(hip.JurassicError):
        hip.formod_host_multi([models[0], models[0]], case.geom)
    for m in models:
        m.close()


@pytest.mark.parametrize("nmodel", [1, 2, 3])
def test_device_multi_collects_on_the_first_models_gpu(hip, nmodel):
    """jur_formod_device_multi on torch-owned HBM buffers: shares 1 .. travel with hipMemcpyPeerAsync on their models'
    streams and come back into the caller's arrays; the caller's stream waits for them.  Balanced bounds and the default
    equal split, bit for bit against jur_formod_host."""
    import torch
    case = scan_case(20000)
    models = [hip.Model(case.ctl, case.lib_tables()) for _ in range(nmodel)]
    hip.models_set_atm(models, case.atm)
    host = models[0].formod_host(case.geom)
    dev = torch.device("cuda", 0)
    nr, nd = len(case.geom), case.ctl.nd
    d_geom = torch.from_numpy(np.ascontiguousarray(case.geom.T)).to(dev)
    for bounds in (None, hip.multi_balance(models[0], case.geom, nmodel)):
        d_rad = torch.zeros((nr, nd), dtype=torch.float64, device=dev)
        d_tau = torch.full((nr, nd), -1.0, dtype=torch.float64, device=dev)
        d_tp = torch.full((3, nr), -1.0, dtype=torch.float64, device=dev)
        d_np = torch.full((nr,), -1, dtype=torch.int32, device=dev)
        d_st = torch.zeros(nmodel, dtype=torch.int32, device=dev)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            hip.formod_device_multi(models, nr, d_geom.data_ptr(), d_rad.data_ptr(), d_tau.data_ptr(), d_tp.data_ptr(),
                                    d_np.data_ptr(), d_st.data_ptr(), s.cuda_stream, bounds=bounds)
        s.synchronize()                                  # the caller's stream alone: it has waited for every share
        assert int(d_st.abs().sum().item()) == 0
        assert same_bits(d_rad.cpu().numpy(), host["rad"]) and same_bits(d_tau.cpu().numpy(), host["tau"])
        assert same_bits(d_tp.cpu().numpy().T, host["tp"]) and np.array_equal(d_np.cpu().numpy(), host["np"])
    for m in models:
        m.close()


def test_multi_device_c_program(hip, tmp_path):
    """tools/multi_device.c: a C caller (no Python in the process) of jur_formod_host_multi with the device listed three
    times; it exits non-zero if any value differs from its own single-model call."""
    case = common.limb_case()
    case.write_files(str(tmp_path), base="boxcar")
    import shutil
    shutil.copy(os.path.join(common.GOLD, "limb", "atm.tab"), tmp_path / "atm.tab")
    exe = str(tmp_path / "multi_device")
    subprocess.check_call(["gcc", "-O2", "-I" + os.path.join(common.ROOT, "include"), os.path.join(common.ROOT, "tools", "multi_device.c"),
                           "-o", exe, "-L" + os.path.dirname(hip.SO), "-ljurassic_hip", "-Wl,-rpath," + os.path.dirname(hip.SO), "-lm"])
    out = subprocess.run([exe, "20000", "0", "0", "0"], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    import json
    doc = json.loads(out.stdout.strip().splitlines()[-1])
    print(doc)
    assert doc["differing_values"] == 0 and doc["models"] == 3 and sum(doc["share_rays"]) == 20000
    assert max(doc["share_los_points"]) <= 1.05 * min(doc["share_los_points"])


def test_arithmetic_is_a_per_model_choice(hip):
    """jur_model_set_arithmetic: EXACT equals the process-wide JUR_EGA_NO_RCP build of the look-up (the reference's
    divisions), FAST the default; both in the fused and in the batched arrangement, which stay bit-identical to each
    other in either mode; the two modes agree to 1e-12 on radiances; switching back returns the first bits."""
    case = common.limb_case(geom=synth.limb_geometry(3000, seed=4, nprofiles=8), nu=common.CTM4_NU, nprofiles=8)
    m = hip.Model(case.ctl, case.lib_tables())
    m.set_atm(case.atm)
    res = {}
    for mode in (hip.ARITH_FAST, hip.ARITH_EXACT, hip.ARITH_FAST):
        m.set_arithmetic(mode)
        m.set_pencil(0)
        batched = m.formod_host(case.geom)
        m.set_pencil(10000, 4)
        fused = m.formod_host(case.geom)
        for k in ("rad", "tau", "tp", "np"):
            assert same_bits(batched[k], fused[k]), (k, mode)
        if mode in res:
            assert same_bits(res[mode]["rad"], batched["rad"]) and same_bits(res[mode]["tau"], batched["tau"])
        res[mode] = batched
    fast, exact = res[hip.ARITH_FAST], res[hip.ARITH_EXACT]
    assert not same_bits(fast["rad"], exact["rad"])      # (they ARE different arithmetics)
    dev = common.rel_err(fast["rad"], exact["rad"]).max()
    print("FAST against EXACT: worst relative radiance deviation %.2e" % dev)
    assert dev < 1e-12 and np.abs(fast["tau"] - exact["tau"]).max() < 1e-13 and np.array_equal(fast["np"], exact["np"])
    with pytest.raises(hip.JurassicError):
        m.set_arithmetic(7)
    m.close()


@pytest.mark.parametrize("nch", [2, 3, 4])
def test_channel_group_lookup_kernel_equals_one_pair_per_workgroup(hip, nch):
    """jur_ega_group_kernel (one lane per (ray, gas) walking up to nch channels that share a (p, T) grid; round-4
    experiment, off by default) against jur_ega_kernel: the same operations on the same operands, so BIT FOR BIT -- full
    and ragged groups (5 channels in groups of 2, 3, 4), a gas without a table for one channel, a channel whose table
    stands on ANOTHER grid (it gets an item of its own), rays that miss the atmosphere, several launches per call."""
    nu = list(np.round(np.linspace(700.0, 2400.0, 5), 4))
    def shapes(g, d):
        return dict(nlev=21, ntemp=6) if (g, d) == (1, 3) else {}      # one pair on a coarser grid
    g = synth.limb_geometry(3000, seed=nch, nprofiles=3)
    extra = np.array([[0, 780.0, 0, 0, 95.0, 0, 20.0], [1, 30.0, 0, 0, 5.0, 0, 3.0]])
    geom = np.vstack([g[:1700], extra, g[1700:], synth.nadir_geometry(90, seed=2, nprofiles=3)])
    case = common.Case(["CO2", "H2O", "O3"], nu, os.path.join(common.GOLD, "limb", "atm.tab"), geom, nprofiles=3,
                       table_kw=shapes, missing={(2, 1)})
    m = hip.Model(case.ctl, case.lib_tables())
    m.set_atm(case.atm)
    m.set_pencil(0)
    ref = m.formod_host(case.geom)
    assert m.set_ega_group(nch) == nch               # the group kernel is what the next call runs
    out = m.formod_host(case.geom)
    m.set_chunk_rays(448)
    chunked = m.formod_host(case.geom)
    assert m.set_ega_group(0) == 0
    back = m.formod_host(case.geom)
    for k in ("rad", "tau", "tp", "np"):
        assert same_bits(out[k], ref[k]) and same_bits(chunked[k], ref[k]) and same_bits(back[k], ref[k]), (k, nch)
    m.close()


def test_workspace_laid_out_by_path_lengths(hip):
    """Calls that need several integration launches pack the transmittance tiles by the longest path of each tile
    instead of JUR_NLOS = 400 points per ray: nadir rays (182 points) then take less than half the launches, a limb scan
    (122 .. 393) fewer -- and every output equals the one-launch result and the JUR_NLOS-strided chunking bit for bit,
    for sorted and unsorted rays, a mix with rays that never enter the atmosphere, and a budget-driven layout."""
    g = synth.limb_geometry(6000, seed=3, nprofiles=2)
    geom = np.vstack([g[:2500], np.array([[0, 780.0, 0, 0, 95.0, 0, 20.0]] * 70), g[2500:], synth.nadir_geometry(3000, seed=2, nprofiles=2)])
    case = common.limb_case(geom=geom, nu=common.CTM4_NU, nprofiles=2)
    m = hip.Model(case.ctl, case.lib_tables())
    m.set_atm(case.atm)
    m.set_pencil(0)
    ref = m.formod_host(case.geom)
    assert m.last_launches() == 1
    res = {}
    for compact in (1, 0):
        m.set_compact_workspace(compact)
        for sort in (1, 0):
            m.set_sort_rays(sort)
            m.set_chunk_rays(1024)
            out = m.formod_host(case.geom)
            res[(compact, sort)] = m.last_launches()
            for k in ("rad", "tau", "tp", "np"):
                assert same_bits(out[k], ref[k]), (k, compact, sort)
    total_points, nr = int(ref["np"].sum()), len(geom)
    print("launches", res, "LOS points", total_points, "rays", nr)
    assert res[(0, 1)] == -(-nr // 1024)                                   # JUR_NLOS points per ray: rays / 1024 launches
    assert res[(1, 1)] <= -(-total_points // (1024 // 64 * 400 * 64 - 64 * 400)) + 1 < res[(0, 1)]   # packed by the points that occur
    # the same through the workspace budget (what a many-channel set runs into): 96 KB per ray at 400 points
    m.set_compact_workspace(1)
    m.set_sort_rays(1)
    m.set_chunk_rays(1 << 21)
    m.set_workspace_budget(200 << 20)
    out = m.formod_host(case.geom)
    assert m.last_launches() > 1
    for k in ("rad", "tau", "tp", "np"):
        assert same_bits(out[k], ref[k]), k
    m.close()


def test_tracer_with_four_lanes_per_ray(hip):
    """The batched ray tracer with a quad of lanes per ray (what launches of up to 65 536 rays with several emitters take):
    the refraction probes of a step side by side, each as the sequential loop sees it -- every output bit for bit equal
    to one lane per ray: limb and nadir rays, refraction off, rays outside / observer inside, NLOS overflow reported."""
    g = synth.limb_geometry(2200, seed=9, nprofiles=3)
    extra = np.array([[0, 780.0, 0, 0, 95.0, 0, 20.0], [1, 30.0, 0, 0, 5.0, 0, 3.0], [2, 780.0, 0, 0, -0.005, 0, 27.0]])
    geom = np.vstack([g[:900], extra, g[900:], synth.nadir_geometry(333, seed=2, nprofiles=3)])
    try:
        for kw in (dict(), dict(refrac=0), dict(raydz=0.2)):
            case = common.limb_case(geom=geom, nprofiles=3, **kw)
            m = hip.Model(case.ctl, case.lib_tables())
            m.set_atm(case.atm)
            m.set_pencil(0)
            res = {}
            for lanes in (1, 4):
                hip.tune_trace(lanes)
                try:
                    res[lanes] = m.formod_host(case.geom)
                except hip.JurassicError as e:            # raydz = 0.2: more than NLOS points, whatever the lanes
                    res[lanes] = str(e)
            if isinstance(res[1], str):
                assert "LOS" in res[1] and res[4] == res[1]
            else:
                for lanes in (4,):
                    for k in ("rad", "tau", "tp", "np"):
                        assert same_bits(res[lanes][k], res[1][k]), (k, lanes, kw)
            m.close()
    finally:
        hip.tune_trace(0)


def test_batched_call_in_a_graph(hip):
    """The batched kernels (ray sort, quad-lane tracer, look-up, radiance update -- no fused kernel) captured into a HIP
    graph after jur_model_reserve and replayed on new inputs: a call that fits the workspace never waits for the host."""
    import torch
    case = common.limb_case(geom=synth.limb_geometry(5000, seed=12, nprofiles=2), nprofiles=2)
    model = hip.Model(case.ctl, case.lib_tables())
    model.set_atm(case.atm)
    model.set_pencil(0)
    dev = torch.device("cuda", 0)
    nr, nd = len(case.geom), case.ctl.nd
    d_geom = torch.from_numpy(np.ascontiguousarray(case.geom.T)).to(dev)
    d_rad = torch.zeros((nr, nd), dtype=torch.float64, device=dev)
    d_tau = torch.zeros((nr, nd), dtype=torch.float64, device=dev)
    d_tp = torch.zeros((3, nr), dtype=torch.float64, device=dev)
    d_st = torch.zeros(1, dtype=torch.int32, device=dev)
    model.reserve(nr)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        d_rad.zero_()
        model.formod_device(nr, d_geom.data_ptr(), d_rad.data_ptr(), d_tau.data_ptr(), d_tp.data_ptr(), 0,
                            d_st.data_ptr(), torch.cuda.current_stream().cuda_stream)
    for seed in (12, 13):
        geom = synth.limb_geometry(nr, seed=seed, nprofiles=2)
        d_geom.copy_(torch.from_numpy(np.ascontiguousarray(geom.T)))
        graph.replay()
        torch.cuda.synchronize()
        ref = model.formod_host(geom)
        assert same_bits(d_rad.cpu().numpy(), ref["rad"]) and same_bits(d_tau.cpu().numpy(), ref["tau"]) and int(d_st.item()) == 0
    model.close()
