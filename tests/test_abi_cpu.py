"""CPU tests of the C-ABI library: it loads, exports what include/*.h declares,
agrees on the struct layout, its host-side table logic works, and compute
entry points fail loudly without a GPU."""
import ctypes as C
import os
import re
import numpy as np
import pytest
import common
from jurassic_hip import abi, lib, synth

ROOT = common.ROOT


def test_library_builds_and_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    L = lib.lib()
    hdr = open(os.path.join(ROOT, "include", "jurassic_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(jur_\w+|formod\w*)\s*\(", hdr))
    assert {"formod", "formod_GPU", "formod_pencil", "jur_formod_device", "jur_formod_host"} <= names
    for n in sorted(names):
        assert hasattr(L, n), f"{n} declared in jurassic_hip.h but not exported"


def test_struct_layout_matches_reference_offsets():
    """sizes/offsets measured on the reference headers."""
    out = (C.c_size_t * 5)()
    lib.lib().jur_abi_sizes(out)
    assert list(out) == [321856, 2841608, 1827848, 100, 30]
    assert abi.ctl_t.nd.offset == 150004 and abi.ctl_t.nu.offset == 150016
    assert abi.ctl_t.tblbase.offset == 151216 and abi.ctl_t.refrac.offset == 156264
    assert abi.ctl_t.write_bbt.offset == 161816 and abi.ctl_t.useGPU.offset == 321828
    assert abi.atm_t.q.offset == 460800 and abi.atm_t.np.offset == 2841600
    assert abi.obs_t.tau.offset == 87040 and abi.obs_t.rad.offset == 957440 and abi.obs_t.nr.offset == 1827840


def test_physical_constants_of_product_and_oracle_equal_the_reference_literals():
    """The product (include/jurassic_abi.h) and the oracle (oracle/oracle_constants.h) carry separate copies of
    the constants on the path, so that a wrong value on one side is a parity failure instead of cancelling.
    Both are held here against literals this test carries itself: C1, C2, P0, RE as reference jurassic.h:109-126
    spells them; N_A, k_B, R as GSL 2.5 (gsl_const_num.h / gsl_const_mksa.h) publishes them -- the reference
    reads them at jr_common.h:330, 450, 744."""
    want = [1.19104259e-8, 1.43877506, 1013.25, 6367.421, 6.02214199e23, 1.3806504e-23, 8.314472]
    got = (C.c_double * 7)()
    lib.lib().jur_abi_constants(got)
    assert list(got) == want
    from oracle import orc
    L = orc.lib()
    L.orc_constants.argtypes = [C.POINTER(C.c_double)]
    L.orc_constants(got)
    assert list(got) == want
    # and as text: the oracle does not see the product's constants (it #undefs them), the product never includes the oracle's
    osrc = open(os.path.join(ROOT, "oracle", "jurassic_oracle.c")).read()
    assert not re.search(r"\bJUR_(C1|C2|P0|RE|AVOGADRO|BOLTZMANN|MOLAR_GAS)\b(?!\n)", re.sub(r"#undef JUR_\w+", "", osrc))
    for root_, _, files in os.walk(os.path.join(ROOT, "jurassic-gpu_amd", "csrc")):
        for f in files:
            if f.endswith((".c", ".h", ".hip")):
                assert "oracle_constants.h" not in open(os.path.join(root_, f)).read()
    # k_B (CODATA 2006, what GSL 2.5 ships) against CODATA 2018: 1.0e-6 apart -- which of the two a build uses is
    # visible at the level of the 1e-6 contract (through the column densities), so this literal is not a detail
    assert 0.9e-6 < abs(want[5] / 1.380649e-23 - 1) < 1.1e-6


def test_table_builder_counts_and_row_rules():
    rows = synth.table_rows("CO2", 792.0)
    tb = lib.Tables(1, 1)
    tb.feed_rows(0, 0, rows)
    assert tb.entries() == len(rows)
    # a row that does not increase eps overwrites instead of extending
    k = 50
    bad = rows[k].copy()
    bad[3] *= 0.9
    tb.feed_rows(0, 0, np.vstack([rows[:k + 1], bad, rows[k + 1:]]))
    assert tb.entries() == len(rows)
    # more than TBLNU entries per curve are dropped
    one = np.array([[100.0, 250.0, 1e15 * 1.01 ** i, 1e-6 * 1.01 ** i] for i in range(400)])
    tb.feed_rows(0, 0, one)
    assert tb.entries() == abi.TBLNU


def test_table_builder_rejects_too_many_levels():
    rows = np.array([[float(p), 250.0 + (p % 2), 1e18, 0.1] for p in range(1, 60)])
    tb = lib.Tables(1, 1)
    with pytest.raises(lib.JurassicError, match="pressure levels"):
        tb.feed_rows(0, 0, rows)
    with pytest.raises(lib.JurassicError):
        tb.feed_rows(3, 0, rows[:2])


def test_ascii_reader_counts_files(tmp_path):
    case = common.limb_case(missing={(3, 0), (4, 1)})
    case.write_files(str(tmp_path))
    tb = lib.Tables(case.ctl.ng, case.ctl.nd)
    assert tb.read_ascii(case.ctl) == 8
    tb.read_filters(case.ctl)
    assert tb.entries() == sum(len(r) for r in case.rows.values())


def test_table_cache_round_trip_and_rejection(tmp_path):
    """parsed tables + source functions -> compact binary cache -> identical tables;
    a cache written for other channels, or a damaged one, is refused."""
    case = common.limb_case(missing={(3, 0)})
    tb = case.lib_tables()
    path = str(tmp_path / "cache.bin")
    tb.save(case.ctl, path)
    assert os.path.getsize(path) < 9 * tb.entries() + 64 * 1024 + 2 * 1201 * 8
    back = lib.Tables.load(case.ctl, path)
    assert back.entries() == tb.entries() and back.checksum() == tb.checksum() != 0
    other = abi.make_ctl(common.LIMB_EMITTERS, [792.0, 833.0])
    with pytest.raises(lib.JurassicError, match="does not match"):
        lib.Tables.load(other, path)
    with pytest.raises(lib.JurassicError, match="no table cache"):
        lib.Tables.load(case.ctl, path + ".missing")
    blob = bytearray(open(path, "rb").read())
    blob[len(blob) // 2] ^= 0x40
    open(path, "wb").write(bytes(blob))
    with pytest.raises(lib.JurassicError, match="does not match"):
        lib.Tables.load(case.ctl, path)


def test_formod_executable_parses_control_file_and_overrides(tmp_path):
    """scan_ctl semantics (jurassic.c:1153-1201): KEY = VALUE lines, KEY[i] / KEY[*], command-line
    overrides, defaults, the automatic continuum switch-off; CHECKMODE keeps the GPU out of it."""
    import subprocess
    exe = os.path.join(ROOT, "jurassic-gpu_amd", "formod")
    assert os.path.exists(exe)
    (tmp_path / "t.ctl").write_text("NG = 1\nEMITTER[0] = CO2\nND = 2\nNU[*] = 700\nTBLBASE = ./x\n")
    out = subprocess.run([exe, "t.ctl", os.path.join(common.GOLD, "nadir", "obs.tab"),
                          os.path.join(common.GOLD, "nadir", "atm.tab"), "rad.tab", "CHECKMODE", "1", "RAYDS", "5"],
                         cwd=tmp_path, capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stdout + out.stderr
    for line in ("NG = 1", "ND = 2", "RAYDS = 5", "RAYDZ = 0.5", "REFRAC = 1", "CTM_CO2 = 1", "READ_BINARY = -1",
                 "No frequency in N2 range, automatically set CTM_N2 = 0", "CHECKMODE = 1 (skip)"):
        assert line in out.stdout, line
    assert not (tmp_path / "rad.tab").exists()
    bad = subprocess.run([exe, "t.ctl", "a", "b", "c", "NG", "2"], cwd=tmp_path, capture_output=True, text=True)
    assert bad.returncode != 0 and "Missing variable EMITTER[1]" in bad.stdout


LIMB_CTL = "TBLBASE = ./boxcar\nNG = 5\nEMITTER[0] = CO2\nEMITTER[1] = H2O\nEMITTER[2] = O3\nEMITTER[3] = F11\n" \
           "EMITTER[4] = CCl4\nND = 2\nNU[0] = 792.0000\nNU[1] = 832.0000\n"
NADIR_CTL = "TBLBASE = ./airs\nNG = 1\nEMITTER[0] = CO2\nND = 3\nNU[0] = 667.7820\nNU[1] = 668.5410\n" \
            "NU[2] = 669.8110\nWRITE_BBT = 1\n"


@pytest.mark.parametrize("name,ctl,obs_tool,obs_args", [
    ("limb", LIMB_CTL, "limb", ["Z0", "3", "Z1", "68", "DZ", "1.0"]),        # example/limb/run.sh:8-11
    ("nadir", NADIR_CTL, "nadir", ["T1", "10"]),                              # example/nadir/run.sh:8-11
])
def test_generators_reproduce_the_shipped_example_inputs(tmp_path, name, ctl, obs_tool, obs_args):
    """`climatology` and `limb`/`nadir` called as the reference's run.sh calls them write files that are
    byte-identical to the atm.tab / obs.tab the reference ships (tests/golden): pins the climatology
    numbers, the profile interpolation, the scan geometry and the %g text format."""
    import subprocess
    bindir = os.path.join(ROOT, "jurassic-gpu_amd")
    (tmp_path / "x.ctl").write_text(ctl)
    for exe, target, extra in (("climatology", "atm.tab", []), (obs_tool, "obs.tab", obs_args)):
        out = subprocess.run([os.path.join(bindir, exe), "x.ctl", target] + extra, cwd=tmp_path,
                             capture_output=True, text=True, timeout=60)
        assert out.returncode == 0, out.stdout + out.stderr
        got = (tmp_path / target).read_bytes()
        want = open(os.path.join(common.GOLD, name, target), "rb").read()
        assert got == want, f"{exe}: {target} differs from the shipped file"


def test_climatology_random_profiles_and_checkmode(tmp_path):
    """RAND=1 perturbs each profile by one (dp, dT) pair within +-5 % / +-30 K (climatology.c:65-78);
    CHECKMODE writes nothing."""
    import subprocess
    exe = os.path.join(ROOT, "jurassic-gpu_amd", "climatology")
    (tmp_path / "x.ctl").write_text(LIMB_CTL)
    run = lambda *a: subprocess.run([exe, "x.ctl", *a], cwd=tmp_path, capture_output=True, text=True, timeout=60)
    assert run("base.tab", "T1", "3").returncode == 0
    assert run("rand.tab", "T1", "3", "RAND", "1").returncode == 0
    from jurassic_hip import textio
    ctl = abi.make_ctl(["CO2", "H2O", "O3", "F11", "CCl4"], [792.0, 832.0])
    a, b = textio.read_atm(str(tmp_path / "base.tab"), ctl), textio.read_atm(str(tmp_path / "rand.tab"), ctl)
    assert a.np == b.np == 4 * 91
    pa, pb = np.ctypeslib.as_array(a.p)[:a.np], np.ctypeslib.as_array(b.p)[:a.np]
    ta, tb = np.ctypeslib.as_array(a.t)[:a.np], np.ctypeslib.as_array(b.t)[:a.np]
    dp, dt = (pb / pa - 1).reshape(4, 91), (tb - ta).reshape(4, 91)
    assert np.all(np.abs(dp) <= 0.05 + 1e-5) and np.all(np.abs(dt) <= 30 + 1e-3)
    assert np.all(np.ptp(dp, axis=1) < 2e-5) and np.all(np.ptp(dt, axis=1) < 2e-3)     # one draw per profile (%g text)
    assert len(set(np.round(dt[:, 0], 3))) == 4                                             # profiles differ
    assert run("none.tab", "CHECKMODE", "1").returncode == 0 and not (tmp_path / "none.tab").exists()


def _fov_obs(nd=3, scans=3, per_scan=30, descending=False, seed=0):
    """obs_t with `scans` limb scans (one time stamp each) and smooth synthetic radiance / transmittance profiles."""
    rng = np.random.default_rng(seed)
    obs = abi.obs_t()
    obs.nr = scans * per_scan
    z = np.linspace(5.0, 63.0, per_scan)
    if descending:
        z = z[::-1]
    for s_ in range(scans):
        for i in range(per_scan):
            ir = s_ * per_scan + i
            obs.time[ir] = float(s_)
            obs.vpz[ir] = z[i] + 0.01 * s_
            for d in range(nd):
                obs.rad[ir][d] = np.exp(-z[i] / (7.0 + d)) * (1 + 0.1 * rng.random())
                obs.tau[ir][d] = 1 - 0.9 * np.exp(-z[i] / (9.0 + d))
    return obs


@pytest.mark.parametrize("descending", [False, True])
def test_fov_convolution_matches_the_restatement(oracle, tmp_path, descending):
    """formod_fov (jurassic.c:214-258): library and oracle restatement agree bit for bit; flat-array entry
    equals the obs_t entry; FOV = '-' is a no-op; a linear profile is a fixed point of a symmetric FOV."""
    import copy
    nd = 3
    ctl = abi.make_ctl(["CO2"], [700.0, 800.0, 900.0])
    dz = np.linspace(-1.5, 1.5, 21)
    w = np.exp(-0.5 * (dz / 0.6) ** 2)
    shape = tmp_path / "fov.shape"
    shape.write_text("# dz [km]  weight\n" + "".join(f"{a:.6f} {b:.8g}\n" for a, b in zip(dz, w)))
    rdz, rw = lib.fov_read_shape(str(shape))
    assert len(rdz) == 21
    a, b = _fov_obs(nd, descending=descending), _fov_obs(nd, descending=descending)
    rad_in = np.ctypeslib.as_array(a.rad)[:a.nr, :nd].copy()
    tau_in = np.ctypeslib.as_array(a.tau)[:a.nr, :nd].copy()
    lib.formod_fov(ctl, a)                                     # ctl.fov == "-": nothing happens
    assert np.array_equal(np.ctypeslib.as_array(a.rad)[:a.nr, :nd], rad_in)
    ctl.fov = str(shape).encode()
    lib.formod_fov(ctl, a)
    assert oracle.formod_fov(ctl, b, rdz, rw) == 0
    for name in ("rad", "tau"):
        x, y = np.ctypeslib.as_array(getattr(a, name))[:a.nr, :nd], np.ctypeslib.as_array(getattr(b, name))[:a.nr, :nd]
        assert np.array_equal(x.view(np.uint64), y.view(np.uint64)), name
    assert not np.array_equal(np.ctypeslib.as_array(a.rad)[:a.nr, :nd], rad_in)
    time = np.ctypeslib.as_array(a.time)[:a.nr].copy()
    vpz = np.ctypeslib.as_array(a.vpz)[:a.nr].copy()
    r2, t2 = rad_in.copy(), tau_in.copy()
    lib.fov_apply(time, vpz, r2, t2, rdz, rw)
    assert np.array_equal(r2, np.ctypeslib.as_array(a.rad)[:a.nr, :nd])
    lin = np.repeat((2.0 + 0.3 * vpz)[:, None], nd, axis=1).copy()        # linear in z: unchanged away from scan ends
    lin2, dummy = lin.copy(), lin.copy()
    lib.fov_apply(time, vpz, lin2, dummy, rdz, rw)
    inner = np.concatenate([np.arange(s_ * 30 + 2, s_ * 30 + 28) for s_ in range(3)])
    assert np.allclose(lin2[inner], lin[inner], rtol=1e-12)
    lone = _fov_obs(nd, scans=1, per_scan=1)                                # a ray alone in its scan: upstream aborts
    assert oracle.formod_fov(ctl, lone, rdz, rw) == -1
    one_r, one_t = np.ones((1, nd)), np.ones((1, nd))
    with pytest.raises(lib.JurassicError, match="Cannot apply FOV"):
        lib.fov_apply(np.zeros(1), np.array([10.0]), one_r, one_t, rdz, rw)


def test_table_number_reader_equals_strtod():
    """jur_parse_number (the reader of the .tab files) returns what strtod / sscanf("%lg") return, bit for bit:
    the fast path for short decimals and the strtod fallback for everything else."""
    L = lib.lib()
    L.jur_parse_number.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_double)]
    L.jur_parse_number.restype = C.c_int
    rng = np.random.default_rng(5)
    vals = np.concatenate([10.0 ** rng.uniform(-30, 30, 20000) * rng.choice([-1, 1], 20000),
                           rng.uniform(0, 1, 5000), rng.integers(0, 10 ** 15, 5000).astype(float),
                           [0.0, 1.0, 1e22, 1e23, 1e-22, 1e-23, 5e-324, 1.7976931348623157e308, 123456789012345.0,
                            1234567890123456.0, 0.1, 1013.25, 6.02214199e23]])
    texts = []
    for v in vals:
        for fmt in ("%g", "%.9g", "%.15g", "%.17g", "%e", "%.3f", "%.12e"):
            texts.append(fmt % v)
    texts += ["  42", "\t-7.5e3  ", "+3.", ".5", "1e5x", "0x1p3", "inf", "-INF", "nan", "1e400", "1e-400", "12abc", "1e", "1e+"]
    for t in texts:
        buf = C.c_char_p(t.encode())
        out = C.c_double(0)
        ok = L.jur_parse_number(C.byref(buf), C.byref(out))
        try:                                                   # Python's float() is strtod minus partial tokens
            want = float(t)
        except ValueError:
            want = None
        if want is not None:
            assert ok == 1, t
            a, b = np.float64(out.value), np.float64(want)
            assert (np.isnan(a) and np.isnan(b)) or a.view(np.uint64) == b.view(np.uint64), (t, out.value, want)
    for t, lead in (("1e5x", 1e5), ("12abc", 12.0), ("1e", 1.0), ("1e+", 1.0)):   # strtod stops where the number ends
        buf = C.c_char_p(t.encode())
        out = C.c_double(0)
        assert L.jur_parse_number(C.byref(buf), C.byref(out)) == 1 and out.value == lead, t
    for t in ("", "   ", "abc", "-", "."):
        buf = C.c_char_p(t.encode())
        assert L.jur_parse_number(C.byref(buf), C.byref(C.c_double(0))) == 0, t


def test_compute_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    case = common.limb_case()
    with pytest.raises(lib.JurassicError, match="no HIP device"):
        lib.Model(case.ctl, case.lib_tables())


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "jurassic-gpu_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".c", ".h", ".hip", "Makefile")):
                txt = open(os.path.join(d, f), errors="ignore").read()
                assert "oracle" not in txt.lower(), f"{f} mentions the oracle"


def test_kernel_register_budgets(tmp_path):
    """The batched kernels' occupancy is part of their measured speed and hangs on a few registers (DESIGN 4.3, 8 vii):
    jur_ega_kernel (strict tables; with the shorter fetch chain of the bracket records the eighth wavefront is worth 4 %)
    must stay within 64 VGPRs (8 wavefronts per SIMD), the radiance-update kernels within 64 (8: a
    workgroup of the grouped one is 8 wavefronts, so a 65th register costs a whole workgroup per CU), the tracer within
    128 (4); jur_ega_kernel and the radiance update use no scratch memory at all, the tracer no more than the few
    doubles it keeps there today (outside its inner loop)."""
    import subprocess
    csrc = os.path.join(common.ROOT, "jurassic-gpu_amd", "csrc")
    asm = tmp_path / "k.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-ffp-contract=off", "-std=c++17",
                           "-I" + os.path.join(common.ROOT, "include"), "-I" + csrc, "-DJUR_ND=100", "-DJUR_NG=30", "-S",
                           "--cuda-device-only", "-o", str(asm), os.path.join(csrc, "jur_kernels.hip")],
                          stderr=subprocess.DEVNULL)
    text = asm.read_text()
    meta = text[text.index("amdhsa.kernels:"):]
    seen = {}
    for block in meta.split("  - .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", block).group(1)
        seen[name] = (int(re.search(r"\.vgpr_count:\s+(\d+)", block).group(1)),
                      int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", block).group(1)))
    budget = {"jur_ega_kernelILb1ELb1ELb1ELb1E": (64, 0),      # the look-up on bracket records: what strict tables run
              "jur_ega_kernelILb1ELb1ELb1ELb0E": (64, 0),      # ... on the two arrays (JUR_EGA_NO_REC)
              "jur_combine_kernel": (64, 0), "jur_combine_group_kernel": (64, 0),
              "jur_trace_kernel": (128, 32), "jur_trace_lanes_kernel": (128, 32)}
    for key, (limit, scratch_limit) in budget.items():
        hits = {n: v for n, v in seen.items() if key in n and "kat" not in n}
        assert len(hits) == 1, (key, list(seen))
        (vgprs, scratch), = hits.values()
        assert vgprs <= limit and scratch <= scratch_limit, (key, vgprs, scratch)

    # the cache policy of the three streams that are written once and read once (profiles/
    # r04_cache_policy_experiment.json): non-temporal stores of the path transmittances and of the tracer's LOS rows,
    # non-temporal loads of the transmittances in the radiance update -- and NOT of the look-up's LOS rows
    def body(key):
        (name,) = [n for n in seen if key in n and "kat" not in n]
        start = text.index("\n" + name + ":")
        return text[start:text.index("s_endpgm", start)]
    nt = lambda code, op: len(re.findall(r"^\s*%s\S*\s.*\bnt\b" % op, code, re.M))
    ega = body("jur_ega_kernelILb1ELb1ELb1ELb1E")
    assert nt(ega, "global_store") >= 1 and nt(ega, "global_load") == 0, (nt(ega, "global_store"), nt(ega, "global_load"))
    assert nt(body("jur_trace_kernel"), "global_store") >= 5      # (the emitter and window loops are rolled in this build)
    assert nt(body("jur_combine_group_kernel"), "global_load") >= 1 and nt(body("jur_combine_kernel"), "global_load") >= 1


def test_los_point_estimate_balances_a_sorted_scan(oracle):
    """jur_estimate_los_points / jur_balance_rays (host arithmetic of the library, no GPU): the closed-form estimate of the
    tracer's point count against the oracle's ACTUAL count for a tangent-height scan, a nadir sweep, rays that miss the
    atmosphere and an observer inside it; shares cut from it carry equal actual points within 6 % where equal ray counts
    are off by a factor of two (what jur_formod_host_multi and shard.balanced_ranges deal by)."""
    from jurassic_hip import lib, shard
    g = synth.limb_geometry(1200, scan=True)
    extra = np.array([[0, 780.0, 0, 0, 95.0, 0, 20.0], [0, 30.0, 0, 0, 5.0, 0, 3.0], [0, 780.0, 0, 0, -0.005, 0, 27.0]])
    geom = np.vstack([g, synth.nadir_geometry(150, seed=3), extra])
    case = common.limb_case(geom=geom)
    ref = oracle.formod_rays(case.ctl, case.atm, case.oracle_tables(oracle), case.geom)["np"].astype(float)
    est = lib.estimate_los_points(case.ctl, case.atm, case.geom)
    assert est.shape == ref.shape and np.all(est >= 0) and est[1350] == 0 and ref[1350] == 0          # the ray that never enters
    live = ref > 0
    rel = np.abs(est[live] - ref[live]) / ref[live]
    print("estimate against the tracer's count: mean %.3f, worst %.3f; totals %.0f / %.0f" % (rel.mean(), rel.max(), est.sum(), ref.sum()))
    assert rel.mean() < 0.03 and rel.max() < 0.15 and abs(est.sum() - ref.sum()) < 0.02 * ref.sum()
    for world in (2, 3, 8):
        ranges = shard.balanced_ranges(case.ctl, case.atm, case.geom, world)
        assert ranges[0][0] == 0 and ranges[-1][1] == len(geom) and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
        pts = [ref[lo:hi].sum() for lo, hi in ranges]
        eq = [ref[slice(*shard.ray_range(r, world, len(geom)))].sum() for r in range(world)]
        assert max(pts) <= 1.06 * min(pts) + 400, (world, pts)        # (refraction lengthens the low rays a little: a systematic 5 %)
        assert max(eq) > 1.5 * min(eq), eq
