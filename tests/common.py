"""Shared builders for the parity tests: one `Case` = control block, atmosphere,
geometry and the same synthetic tables fed to both the oracle and the library."""
import os
import numpy as np
from jurassic_hip import abi, synth, textio

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")

LIMB_EMITTERS = ["CO2", "H2O", "O3", "F11", "CCl4"]
LIMB_NU = [792.0, 832.0]
NADIR_EMITTERS = ["CO2"]
NADIR_NU = [667.782, 668.541, 669.811]
CTM4_NU = [792.0, 832.0, 1450.0, 2150.0]     # keeps all four continua switched on


class Case:
    def __init__(self, emitters, nu, atm_file, geom, nprofiles=1, table_kw=None, missing=(), filt=None, **ctl_kw):
        self.ctl = abi.make_ctl(emitters, nu, **ctl_kw)
        base = textio.read_atm(atm_file, self.ctl)
        self.atm = synth.stack_profiles(base, self.ctl, nprofiles, seed=7) if nprofiles > 1 else base
        self.geom = np.asarray(geom, dtype=np.float64)
        self.rows = {}
        kw = table_kw or {}
        for g, em in enumerate(emitters):
            for d, v in enumerate(nu):
                if (g, d) in missing:
                    continue
                pkw = kw(g, d) if callable(kw) else kw          # per-pair table shapes when given a function
                self.rows[(g, d)] = synth.table_rows(em, v, id_=d, **pkw)
        self.filters = [filt(v) if filt else synth.boxcar_filter(v) for v in nu]

    def oracle_tables(self, orc, reference_layout=False):
        """reference_layout: channel stride ND as in the reference's tbl_t (jurassic.h:408-411: u/eps[...][ND], channel
        index fastest) instead of the number of channels in use -- same results, the reference's memory behaviour."""
        tb = orc.Tables(self.ctl.ng, abi.ND if reference_layout else self.ctl.nd)
        for (g, d), r in self.rows.items():
            tb.feed_rows(g, d, r)
        for d, (x, f) in enumerate(self.filters):
            tb.planck_shape(d, x, f)
        return tb

    def lib_tables(self):
        from jurassic_hip import lib
        tb = lib.Tables(self.ctl.ng, self.ctl.nd)
        for (g, d), r in self.rows.items():
            tb.feed_rows(g, d, r)
        for d, (x, f) in enumerate(self.filters):
            tb.set_filter(d, x, f)
        return tb

    def write_files(self, dirname, base="tbl"):
        """ASCII tables + filter files the way the reference expects them."""
        self.ctl.tblbase = os.path.join(dirname, base).encode()
        for (g, d), r in self.rows.items():
            name = "%s_%.4f_%s.tab" % (self.ctl.tblbase.decode(), self.ctl.nu[d], self.ctl.emitter[g].value.decode())
            synth.write_table_file(name, r)
        for d, (x, f) in enumerate(self.filters):
            synth.write_filter_file("%s_%.4f.filt" % (self.ctl.tblbase.decode(), self.ctl.nu[d]), x, f)


def golden_geometry(case):
    """(nr, 7) geometry of the reference example's rad.org rows."""
    nd = {"limb": 2, "nadir": 3}[case]
    return textio.read_obs_array(os.path.join(GOLD, case, "rad.org"), nd)[:, :7]


def limb_case(geom=None, **kw):
    g = golden_geometry("limb") if geom is None else geom
    return Case(LIMB_EMITTERS, kw.pop("nu", LIMB_NU), os.path.join(GOLD, "limb", "atm.tab"), g, **kw)


def nadir_case(geom=None, **kw):
    g = golden_geometry("nadir") if geom is None else geom
    kw.setdefault("write_bbt", 1)
    return Case(NADIR_EMITTERS, NADIR_NU, os.path.join(GOLD, "nadir", "atm.tab"), g, **kw)


def tau_atol(ref_tau):
    """Absolute allowance for a path transmittance next to the relative 1e-9.  The algorithm forms a segment's
    transmittance as (1 - eps) / tau: every look-up leaves ~1e-15 of ABSOLUTE rounding in eps, i.e. ~1e-15 / tau
    relative in the path transmittance, over some hundred segments -- and the last-bit differences between the device's
    and the host's libm in the ray tracer (exp, sin, cos, asin, atan2) enter every one of them.  So the allowance grows
    as tau falls: 1.5e-13 for tau of order one, 5e-14 / tau below, at most 5e-12 (tau <= 1e-2).  What set it: two of
    6 000 random configurations reach 1.2e-12 at tau ~ 1e-4, seed 62207 of tools/fuzz_parity.py 8.1e-12 at tau = 7.6e-3
    (1.06e-9 relative) -- each with every arrangement and every arithmetic of the kernels to the same bits, radiances
    within 1e-12 (profiles/r03_fuzz_failing_seeds_debug.log); the oracle's own transmittance of such rays moves by that
    much when the view latitude changes by 1e-13 degrees."""
    t = np.maximum(np.abs(np.asarray(ref_tau, dtype=np.float64)), 1e-300)
    return 1e-13 + np.minimum(5e-14 / t, 5e-12)


def rel_err(a, b):
    a, b = np.asarray(a), np.asarray(b)
    den = np.maximum(np.abs(b), 1e-300)
    return np.abs(a - b) / den


def oracle_goldens():
    """Committed oracle results for the reference examples with the seeded synthetic tables
    (tests/golden/oracle_examples.json, written by tools/make_oracle_goldens.py): name -> (case, dict of arrays)."""
    import json
    doc = json.load(open(os.path.join(GOLD, "oracle_examples.json")))
    cases = {"limb": limb_case(), "nadir": nadir_case(), "limb_four_continua": limb_case(nu=CTM4_NU)}
    out = {}
    for name, g in doc["cases"].items():
        arr = {k: np.array([[float.fromhex(v) for v in row] for row in g[k]]) for k in ("rad", "tau", "tp")}
        arr["np"] = np.array(g["np"], dtype=np.int32)
        out[name] = (cases[name], arr)
    return out
