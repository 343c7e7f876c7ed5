"""bench.py's roofline block on the CPU: which counter summaries it accepts, and that what it prints is a fraction
of a stated peak, recomputable from the same inputs."""
import json
import pytest
import bench

KMS = dict(trace_ms=44.0, ega_ms=281.0, combine_ms=91.0, trace_launches=5, ega_launches=5, combine_launches=5)
SHAPE = (5, 4, 1, 20, 5)          # ng, nd, nw, pairs with a table, gases with a table
AB = dict(ega=3.03e6 * 4096, total=3.2e6 * 4096, rays=4096)
SUM_NP = 253e6                     # LOS points of the 1e6-ray batch


def _summary(path, sha):
    doc = {"kernel_source_sha256": sha, "workloads": {"limb_1e6": {"rays_per_launch": 1e6, "kernels": {
        "trace": {"SQ_INSTS_VALU": 2.38e9, "FETCH_SIZE": 5.0e5, "WRITE_SIZE": 2.0e7},
        "ega": {"SQ_INSTS_VALU": 2.649e10, "FETCH_SIZE": 4.12e7, "WRITE_SIZE": 3.95e7},
        "combine": {"SQ_INSTS_VALU": 5.49e9, "FETCH_SIZE": 4.2e7, "WRITE_SIZE": 3.4e5}}}}}
    path.write_text(json.dumps(doc))


def test_counters_of_other_kernel_sources_are_refused(tmp_path, monkeypatch):
    f = tmp_path / "pmc.json"
    monkeypatch.setattr(bench, "PMC_SUMMARY", str(f))
    assert bench.load_pmc("limb_1e6") == (None, "no profiles/pmc_current.json")
    _summary(f, "0" * 64)
    pmc, why = bench.load_pmc("limb_1e6")
    assert pmc is None and "sha mismatch" in why
    r = bench.roofline_block("limb_1e6", KMS, 1_000_000, 5, SUM_NP, SHAPE, AB)
    assert r["bound"] == "hbm" and r["pmc_source"] is None and r["traffic"] is None and "sha mismatch" in r["pmc_missing"]
    assert 0 < r["frac"] < 1 and r["frac"] == r["hbm_compulsory_frac"]      # the fraction a run can measure by itself
    _summary(f, bench.kernel_source_sha())
    pmc, why = bench.load_pmc("limb_1e6")
    assert why is None and pmc["rays_per_launch"] == 1e6
    assert bench.load_pmc("nadir_1e5")[0] is None                            # no pass for that workload in this file


def test_roofline_fraction_is_recomputable_and_below_one(tmp_path, monkeypatch):
    f = tmp_path / "pmc.json"
    monkeypatch.setattr(bench, "PMC_SUMMARY", str(f))
    _summary(f, bench.kernel_source_sha())
    r = bench.roofline_block("limb_1e6", KMS, 1_000_000, 5, SUM_NP, SHAPE, AB)
    assert r["bound"] == "valu_fp64_issue" and r["kernel"] == "jur_ega_kernel" and r["unit"].startswith("G wavefront")
    avg_s = 281.0 / 5 * 1e-3
    assert r["frac"] == pytest.approx(2.649e10 / avg_s / (256 * 4 * 2.4e9 / 4))
    assert r["achieved"] == pytest.approx(r["frac"] * r["peak"]) and r["peak"] == pytest.approx(614.4)
    assert r["traffic"] == pytest.approx((2 * 4.12e7 + 3.95e7) * 1024)        # gfx950: FETCH_SIZE counts half
    assert r["hbm_frac"] == pytest.approx(r["traffic"] / avg_s / 8e12)
    for k in ("trace", "ega", "combine"):
        e = r["kernels"][k]
        assert 0 < e["valu_issue_frac"] < 1 and 0 < e["hbm_frac"] < 1 and 0 < e["hbm_compulsory_frac"] < 1
    assert r["algorithmic_frac"] > 1                                          # kept, labelled: not a fraction of peak
    # half the rays per launch: half the instructions, same fraction at half the time
    kms = dict(KMS, ega_ms=281.0, ega_launches=10)
    r2 = bench.roofline_block("limb_1e7_sharded", kms, 1_000_000, 5, SUM_NP, SHAPE, AB)
    assert r2["bound"] == "valu_fp64_issue" and r2["frac"] == pytest.approx(r["frac"])
