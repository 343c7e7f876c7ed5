"""N > 1 path on CPU (gloo): the sharding / gather code bench.py runs, and bench.py's own launcher.

* two gloo ranks shard a ragged ray set with shard.ray_range, compute their shard (the oracle standing in for
  the GPU kernels) and gather obs.rad on rank 0 with shard.gather_rows;
* `python3 bench.py --gpus 2 --dry-run` must start its two ranks itself -- as child processes, from a parent
  that never imported torch -- shard one global ray set, gather, verify and relay ONE JSON line with n_gpus 2;
* a world size that differs from --gpus is an error, never a silent 1-GPU run.
"""
import json
import os
import subprocess
import sys
import textwrap
import numpy as np
import common

WORKER = textwrap.dedent("""
    import os, sys
    sys.path[:0] = [%(root)r, os.path.join(%(root)r, 'jurassic-gpu_amd'), os.path.join(%(root)r, 'tests')]
    import numpy as np, torch, torch.distributed as dist
    import common
    from oracle import orc
    from jurassic_hip import shard, synth
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    n = 301                                   # ragged split on purpose
    geom = synth.limb_geometry(n, seed=21)
    case = common.limb_case(geom=geom)
    lo, hi = shard.ray_range(rank, world, n)
    res = orc.formod_rays(case.ctl, case.atm, case.oracle_tables(orc), geom[lo:hi])
    counts = shard.ray_counts(world, n)
    out = torch.full((n, 2), -1.0, dtype=torch.float64) if rank == 0 else None
    full = shard.gather_rows(torch.from_numpy(res['rad']), counts, dst=0, out=out)
    again = shard.gather_rows(torch.from_numpy(res['rad']), counts, dst=0)          # without a preallocated buffer
    if rank == 0:
        ref = orc.formod_rays(case.ctl, case.atm, case.oracle_tables(orc), geom)
        assert full is out and full.shape == (n, 2), full.shape
        assert np.array_equal(full.numpy(), ref['rad'])
        assert np.array_equal(again.numpy(), ref['rad'])
        print('GATHER_OK', counts)
    else:
        assert full is None and again is None
    dist.destroy_process_group()
""")


def test_two_rank_shard_and_gather(tmp_path, oracle):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % dict(root=common.ROOT))
    env = dict(os.environ, OMP_NUM_THREADS="2", MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29653", str(script)],
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "GATHER_OK [150, 151]" in out.stdout


def test_ray_ranges_partition_exactly():
    from jurassic_hip import shard
    for n in (0, 1, 7, 1088, 10_000_000):
        for w in (1, 2, 4, 8):
            r = [shard.ray_range(k, w, n) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1
            assert shard.ray_counts(w, n) == [h - l for l, h in r]


def _bench(*argv, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    e.update({"OMP_NUM_THREADS": "2", **(env or {})})
    return subprocess.run([sys.executable, os.path.join(common.ROOT, "bench.py"), *argv], capture_output=True, text=True,
                          env=e, timeout=900)


def test_bench_starts_its_own_ranks():
    """What the round-1 bench could not do: `bench.py --gpus 2` without torchrun."""
    out = _bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--rays", "2001", "--dry-run")
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout                               # exactly one JSON line on stdout
    doc = json.loads(lines[0])
    assert doc["n_gpus"] == 2 and doc["dry_run"] is True
    assert doc["config"]["workload"] == "limb_1e7_sharded"           # configs[3]: one global set, sharded
    assert doc["config"]["rays_total"] == 2001 and doc["config"]["rays_per_gpu"] == [1000, 1001]
    kids = doc["launcher"]["children"]
    assert [k["rank"] for k in kids] == [0, 1] and [k["local_rank"] for k in kids] == [0, 1]
    assert all(k["world_size"] == 2 for k in kids)
    assert [k["rays"] for k in kids] == [[0, 1000], [1000, 2001]]
    pids = {k["pid"] for k in kids}
    assert len(pids) == 2 and doc["launcher"]["parent_pid"] not in pids      # two children, neither is the parent
    assert doc["launcher"]["parent_imported_torch"] is False                 # the parent never came near a GPU
    assert doc["gather_check"]["differing_values"] == 0 and doc["gather_check"]["sampled_rays"] > 1000
    assert doc["scaling"] == "strong" and doc["value"] > 0


def test_bench_refuses_a_world_that_differs_from_gpus():
    """--gpus 8 inside a 1-rank world (or the reverse) must fail, not measure one GPU."""
    out = _bench("--gpus", "2", "--dry-run", "--rays", "100",
                 env=dict(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29655"))
    assert out.returncode != 0 and "WORLD_SIZE=1" in out.stderr
    assert not out.stdout.strip()
    out = _bench("--gpus", "1", "--dry-run", "--rays", "100",
                 env=dict(RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29656"))
    assert out.returncode != 0 and not out.stdout.strip()


def test_bench_dry_run_single_rank_line():
    out = _bench("--dry-run", "--rays", "500", "--steps", "1", "--warmup", "0")
    assert out.returncode == 0, out.stdout + out.stderr
    doc = json.loads(out.stdout.strip())
    assert doc["n_gpus"] == 1 and doc["config"]["workload"] == "limb_1e6" and doc["config"]["rays_per_gpu"] == [500]


def test_bench_airs_workload_starts_the_wide_build_and_shards_by_index():
    """configs[4] through the launcher: `bench.py --gpus 2 --workload airs_2378_sharded --dry-run` starts its two ranks
    with the ND = 2378 dimensions exported (the ranks import 2378-channel structs), every rank builds only its own rows
    of the index-addressable observation set, rank 0 re-builds the sampled rows from their indices and finds them
    equal to what was gathered."""
    out = _bench("--gpus", "2", "--workload", "airs_2378_sharded", "--steps", "1", "--warmup", "0", "--rays", "301", "--dry-run")
    assert out.returncode == 0, out.stdout + out.stderr
    doc = json.loads(out.stdout.strip())
    assert doc["n_gpus"] == 2 and doc["dry_run"] is True and doc["scaling"] == "weak"
    assert doc["config"]["workload"].startswith("airs_2378_sharded") and doc["config"]["channels"] == 2378
    assert doc["config"]["rays_per_gpu"] == [150, 151] and doc["gather_check"]["differing_values"] == 0
    # and a single rank that finds itself without the dimensions re-runs itself with them (one line, same contract)
    out = _bench("--workload", "airs_2378_sharded", "--steps", "1", "--warmup", "0", "--rays", "40", "--dry-run")
    assert out.returncode == 0, out.stdout + out.stderr
    doc = json.loads(out.stdout.strip())
    assert doc["n_gpus"] == 1 and doc["config"]["channels"] == 2378 and doc["config"]["rays_total"] == 40
    assert "1 of the 8 GPU shares" in doc["config"]["workload"]


def test_bench_eight_ranks_ragged_dry_run():
    """First contact with an 8-GPU node should be boring: `bench.py --gpus 8 --dry-run` through its own launcher for
    both sharded workloads with ray counts that do not divide by eight -- eight children, eight contiguous ranges that
    tile the set, the gather verified on sampled rows (shard boundaries included), the backend recorded per rank."""
    for workload, rays, nd in (("limb_1e7_sharded", 100_003, 4), ("airs_2378_sharded", 1_003, 2378)):
        out = _bench("--gpus", "8", "--workload", workload, "--steps", "1", "--warmup", "1", "--rays", str(rays), "--dry-run",
                     env=dict(OMP_NUM_THREADS="1"))
        assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
        doc = json.loads(out.stdout.strip())
        assert doc["n_gpus"] == 8 and doc["dry_run"] is True and doc["config"]["channels"] == nd
        counts = doc["config"]["rays_per_gpu"]
        assert len(counts) == 8 and sum(counts) == rays and max(counts) - min(counts) == 1
        kids = doc["launcher"]["children"]
        assert [k["rank"] for k in kids] == list(range(8)) and len({k["pid"] for k in kids}) == 8
        assert kids[0]["rays"][0] == 0 and kids[-1]["rays"][1] == rays
        assert all(a["rays"][1] == b["rays"][0] for a, b in zip(kids, kids[1:]))
        assert all(k["backend"] == "gloo" for k in kids)          # (on the GPUs: "nccl", which is RCCL on ROCm)
        assert doc["gather_check"]["differing_values"] == 0
        assert doc["scaling"] == ("strong" if workload.startswith("limb") else "weak")


def test_geometry_is_index_addressable():
    """Ray i of a workload is a function of i alone: any slice, any order, any rank builds the same rows; the uniform
    stream is splitmix64 (known answer: the generator's published first output for seed 1234567)."""
    import bench
    from jurassic_hip import synth
    z = int(synth.splitmix64_uniform(1234567, [0])[0] * 2 ** 53)
    assert z == 6457827717110365317 >> 11
    full = bench.workload_rays("limb_1e7_sharded", np.arange(5000))
    assert np.array_equal(full[1234:2345], bench.workload_rays("limb_1e7_sharded", np.arange(1234, 2345)))
    idx = np.array([4999, 0, 77, 4096])
    assert np.array_equal(full[idx], bench.workload_rays("limb_1e7_sharded", idx))
    far = bench.workload_rays("limb_1e7_sharded", np.array([9_999_999]))            # no need to build 1e7 rows for the last one
    assert far.shape == (1, 7) and 3.0 <= far[0, 4] <= 68.0 and far[0, 0] == 9_999_999 % 64
    assert 3.0 <= full[:, 4].min() < 3.1 and 67.9 < full[:, 4].max() <= 68.0 and abs(full[:, 4].mean() - 35.5) < 1.0
    nad = bench.workload_rays("nadir_1e5", np.arange(2000))
    assert np.all(np.abs(nad[:, 6]) <= 8.01) and nad[:, 6].std() > 4.0
