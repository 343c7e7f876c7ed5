"""N > 1 path on CPU: two gloo ranks shard the rays, compute their shard (with
the oracle standing in for the GPU kernels) and gather obs.rad on rank 0."""
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
    counts = [shard.ray_range(r, world, n)[1] - shard.ray_range(r, world, n)[0] for r in range(world)]
    full = shard.gather_rows(torch.from_numpy(res['rad']), counts, dst=0)
    if rank == 0:
        ref = orc.formod_rays(case.ctl, case.atm, case.oracle_tables(orc), geom)
        assert full.shape == (n, 2), full.shape
        assert np.array_equal(full.numpy(), ref['rad'])
        print('GATHER_OK', counts)
    else:
        assert full is None
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
