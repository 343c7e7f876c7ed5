#!/usr/bin/env python3
"""Headline benchmark: limb EGA forward model, rays/s on N MI355X.

One "step" = one pass of the hot path (ray sort + ray tracing + along-path EGA
integration + epilogue) over one batch of synthetic rays already resident in
HBM, followed (N > 1) by the RCCL gather of the per-detector radiances to
rank 0.  Contract: see the task statement; ONE JSON line on stdout.

Workloads
  N = 1  "limb_1e6"  BASELINE.json configs[2] (SURVEY.md 8d "C3"): 1e6 limb rays, view-point altitude
         ~ U[3, 68] km from 780 km, 5 emitters (CO2, H2O, O3, F11, CCl4), 4 channels {792, 832, 1450, 2150}
         cm^-1 (all four continua active), 64 perturbed atmosphere profiles, synthetic emissivity tables
         33 p x 10 T x ~203 u.  (--workload nadir_1e5 is configs[1].)
  N > 1  "limb_1e7_sharded"  configs[3]: ONE global seeded set of 1e7 such rays; rank r owns the contiguous
         range shard.ray_range(r, N, 1e7), the radiances are gathered with shard.gather_rows, and rank 0
         recomputes a sample of global ray indices on its own GPU and compares them bit for bit with the
         gathered rows.  Total work is fixed as N grows ("scaling": "strong").
  any N  --workload airs_2378_sharded  configs[4] (SURVEY 8d "C5"): AIRS-like nadir sounder, 2378 channels
         650 .. 2665 cm^-1, CO2 / H2O / O3, full-size synthetic tables (7134 tables, 4.8e8 entries, 3.8 GB per GPU),
         125 000 observations PER GPU -- at N = 8 the configuration's 1e6 observations, at N = 1 one GPU's share of
         it ("scaling": "weak"); same sharding, gather and sampled bit-for-bit check.  Needs the ND = 2378 build of
         the library (the reference's -D ND / -D NG): bench.py starts itself again with those dimensions exported.
  The ray sets are index-addressable (synth.limb_rays / nadir_rays: ray i is a function of (seed, i), the splitmix64
  stream SURVEY 8d names): a rank builds its own rows [lo, hi) only, rank 0 the few sampled rows it re-computes.

`python3 bench.py --gpus N` with N > 1 and no RANK in the environment starts the N ranks itself, as child
processes (python -m torch.distributed.run ... bench.py ...), before anything in this process has imported
torch or touched a GPU, and relays rank 0's JSON line; under an external torchrun (RANK set) it is a rank.
It exits non-zero whenever the world size differs from --gpus.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, "jurassic-gpu_amd"), os.path.join(ROOT, "tests")]

HBM_PEAK = 8.0e12                    # B/s, MI355X_MICROARCH.md
VALU_PEAK = 256 * 4 * 2.4e9 / 4      # wavefront VALU instructions/s: 1024 SIMDs, one fp64 instruction per 4 cycles
# what the device code is built from: the kernels, the structures they share with the host, the compile-time dimensions
# and constants, the build flags.  (include/jurassic_hip.h -- prototypes of the C-ABI only -- was part of this list until
# round 4; a new entry point there does not change a kernel.)
KERNEL_SOURCES = ["jurassic-gpu_amd/csrc/jur_kernels.hip", "jurassic-gpu_amd/csrc/jur_internal.h",
                  "jurassic-gpu_amd/csrc/Makefile", "include/jurassic_abi.h"]
PMC_SUMMARY = os.path.join(ROOT, "profiles", "pmc_current.json")


def kernel_source_sha():
    """Identifies the device code a PMC summary was measured on: sha256 over the files the kernels are built from."""
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        h.update(rel.encode())
        h.update(open(os.path.join(ROOT, rel), "rb").read())
    return h.hexdigest()


# total: rays of the whole job (per_gpu: ... of one rank; the job has N times that)
WORKLOADS = {
    "limb_1e6": dict(kind="limb", total=1_000_000, scaling="weak"),            # configs[2]
    "limb_1e7_sharded": dict(kind="limb", total=10_000_000, scaling="strong"),   # configs[3]
    "nadir_1e5": dict(kind="nadir", total=100_000, scaling="weak"),              # configs[1]
    "airs_2378_sharded": dict(kind="airs", per_gpu=125_000, scaling="weak"),     # configs[4]: 1e6 observations at N = 8
}
WIDE_DIMS = dict(JUR_ND="2378", JUR_NG="3", JUR_SUFFIX="_nd2378")             # the build airs_2378_sharded needs


def needs_wide_build(workload):
    return WORKLOADS[workload]["kind"] == "airs" and any(os.environ.get(k) != v for k, v in WIDE_DIMS.items())


def parse_args(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS))
    ap.add_argument("--rays", type=int, default=0, help="rays of the whole job (default: the workload's size)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-inclusive", action="store_true")
    ap.add_argument("--no-package-api", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the configs[1] side measurement of the N = 1 line")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / sharding / gather rehearsal on the CPU (gloo): no library, no GPU, the forward "
                         "model replaced by a checksum of each ray's geometry; the line says dry_run")
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------------------------------------
# parent: start the N ranks as children.  Nothing here may import torch or touch the GPU.
# ----------------------------------------------------------------------------------------------------------
def launch_ranks(args, argv):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", JUR_BENCH_LAUNCHER_PID=str(os.getpid()))
    if args.workload and WORKLOADS[args.workload]["kind"] == "airs":
        env.update(WIDE_DIMS)                    # the ranks import the ND = 2378 structs and load that build
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)
    lines = [ln for ln in proc.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
    if proc.returncode != 0 or not lines:
        sys.stderr.write("bench.py: the %d-rank run failed (exit %d)\n" % (args.gpus, proc.returncode))
        return proc.returncode or 1
    doc = json.loads(lines[-1])
    if doc.get("n_gpus") != args.gpus:
        sys.stderr.write("bench.py: asked for %d ranks, the run reports %r\n" % (args.gpus, doc.get("n_gpus")))
        return 1
    doc.setdefault("launcher", {}).update(parent_pid=os.getpid(), parent_imported_torch="torch" in sys.modules,
                                          command=" ".join(cmd[1:6]) + " ... bench.py " + " ".join(argv))
    print(json.dumps(doc), flush=True)
    return 0


# ----------------------------------------------------------------------------------------------------------
# workload
# ----------------------------------------------------------------------------------------------------------
NPROF = 64


def workload_rays(workload, idx):
    """(len(idx), 7) rows number idx of the ONE global seeded ray set of a workload -- a function of the indices
    alone (synth.splitmix64_uniform), so every rank builds just its own rows."""
    from jurassic_hip import synth
    if WORKLOADS[workload]["kind"] == "limb":
        return synth.limb_rays(idx, nprofiles=NPROF)
    return synth.nadir_rays(idx)


def global_geometry(workload, nrays, seed=None):
    """Rows 0 .. nrays-1 of the workload's ray set (tools and tests; bench ranks use workload_rays on their range)."""
    import numpy as np
    return workload_rays(workload, np.arange(nrays))


class AirsCase:
    """configs[4]: 2378 channels 650 .. 2665 cm^-1, CO2 / H2O / O3, the limb example's atmosphere,
    full-size synthetic tables 33 p x 10 T x ~203 u per pair.  The 7134 tables (4.8e8 rows) are generated pair by
    pair and fed straight to the library (and, for the CPU legs, to the oracle): nothing is kept in Python."""
    EMITTERS = ["CO2", "H2O", "O3"]

    def __init__(self, geom):
        import common
        from jurassic_hip import abi, textio
        assert (abi.ND, abi.NG) == (2378, 3), "the airs workload needs JUR_ND=2378 JUR_NG=3 JUR_SUFFIX=_nd2378"
        self.nu = [650.0 + i * (2665.0 - 650.0) / 2377 for i in range(2378)]
        self.ctl = abi.make_ctl(self.EMITTERS, self.nu)
        self.atm = textio.read_atm(os.path.join(common.GOLD, "limb", "atm.tab"), self.ctl)
        self.geom = geom
        self.rows = {(g, d): None for g in range(3) for d in range(2378)}       # which pairs have a table
        self._lib = self._orc = None

    def _feed(self, lib_tb, orc_tb):
        from jurassic_hip import synth
        for g, em in enumerate(self.EMITTERS):
            for d, v in enumerate(self.nu):
                rows = synth.table_rows(em, v, id_=d % 7)
                for tb in (lib_tb, orc_tb):
                    if tb is not None:
                        tb.feed_rows(g, d, rows)
                if g == 0:
                    x, f = synth.boxcar_filter(v)
                    if lib_tb is not None:
                        lib_tb.set_filter(d, x, f)
                    if orc_tb is not None:
                        orc_tb.planck_shape(d, x, f)

    def lib_tables(self):
        if self._lib is None:
            from jurassic_hip import lib
            self._lib = lib.Tables(3, 2378)
            self._feed(self._lib, None)
        return self._lib

    def oracle_tables(self, orc, reference_layout=False):
        if self._orc is None:
            self._orc = orc.Tables(3, 2378)       # stride 2378 either way: ctl.nd == ND in this build
            self._feed(None, self._orc)
        return self._orc


def build_case(workload, geom):
    """Control block, atmosphere and tables of a workload around the given geometry rows."""
    import common
    kind = WORKLOADS[workload]["kind"]
    if kind == "limb":
        return common.limb_case(geom=geom, nu=common.CTM4_NU, nprofiles=NPROF)
    if kind == "airs":
        return AirsCase(geom)
    return common.nadir_case(geom=geom)


def usable_cores():
    """Cores this process may actually use: the affinity mask capped by the cgroup CPU quota (a box may show all
    host threads in the mask and still be limited to a share of them)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]                   # cgroup v2
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())                   # cgroup v1
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0 and period > 0:
                n = min(n, max(1, -(-quota // period)))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(case, target_s=15.0, n0=16384, pkg=1088):
    """Oracle (CPU restatement of CPUdrivers.c) timed on bounded samples of the same workload.
    value: all host cores, every thread tracing and integrating its own rays (the fair many-core arrangement);
    as_reference_value: the reference's arrangement AND memory layout -- tables with the channel index fastest at
      stride ND (jurassic.h:408-411: every probe of a bisection is its own cache line), one formod call per package
      of <= NR = 1088 rays with the LOS buffers allocated inside the call (CPUdrivers.c:121-123), ray tracing serial
      (the orphaned `omp for`, CPUdrivers.c:91,137), OpenMP over the rays of the package for the integration (:138-142);
    one_thread_value: one such package on one thread.  A reported baseline, not the target."""
    from oracle import orc
    orc.build()
    ot = case.oracle_tables(orc)
    cores = usable_cores()
    orc.set_threads(cores)

    def timed(n, mode, tables, first=0):
        t0 = time.perf_counter()
        orc.formod_rays(case.ctl, case.atm, tables, case.geom[first:first + n], serial_trace=mode)
        return time.perf_counter() - t0

    n0 = min(len(case.geom), n0)
    dt = timed(n0, 2, ot)
    n1 = int(min(len(case.geom), max(n0, n0 * target_s / max(dt, 1e-3))))
    dt = timed(n1, 2, ot)
    # the reference's arrangement on the reference's table layout, package by package (each call allocates and frees
    # its LOS buffers, as formod_CPU does)
    rt = case.oracle_tables(orc, reference_layout=True)
    npk = max(1, min(len(case.geom) // pkg, 8))
    n_pk = min(pkg, len(case.geom))
    t0 = time.perf_counter()
    for k in range(npk):
        timed(n_pk, 1, rt, first=k * n_pk)
    dta, na = time.perf_counter() - t0, npk * n_pk
    dtc = 0.0
    for k in range(npk):                                   # the same packages on the compact stride, for the ratio
        dtc += timed(n_pk, 1, ot, first=k * n_pk)
    threads = orc.set_threads(1)
    n2 = min(len(case.geom), max(pkg // 8, 1))
    dt1 = timed(n2, 1, rt)
    orc.set_threads(threads)
    from jurassic_hip import abi
    return dict(value=n1 / dt, unit="rays/s", cores=cores, kind="port",
                note="oracle restatement of CPUdrivers.c/jr_common.h (the reference itself needs GSL and cannot be built "
                     "here).  `value` runs it on a compact table stride with every thread tracing its own rays: faster "
                     "than the reference would be.  `as_reference_value` is the reference's own arrangement on its own "
                     "table layout (stride ND * 4 B)",
                sample="first %d rays of the workload, OpenMP over rays (each thread traces and integrates its own), %.1f s"
                       % (n1, dt),
                as_reference_value=na / dta,
                as_reference_layout="ND-strided (jurassic.h:408-411): u/eps[gas][p][T][u][ND = %d], %d channels used"
                                    % (abi.ND, case.ctl.nd),
                as_reference_sample="first %d rays as %d formod calls of %d rays: LOS buffers allocated per call, serial "
                                    "ray tracing, OpenMP inside each call for the integration, %.1f s" % (na, npk, n_pk, dta),
                as_reference_vs_compact_stride=dta / dtc,
                one_thread_value=n2 / dt1,
                one_thread_sample="first %d rays, 1 thread, reference layout and arrangement, %.1f s" % (n2, dt1))


def algorithmic_bytes(case, n=4096):
    from oracle import orc
    orc.build()
    orc.set_threads(usable_cores())
    return orc.algorithmic_bytes(case.ctl, case.atm, case.oracle_tables(orc), case.geom[:n])


def load_pmc(workload):
    """PMC summary of the current device code, or (None, reason).  The summary (tools/pmc_profile.sh ->
    tools/pmc_summary.py) records the sha256 of the kernel sources it was measured on; one measured on other
    code is not used."""
    if not os.path.exists(PMC_SUMMARY):
        return None, "no profiles/pmc_current.json"
    try:
        doc = json.load(open(PMC_SUMMARY))
    except ValueError:
        return None, "profiles/pmc_current.json unreadable"
    if doc.get("kernel_source_sha256") != kernel_source_sha():
        return None, "profiles/pmc_current.json was measured on other kernel sources (sha mismatch)"
    w = doc.get("workloads", {}).get(workload)
    if not w:
        return None, "profiles/pmc_current.json has no pass for workload %s" % workload
    return w, None


def roofline_block(workload, kms, nrays_step, steps, sum_np, shape, ab):
    """Per-kernel roofline figures from the event-timed launch durations of THIS run.
    bound valu_fp64_issue: wavefront VALU instructions per launch (PMC SQ_INSTS_VALU of the same device code,
    scaled by rays) / launch duration / (1024 SIMDs x 2.4 GHz / 4 cycles).  hbm_frac: PMC HBM bytes per launch
    (2*FETCH_SIZE + WRITE_SIZE, KiB, gfx950 correction) / duration / 8 TB/s.  hbm_compulsory_frac: the bytes the
    kernel cannot avoid moving (its LOS rows in, its results out, no table traffic), counted from this run's
    LOS point total.  algorithmic_frac: SURVEY 8d's byte count of the REFERENCE algorithm (no cache credit)
    over the same duration -- kept for comparison, exceeds 1 because warm-started searches replace most probes."""
    ng, nd, nw, npair_tab, ng_tab = shape
    # the sharded 1e7-ray set is the 1e6-ray workload's ray distribution at another size: its counters scale by rays
    pmc, why = load_pmc("limb_1e6" if workload.startswith("limb") else workload)
    compulsory = {  # bytes per LOS point (ray, point): reads + writes each kernel must do
        "trace": 8.0 * (4 + nw + ng),                                   # writes p, T, ds, q_H2O, k[nw], u[ng]
        "ega": 8.0 * (2 + ng_tab + npair_tab),                          # reads p, T, u[g]; writes one double per pair
        "combine": 8.0 * (4 + nw + min(ng, 2) + npair_tab),             # reads the rows the continua use + every pair's transmittance
    }
    out = {}
    for k in ("trace", "ega", "combine"):
        n = max(1, kms[k + "_launches"])
        avg_s = kms[k + "_ms"] / n * 1e-3
        if avg_s <= 0:
            continue
        rays_per_launch = nrays_step * steps / n
        e = {"avg_launch_ms": avg_s * 1e3, "launches": n, "rays_per_launch": rays_per_launch,
             "hbm_compulsory_frac": compulsory[k] * sum_np * steps / n / avg_s / HBM_PEAK}
        if pmc and k in pmc["kernels"]:
            c = pmc["kernels"][k]
            scale = rays_per_launch / c.get("rays_per_launch", pmc["rays_per_launch"])
            e["valu_insts_per_launch"] = c["SQ_INSTS_VALU"] * scale
            e["valu_issue_frac"] = c["SQ_INSTS_VALU"] * scale / avg_s / VALU_PEAK
            e["traffic"] = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024 * scale
            e["hbm_frac"] = e["traffic"] / avg_s / HBM_PEAK
        out[k] = e
    dom = out["ega"] if "ega" in out else out["combine"]
    a_ega = ab["ega"] / ab["rays"]
    block = {"kernel": "jur_ega_kernel" if "ega" in out else "jur_combine_kernel",
             "avg_launch_ms": dom["avg_launch_ms"], "rays_per_launch": dom["rays_per_launch"],
             "hbm_compulsory_frac": dom["hbm_compulsory_frac"],
             "algorithmic_bytes_per_ray": a_ega,
             "algorithmic_frac": a_ega * dom["rays_per_launch"] / (dom["avg_launch_ms"] * 1e-3) / HBM_PEAK,
             "kernels": out}
    if "valu_issue_frac" in dom:
        block.update(bound="valu_fp64_issue", achieved=dom["valu_insts_per_launch"] / (dom["avg_launch_ms"] * 1e-3) / 1e9,
                     peak=VALU_PEAK / 1e9, unit="G wavefront-instructions/s", frac=dom["valu_issue_frac"],
                     hbm_frac=dom["hbm_frac"], traffic=dom["traffic"],
                     pmc_source="profiles/pmc_current.json (sha256 of the kernel sources matches this build)")
    else:   # no counters of this code version: the one fraction this run can measure by itself
        block.update(bound="hbm", achieved=dom["hbm_compulsory_frac"] * HBM_PEAK / 1e9, peak=HBM_PEAK / 1e9, unit="GB/s",
                     frac=dom["hbm_compulsory_frac"], traffic=None, pmc_source=None, pmc_missing=why)
    block["note"] = ("frac is a fraction of the named bound's peak for the dominant kernel (valu_fp64_issue: one wavefront "
                     "instruction per 4 cycles and SIMD, the fp64 rate -- 32-bit instructions issue in 2 cycles, so a stream "
                     "with many of them can read slightly above 1: the many-channel look-up does, 1.03).  Durations: THIS run's "
                     "event-timed launches.  VALU instruction and HBM byte counts per launch: NOT counters of this run -- "
                     "a rocprofv3 --pmc pass of the same device code on the builder's box (profiles/pmc_current.json, "
                     "tools/pmc_profile.sh), scaled by rays per launch and refused when the kernel sources changed since")
    return block


# ----------------------------------------------------------------------------------------------------------
# one rank
# ----------------------------------------------------------------------------------------------------------
def main(argv):
    args = parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "RANK" not in os.environ:
        return launch_ranks(args, argv)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d\n" % (args.gpus, world))
        return 2
    workload = args.workload or ("limb_1e6" if world == 1 else "limb_1e7_sharded")
    if needs_wide_build(workload):
        # The struct dimensions are fixed when jurassic_hip.abi is imported and select the library build: run this
        # very command again as a CHILD with them exported (nothing here has touched the GPU yet) and hand on its
        # line and exit code.  Under torchrun every rank does this; RANK / MASTER_* travel with the environment.
        proc = subprocess.run([sys.executable, os.path.abspath(__file__)] + argv, env=dict(os.environ, **WIDE_DIMS))
        return proc.returncode
    # stdout carries exactly one JSON line: whatever libraries print to file descriptor 1 on the way (RCCL's
    # version banner, for one) is sent to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist
    from jurassic_hip import shard

    dry = args.dry_run
    use_dist = world > 1 or "RANK" in os.environ        # under torchrun the collective path runs even at N = 1
    if dry:
        dev = torch.device("cpu")
        if use_dist:
            dist.init_process_group("gloo")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the library has no CPU path")
        # rehearsal on a box with fewer GPUs than ranks (JUR_BENCH_REHEARSAL=1): every rank on a device it shares,
        # gloo instead of RCCL (which refuses two ranks on one device); the line says so and is not a measurement
        rehearsal = os.environ.get("JUR_BENCH_REHEARSAL") == "1"
        gpu = local_rank % torch.cuda.device_count() if rehearsal else local_rank
        torch.cuda.set_device(gpu)
        dev = torch.device("cuda", gpu)
        if use_dist:
            if rehearsal:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=dev)
    if use_dist and dist.get_world_size() != args.gpus:
        raise SystemExit("process group has %d ranks, --gpus %d" % (dist.get_world_size(), args.gpus))

    spec = WORKLOADS[workload]
    total = args.rays or spec.get("total") or spec["per_gpu"] * world
    lo, hi = shard.ray_range(rank, world, total)
    counts = shard.ray_counts(world, total)
    nrays = hi - lo
    case = build_case(workload, workload_rays(workload, np.arange(lo, hi)))   # this rank's rows of the ONE global set
    nd = case.ctl.nd

    memory = None
    if dry:
        model = None

        def forward(geom_rows):    # stands in for the forward model: any function of the ray alone
            g = torch.from_numpy(np.ascontiguousarray(geom_rows))
            return torch.stack([(g * (d + 1.0)).sum(dim=1) for d in range(nd)], dim=1)
        d_rad = torch.zeros((nrays, nd), dtype=torch.float64)
        geom_local = case.geom

        def step():
            d_rad.copy_(forward(geom_local))
    else:
        from jurassic_hip import lib
        model = lib.Model(case.ctl, case.lib_tables(), device=dev.index)
        model.set_atm(case.atm)
        # Memory plan of this rank, made while only the tables are resident: what the run itself will allocate beside
        # the model's workspace (the call's arrays, the copies of the determinism check, rank 0's gather target and its
        # sampled re-computation) is set aside, the workspace budget takes the rest up to the library's default.
        memory = plan_memory(lib, model, dev.index, nrays, nd, total if (use_dist and rank == 0) else 0)
        d_geom = torch.from_numpy(np.ascontiguousarray(case.geom.T)).to(dev)       # [7][nr]
        d_rad = torch.zeros((nrays, nd), dtype=torch.float64, device=dev)
        d_tau = torch.zeros((nrays, nd), dtype=torch.float64, device=dev)
        d_tp = torch.zeros((3, nrays), dtype=torch.float64, device=dev)
        d_np = torch.zeros(nrays, dtype=torch.int32, device=dev)
        d_status = torch.zeros(1, dtype=torch.int32, device=dev)
        model.reserve(nrays)

        def step():
            d_rad.zero_()          # input rad carries the NaN mask; all finite here
            stream = torch.cuda.current_stream().cuda_stream
            model.formod_device(nrays, d_geom.data_ptr(), d_rad.data_ptr(), d_tau.data_ptr(), d_tp.data_ptr(),
                                d_np.data_ptr(), d_status.data_ptr(), stream)
    gathered = torch.empty((total, nd), dtype=torch.float64, device=dev) if (use_dist and rank == 0) else None

    def full_step():
        step()
        if use_dist:                # per-detector radiances to rank 0: each peer sends its block straight to the root
            if not dry and os.environ.get("JUR_BENCH_REHEARSAL") == "1":
                torch.cuda.synchronize()    # gloo is not ordered behind the compute stream the way RCCL is
            shard.gather_rows(d_rad, counts, dst=0, out=gathered)

    def fence():
        if use_dist:
            dist.barrier()
        if not dry:
            torch.cuda.synchronize()

    if use_dist:    # RCCL sets up a peer-to-root connection on first use: do that before any step, also with --warmup 0
        shard.gather_rows(torch.zeros((1, nd), dtype=torch.float64, device=dev), [1] * world, dst=0)
    for _ in range(args.warmup):
        full_step()
    fence()
    if model:
        model.enable_timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        full_step()
    fence()
    dt = time.perf_counter() - t0
    kms = model.kernel_ms() if model else None
    if model:
        model.enable_timing(False)
        if int(d_status.item()) != 0:
            raise SystemExit("a ray overflowed NLOS")
        if not bool(torch.isfinite(d_rad).all()):
            raise SystemExit("non-finite radiance in the benchmark output")

    # run-to-run determinism, as the reference's own benchmark harness checks it (formod.c:107-157): run again,
    # count every element that differs from the previous result
    mismatches = 0
    if model:
        first_rad, first_tau = d_rad.clone(), d_tau.clone()
        for _ in range(2):
            step()
            torch.cuda.synchronize()
            mismatches += int((d_rad != first_rad).sum().item()) + int((d_tau != first_tau).sum().item())
        del first_rad, first_tau
    t = torch.tensor([dt, float(mismatches)], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt, mismatches = float(t[0].item()), int(t[1].item())
    peers = [None] * world
    me = dict(rank=rank, local_rank=local_rank, world_size=world, pid=os.getpid(), rays=[lo, hi],
              backend=(dist.get_backend() if use_dist else None))
    if not dry:     # which physical device this rank computed on: N ranks must show N distinct PCI bus ids
        props = torch.cuda.get_device_properties(dev)
        me["device"] = dict(index=dev.index, name=props.name, pci_bus_id=memory["pci_bus_id"],
                            uuid=str(getattr(props, "uuid", "")) or None)
        me["memory"] = {k: v for k, v in memory.items() if k != "pci_bus_id"}
    if use_dist:
        dist.all_gather_object(peers, me)
    else:
        peers = [me]

    # rank 0: the gathered rows must be what ONE process computes for those global ray indices
    verified = None
    if rank == 0 and use_dist:
        rng = np.random.default_rng(5)
        idx = np.unique(np.concatenate([rng.integers(0, total, 4096), [0, total - 1],
                                        np.cumsum(counts)[:-1], np.cumsum(counts)[:-1] - 1]))
        idx = idx[(idx >= 0) & (idx < total)]
        if not torch.equal(gathered[lo:hi], d_rad):
            raise SystemExit("rank 0's own block of the gathered radiances differs from its local result")
        sample_geom = workload_rays(workload, idx)           # the sampled global rows, built from their indices
        if dry:
            sample = forward(sample_geom)
        else:
            s_geom = torch.from_numpy(np.ascontiguousarray(sample_geom.T)).to(dev)
            s_rad = torch.zeros((len(idx), nd), dtype=torch.float64, device=dev)
            s_tau, s_tp = torch.zeros_like(s_rad), torch.zeros((3, len(idx)), dtype=torch.float64, device=dev)
            model.formod_device(len(idx), s_geom.data_ptr(), s_rad.data_ptr(), s_tau.data_ptr(), s_tp.data_ptr(), 0,
                                d_status.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            sample = s_rad
        got = gathered[torch.from_numpy(idx).to(dev)]
        bad = int((got != sample).sum().item())
        if bad:
            raise SystemExit("%d of %d sampled gathered radiances differ from rank 0's own recomputation" % (bad, got.numel()))
        verified = dict(sampled_rays=int(len(idx)), differing_values=0,
                        how="rank 0 recomputed these global ray indices on its own device; bit-for-bit equal")

    if rank == 0:
        out = {
            "metric": "rays/s (radiance spectra/s) for limb EGA forward model",
            "value": total * args.steps / dt,
            "unit": "rays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True,
            "scaling": spec["scaling"] if world > 1 else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": workload if spec["kind"] != "airs" or world == 8 else
                       "%s: %d of the 8 GPU shares of configs[4] (125 000 of 1e6 observations each)" % (workload, world),
                       "rays_total": total, "rays_per_gpu": counts, "channels": nd,
                       "emitters": case.ctl.ng, "tables": "synthetic 33p x 10T x ~203u per (gas, channel), fp32",
                       "atm_profiles": int(case.atm.np // 91) if spec["kind"] == "limb" else 1,
                       "geometry": "index-addressable: ray i = f(splitmix64 seed 0x4A55524153534943, i); each rank "
                                   "builds its own rows only",
                       "sharding": "one global seeded ray set, contiguous range per rank (shard.ray_range), "
                                   "obs.rad gathered to rank 0 peer-to-root (shard.gather_rows)"},
            "rerun_mismatches": mismatches,
            "launcher": {"children": peers},
            "note": "radiance parity is against oracle/ (a CPU restatement of the reference: the reference's own "
                    "emissivity tables and GSL are missing blobs, so only its ray-tracing columns are pinned to "
                    "reference-produced data)",
        }
        if verified:
            out["gather_check"] = verified
        if os.environ.get("JUR_BENCH_REHEARSAL") == "1" and not dry:
            out["rehearsal"] = "ranks share GPUs, gloo instead of RCCL: exercises the N-rank code path, measures nothing"
        if dry:
            out["dry_run"] = True
        else:
            sum_np = float(d_np.sum(dtype=torch.int64).item())
            pairs = [(g, d) for (g, d) in case.rows]
            shape = (case.ctl.ng, nd, max(case.ctl.nw, 1), len(pairs), len({g for g, _ in pairs}))
            airs = spec["kind"] == "airs"     # one observation there is 1.3e6 look-ups: samples of tens, not thousands
            if world == 1 and not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(case, n0=32, pkg=64) if airs else cpu_baseline(case)
            ab = algorithmic_bytes(case, 48 if airs else 4096)
            out["roofline"] = roofline_block(workload, kms, nrays, args.steps, sum_np, shape, ab)
            out["roofline"]["whole_path_bytes_per_ray"] = ab["total"] / ab["rays"]
            if world == 1 and not args.no_host_inclusive and hasattr(model, "host_buffers"):
                out["host_inclusive"] = host_inclusive(model, case, args.steps, out["ms_per_step"])
                # SURVEY 8d's metric as written (geometry starts in host memory, radiances end there).  `value` above
                # is the device-resident rate the bench contract defines (inputs in HBM when the timed region starts;
                # the PCIe-inclusive rate is reported, never as `value`): both are here, named for what they are.
                out["value_host_inclusive"] = out["host_inclusive"]["pinned"]["value"]
                out["value_device_resident"] = out["value"]
            if world == 1 and not args.no_package_api:
                out["package_api"] = package_api(model, case)
            if world == 1 and workload == "limb_1e6" and not args.no_extra:
                # configs[1] beside the headline, driver-timed like it (about a second of GPU): the mid-size regime,
                # where a launch does not fill the chip
                out["extra"] = {"nadir_1e5": side_workload("nadir_1e5", dev, steps=20, warmup=3)}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.barrier()          # rank 0 may still have been in its CPU-side bookkeeping
        dist.destroy_process_group()
    return 0


def plan_memory(lib, model, device, nrays, nd, gathered_rows):
    """Sets the model's workspace budget from what the device has free right now minus what this run allocates itself;
    returns the figures for the line.  Refuses a run that cannot fit instead of letting an allocation fail mid-way."""
    info = lib.device_info(device)
    own = nrays * (7 + 2 * nd + 3) * 8 + nrays * 4            # geometry, rad, tau, tp, np of the call
    own += 2 * nrays * nd * 8                                 # the copies the determinism check keeps
    own += gathered_rows * nd * 8                             # rank 0: the gather target
    own += 4200 * (7 + 2 * nd + 3) * 8 * 2                    # rank 0: the sampled rows it re-computes
    reserve = own + (4 << 30)                                 # allocator slack, sort buffers, RCCL's own buffers
    default = 128 << 30
    budget = min(default, info["free"] - reserve)
    if budget < (1 << 30):
        raise SystemExit("bench.py: device %d has %.1f GiB free, the run needs %.1f GiB beside a workspace of at least 1 GiB"
                         % (device, info["free"] / 2**30, reserve / 2**30))
    model.set_workspace_budget(budget)
    return dict(pci_bus_id=info["pci_bus_id"], device_total_bytes=info["total"], free_bytes_after_tables=info["free"],
                table_bytes=model.table_bytes(), own_arrays_bytes=own, workspace_budget_bytes=budget,
                workspace_budget_lowered=budget < default)


def side_workload(workload, dev, steps, warmup):
    """A second workload measured in the same process after the headline (device-resident, event-timed kernels)."""
    import numpy as np
    import torch
    from jurassic_hip import lib
    n = WORKLOADS[workload]["total"]
    case = build_case(workload, workload_rays(workload, np.arange(n)))
    nd = case.ctl.nd
    model = lib.Model(case.ctl, case.lib_tables(), device=dev.index)
    model.set_atm(case.atm)
    d_geom = torch.from_numpy(np.ascontiguousarray(case.geom.T)).to(dev)
    d_rad = torch.zeros((n, nd), dtype=torch.float64, device=dev)
    d_tau, d_tp = torch.zeros_like(d_rad), torch.zeros((3, n), dtype=torch.float64, device=dev)
    d_status = torch.zeros(1, dtype=torch.int32, device=dev)
    model.reserve(n)

    def step():
        d_rad.zero_()
        model.formod_device(n, d_geom.data_ptr(), d_rad.data_ptr(), d_tau.data_ptr(), d_tp.data_ptr(), 0,
                            d_status.data_ptr(), torch.cuda.current_stream().cuda_stream)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    model.enable_timing(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kms = model.kernel_ms()
    ok = int(d_status.item()) == 0 and bool(torch.isfinite(d_rad).all())
    model.close()
    if not ok:
        raise SystemExit("side workload %s: NLOS overflow or a non-finite radiance" % workload)
    return {"value": n * steps / dt, "unit": "rays/s", "ms_per_step": 1e3 * dt / steps, "steps": steps, "warmup": warmup,
            "config": {"workload": workload, "rays_total": n, "channels": nd, "emitters": case.ctl.ng},
            "kernels": {k: {"avg_launch_ms": kms[k + "_ms"] / max(1, kms[k + "_launches"]), "launches": kms[k + "_launches"]}
                        for k in ("trace", "ega", "combine")}}


def host_inclusive(model, case, steps, device_ms):
    """SURVEY 8d's metric as written: geometry starts in host memory, radiances end in host memory
    (jur_formod_host).  Reported beside `value`, never as `value`."""
    import numpy as np
    res = {}
    for kind in ("pinned", "pageable"):
        bufs = model.host_buffers(len(case.geom), pinned=(kind == "pinned"))
        bufs.set_geometry(case.geom)
        model.formod_host_buffers(bufs)                # warm-up: staging buffers are allocated here
        t0 = time.perf_counter()
        n = max(2, min(steps, 5))
        for _ in range(n):                             # rad holds the previous (finite) result: no channel is masked
            model.formod_host_buffers(bufs)
        dt = (time.perf_counter() - t0) / n
        assert np.isfinite(bufs.rad).all()
        res[kind] = {"value": len(case.geom) / dt, "ms_per_step": 1e3 * dt,
                     "overhead_vs_device_resident": 1e3 * dt / device_ms - 1.0}
        bufs.close()
    return {"unit": "rays/s", "entry": "jur_formod_host (host arrays in, host arrays out)", **res}


def package_api(model, case, nr=1088, calls=40):
    """the same rays in packages of <= NR = 1088 through the host entry the drop-in formod() uses
    (one fused kernel per package), one caller thread.  Beside `value`, never `value`."""
    n = min(nr, len(case.geom))
    bufs = model.host_buffers(n, pinned=False)          # ordinary host arrays, as an obs_t's are; allocated once
    bufs.set_geometry(case.geom[:n])
    model.formod_host_buffers(bufs)                     # warm-up: the model's pinned image of a package
    t0 = time.perf_counter()
    inside = 0.0
    for i in range(calls):
        lo = (i * n) % max(1, len(case.geom) - n)
        bufs.set_geometry(case.geom[lo:lo + n])
        inside += model.formod_host_buffers(bufs)
    dt = inside / calls                                  # what a C caller sees (the reference's callers are C)
    dt_py = (time.perf_counter() - t0) / calls           # with this script's numpy marshalling around each call
    bufs.close()
    return {"rays_per_call": n, "ms_per_call": 1e3 * dt, "value": n / dt, "unit": "rays/s", "callers": 1,
            "ms_per_call_with_python_marshalling": 1e3 * dt_py,
            "note": "16 concurrent callers: profiles/r03_lanes_dropin_throughput.json"}


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
