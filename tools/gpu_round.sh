#!/bin/bash
# One gpurun call of a development round: GPU tests, the bench line, latency / concurrency tables, optional PMC
# passes.  A step that had to be killed ends the call (no further GPU work on a device in an unknown state).
#   usage: tools/gpu_round.sh <tag> [tests] [bench] [small] [lanes] [pmc] [micro]      (default: tests bench)
ulimit -c 0          # a faulting kernel must not leave a core file of the whole address space on the box
make -s -C jurassic-gpu_amd/csrc clean-variants 2>/dev/null   # stale experiment builds do not belong to the measured tree
mkdir -p gpurun_out
OUTDIR=$(pwd)/gpurun_out
TAG=${1:-r02}; shift
WHAT="${@:-tests bench}"
step() { # name, seconds, command...
  name=$1; secs=$2; shift 2
  echo "== $name"
  timeout -k 10 "$secs" "$@" > "$OUTDIR/${TAG}_$name.log" 2>&1
  rc=$?
  tail -${TAIL:-4} "$OUTDIR/${TAG}_$name.log"
  if [ $rc -ge 124 ]; then echo "step $name killed (rc $rc): stopping"; exit $rc; fi
  return 0
}
for w in $WHAT; do
  case $w in
    tests) step gpu_tests 1100 python3 -m pytest tests -q -m gpu -p no:cacheprovider ;;
    bench) TAIL=1 step bench 600 python3 bench.py --steps 5 ;;
    airs) TAIL=1 step bench_airs 900 python3 bench.py --workload airs_2378_sharded --steps 3 --warmup 1 --no-cpu-baseline --no-host-inclusive --no-package-api ;;
    nadir) TAIL=1 step bench_nadir 300 python3 bench.py --workload nadir_1e5 --steps 20 --no-cpu-baseline ;;
    small) TAIL=1 step small 300 python3 tools/bench_small.py ;;
    lanes) TAIL=5 step lanes 300 bash tools/run_lanes_bench.sh ;;
    micro) step microbench_build 120 /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -Wno-unused-result -o /tmp/microbench_valu tools/microbench_valu.hip
           TAIL=1 step microbench 120 /tmp/microbench_valu ;;
    pmc)   PMC_SHORT=${PMC_SHORT-1} step pmc 900 bash tools/pmc_profile.sh gpurun_out/${TAG}_pmc --rays 1000000 --steps 1 --warmup 0
           step pmc_summary 60 python3 tools/pmc_summary.py gpurun_out/${TAG}_pmc limb_1e6 1000000 gpurun_out/${TAG}_pmc_current.json
           PMC_SHORT=1 step pmc_nadir 900 bash tools/pmc_profile.sh gpurun_out/${TAG}_pmc_nadir --workload nadir_1e5 --steps 1 --warmup 0
           step pmc_nadir_summary 60 python3 tools/pmc_summary.py gpurun_out/${TAG}_pmc_nadir nadir_1e5 100000 gpurun_out/${TAG}_pmc_current.json ;;
    pmcairs) PMC_SHORT=1 step pmc_airs 1100 bash tools/pmc_profile.sh gpurun_out/${TAG}_pmc_airs --workload airs_2378_sharded --steps 1 --warmup 0
           step pmc_airs_summary 60 python3 tools/pmc_summary.py gpurun_out/${TAG}_pmc_airs airs_2378_sharded 125000 gpurun_out/${TAG}_pmc_current.json ;;
    lanesmode) for md in "JUR_NO_ZERO_COPY=1" "GPU_MAX_HW_QUEUES=8" "JUR_PENCIL_RAYS=0"; do
             TAIL=6 EXTRA_ENV="env $md" CALLS=16 step lanesmode_${md%%=*} 300 bash tools/run_lanes_bench.sh
           done ;;
    torchrun1) TAIL=1 step bench_torchrun1 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --no-cpu-baseline --no-host-inclusive --no-package-api ;;
    wide) TAIL=1 JUR_ND=2378 JUR_NG=3 JUR_SUFFIX=_nd2378 step wide 900 python3 tools/bench_wide.py 4096 ;;
    widestats) cd /tmp; export TMPDIR=/tmp
           JUR_ND=2378 JUR_NG=3 JUR_SUFFIX=_nd2378 step wide_stats 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUTDIR/${TAG}_wide_stats -- python3 $GRAFT_REPO_ROOT/tools/bench_wide.py 4096
           for pass in "sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "fetch FETCH_SIZE" "write WRITE_SIZE" "tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
             set -- $pass; name=$1; shift
             JUR_ND=2378 JUR_NG=3 JUR_SUFFIX=_nd2378 step wide_pmc_$name 900 rocprofv3 --pmc "$@" --output-format csv -d $OUTDIR/${TAG}_wide_pmc/$name -- python3 $GRAFT_REPO_ROOT/tools/bench_wide.py 4096
           done
           cd $GRAFT_REPO_ROOT
           step wide_pmc_summary 60 python3 tools/pmc_summary.py gpurun_out/${TAG}_wide_pmc wide_2378ch_nadir_4096 4096 gpurun_out/${TAG}_wide_pmc_summary.json ;;
    nadirstats) cd /tmp; export TMPDIR=/tmp
           step nadir_stats 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUTDIR/${TAG}_nadir_stats -- python3 $GRAFT_REPO_ROOT/bench.py --workload nadir_1e5 --steps 20 --no-cpu-baseline --no-host-inclusive --no-package-api
           cd $GRAFT_REPO_ROOT
           PMC_SHORT=1 step nadir_pmc 900 bash tools/pmc_profile.sh gpurun_out/${TAG}_nadir_pmc --workload nadir_1e5 --steps 1 --warmup 0
           step nadir_pmc_summary 60 python3 tools/pmc_summary.py gpurun_out/${TAG}_nadir_pmc nadir_1e5 100000 gpurun_out/${TAG}_nadir_pmc_summary.json ;;
    cmpbuilds) TAIL=2 step compare_builds 600 python3 tools/compare_builds.py jurassic-gpu_amd/libjurassic_hip_prev.so jurassic-gpu_amd/libjurassic_hip.so 300000 ;;
    kat) step kat 600 python3 -m pytest tests/test_kat_gpu.py -q -p no:cacheprovider ;;
    ab) TAIL=20 step ab 900 bash tools/ab_env.sh $AB ;;
    fuzz) TAIL=3 step fuzz 1100 python3 tools/fuzz_parity.py ${FUZZ:-3000 400} ;;
    rehearse2) TAIL=1 JUR_BENCH_REHEARSAL=1 step rehearse2 900 python3 bench.py --gpus 2 --rays 600000 --steps 2 --warmup 1 ;;
    abso) TAIL=20 step abso 900 bash tools/ab_env.sh "JURASSIC_HIP_SO=$GRAFT_REPO_ROOT/jurassic-gpu_amd/libjurassic_hip.so" "JURASSIC_HIP_SO=$GRAFT_REPO_ROOT/jurassic-gpu_amd/libjurassic_hip$ABSUF.so"
          TAIL=2 step abso_cmp 600 python3 tools/compare_builds.py jurassic-gpu_amd/libjurassic_hip.so jurassic-gpu_amd/libjurassic_hip$ABSUF.so 300000 ;;
    abv) TAIL=30 step abv 1100 bash tools/ab_so.sh $ABV ;;
    jac) TAIL=1 step jacobian 600 python3 tools/bench_jacobian.py ;;
    conc) TAIL=1 step concurrent 300 python3 tools/bench_concurrent.py ;;
    pencil) step pencil_tests 600 python3 -m pytest tests/test_pencil_gpu.py -q -p no:cacheprovider ;;
    lanestrace) D=$(mktemp -d); ( cd $D && python3 - <<PY
import sys
sys.path[:0] = ["$GRAFT_REPO_ROOT", "$GRAFT_REPO_ROOT/jurassic-gpu_amd", "$GRAFT_REPO_ROOT/tests"]
import common
common.limb_case().write_files("$D", base="boxcar")
PY
           cp $GRAFT_REPO_ROOT/tests/golden/limb/atm.tab $D/
           gcc -O2 -fopenmp -I$GRAFT_REPO_ROOT/include $GRAFT_REPO_ROOT/tools/lanes_bench.c -o $D/lanes_bench -L$GRAFT_REPO_ROOT/jurassic-gpu_amd -ljurassic_hip -Wl,-rpath,$GRAFT_REPO_ROOT/jurassic-gpu_amd -lm )
           export TMPDIR=/tmp
           cd $D
           JUR_LANES=4 step lanes_trace 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_lanestrace -- $D/lanes_bench 4 8
           TAIL=1 JUR_PENCIL_RAYS=0 JUR_LANES=16 step lanes_hwq 300 $D/lanes_bench 16 16      # batched kernels, 16 lanes
           cd $GRAFT_REPO_ROOT ;;
    stats) cd /tmp; export TMPDIR=/tmp
           step kernel_stats 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --no-cpu-baseline --no-host-inclusive --no-package-api --no-extra
           cd $GRAFT_REPO_ROOT ;;
  esac
done
