#!/bin/bash
# PMC passes for the bench kernels (run on the GPU box through gpurun).
# Each pass is its own rocprofv3 run (counter slots: SQ 8, TCC 4; FETCH_SIZE and WRITE_SIZE do
# not fit one pass) with --pmc only -- no trace domains next to counters; the program follows `--` directly.
# usage: tools/pmc_profile.sh <outdir> [bench args...]     then: tools/pmc_summary.py <outdir> <workload> <rays/launch>
set -e
OUT=$(realpath -m "$1"); shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH_ARGS="${@:---rays 1000000 --steps 1 --warmup 0}"
pass() {
  name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-host-inclusive --no-package-api --no-extra $BENCH_ARGS > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$OUT/$name.log"; }
  echo "pass $name done"
}
pass sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass fetch FETCH_SIZE
pass write WRITE_SIZE
if [ -z "$PMC_SHORT" ]; then
pass sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum
fi
