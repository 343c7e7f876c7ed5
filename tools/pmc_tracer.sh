#!/bin/bash
# Two PMC passes that say what jur_trace_kernel's wavefronts do with their cycles (instruction classes; active / waiting
# cycles).  usage (on the GPU box): tools/pmc_tracer.sh <outdir>;  summary: per-dispatch means of the tracer's rows.
OUT=$(realpath -m "$1"); ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
pass() {
  name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-host-inclusive --no-package-api --no-extra --rays 1000000 --steps 1 --warmup 0 > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$OUT/$name.log"; }
  echo "pass $name done"
}
pass mix SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_SALU
pass act SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
pass mem SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_ACTIVE_INST_SCA
python3 - "$OUT" <<'PY'
import csv, glob, sys, json, collections
out = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        name = "trace" if "jur_trace" in k else "ega" if "jur_ega" in k else "combine" if "jur_combine" in k else None
        if name: out[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
print(json.dumps({k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in out.items()}, indent=1))
PY
