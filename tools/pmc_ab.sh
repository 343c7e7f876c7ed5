#!/bin/bash
# Counters of the look-up kernel for several builds of the library side by side (run on the GPU box through gpurun):
#   tools/pmc_ab.sh <outdir> "<suffix> ..."  "<pass-name counter counter ...>" ...      ("" = the default build)
# One rocprofv3 --pmc run per (build, pass), nothing but counters in it, bench.py directly after `--`.
ulimit -c 0
OUT=$(realpath -m "$1"); SUFS="$2"; shift 2
PASSES=("$@")
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
for s in $SUFS; do
  [ "$s" = "-" ] && s=""
  for pass in "${PASSES[@]}"; do
    name=${pass%% *}
    JURASSIC_HIP_SO=$ROOT/jurassic-gpu_amd/libjurassic_hip$s.so timeout -k 10 300 rocprofv3 --pmc ${pass#* } --output-format csv -d "$OUT/so$s/$name" -- \
      python3 "$ROOT/bench.py" --no-cpu-baseline --no-host-inclusive --no-package-api --rays ${RAYS:-1000000} --steps 1 --warmup 0 > "$OUT/so${s}_$name.log" 2>&1 || { echo "pass $name of $s failed"; tail -3 "$OUT/so${s}_$name.log"; exit 1; }
  done
  python3 - "$OUT/so$s" "${KERNEL:-jur_ega_kernel}" "libjurassic_hip$s.so" <<'PY'
import collections, csv, glob, os, sys
agg = collections.defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if sys.argv[2] in row["Kernel_Name"]:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
print(sys.argv[3], {k: float("%.4g" % (sum(v) / len(v))) for k, v in sorted(agg.items())})
PY
done
