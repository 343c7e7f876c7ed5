#!/bin/bash
# Counters of the look-up kernel under several settings of the library's environment switches, side by side (run on
# the GPU box through gpurun):   tools/pmc_env.sh <outdir> "<name>:<ENV=..> <ENV=..>" ... -- "<pass-name counter ...>" ...
# One rocprofv3 --pmc run per (setting, pass), nothing but counters in it, bench.py directly after `--`.
ulimit -c 0
OUT=$(realpath -m "$1"); shift
SETS=()
while [ "$1" != "--" ]; do SETS+=("$1"); shift; done
shift
PASSES=("$@")
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
for set in "${SETS[@]}"; do
  name=${set%%:*}; envs=${set#*:}
  for pass in "${PASSES[@]}"; do
    pn=${pass%% *}
    ( export $envs; timeout -k 10 300 rocprofv3 --pmc ${pass#* } --output-format csv -d "$OUT/$name/$pn" -- \
      python3 "$ROOT/bench.py" --no-cpu-baseline --no-host-inclusive --no-package-api --no-extra --workload ${WORKLOAD:-limb_1e6} --rays ${RAYS:-1000000} --steps 1 --warmup 0 > "$OUT/${name}_$pn.log" 2>&1 ) || { echo "pass $pn of $name failed"; tail -3 "$OUT/${name}_$pn.log"; exit 1; }
  done
  python3 - "$OUT/$name" "${KERNEL:-jur_ega}" "$name" <<'PY'
import collections, csv, glob, os, sys
agg = collections.defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if sys.argv[2] in row["Kernel_Name"]:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
print(sys.argv[3], {k: float("%.4g" % (sum(v) / len(v))) for k, v in sorted(agg.items())})
PY
done
