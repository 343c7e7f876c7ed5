#!/bin/bash
# vector instructions per launch of the three batched kernels for several builds of the library (one rocprofv3 --pmc
# pass each, counters only; PMC="..." picks other counters):   tools/ab_valu.sh _suffix1 _suffix2 ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
for s in "" "$@"; do
  d=$R/gpurun_out/abvalu$s
  rm -rf $d
  ( cd /tmp && JURASSIC_HIP_SO=$R/jurassic-gpu_amd/libjurassic_hip$s.so rocprofv3 --pmc ${PMC:-SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU} --output-format csv -d $d -- python3 $R/bench.py --rays 250000 --steps 1 --warmup 0 --no-cpu-baseline --no-host-inclusive --no-package-api > /dev/null 2>&1 )
  python3 - "$d" "libjurassic_hip$s.so" <<'PY'
import sys, csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        name = "ega" if "jur_ega" in k else "combine" if "jur_combine" in k else "trace" if "jur_trace" in k else None
        if name:
            acc[name][row["Counter_Name"]] += float(row["Counter_Value"]); n[(name, row["Counter_Name"])] += 1
print(sys.argv[2], {k: {c: round(v / max(n[(k, c)], 1) / 1e6, 1) for c, v in d.items()} for k, d in acc.items()}, "(millions per launch of 250 000 rays)")
PY
done
