#!/bin/bash
# builds the limb example's tables/filters in a scratch dir and runs tools/lanes_bench with 1..16 threads
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
D=$(mktemp -d)
cd "$D"
python3 - <<PY
import sys
sys.path[:0] = ["$ROOT", "$ROOT/jurassic-gpu_amd", "$ROOT/tests"]
import common
c = common.limb_case()
c.write_files("$D", base="boxcar")
PY
cp "$ROOT/tests/golden/limb/atm.tab" .
gcc -O2 -fopenmp -I"$ROOT/include" "$ROOT/tools/lanes_bench.c" -o "$D/lanes_bench" -L"$ROOT/jurassic-gpu_amd" -ljurassic_hip -Wl,-rpath,"$ROOT/jurassic-gpu_amd" -lm
for t in 1 2 4 8 16; do JUR_LANES=$t $EXTRA_ENV "$D/lanes_bench" $t ${CALLS:-32} | tail -1; done
