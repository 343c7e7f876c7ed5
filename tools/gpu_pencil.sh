#!/bin/bash
# First contact of the fused kernel with the GPU: one small case under a short limit, then its whole test
# file, then the latency table.  A step that had to be killed ends the call.
mkdir -p gpurun_out
TAG=${1:-r02p}
step() { name=$1; secs=$2; shift 2; echo "== $name"; timeout -k 10 "$secs" "$@" > "gpurun_out/${TAG}_$name.log" 2>&1; rc=$?; tail -${TAIL:-6} "gpurun_out/${TAG}_$name.log"; if [ $rc -ge 124 ]; then echo "step $name killed (rc $rc): stopping"; exit $rc; fi; return 0; }
step first 180 python3 -m pytest tests/test_pencil_gpu.py -q -x -k "nadir_package" -p no:cacheprovider
grep -q "passed" gpurun_out/${TAG}_first.log || { echo "first case not green: stopping"; exit 1; }
step pencil_tests 600 python3 -m pytest tests/test_pencil_gpu.py -q -p no:cacheprovider
TAIL=3 step small 300 python3 tools/bench_small.py
