#!/bin/bash
# One gpurun call of a development round: GPU tests, instruction micro-benchmark, the bench line.
# A step that had to be killed ends the call (no further GPU work on a device in an unknown state).
mkdir -p gpurun_out
TAG=${1:-r02}
step() { # name, seconds, command...
  name=$1; secs=$2; shift 2
  echo "== $name"
  timeout -k 10 "$secs" "$@" > "gpurun_out/${TAG}_$name.log" 2>&1
  rc=$?
  tail -4 "gpurun_out/${TAG}_$name.log"
  if [ $rc -ge 124 ]; then echo "step $name killed (rc $rc): stopping"; exit $rc; fi
  return 0
}
step gpu_tests 900 python3 -m pytest tests -q -m gpu
step microbench_build 120 /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -Wno-unused-result -o /tmp/microbench_valu tools/microbench_valu.hip
step microbench 120 /tmp/microbench_valu
step bench 600 python3 bench.py --steps 5
