#!/bin/bash
# bench.py with several builds of the library (A/B of compile-time variants), kernel times side by side, and a bitwise
# comparison of each against the default build:   tools/ab_so.sh _suffix1 _suffix2 ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
for s in "" "$@"; do
  echo "libjurassic_hip$s.so"
  JURASSIC_HIP_SO=$R/jurassic-gpu_amd/libjurassic_hip$s.so python3 bench.py --steps 5 --no-cpu-baseline --no-host-inclusive --no-package-api --no-extra 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
k = d['roofline']['kernels']
print(json.dumps({'value': round(d['value']), 'ms_per_step': round(d['ms_per_step'], 2), 'kernel_ms': {a: round(b['avg_launch_ms'], 2) for a, b in k.items()}}))"
  JURASSIC_HIP_SO=$R/jurassic-gpu_amd/libjurassic_hip$s.so python3 bench.py --workload nadir_1e5 --steps 20 --no-cpu-baseline --no-host-inclusive --no-package-api --no-extra 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
k = d['roofline']['kernels']
print(json.dumps({'nadir_value': round(d['value']), 'kernel_ms': {a: round(b['avg_launch_ms'], 3) for a, b in k.items()}}))"
done
for s in "$@"; do echo "compare $s"; python3 tools/compare_builds.py jurassic-gpu_amd/libjurassic_hip.so jurassic-gpu_amd/libjurassic_hip$s.so 200000 | cut -c1-300; done
