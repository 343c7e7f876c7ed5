#!/bin/bash
# bench.py under each of the given environment settings (A/B of an experiment switch), kernel times side by side
#   tools/ab_env.sh "JUR_X=0" "JUR_X=1" ...
for e in "$@"; do
  echo "$e"
  env $e python3 bench.py --steps 5 --no-cpu-baseline --no-host-inclusive --no-package-api --no-extra 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
k = d['roofline']['kernels']
print(json.dumps({'value': round(d['value']), 'ms_per_step': round(d['ms_per_step'], 2), 'rerun_mismatches': d['rerun_mismatches'],
      'kernel_ms': {a: round(b['avg_launch_ms'], 2) for a, b in k.items()}}))"
done
