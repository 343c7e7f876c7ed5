#!/bin/bash
# bench.py with the sub-chunk pipeline at several sub-chunk sizes (0 = off)
for p in 0 500000 333312 250000 200000 125000 62464; do
  echo "JUR_PIPE_RAYS=$p"
  JUR_PIPE_RAYS=$p python3 bench.py --steps 5 --no-cpu-baseline --no-host-inclusive 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
k = d['roofline']['kernels']
print(json.dumps({'value': round(d['value']), 'ms_per_step': round(d['ms_per_step'], 2), 'rerun_mismatches': d['rerun_mismatches'],
      'kernel_sum_ms': {a: round(b['avg_launch_ms'] * b['launches'] / d['steps'], 2) for a, b in k.items()}}))"
done
