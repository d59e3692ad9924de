#!/bin/bash
# tools/gpu_path_sweep.sh — bench.py --mode path under refill thresholds / frames in flight (same box)
run() { # label, env..., -- args
  label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --mode path --no-cpu-baseline --steps 400 --no-extras $ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$label', 'Mrays/s=%.0f' % d['value'], 'ms=%.4f' % d['ms_per_step'])"
}
export VRT_PATH_SORT=0
for r in 8 16 24 32; do ARGS="" run "refill=$r" VRT_PATH_POOL_REFILL=$r; done
for f in 1 2 3 4; do ARGS="--frames-in-flight $f" run "in_flight=$f" X=1; done
ARGS="" run "again refill=16" VRT_PATH_POOL_REFILL=16
