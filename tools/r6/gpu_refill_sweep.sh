#!/bin/bash
# tools/r6/gpu_refill_sweep.sh — the bounce launch's refill threshold (VRT_PATH_POOL_REFILL: a wave goes back to its pool once this many lanes
# are idle) re-swept after the march loop became an asm statement: C4 and C5, the shipped value among its neighbours.
cd $GRAFT_REPO_ROOT
for r in 4 8 12 16 24 32; do
  c4=$(VRT_PATH_POOL_REFILL=$r timeout -k 10 300 python bench.py --mode path --no-cpu-baseline --steps 500 --warmup 50 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('C4 %.0f (1 in flight %.0f, bounce launch %.1f us)' % (d['value'], d['value_1_in_flight'], d['avg_bounce_launches_ms_1_in_flight']*1e3))")
  c5=$(VRT_PATH_POOL_REFILL=$r timeout -k 10 300 python bench.py --mode path --chunks 32 --width 3840 --height 2160 --spp 16 --steps 12 --warmup 4 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('C5 %.0f (%.2f ms)' % (d['value'], d['ms_per_step']))")
  echo "refill at $r idle lanes: $c4 | $c5"
done
