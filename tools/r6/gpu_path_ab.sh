#!/bin/bash
# tools/r6/gpu_path_ab.sh A.so B.so — same-box A/B of two builds on the path trace: C4 (two in flight / one at a time / the bounce launch alone),
# C4 at 4 spp, C5; then the path trace's parity tests on B.
R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2; do
  for lib in $1 $2; do
    c4=$(VRT_LIB=$lib timeout -k 10 300 python bench.py --mode path --no-cpu-baseline --steps 500 --warmup 50 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('C4 %.0f (1 in flight %.0f, bounce launch %.1f us)' % (d['value'], d['value_1_in_flight'], d['avg_bounce_launches_ms_1_in_flight']*1e3))")
    c5=$(VRT_LIB=$lib timeout -k 10 300 python bench.py --mode path --chunks 32 --width 3840 --height 2160 --spp 16 --steps 12 --warmup 4 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('C5 %.0f (%.2f ms)' % (d['value'], d['ms_per_step']))")
    echo "$lib $c4 | $c5"
  done
done
VRT_LIB=$2 timeout -k 10 600 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -x -q -m gpu -k "path or c4 or c5 or bounce or samples or pool" 2>&1 | tail -2
