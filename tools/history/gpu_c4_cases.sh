#!/bin/bash
# tools/gpu_c4_cases.sh — bench.py --mode path (C4) under the switches given as "VAR=val ...;VAR=val ..." in $CASES, twice each
IFS=';' read -ra CS <<< "${CASES:-X=1}"
for rep in 1 2; do
for cs in "${CS[@]}"; do
  env $cs timeout -k 10 300 python bench.py --mode path --no-cpu-baseline --steps 500 --no-extras $BENCH_ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('C4 [$cs]', 'Mrays/s=%.0f' % d['value'], 'ms=%.4f' % d['ms_per_step'])" || exit 1
done
done
