#!/bin/bash
# tools/gpu_path_quick.sh — the path-trace tests, then bench.py --mode path (C4) under the switches given as "VAR=val ..." lines in $CASES
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "path or operating" > gpurun_out/path_tests.log 2>&1 || { tail -30 gpurun_out/path_tests.log; exit 1; }
tail -2 gpurun_out/path_tests.log
IFS=';' read -ra CS <<< "${CASES:-X=1}"
for rep in 1 2; do
for cs in "${CS[@]}"; do
  for a in "" "--frames-in-flight 1"; do
  env $cs timeout -k 10 300 python bench.py --mode path --no-cpu-baseline --steps 500 --no-extras $a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('C4 [$cs] $a', 'Mrays/s=%.0f' % d['value'], 'ms=%.4f' % d['ms_per_step'])"
  done
done
done
