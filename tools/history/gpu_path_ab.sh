#!/bin/bash
# tools/gpu_path_ab.sh — path-trace tests, then bench.py --mode path under the backend's path-kernel switches (same box)
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "path" > gpurun_out/path_tests.log 2>&1 || { tail -30 gpurun_out/path_tests.log; exit 1; }
tail -3 gpurun_out/path_tests.log
for cfg in "0 1" "1 0" "1 1"; do
  set -- $cfg
  for rep in 1 2; do
    VRT_PATH_CELLS=$1 VRT_PATH_SORT=$2 timeout -k 10 300 python bench.py --mode path --no-cpu-baseline --steps 500 --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('cells=$1 sort=$2', 'Mrays/s=%.0f' % d['value'], 'ms=%.4f' % d['ms_per_step'])"
  done
done
