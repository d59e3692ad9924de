#!/bin/bash
# tools/gpu_c5.sh — C5 (3840x2160, 16 spp, 32^3 world, 4 bounces) on one GPU under the given switches ("VAR=val;VAR=val")
IFS=';' read -ra CS <<< "${CASES:-X=1}"
for cs in "${CS[@]}"; do
  env $cs timeout -k 10 500 python bench.py --mode path --chunks 32 --width 3840 --height 2160 --spp 16 --steps 20 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('C5 [$cs]', 'Mrays/s=%.0f' % d['value'], 'ms=%.3f' % d['ms_per_step'])"
done
