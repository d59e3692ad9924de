#!/bin/bash
# tools/gpu_r4_ab.sh — C4 path bench (two in flight / one at a time) of the builds in $LIBS, two rounds; C5 of $LIBS5
mkdir -p gpurun_out
for rep in 1 2; do
for lib in $LIBS; do
  for a in "" "--frames-in-flight 1"; do
  VRT_LIB=$lib timeout -k 10 300 python bench.py --mode path --no-cpu-baseline --steps 500 --no-extras $a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('C4 $lib $a', 'Mrays/s=%.0f' % d['value'], 'ms=%.4f' % d['ms_per_step'])" | tee -a gpurun_out/r04_ab.txt
  done
done
done
for lib in $LIBS5; do
  VRT_LIB=$lib timeout -k 10 300 python bench.py --mode path --chunks 32 --width 3840 --height 2160 --spp 16 --steps 20 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('C5 $lib', 'Mrays/s=%.0f' % d['value'], 'ms=%.4f' % d['ms_per_step'])" | tee -a gpurun_out/r04_ab.txt
done
