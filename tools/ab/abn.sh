#!/bin/bash
# tools/ab/abn.sh "<bench args>" lib1 lib2 ... — same-box comparison of several builds, two rounds, 1 and 2 frames in flight
ARGS=$1; shift
for rep in 1 2; do
  for lib in "$@"; do
    for fif in 2 1; do
      VRT_LIB=$lib python bench.py --steps 3000 --no-cpu-baseline --no-extras --fixed-camera --frames-in-flight $fif $ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$lib', 'in_flight=$fif', 'ms_per_step=%.5f' % d['ms_per_step'], 'launch_ms=%.5f' % d['roofline']['avg_launch_ms'])"
    done
  done
done
