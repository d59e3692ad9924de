#!/bin/bash
# Same-box A/B of two builds of libvrt.so: tools/ab/ab.sh <libA> <libB> [extra bench args]; prints ms_per_step per run.
A=$1; B=$2; shift 2
for rep in 1 2 3; do
  for lib in "$A" "$B"; do
    for fif in 2 1; do
      VRT_LIB=$lib python bench.py --steps 3000 --no-cpu-baseline --frames-in-flight $fif "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$lib', 'in_flight=$fif', 'ms_per_step=%.5f' % d['ms_per_step'], 'launch_ms=%.5f' % d['roofline']['avg_launch_ms'])"
    done
  done
done
