#!/bin/bash
# tools/r6/gpu_loop_ab.sh A.so B.so [reps] — same-box A/B of two builds on the headline loop: C2 with two frames in flight, the standing
# camera, one frame at a time (lone launch) -> stdout
R=$GRAFT_REPO_ROOT; cd $R
A=$1; B=$2; N=${3:-3}
for rep in $(seq $N); do
  for lib in $A $B; do
    VRT_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --steps 3000 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$lib', 'C2 %.0f' % d['value'], 'fixed %.0f' % d['value_fixed_camera'], '1-in-flight %.0f orbit %.0f' % (d['value_1_in_flight'], d['value_1_in_flight_orbit']), 'lone launch %.2f us' % (d['avg_launch_ms_1_in_flight']*1e3))"
  done
done
