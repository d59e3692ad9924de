#!/bin/bash
# tools/gpu_ab_pmc.sh "<libA> <libB> ..." — C4 path bench per build (same box, two rounds) and the L1 counters of the bounce launch
for rep in 1 2; do for lib in $LIBS; do
  VRT_LIB=$lib timeout -k 10 300 python bench.py --mode path --no-cpu-baseline --steps 500 --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$lib', 'Mrays/s=%.0f' % d['value'], 'ms=%.4f' % d['ms_per_step'])"
done; done
for lib in $LIBS; do
  VRT_LIB=$lib PMC_GROUPS="7" bash tools/pmc.sh ab_$(basename $lib .so) --mode path > /dev/null 2>&1
  echo "== $lib"; grep -A6 "path_bounce_cells" gpurun_out/pmc_ab_$(basename $lib .so)/summary.txt | grep TCP
done
