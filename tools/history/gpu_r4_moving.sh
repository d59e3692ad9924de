#!/bin/bash
# tools/gpu_r4_moving.sh — the tile order under a moving camera (round 4, review item 4): its tests, then the one-at-a-time orbit with and without
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04m; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "tiles_ordered or longest_tiles" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
for rep in 1 2; do
 for mv in 1 0; do
  VRT_TILE_ORDER_MOVING=$mv timeout -k 10 300 python bench.py --steps 3000 --no-cpu-baseline --frames-in-flight 1 --no-extras 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('moving=$mv orbit 1 in flight: %.1f Mrays/s  %.2f us' % (d['value'], d['ms_per_step']*1e3))" | tee -a $O/ab.txt
  VRT_TILE_ORDER_MOVING=$mv timeout -k 10 300 python bench.py --steps 3000 --no-cpu-baseline --frames-in-flight 1 --no-extras --fixed-camera 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('moving=$mv standing 1 in flight: %.1f Mrays/s  %.2f us' % (d['value'], d['ms_per_step']*1e3))" | tee -a $O/ab.txt
 done
done
