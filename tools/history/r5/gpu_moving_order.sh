#!/bin/bash
# tools/r5/gpu_moving_order.sh — the moving-camera tile order (experiments build): its tests, then bench.py one frame at a time with the
# order off (0), as one launch over blocks behind the frame (1), made beside the next frame for the one after it (2), as round 4's six launches (6)
mkdir -p gpurun_out
VRT_LIB=tools/ab/libvrt_exp.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "moving_camera or block_order or longest_tiles" > gpurun_out/r5_moving_tests_exp.log 2>&1 || { tail -30 gpurun_out/r5_moving_tests_exp.log; exit 1; }
tail -1 gpurun_out/r5_moving_tests_exp.log
for rep in 1 2; do
for m in 0 1 2 6; do
  lib=tools/ab/libvrt_exp.so
  VRT_LIB=$lib VRT_TILE_ORDER_MOVING=$m timeout -k 10 300 python bench.py --steps 3000 --warmup 100 --frames-in-flight 1 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('orbit, one frame at a time, moving order $m:', round(d['value']), 'Mrays/s', round(d['ms_per_step']*1e3,2), 'us per frame')" || exit 1
done
done 2>&1 | tee gpurun_out/r5_moving_order.txt
VRT_LIB=tools/ab/libvrt_exp.so VRT_TILE_ORDER_MOVING=1 timeout -k 10 300 python bench.py --steps 2000 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('two in flight (the order is not used there), moving order 1:', round(d['value']))" | tee -a gpurun_out/r5_moving_order.txt
