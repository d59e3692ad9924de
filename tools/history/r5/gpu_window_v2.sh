#!/bin/bash
# tools/r5/gpu_window_v2.sh — path tests (skipping the register-metadata test), C4 rates per window shape, then the probe
mkdir -p gpurun_out
export VRT_LIB=${VRT_LIB:-tools/ab/libvrt_exp.so}   # the window launch lives in the experiments build (make -C voxelraytracing_amd/csrc experiments)
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "path or c4 or c5" > gpurun_out/r5_window_tests.log 2>&1 || { tail -40 gpurun_out/r5_window_tests.log; exit 1; }
tail -2 gpurun_out/r5_window_tests.log
for cs in "VRT_PATH_WINDOW=0" "VRT_PATH_WINDOW=1 VRT_PATH_WINDOW_SHAPE=0" "VRT_PATH_WINDOW=1 VRT_PATH_WINDOW_SHAPE=1" "VRT_PATH_WINDOW=1 VRT_PATH_WINDOW_SHAPE=2" "VRT_PATH_WINDOW=1 VRT_PATH_WINDOW_SHAPE=3"; do
  for a in "" "--frames-in-flight 1"; do
  env $cs timeout -k 10 300 python bench.py --mode path --no-cpu-baseline --steps 500 --no-extras $a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('C4 [$cs] $a', 'Mrays/s=%.0f' % d['value'], 'ms=%.4f' % d['ms_per_step'])" || exit 1
  done
done 2>&1 | tee gpurun_out/r5_window_v2_shapes.txt
SHAPES="0 2 3" bash tools/r5/gpu_window_probe2.sh
