#!/bin/bash
# tools/r5/gpu_window_probe.sh — tools/window_probe.py per window shape (the probe build), then the plain build's C4 rates per shape
mkdir -p gpurun_out
export VRT_LIB=${VRT_LIB:-tools/ab/libvrt_exp.so}   # the window launch lives in the experiments build (make -C voxelraytracing_amd/csrc experiments)
for sh in 0 1 2 3; do
  echo "== shape $sh"
  VRT_LIB=tools/ab/libvrt_windbg.so VRT_PATH_WINDOW_SHAPE=$sh timeout -k 10 200 python tools/window_probe.py || exit 1
done 2>&1 | tee gpurun_out/r5_window_probe.txt
for cs in "VRT_PATH_WINDOW=0" "VRT_PATH_WINDOW=1 VRT_PATH_WINDOW_SHAPE=0" "VRT_PATH_WINDOW=1 VRT_PATH_WINDOW_SHAPE=1" "VRT_PATH_WINDOW=1 VRT_PATH_WINDOW_SHAPE=2" "VRT_PATH_WINDOW=1 VRT_PATH_WINDOW_SHAPE=3"; do
  for a in "" "--frames-in-flight 1"; do
  env $cs timeout -k 10 300 python bench.py --mode path --no-cpu-baseline --steps 500 --no-extras $a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('C4 [$cs] $a', 'Mrays/s=%.0f' % d['value'], 'ms=%.4f' % d['ms_per_step'])" || exit 1
  done
done 2>&1 | tee gpurun_out/r5_window_shapes.txt
