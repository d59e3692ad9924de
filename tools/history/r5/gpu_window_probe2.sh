#!/bin/bash
mkdir -p gpurun_out
for sh in ${SHAPES:-0 2 3}; do
  echo "== shape $sh"
  VRT_LIB=tools/ab/libvrt_windbg.so VRT_PATH_WINDOW_SHAPE=$sh timeout -k 10 200 python tools/window_probe.py 2>/dev/null | head -12 || exit 1
done 2>&1 | tee gpurun_out/r5_window_probe2.txt
