#!/bin/bash
# tools/r5/gpu_deep.sh — path tests on the deep launch (shape 4), then C4, C4 at 4 spp and C5 under the old cells kernel, the window
# launch and the deep launch
mkdir -p gpurun_out
export VRT_LIB=${VRT_LIB:-tools/ab/libvrt_exp.so}   # the window launch lives in the experiments build (make -C voxelraytracing_amd/csrc experiments)
if [ -z "$SKIP_TESTS" ]; then
VRT_PATH_WINDOW_SHAPE=4 timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "path or c4 or c5" > gpurun_out/r5_deep_tests.log 2>&1 || { tail -40 gpurun_out/r5_deep_tests.log; exit 1; }
tail -2 gpurun_out/r5_deep_tests.log
fi
run() { local label="$2 [$1] ${*:3}"; env $1 timeout -k 10 400 python bench.py --mode path --no-cpu-baseline --no-extras ${@:3} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$label', 'Mrays/s=%.0f' % d['value'], 'ms=%.4f' % d['ms_per_step'])" || exit 1; }
for cs in "VRT_PATH_WINDOW=0" "VRT_PATH_WINDOW=1 VRT_PATH_WINDOW_SHAPE=4" "VRT_PATH_WINDOW=1 VRT_PATH_WINDOW_SHAPE=2"; do
  run $cs C4 --steps 500
  run $cs C4 --steps 500 --frames-in-flight 1
  run $cs C4x4 --spp 4 --steps 300
  run $cs C5 --chunks 32 --width 3840 --height 2160 --spp 16 --steps 20 --warmup 10
done 2>&1 | tee gpurun_out/r5_deep.txt
