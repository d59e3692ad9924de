#!/bin/bash
# tools/r5/gpu_k5.sh — pools of 320 rays a wave (tools/ab/build_variant.sh k5 "-DVRT_POOL_K=5": 28 waves per CU) against the shipping 256
mkdir -p gpurun_out
run() { local label="$2 [$1] ${*:3}"; VRT_LIB=$1 timeout -k 10 400 python bench.py --mode path --no-cpu-baseline --no-extras ${@:3} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$label', 'Mrays/s=%.0f' % d['value'], 'ms=%.4f' % d['ms_per_step'])" || exit 1; }
for rep in 1 2; do
for lib in voxelraytracing_amd/libvrt.so tools/ab/libvrt_k5.so; do
  run $lib C4 --steps 1000
  run $lib C4 --steps 1000 --frames-in-flight 1
  run $lib C4x4 --spp 4 --steps 300
done
done 2>&1 | tee gpurun_out/r5_k5.txt
for lib in voxelraytracing_amd/libvrt.so tools/ab/libvrt_k5.so; do run $lib C5 --chunks 32 --width 3840 --height 2160 --spp 16 --steps 10 --warmup 4; done 2>&1 | tee -a gpurun_out/r5_k5.txt
