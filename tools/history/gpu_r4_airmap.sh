#!/bin/bash
# tools/gpu_r4_airmap.sh — the air map in LDS (round 4): path-trace parity tests on the default build, then C4 / C5 A/B against
# the builds without the map and with 4 waves per workgroup
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "path or accel or c4 or c5" > gpurun_out/r04_airmap_tests.log 2>&1; tail -3 gpurun_out/r04_airmap_tests.log
for rep in 1 2; do
for lib in voxelraytracing_amd/libvrt.so tools/ab/libvrt_w7noair.so tools/ab/libvrt_w4air.so tools/ab/libvrt_w4noair.so; do
  for a in "" "--frames-in-flight 1"; do
  VRT_LIB=$lib timeout -k 10 300 python bench.py --mode path --no-cpu-baseline --steps 500 --no-extras $a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('C4 $lib $a', 'Mrays/s=%.0f' % d['value'], 'ms=%.4f' % d['ms_per_step'])" | tee -a gpurun_out/r04_airmap_ab.txt
  done
done
done
for lib in voxelraytracing_amd/libvrt.so tools/ab/libvrt_w7noair.so tools/ab/libvrt_w4noair.so; do
  VRT_LIB=$lib timeout -k 10 300 python bench.py --mode path --chunks 32 --width 3840 --height 2160 --spp 16 --steps 20 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('C5 $lib', 'Mrays/s=%.0f' % d['value'], 'ms=%.4f' % d['ms_per_step'])" | tee -a gpurun_out/r04_airmap_ab.txt
done
