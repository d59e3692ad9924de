#!/bin/bash
# tools/gpu_r4_marginal.sh — the marginal cost of one more instruction per step of the bounce launch's march loop: builds with 8 / 16
# extra full-rate VALU, 16 scalar, one more 16-byte load (the same cell: L1 hit; the neighbouring line; a random line), C4, one frame
# at a time (the launches alone on the GPU) and two in flight
mkdir -p gpurun_out
for rep in 1 2; do
for lib in voxelraytracing_amd/libvrt.so tools/ab/libvrt_mc_valu8.so tools/ab/libvrt_mc_valu16.so tools/ab/libvrt_mc_salu16.so tools/ab/libvrt_mc_load1.so tools/ab/libvrt_mc_load3.so tools/ab/libvrt_mc_load2.so; do
  for a in "--frames-in-flight 1" ""; do
  VRT_LIB=$lib timeout -k 10 300 python bench.py --mode path --no-cpu-baseline --steps 500 --no-extras --fixed-camera $a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['roofline']['kernels_ms']; print('C4 $lib $a', 'Mrays/s=%.0f' % d['value'], 'ms=%.4f' % d['ms_per_step'], 'bounce launch ms=%.4f' % k['path_bounce_marches'], 'primary=%.4f' % k['path_primary_march'])" | tee -a gpurun_out/r04_marginal.txt
  done
done
done
