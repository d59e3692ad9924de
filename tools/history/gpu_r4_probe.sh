#!/bin/bash
# tools/gpu_r4_probe.sh — round 4, first look at the bounce launch: what its lookups find (probe build) and how its time
# depends on the waves resident per CU (experiments build, LDS claimed beyond the pools)
mkdir -p gpurun_out
VRT_LIB=tools/ab/libvrt_celldbg.so timeout -k 10 300 python tools/cells_probe.py > gpurun_out/r04_cells_probe.txt 2>&1; tail -12 gpurun_out/r04_cells_probe.txt
for rep in 1 2; do
for pad in 0 4000 8000 13000 21000 34000 45000; do
  for a in "" "--frames-in-flight 1"; do
  VRT_LIB=tools/ab/libvrt_exp.so VRT_PATH_LDS_PAD=$pad timeout -k 10 300 python bench.py --mode path --no-cpu-baseline --steps 500 --no-extras $a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('C4 lds_pad=$pad $a', 'Mrays/s=%.0f' % d['value'], 'ms=%.4f' % d['ms_per_step'])" | tee -a gpurun_out/r04_occupancy_sweep.txt
  done
done
done
