#!/bin/bash
# tools/gpu_path_prof.sh — path trace: kernel-trace stats one frame at a time, then PMC groups
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
for lib in "" tools/ab/libvrt_w7.so; do
  for rep in 1 2; do
  VRT_LIB=$lib VRT_PATH_SORT=0 timeout -k 10 300 python bench.py --mode path --no-cpu-baseline --steps 500 --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('lib=$lib', 'Mrays/s=%.0f' % d['value'], 'ms=%.4f' % d['ms_per_step'])"
  done
done
cd /tmp && export TMPDIR=/tmp
VRT_PATH_SORT=0 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_path1 -- python3 $R/bench.py --mode path --steps 100 --warmup 10 --no-cpu-baseline --fixed-camera --no-extras --settle-seconds 0 --frames-in-flight 1 > $R/gpurun_out/kt_path1.log 2>&1
find $R/gpurun_out/kt_path1 -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-160
cd $R
VRT_PATH_SORT=0 PMC_GROUPS="1 2 7" bash tools/pmc.sh path_cells --mode path > /dev/null 2>&1
cat gpurun_out/pmc_path_cells/summary.txt | grep -A30 "path_bounce_cells" | head -40
