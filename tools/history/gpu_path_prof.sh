#!/bin/bash
# tools/gpu_path_prof.sh — path trace (C4): kernel-trace stats one frame at a time, then PMC groups of the bounce launch
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_path1 -- python3 $R/bench.py --mode path --steps 100 --warmup 10 --no-cpu-baseline --fixed-camera --no-extras --settle-seconds 0 --frames-in-flight 1 > $R/gpurun_out/kt_path1.log 2>&1
find $R/gpurun_out/kt_path1 -name "*kernel_stats.csv" | sort | tail -1 | xargs cat | cut -c1-160
cd $R
PMC_GROUPS="${PMC_GROUPS:-1 2 7}" bash tools/pmc.sh path_cells --mode path > /dev/null 2>&1
grep -A30 "path_bounce_cells" gpurun_out/pmc_path_cells/summary.txt
