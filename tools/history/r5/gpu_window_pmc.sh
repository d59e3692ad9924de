#!/bin/bash
# tools/r5/gpu_window_pmc.sh — the bounce launch of C4 under rocprofv3: kernel-trace stats one frame at a time, then the PMC groups of
# profiles/r04_pmc_summary_path8.txt, for the shipping kernel (VRT_PATH_WINDOW=0) and the window / deep launches
mkdir -p gpurun_out
export VRT_LIB=${VRT_LIB:-tools/ab/libvrt_exp.so}   # the window launch lives in the experiments build (make -C voxelraytracing_amd/csrc experiments)
R=$GRAFT_REPO_ROOT
CASES="VRT_PATH_WINDOW=0;VRT_PATH_WINDOW=1 VRT_PATH_WINDOW_SHAPE=0;VRT_PATH_WINDOW=1 VRT_PATH_WINDOW_SHAPE=2;VRT_PATH_WINDOW=1 VRT_PATH_WINDOW_SHAPE=3;VRT_PATH_WINDOW=1 VRT_PATH_WINDOW_SHAPE=4" bash tools/gpu_kt.sh > gpurun_out/r5_window_kt.txt 2>&1 || { tail gpurun_out/r5_window_kt.txt; exit 1; }
cat gpurun_out/r5_window_kt.txt
for cs in "VRT_PATH_WINDOW=0" "VRT_PATH_WINDOW=1 VRT_PATH_WINDOW_SHAPE=2" "VRT_PATH_WINDOW=1 VRT_PATH_WINDOW_SHAPE=3" "VRT_PATH_WINDOW=1 VRT_PATH_WINDOW_SHAPE=4"; do
  tag=$(echo $cs | tr '= ' '__')
  ( export $cs; cd $R; PMC_GROUPS="1 2 7 8" bash tools/pmc.sh $tag --mode path > /dev/null 2>&1 )
  echo "== $cs"; grep -A40 "path_bounce_window_kernel\|path_bounce_cells_kernel" gpurun_out/pmc_$tag/summary.txt | grep -v "^==.*path_bounce_kernel" | head -32
done > gpurun_out/r5_window_pmc.txt 2>&1
tail -5 gpurun_out/r5_window_pmc.txt
