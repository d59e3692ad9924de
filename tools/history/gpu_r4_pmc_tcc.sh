#!/bin/bash
# tools/gpu_r4_pmc_tcc.sh — the XCD L2 under a C4 path frame: hits, misses, what goes on to the fabric, per build in $LIBS
mkdir -p gpurun_out
for lib in $LIBS; do
  tag=$(basename $lib .so)
  VRT_LIB=$GRAFT_REPO_ROOT/$lib PMC_GROUPS="4" PMC_EXTRA="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum;TCC_TAG_STALL_sum TCC_BUBBLE_sum TCC_REQ_sum TCC_NORMAL_WRITEBACK_sum" bash tools/pmc.sh r04tcc_$tag --mode path > /dev/null 2>&1
  echo "=== $lib"; grep -A14 "path_bounce_cells" gpurun_out/pmc_r04tcc_$tag/summary.txt | head -16
  grep -l "failed\|rror" gpurun_out/pmc_r04tcc_$tag/*.log | head -3
done
