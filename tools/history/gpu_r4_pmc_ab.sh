#!/bin/bash
# tools/gpu_r4_pmc_ab.sh "<lib> <lib> ..." — PMC groups 1, 2, 7 (instructions, waits, L1) of a C4 path frame per build
mkdir -p gpurun_out
for lib in $LIBS; do
  tag=$(basename $lib .so)
  VRT_LIB=$GRAFT_REPO_ROOT/$lib PMC_GROUPS="1 2 7" bash tools/pmc.sh r04_$tag --mode path > /dev/null 2>&1
  echo "=== $lib"; grep -A30 "path_bounce_cells" gpurun_out/pmc_r04_$tag/summary.txt | grep -E "path_bounce_cells|SQ_INSTS_VALU |SQ_INSTS_SALU|SQ_INSTS_VMEM_RD|SQ_INSTS_LDS|SQ_WAVE_CYCLES|SQ_BUSY_CYCLES|SQ_WAIT_INST_ANY|SQ_ACTIVE_INST_VALU|TCP_|SQ_WAVES|SQ_THREAD" 
done
