#!/bin/bash
# tools/r5/gpu_moving_kt.sh — kernel-trace stats of the orbit, one frame at a time, with the one-launch moving order
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/kt_mov
VRT_LIB=$GRAFT_REPO_ROOT/tools/ab/libvrt_exp.so VRT_TILE_ORDER_MOVING=${MOV:-1} timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/kt_mov -- python3 $GRAFT_REPO_ROOT/bench.py --steps 500 --warmup 50 --frames-in-flight 1 --no-cpu-baseline --no-extras --settle-seconds 0 > /dev/null 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/kt_mov -name "*kernel_stats.csv" | head -1 | xargs cat | grep "false, 4, false\|tile_order" | sed 's/^"\([^(]*\).*)",/\1 /' | cut -c1-140
