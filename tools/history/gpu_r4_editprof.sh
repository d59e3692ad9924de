#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04e; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/tools/edit_cost.py 8 > $O/edit_under_rocprof.txt 2> $O/kt.err
f=$(find $O/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_edit.csv && head -12 $f | cut -c1-200
rm -rf $O/kt
