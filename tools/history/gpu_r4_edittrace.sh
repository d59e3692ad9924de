#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04e; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt2 -- python3 $R/tools/edit_cost.py 8 > $O/edit_under_trace.txt 2> $O/kt2.err
f=$(find $O/kt2 -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_trace_edit.csv && wc -l $f
rm -rf $O/kt2
