#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04o; mkdir -p $O; cd $R
python tools/op_point.py 2 1 2>&1 | grep -v amdgpu.ids | tee $O/op.txt
python tools/op_point.py 2 0 2>&1 | grep -v amdgpu.ids | tee -a $O/op.txt
python tools/op_point.py 1 1 2>&1 | grep -v amdgpu.ids | tee -a $O/op.txt
python tools/op_point.py 1 0 2>&1 | grep -v amdgpu.ids | tee -a $O/op.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/tools/op_point.py 2 1 > $O/op_under_rocprof.txt 2> $O/kt.err
f=$(find $O/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_op.csv && head -6 $f | cut -c1-160
rm -rf $O/kt
