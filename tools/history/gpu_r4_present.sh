#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04o; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_api.py tests/test_gpu_reference_wgsl.py tests/test_gpu_operating_point.py -x -q -k "present or any_size or operating or wgsl" > $O/tests_present.txt 2>&1; tail -2 $O/tests_present.txt
python tools/op_point.py 2 1 2>&1 | grep -v amdgpu.ids | tee $O/op2.txt
python tools/op_point.py 2 0 2>&1 | grep -v amdgpu.ids | tee -a $O/op2.txt
python tools/op_point.py 1 1 2>&1 | grep -v amdgpu.ids | tee -a $O/op2.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/tools/op_point.py 1 1 > $O/op_under_rocprof.txt 2> $O/kt.err
f=$(find $O/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_op_1.csv && head -3 $f | cut -c1-200
rm -rf $O/kt
