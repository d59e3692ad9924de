#!/bin/bash
# the chunk rebuild as a wave per split cell: its tests, the probe, the costs
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04c; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_accel.py tests/test_gpu_operating_point.py tests/test_gpu_api.py -x -q > $O/tests.txt 2>&1; tail -3 $O/tests.txt
VRT_LIB=$R/tools/ab/libvrt_chunkdbg.so timeout -k 10 300 python tools/chunk_probe.py 8 2>&1 | grep -v amdgpu.ids | tee $O/chunk_probe.txt
python tools/edit_cost.py 8 2>&1 | grep -v amdgpu.ids > $O/edit_cost.txt; python tools/edit_cost.py 32 2>&1 | grep -v amdgpu.ids >> $O/edit_cost.txt; cat $O/edit_cost.txt
python tools/stream_cost.py 2>&1 | grep -v amdgpu.ids > $O/stream_cost.txt; tail -6 $O/stream_cost.txt
