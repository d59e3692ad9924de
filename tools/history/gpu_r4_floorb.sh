#!/bin/bash
# A/B: coordinates as biased float bits, floors as adds under round-to-minus-infinity (-DVRT_FLOOR_BIASED) against the tree's build
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04f; mkdir -p $O; cd $R
VRT_LIB=$R/tools/ab/libvrt_floorb.so timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_wgsl.py tests/test_gpu_accel.py -x -q > $O/tests_floorb.txt 2>&1; tail -3 $O/tests_floorb.txt
bash tools/ab/ab.sh $R/voxelraytracing_amd/libvrt.so $R/tools/ab/libvrt_floorb.so --no-extras 2>&1 | tee $O/ab.txt
