#!/bin/bash
# tools/sanitize_cpu.sh — AddressSanitizer + UBSan over the CPU-side native code (the C++ host mirror and the C oracle):
# builds both with -fsanitize=address,undefined into a scratch directory, swaps them in for the CPU test suite and
# restores the regular builds.  (GPU sanitizers are not available on this pool; the kernels are covered by parity tests.)
set -e
cd "$(dirname "$0")/.."
T=$(mktemp -d)
g++ -O1 -g -std=c++17 -fPIC -shared -pthread -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off \
    -o $T/libvrt_host.so voxelraytracing_amd/csrc/host/host_capi.cpp
gcc -O1 -g -std=c99 -ffp-contract=off -fno-fast-math -fopenmp -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -shared \
    -o $T/libvrt_oracle.so oracle/vrt_oracle.c -lm
cp voxelraytracing_amd/libvrt_host.so $T/host.bak; cp oracle/libvrt_oracle.so $T/oracle.bak
restore() { cp $T/host.bak voxelraytracing_amd/libvrt_host.so; cp $T/oracle.bak oracle/libvrt_oracle.so; touch oracle/libvrt_oracle.so; rm -rf $T; }
trap restore EXIT
cp $T/libvrt_host.so voxelraytracing_amd/libvrt_host.so; cp $T/libvrt_oracle.so oracle/libvrt_oracle.so; touch oracle/libvrt_oracle.so
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 OMP_NUM_THREADS=4 \
    python -m pytest tests/test_oracle_kat.py tests/test_golden.py tests/test_host_world.py tests/test_regionfile.py tests/test_netmsg.py tests/test_formats_fuzz.py \
    -x -q -m "not gpu"
