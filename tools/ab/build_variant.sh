#!/bin/bash
# tools/ab/build_variant.sh NAME "-DFLAG ..." — builds tools/ab/libvrt_NAME.so from the working tree with extra compiler flags
# (same-box A/B experiments: tools/ab/ab.sh).  Objects go to /tmp, the tree's own build is untouched.
set -e
NAME=$1; FLAGS=$2
SRC=$(cd "$(dirname "$0")/../../voxelraytracing_amd/csrc" && pwd)
OUT=$(cd "$(dirname "$0")" && pwd)
T=/tmp/vrt_ab_$NAME; mkdir -p $T
HIPFLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -fno-slp-vectorize -Wno-unused-value $FLAGS"
FILES="vrt_kernels vrt_path vrt_accel vrt_frames vrt_order vrt_uploads vrt_present vrt_group"
EXP="vrt_exp_register vrt_path_window"   # (csrc/experiments/: every variant is an experiments build)
for f in $FILES; do /opt/rocm/bin/hipcc $HIPFLAGS -c -o $T/$f.o $SRC/$f.hip & done
for f in $EXP; do /opt/rocm/bin/hipcc $HIPFLAGS -c -o $T/$f.o $SRC/experiments/$f.hip & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o $OUT/libvrt_$NAME.so $(for f in $FILES $EXP; do echo $T/$f.o; done)
echo built $OUT/libvrt_$NAME.so
