#!/bin/bash
# tools/gpu_kt.sh — rocprofv3 kernel-trace stats of bench.py --mode path (C4, one frame at a time, standing camera) under the
# environment given as "VAR=val ..." lines in $CASES (the variables are exported: rocprofv3 wants the program itself after --)
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
IFS=';' read -ra CS <<< "${CASES:-X=1}"
i=0
for cs in "${CS[@]}"; do
  i=$((i+1))
  ( export $cs; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_$i -- python3 $R/bench.py --mode path --steps 100 --warmup 10 --no-cpu-baseline --fixed-camera --no-extras --settle-seconds 0 --frames-in-flight ${FIF:-1} > $R/gpurun_out/kt_$i.log 2>&1 ) || exit 1
  echo "== $cs"
  find $R/gpurun_out/kt_$i -name "*kernel_stats.csv" | sort | tail -1 | xargs cat | cut -c1-150 | grep -v "rocclr\|accel_\|upload_" 
done
