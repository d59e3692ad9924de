#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04m; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for mv in 1 0; do
  export VRT_TILE_ORDER_MOVING=$mv
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$mv -- python3 $R/bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-extras --frames-in-flight 1 > $O/under_$mv.json 2> $O/kt_$mv.err
  f=$(find $O/kt_$mv -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_moving_$mv.csv && head -9 $f | cut -c1-170
  rm -rf $O/kt_$mv
done
