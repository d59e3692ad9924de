#!/bin/bash
# tools/r5/gpu_row_order.sh — the moving view's order once more, cheaper: made once in K camera steps (VRT_TILE_ORDER_EVERY) and, with
# (history: VRT_TILE_ORDER_EVERY and the row-wise radius codes existed in the experiments build of commit 2325149+ only; the kept order that came of the sweep is the default now — tools/r5/gpu_kept_order.sh)
# VRT_TILE_ORDER_RADIUS >= 100, by block ROWS (a turn of the view leaves a row's cost alone).  Experiments build, one frame at a time.
mkdir -p gpurun_out
run() {   # moving-mode every radius
  VRT_LIB=tools/ab/libvrt_exp.so VRT_TILE_ORDER_MOVING=$1 VRT_TILE_ORDER_EVERY=$2 VRT_TILE_ORDER_RADIUS=$3 timeout -k 10 300 python bench.py --steps 3000 --warmup 100 --frames-in-flight 1 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('moving order $1 every $2 radius $3:', round(d['value']), 'Mrays/s', round(d['ms_per_step']*1e3,2), 'us per frame')" || exit 1
}
{
run 0 1 3
run 1 8 4
run 1 8 5
run 1 8 6
run 1 12 6
run 1 16 8
run 1 12 8
run 1 6 3
run 1 6 4
run 1 10 5
} 2>&1 | tee gpurun_out/r5_row_order.txt
