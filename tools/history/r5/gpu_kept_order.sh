#!/bin/bash
# tools/r5/gpu_kept_order.sh — the kept block order of a moving view (the default since round 5): its tests on both builds, then
# bench.py one frame at a time with it (default), without (VRT_TILE_ORDER_MOVING=0), and the default line (two in flight: not used)
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "moving_camera or block_order or longest_tiles or kept_block" > gpurun_out/r5_kept_tests.log 2>&1 || { tail -30 gpurun_out/r5_kept_tests.log; exit 1; }
tail -1 gpurun_out/r5_kept_tests.log
VRT_LIB=tools/ab/libvrt_exp.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "moving_camera or block_order or longest_tiles or kept_block" > gpurun_out/r5_kept_tests_exp.log 2>&1 || { tail -30 gpurun_out/r5_kept_tests_exp.log; exit 1; }
tail -1 gpurun_out/r5_kept_tests_exp.log
run() {
  VRT_TILE_ORDER_MOVING=$1 VRT_TILE_ORDER_RADIUS=$2 timeout -k 10 300 python bench.py --steps 3000 --warmup 100 --frames-in-flight 1 --no-cpu-baseline --no-extras $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('one frame at a time, moving order $1 radius $2 $3:', round(d['value']), 'Mrays/s', round(d['ms_per_step']*1e3,2), 'us per frame')" || exit 1
}
{
run 0 5
run 1 5
run 1 4
run 1 6
run 0 5 "--width 3840 --height 2160"
run 1 5 "--width 3840 --height 2160"
run 0 5 "--mode primary"
run 1 5 "--mode primary"
} 2>&1 | tee gpurun_out/r5_kept_order.txt
timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('default line:', round(d['value']), 'standing', round(d['value_fixed_camera']), 'one at a time', round(d['value_1_in_flight']), 'orbit', round(d['value_1_in_flight_orbit']))" | tee -a gpurun_out/r5_kept_order.txt
