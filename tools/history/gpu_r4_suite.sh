#!/bin/bash
# tools/gpu_r4_suite.sh — the whole -m gpu suite, then the default bench line (no CPU baseline)
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r04_gpu_tests.log 2>&1; tail -3 gpurun_out/r04_gpu_tests.log
timeout -k 10 400 python bench.py --no-cpu-baseline > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.err; python -c "
import json; d=json.loads(open('gpurun_out/r04_bench.json').readline()); print('bench', round(d['value']), 'ms', round(d['ms_per_step'],4), 'fixed', round(d.get('value_fixed_camera',0)), '1if', round(d.get('value_1_in_flight',0)), 'orbit1', round(d.get('value_1_in_flight_orbit',0)), 'op', {k: round(v['value']) for k, v in d.get('operating_point', {}).items() if isinstance(v, dict)})" || tail -5 gpurun_out/r04_bench.err
