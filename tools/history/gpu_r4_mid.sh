#!/bin/bash
# tools/gpu_r4_mid.sh — round 4, after the host-side changes: the new GPU tests, the lone edit's cost, the default bench line
# with its operating_point leg
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_api.py tests/test_gpu_multidevice.py tests/test_gpu_operating_point.py -x -q -m gpu > gpurun_out/r04_mid_tests.log 2>&1; tail -3 gpurun_out/r04_mid_tests.log
timeout -k 10 300 python tools/edit_cost.py 8 > gpurun_out/r04_edit_cost.txt 2>&1; grep -v amdgpu.ids gpurun_out/r04_edit_cost.txt
timeout -k 10 400 python bench.py --no-cpu-baseline > gpurun_out/r04_mid_bench.json 2> gpurun_out/r04_mid_bench.err; python -c "
import json; d=json.loads(open('gpurun_out/r04_mid_bench.json').readline()); print('bench', round(d['value']), 'fixed', round(d.get('value_fixed_camera',0)), '1if', round(d.get('value_1_in_flight',0)), 'orbit1', round(d.get('value_1_in_flight_orbit',0)), d['roofline'].get('pmc_note','')[:60]); print(json.dumps(d.get('operating_point'), indent=1))" || tail -5 gpurun_out/r04_mid_bench.err
