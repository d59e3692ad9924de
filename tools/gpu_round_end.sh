#!/bin/bash
# tools/gpu_round_end.sh — what the driver runs at round end, rehearsed on one GPU: the whole -m gpu suite, the experiments
# build's tests, smoke(), the default bench line, and both N > 1 modes with every rank on the one device.
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1; tail -2 gpurun_out/gpu_tests.log
VRT_LIB=tools/ab/libvrt_exp.so timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu > gpurun_out/gpu_tests_exp.log 2>&1; tail -1 gpurun_out/gpu_tests_exp.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout -k 10 300 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; python -c "
import json; d=json.loads(open('gpurun_out/bench_default.json').readline()); print('bench', round(d['value']), d['unit'], 'frac', d['roofline']['frac'], d['roofline'].get('pmc_note',''), 'cpu', round(d['cpu_baseline']['value'],1), 'ranks', d.get('ranks_seen'))"
timeout -k 10 300 python bench.py --gpus 2 --rehearse-on-one-gpu --steps 200 --warmup 50 --no-cpu-baseline > gpurun_out/bench_2ranks.json 2> gpurun_out/bench_2ranks.err; python -c "
import json; d=json.loads(open('gpurun_out/bench_2ranks.json').readline()); print('2 ranks on one GPU:', round(d['value']), 'ranks_seen', d.get('ranks_seen'), 'links', d.get('links'))" || tail -5 gpurun_out/bench_2ranks.err
timeout -k 10 300 python bench.py --gpus 2 --single-process --rehearse-on-one-gpu --steps 200 --warmup 50 --no-cpu-baseline > gpurun_out/bench_2dev.json 2> gpurun_out/bench_2dev.err; python -c "
import json; d=json.loads(open('gpurun_out/bench_2dev.json').readline()); print('one context over 2 devices (one GPU):', round(d['value']), 'links', d.get('links'))" || tail -5 gpurun_out/bench_2dev.err
