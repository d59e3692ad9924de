#!/bin/bash
# tools/r6/gpu_check.sh [tag] — the -m gpu suite (default and experiments build), smoke(), and short bench lines of C2 and C4
# (2 000 / 500 steps, no CPU baseline) -> gpurun_out/r06/<tag>_*.  The quick confirm-on-the-GPU step of round 6.
T=${1:-check}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/${T}_gpu_tests.log 2>&1; tail -2 $O/${T}_gpu_tests.log
VRT_LIB=tools/ab/libvrt_exp.so timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bounce_launch" > $O/${T}_gpu_tests_exp.log 2>&1; tail -1 $O/${T}_gpu_tests_exp.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
b() { n=$1; shift; timeout -k 10 600 python bench.py "$@" > $O/${T}_bench_$n.json 2> $O/${T}_bench_$n.err; python -c "
import json; d=json.loads(open('$O/${T}_bench_$n.json').readline()); r=d['roofline']; print('$n', round(d['value']), 'Mrays/s', round(d['ms_per_step'],4), 'ms', 'frac', r.get('frac'), r.get('pmc_note','')[:60])" || tail -3 $O/${T}_bench_$n.err; }
b c2 --steps 2000 --warmup 200 --no-cpu-baseline
b c4 --mode path --no-cpu-baseline --steps 500 --warmup 50
