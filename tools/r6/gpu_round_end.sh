#!/bin/bash
# tools/r6/gpu_round_end.sh — what the driver runs at round end, rehearsed on one GPU: the whole -m gpu suite, the experiments build's
# tests, smoke(), the driver's N = 1 command, and its N > 1 command (torch.distributed.run, one rank per GPU) with both ranks on the one device.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/end_gpu_tests.log 2>&1; tail -2 $O/end_gpu_tests.log
VRT_LIB=tools/ab/libvrt_exp.so timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bounce_launch" > $O/end_gpu_tests_exp.log 2>&1; tail -1 $O/end_gpu_tests_exp.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/end_bench_driver.json 2> $O/end_bench_driver.err; python -c "
import json; d=json.loads(open('$O/end_bench_driver.json').readline()); c=d['config']; print('driver N=1:', round(d['value']), d['unit'], 'ms_per_step', round(d['ms_per_step'],4), 'steps', d['steps'], 'config.steps_timed', c['steps_timed'], 'config.value_requested_steps', round(c['value_requested_steps']), 'frac', round(d['roofline']['frac'],3), 'cpu', round(d['cpu_baseline']['value'],1), d['cpu_baseline']['cores'])"
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 --rehearse-on-one-gpu > $O/end_bench_torchrun2.json 2> $O/end_bench_torchrun2.err; python -c "
import json; d=[json.loads(l) for l in open('$O/end_bench_torchrun2.json') if l.startswith('{')][0]; c=d['config']; print('torchrun N=2 (both ranks on one GPU):', round(d['value']), 'ranks_seen', d['ranks_seen'], 'expected_speedup', round(c['expected_speedup'],2), c['predicted_faster_mode'], 'degenerate', c['degenerate_scaling_point'], 'root_weight', c['root_weight'])" || tail -5 $O/end_bench_torchrun2.err
