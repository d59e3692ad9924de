#!/bin/bash
# tools/r6/gpu_step1.sh — round 6, first measurement of the fused presentation and the bounce launch's banded logarithm / square
# roots: the -m gpu suite, the default bench line (operating_point: the client's frame with the window's image stored by the frame's
# own launch), C4 / C5, the VALU counters of the path frame, the extended issue-rate table.  -> gpurun_out/r06/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
bash tools/r6/gpu_check.sh step1 || exit 1
timeout -k 10 400 python bench.py > $O/step1_bench_default.json 2> $O/step1_bench_default.err; python -c "
import json; d=json.loads(open('$O/step1_bench_default.json').readline()); print('default', round(d['value']), d['steps_timed']); op=d['operating_point']; print({k:(round(v['ms_per_frame']*1e3,1), round(v['value'])) for k,v in op.items() if isinstance(v,dict)})"
timeout -k 10 300 python bench.py --mode path --chunks 32 --width 3840 --height 2160 --spp 16 --steps 20 --warmup 20 --no-cpu-baseline > $O/step1_bench_c5.json 2> $O/step1_bench_c5.err; python -c "
import json; d=json.loads(open('$O/step1_bench_c5.json').readline()); print('c5', round(d['value']), round(d['ms_per_step'],3))"
PMC_GROUPS="1 2" bash tools/pmc.sh r06_step1_path8 --mode path > /dev/null 2>&1; cp gpurun_out/pmc_r06_step1_path8/summary.txt $O/step1_pmc_path8.txt; rm -rf gpurun_out/pmc_r06_step1_path8/g*/; grep -E "SQ_INSTS_VALU|SQ_INSTS_SALU|SQ_WAIT_ANY|SQ_WAVE_CYCLES|kernel" $O/step1_pmc_path8.txt | head -20
./tools/valu_rates > $O/valu_issue_rates.txt 2>&1; grep -E "mul_lo|mul_hi|mad_u64|mul_u24|div_scale|div_fmas|pk_fma|cvt_f32_u32|bpermute" $O/valu_issue_rates.txt
