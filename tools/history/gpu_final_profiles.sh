#!/bin/bash
# tools/gpu_final_profiles.sh A|B — the round's evidence.  A: rocprofv3 kernel stats (standing camera 2 / 1 in flight, the
# orbit, the path trace) and the PMC passes -> gpurun_out/r03/.  B (after profiles/traffic_latest.json was derived from A's
# summary for this very build): the bench lines.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
if [ "$1" = "A" ]; then
  cd /tmp && export TMPDIR=/tmp
  ks() { # name, bench args...
    n=$1; shift
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$n -- python3 $R/bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-extras "$@" > $O/bench_under_rocprof_$n.json 2> $O/kt_$n.err
    f=$(find $O/kt_$n -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_$n.csv && head -3 $f | cut -c1-150
  }
  ks standing_2_in_flight --fixed-camera
  ks standing_1_in_flight --fixed-camera --frames-in-flight 1
  ks orbit_2_in_flight
  ks orbit_1_in_flight --frames-in-flight 1
  ks path_2_in_flight --mode path --fixed-camera --steps 500
  ks path_1_in_flight --mode path --fixed-camera --steps 300 --frames-in-flight 1
  cd $R
  bash tools/pmc.sh r03final > /dev/null 2>&1; cp gpurun_out/pmc_r03final/summary.txt $O/pmc_summary.txt
  PMC_GROUPS="1 2 3 7" bash tools/pmc.sh r03path --mode path > /dev/null 2>&1; cp gpurun_out/pmc_r03path/summary.txt $O/path_pmc_summary.txt
  ./tools/valu_rates > $O/valu_issue_rates.txt 2>&1
  grep -A3 "primary_shadow_wave_kernel<0, false, false, 4, false>" $O/pmc_summary.txt | head -5
else
  cd $R
  b() { n=$1; shift; timeout -k 10 600 python bench.py "$@" > $O/bench_$n.json 2> $O/bench_$n.err; python -c "
import json; d=json.loads(open('$O/bench_$n.json').readline()); print('$n', round(d['value']), 'Mrays/s', round(d['ms_per_step'],4), 'ms', 'frac', d['roofline'].get('frac'), d['roofline'].get('pmc_note',''))"; }
  b final
  b final_20_steps --steps 20 --warmup 5
  b path --mode path --no-cpu-baseline
  b path_20_steps --mode path --no-cpu-baseline --steps 20 --warmup 5
  b c3shape --chunks 16 --no-cpu-baseline
  b c5shape --chunks 32 --width 3840 --height 2160 --no-cpu-baseline
  b primary --mode primary --no-cpu-baseline
  b c5 --mode path --chunks 32 --width 3840 --height 2160 --spp 16 --steps 20 --warmup 20 --no-cpu-baseline
  b c4_4spp --mode path --spp 4 --steps 300 --no-cpu-baseline
  python tools/fixed_cost.py > $O/fixed_cost.txt 2>/dev/null; tail -3 $O/fixed_cost.txt
fi
