#!/bin/bash
# tools/r5/gpu_final.sh A|B|C — the round's evidence (gpurun_out/r05/).
#  A: rocprofv3 kernel stats (standing camera and orbit, 1 / 2 frames in flight; the path trace) and the PMC passes of every
#     workload README quotes.  Then, here: tools/isa_mix.py and tools/traffic_from_pmc.py per workload -> profiles/traffic_latest.json.
#  B: the bench lines (they print the PMC-derived fields only for the build and workloads traffic_latest.json holds).
#  C: the costs (edit, stream, fixed) and the N > 1 rehearsals.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O
if [ "$1" = "A" ] || [ "$1" = "A2" ]; then
  cd /tmp && export TMPDIR=/tmp
  ks() { # name, bench args...
    n=$1; shift
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$n -- python3 $R/bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-extras "$@" > $O/bench_under_rocprof_$n.json 2> $O/kt_$n.err
    f=$(find $O/kt_$n -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_$n.csv && head -3 $f | cut -c1-150
    rm -rf $O/kt_$n
  }
  if [ "$1" = "A" ]; then
  ks standing_2_in_flight --fixed-camera
  ks standing_1_in_flight --fixed-camera --frames-in-flight 1
  ks orbit_2_in_flight
  ks orbit_1_in_flight --frames-in-flight 1
  ks path_2_in_flight --mode path --fixed-camera --steps 500
  ks path_1_in_flight --mode path --fixed-camera --steps 300 --frames-in-flight 1
  fi
  cd $R
  pm() { # tag, groups, bench args...
    t=$1; g=$2; shift 2
    PMC_GROUPS="$g" bash tools/pmc.sh r05_$t "$@" > /dev/null 2>&1; cp gpurun_out/pmc_r05_$t/summary.txt $O/pmc_summary_$t.txt; rm -rf gpurun_out/pmc_r05_$t/g*/
    echo "pmc $t: $(grep -c mean/dispatch $O/pmc_summary_$t.txt) counter lines"
  }
  pm shadow8 "1 2 3 4 5 6 7 8"
  pm shadow16 "1 2 3 5 6 7" --chunks 16
  pm shadow32_4k "1 2 3 5 6 7" --chunks 32 --width 3840 --height 2160
  pm primary8 "1 2 3 5 6 7" --mode primary
  pm path8 "1 2 3 4 5 6 7" --mode path
  pm path8_4spp "1 2 3 5 6 7" --mode path --spp 4
  pm path32_4k_16spp "1 2 3 5 6 7" --mode path --chunks 32 --width 3840 --height 2160 --spp 16
  ./tools/valu_rates > $O/valu_issue_rates.txt 2>&1
elif [ "$1" = "B" ]; then
  cd $R
  b() { n=$1; shift; timeout -k 10 600 python bench.py "$@" > $O/bench_$n.json 2> $O/bench_$n.err; python -c "
import json; d=json.loads(open('$O/bench_$n.json').readline()); r=d['roofline']; print('$n', round(d['value']), 'Mrays/s', round(d['ms_per_step'],4), 'ms', 'frac', r.get('frac'), 'traffic', r.get('traffic'), r.get('pmc_note','')[:50])"; }
  b final
  b final_20_steps --steps 20 --warmup 5
  b path --mode path --no-cpu-baseline
  b path_20_steps --mode path --no-cpu-baseline --steps 20 --warmup 5
  b c3shape --chunks 16 --no-cpu-baseline
  b c5shape --chunks 32 --width 3840 --height 2160 --no-cpu-baseline
  b primary --mode primary --no-cpu-baseline
  b c5 --mode path --chunks 32 --width 3840 --height 2160 --spp 16 --steps 20 --warmup 20 --no-cpu-baseline
  b c4_4spp --mode path --spp 4 --steps 300 --no-cpu-baseline
else
  cd $R
  python tools/fixed_cost.py > $O/fixed_cost.txt 2>/dev/null; tail -3 $O/fixed_cost.txt
  python tools/edit_cost.py 8 2>&1 | grep -v amdgpu.ids > $O/edit_cost.txt; python tools/edit_cost.py 32 2>&1 | grep -v amdgpu.ids >> $O/edit_cost.txt; cat $O/edit_cost.txt
  python tools/stream_cost.py 2>&1 | grep -v amdgpu.ids > $O/stream_cost.txt; tail -6 $O/stream_cost.txt
  bash tools/r5/gpu_rehearse.sh > $O/rehearse.txt 2>&1; cat $O/rehearse.txt
  for n in 2 3 8; do timeout -k 10 300 python bench.py --gpus $n --single-process --rehearse-on-one-gpu --no-cpu-baseline > $O/bench_sp_$n.json 2> $O/bench_sp_$n.err; python -c "
import json; d=json.loads(open('$O/bench_sp_$n.json').readline()); print('one context over $n device contexts on one GPU (2000 steps):', round(d['value']), 'Mrays/s, host', round(d['host_submit_ms_per_step']*1e3,1), 'us per frame')"; done
fi
