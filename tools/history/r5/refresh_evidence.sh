#!/bin/bash
# tools/refresh_evidence.sh A|B — here, in the container, after `gpurun -- bash tools/r5/gpu_final.sh A` (resp. B):
#  A: gpurun_out/r05's kernel stats, PMC summaries and issue rates -> profiles/r05_*, the march's ISA mix, and
#     profiles/traffic_latest.json made anew from them (one entry per workload README quotes, stamped with this build's code object)
#  B: the bench lines -> profiles/r05_final_bench*.json
set -e
cd "$(dirname "$0")/../.."; O=gpurun_out/r05
if [ "$1" = "A" ]; then
  for n in standing orbit; do for f in 1 2; do
    cp $O/kernel_stats_${n}_${f}_in_flight.csv profiles/r05_final_kernel_stats_${n}_${f}_in_flight.csv
    cp $O/bench_under_rocprof_${n}_${f}_in_flight.json profiles/r05_final_bench_under_rocprof_${n}_${f}_in_flight.json
  done; done
  for f in 1 2; do cp $O/kernel_stats_path_${f}_in_flight.csv profiles/r05_path_kernel_stats_${f}_in_flight.csv; cp $O/bench_under_rocprof_path_${f}_in_flight.json profiles/r05_path_bench_under_rocprof_${f}_in_flight.json; done
  cp $O/valu_issue_rates.txt profiles/r05_valu_issue_rates.txt
  python tools/isa_mix.py --out profiles/r05_isa_mix > /dev/null
  rm -f profiles/traffic_latest.json
  while read t key; do
    cp $O/pmc_summary_$t.txt profiles/r05_pmc_summary_$t.txt
    python tools/traffic_from_pmc.py profiles/r05_pmc_summary_$t.txt profiles/r05_isa_mix.json profiles/r05_valu_issue_rates.txt profiles/traffic_latest.json $key > /dev/null
  done <<K
shadow8 shadow:8:1920x1080:v0
shadow16 shadow:16:1920x1080:v0
shadow32_4k shadow:32:3840x2160:v0
primary8 primary:8:1920x1080:v0
path8 path:8:1920x1080:v0:1spp:4b
path8_4spp path:8:1920x1080:v0:4spp:4b
path32_4k_16spp path:32:3840x2160:v0:16spp:4b
K
  python -c "
import json; d=json.load(open('profiles/traffic_latest.json')); print('traffic_latest.json:', d['code_object_sha256'][:16], sorted(d['workloads']))"
else
  cp $O/bench_final.json profiles/r05_final_bench.json
  for n in final_20_steps path path_20_steps c3shape c5shape primary c5 c4_4spp; do cp $O/bench_$n.json profiles/r05_final_bench_${n#final_}.json; done
  ls -la profiles/r05_final_bench*.json | wc -l
fi
