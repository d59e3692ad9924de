#!/bin/bash
# tools/gpu_bench_modes.sh — bench.py as the driver runs it, then both N > 1 modes rehearsed on the one GPU
mkdir -p gpurun_out
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err || { tail -5 gpurun_out/bench_n1.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_n1.json").readline())
print("N=1:", round(d["value"]), "Mrays/s", d["ms_per_step"], "ranks_seen", d["ranks_seen"], "links", d["links"], "frac", d["roofline"]["frac"], d["roofline"].get("frac_fixed_camera"),
      "1-in-flight", d.get("value_1_in_flight"), d.get("value_1_in_flight_orbit"), "cpu", d.get("cpu_baseline", {}).get("value"))
PY
for n in 2 3; do
  timeout -k 10 300 python bench.py --gpus $n --rehearse-on-one-gpu --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/bench_reh_$n.json 2> gpurun_out/bench_reh_$n.err || { echo "rehearsal $n failed"; tail -5 gpurun_out/bench_reh_$n.err; exit 1; }
  python -c "
import json; d=json.loads(open('gpurun_out/bench_reh_$n.json').readline()); print('N=$n processes (rehearsal):', round(d['value']), 'ranks_seen', d['ranks_seen'], d['links'])"
  timeout -k 10 300 python bench.py --gpus $n --single-process --rehearse-on-one-gpu --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/bench_sp_$n.json 2> gpurun_out/bench_sp_$n.err || { echo "single-process $n failed"; tail -5 gpurun_out/bench_sp_$n.err; exit 1; }
  python -c "
import json; d=json.loads(open('gpurun_out/bench_sp_$n.json').readline()); print('N=$n one context (rehearsal):', round(d['value']), 'ranks_seen', d['ranks_seen'], d['links'])"
done
# a rank that cannot reach its peers: WORLD_SIZE says 2, only rank 0 exists -> status 3 within the timeout, the rank named
RANK=0 LOCAL_RANK=0 WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=29655 timeout -k 10 120 python bench.py --gpus 2 --rehearse-on-one-gpu --init-timeout 8 --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/bench_lonely.err; echo "lonely rank exit status $?"; tail -2 gpurun_out/bench_lonely.err
# the driver's own launch form for N > 1 (one process per rank started by torch.distributed.run), every rank on the one GPU
for n in 2 4; do
  timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29700 + n)) bench.py --gpus $n --steps 40 --warmup 10 --rehearse-on-one-gpu --no-cpu-baseline > gpurun_out/bench_torchrun_$n.json 2> gpurun_out/bench_torchrun_$n.err || { echo "torchrun $n failed"; tail -8 gpurun_out/bench_torchrun_$n.err; exit 1; }
  python -c "
import json
lines=[l for l in open('gpurun_out/bench_torchrun_$n.json') if l.startswith('{')]
assert len(lines)==1, lines
d=json.loads(lines[0]); print('torch.distributed.run N=$n (rehearsal):', round(d['value']), 'n_gpus', d['n_gpus'], 'ranks_seen', d['ranks_seen'], 'scaling', d['scaling'])"
done
