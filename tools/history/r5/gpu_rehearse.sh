#!/bin/bash
# tools/r5/gpu_rehearse.sh — bench.py as the driver runs it (N = 1: the 20-step request, the long window beside it), then both N > 1
# modes rehearsed on the one GPU for C3 (1080p, 16^3 chunks, primary + shadow) and C5 (4K, 32^3 chunks, 16 spp path trace): every line
# carries config.expected_scaling (the prediction for N distinct devices beside the rehearsal's own figure)
mkdir -p gpurun_out/r5_reh
O=gpurun_out/r5_reh
if [ -z "$SKIP_N1" ]; then
python bench.py --steps 20 --warmup 5 > $O/n1_20steps.json 2> $O/n1_20steps.err || { tail -5 $O/n1_20steps.err; exit 1; }
python bench.py --steps 2000 --warmup 50 --no-cpu-baseline > $O/n1_2000steps.json 2> $O/n1_2000steps.err || { tail -5 $O/n1_2000steps.err; exit 1; }
python - <<'PY'
import json
a = json.loads(open("gpurun_out/r5_reh/n1_20steps.json").readline()); b = json.loads(open("gpurun_out/r5_reh/n1_2000steps.json").readline())
print("N=1 --steps 20: value", round(a["value"]), "over", a["steps_timed"], "frames; the 20 steps alone", round(a["value_requested_steps"]), "| --steps 2000:", round(b["value"]),
      "| driver-style / 2000-step:", round(a["value"] / b["value"], 4))
PY
fi
show() { python -c "
import json,sys
d=json.loads(open('$1').readline()); e=d['config'].get('expected_scaling') or {}
print('$2', round(d['value']), 'Mrays/s', round(d['ms_per_step'],4), 'ms; one GPU', round(e.get('frame_ms_1gpu_measured_in_this_run',0),4), 'ms; predicted for N devices', round(e.get('predicted_ms',0),4), 'ms = x', round(e.get('speedup',0),2), 'bound', e.get('bound'), '; host', round(d['host_submit_ms_per_step']*1e3,1), 'us/frame')"; }
C3="--chunks 16 --steps 200 --warmup 20 --no-cpu-baseline --no-extras"
C5="--mode path --chunks 32 --width 3840 --height 2160 --spp 16 --steps 6 --warmup 2 --no-cpu-baseline --no-extras"
for n in 2 4; do
  timeout -k 10 400 python bench.py --gpus $n --rehearse-on-one-gpu $C3 > $O/c3_mp_$n.json 2> $O/c3_mp_$n.err || { echo "C3 multi-process $n failed"; tail -5 $O/c3_mp_$n.err; exit 1; }
  show $O/c3_mp_$n.json "C3 N=$n processes (rehearsal):"
done
for n in 2 8; do
  timeout -k 10 400 python bench.py --gpus $n --single-process --rehearse-on-one-gpu $C3 > $O/c3_sp_$n.json 2> $O/c3_sp_$n.err || { echo "C3 single-process $n failed"; tail -5 $O/c3_sp_$n.err; exit 1; }
  show $O/c3_sp_$n.json "C3 N=$n one context (rehearsal):"
done
for n in 2 4; do
  timeout -k 10 600 python bench.py --gpus $n --rehearse-on-one-gpu $C5 > $O/c5_mp_$n.json 2> $O/c5_mp_$n.err || { echo "C5 multi-process $n failed"; tail -5 $O/c5_mp_$n.err; exit 1; }
  show $O/c5_mp_$n.json "C5 N=$n processes (rehearsal):"
done
for n in 2 8; do
  timeout -k 10 600 python bench.py --gpus $n --single-process --rehearse-on-one-gpu $C5 > $O/c5_sp_$n.json 2> $O/c5_sp_$n.err || { echo "C5 single-process $n failed"; tail -5 $O/c5_sp_$n.err; exit 1; }
  show $O/c5_sp_$n.json "C5 N=$n one context (rehearsal):"
done
