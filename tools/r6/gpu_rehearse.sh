#!/bin/bash
# tools/r6/gpu_rehearse.sh — both N > 1 modes rehearsed on the one GPU with the HOST TERM MEASURED (round 6): every line carries
# config.expected_scaling.host_term (vrt_get_issue_profile / FrameGather.host_profile of the timed frames) beside the model's
# prediction for N distinct devices.  One context over N device contexts: issued by one thread in turn (the default when the
# ordinals repeat) and with one issuing thread per device (VRT_GROUP_THREADS=1: on one GPU they contend for the one device's
# queues).  One process per GPU: the gloo stand-in's line, and RCCL's own cost per collective from a one-rank group (--force-gather).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06/reh; mkdir -p $O; cd $R
show() { python -c "
import json,sys
d=json.loads(open('$1').readline()); e=d['config'].get('expected_scaling') or {}; h=e.get('host_term') or {}; ip=h.get('issue_profile') or {}
s='$2 %d Mrays/s %.4f ms; one GPU %.4f ms; predicted for N devices %.4f ms = x %.2f bound %s; host: loop %.1f us/frame, render call %.1f us' % (d['value'], d['ms_per_step'], e.get('frame_ms_1gpu_measured_in_this_run',0), e.get('predicted_ms',0), e.get('speedup',0), e.get('bound'), d['host_submit_ms_per_step']*1e3, h.get('render_call_us',0))
if ip: s+=' = root %.1f + %d x shard %.1f (max %.1f) + join %.1f + tail %.1f (of it %.1f the message waits) [sum of parts %.1f]; with issuing threads %.1f; threads %d' % (ip['root_issue_us'], ip['devices']-1, ip['shard_issue_us_mean'], ip['shard_issue_us_max'], ip['join_wait_us'], ip['tail_us'], ip['message_waits_us'], h.get('serial_sum_of_parts_us',0), h.get('with_issuing_threads_us',0), ip['issuing_threads'])
c=h.get('collective')
if c: s+='; collective: gather call %.1f + wait %.1f + assemble %.1f us per %.1f frames' % (c['gather_call_us'], c['wait_us'], c['assemble_us'], c['frames_per_collective'])
s+='; term used %.1f us; faster by the model: %s' % (h.get('host_us_per_frame_used',0), (e.get('modes') or {}).get('predicted_faster'))
print(s)"; }
C3="--chunks 16 --steps 400 --warmup 50 --no-cpu-baseline --no-extras"
C5="--mode path --chunks 32 --width 3840 --height 2160 --spp 16 --steps 6 --warmup 2 --no-cpu-baseline --no-extras"
for n in 2 4 8; do
  timeout -k 10 400 python bench.py --gpus $n --single-process --rehearse-on-one-gpu $C3 > $O/c3_sp_$n.json 2> $O/c3_sp_$n.err || { echo "C3 one context $n failed"; tail -5 $O/c3_sp_$n.err; exit 1; }
  show $O/c3_sp_$n.json "C3 N=$n one context, issued in turn (rehearsal):"
done
for n in 2 8; do
  VRT_GROUP_THREADS=1 timeout -k 10 400 python bench.py --gpus $n --single-process --rehearse-on-one-gpu $C3 > $O/c3_spt_$n.json 2> $O/c3_spt_$n.err || { echo "C3 one context + threads $n failed"; tail -5 $O/c3_spt_$n.err; exit 1; }
  show $O/c3_spt_$n.json "C3 N=$n one context, issuing threads (rehearsal):"
done
for n in 2 4; do
  timeout -k 10 400 python bench.py --gpus $n --rehearse-on-one-gpu $C3 > $O/c3_mp_$n.json 2> $O/c3_mp_$n.err || { echo "C3 processes $n failed"; tail -5 $O/c3_mp_$n.err; exit 1; }
  show $O/c3_mp_$n.json "C3 N=$n processes (rehearsal, gloo stand-in):"
done
# RCCL's own host cost per collective: a one-rank nccl group, the pipelined gather + assemble path, batches of 1 and 4 frames
for b in 1 4; do
  timeout -k 10 400 python bench.py --gpus 1 --force-gather --gather-batch $b $C3 > $O/c3_fg_$b.json 2> $O/c3_fg_$b.err || { echo "force-gather failed"; tail -5 $O/c3_fg_$b.err; exit 1; }
  python -c "
import json
d=json.loads(open('$O/c3_fg_$b.json').readline()); print('C3 N=1 through a one-rank RCCL group, $b frame(s) per gather: %d Mrays/s %.4f ms; host loop %.1f us/frame; %s' % (d['value'], d['ms_per_step'], d['host_submit_ms_per_step']*1e3, json.dumps(d.get('gather_host_profile'))))"
done
for n in 2 8; do
  timeout -k 10 600 python bench.py --gpus $n --single-process --rehearse-on-one-gpu $C5 > $O/c5_sp_$n.json 2> $O/c5_sp_$n.err || { echo "C5 one context $n failed"; tail -5 $O/c5_sp_$n.err; exit 1; }
  show $O/c5_sp_$n.json "C5 N=$n one context (rehearsal):"
done
