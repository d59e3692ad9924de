#!/bin/bash
# tools/r6/gpu_assemble_ab.sh — the gather root's assembly kernels with plain against non-temporal texel stores (tools/ab/build_variant.sh nt_assemble
# "-DVRT_AB_NT_ASSEMBLE"): kernel durations under rocprofv3 --kernel-trace in the one-context rehearsal (C3's shape, N = 2 and 8: 8-byte records
# shaded at the root) and with texel messages (the path trace's form of the message, here for primary + shadow frames of variant 3).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for lib in voxelraytracing_amd/libvrt.so tools/ab/libvrt_nt_assemble.so; do
  for n in 2 8; do
    VRT_LIB=$R/$lib timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_asm -- python3 $R/bench.py --gpus $n --single-process --rehearse-on-one-gpu --chunks 16 --steps 400 --warmup 50 --no-cpu-baseline --no-extras > $O/asm.json 2> $O/asm.err
    f=$(find $O/kt_asm -name "*kernel_stats.csv" | head -1)
    echo "$lib N=$n: $(python3 -c "
import json; d=json.loads(open('$O/asm.json').readline()); print('%.0f Mrays/s %.4f ms' % (d['value'], d['ms_per_step']))") | $(python3 -c "
import csv
for r in csv.DictReader(open('$f')):
    if 'assemble' in r['Name']: print(r['Name'].split('(')[0].split('::')[-1], r['Calls'], 'calls', '%.2f us' % (float(r['AverageNs'])/1e3))")"
    rm -rf $O/kt_asm
  done
done
