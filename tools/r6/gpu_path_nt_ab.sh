#!/bin/bash
# tools/r6/gpu_path_nt_ab.sh — same-box A/B of the path trace's streaming accesses as non-temporal ones (tools/ab/build_variant.sh
# ntp_rmw "-DVRT_AB_NT_RMW": the texel read-modify-write of a path that ends; ntp_rmw_fin: + the sample planes' finishing pass;
# ntp_all: + the primary launch's texel / plane stores): C4 (1 and 2 frames in flight), C4 at 4 spp, C5.
R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2; do
  for lib in voxelraytracing_amd/libvrt.so tools/ab/libvrt_ntp_rmw.so tools/ab/libvrt_ntp_rmw_fin.so tools/ab/libvrt_ntp_all.so; do
    c4=$(VRT_LIB=$lib timeout -k 10 300 python bench.py --mode path --no-cpu-baseline --steps 500 --warmup 50 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('C4 %.0f (1 in flight %.0f, bounce launch %.1f us)' % (d['value'], d['value_1_in_flight'], d['avg_bounce_launches_ms_1_in_flight']*1e3))")
    c44=$(VRT_LIB=$lib timeout -k 10 300 python bench.py --mode path --spp 4 --no-cpu-baseline --no-extras --steps 200 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('C4 4spp %.0f' % d['value'])")
    c5=$(VRT_LIB=$lib timeout -k 10 300 python bench.py --mode path --chunks 32 --width 3840 --height 2160 --spp 16 --steps 12 --warmup 4 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('C5 %.0f (%.2f ms)' % (d['value'], d['ms_per_step']))")
    echo "$lib $c4 | $c44 | $c5"
  done
done
