#!/bin/bash
# tools/r6/gpu_pad_ab.sh [libs...] — what a march step's instructions cost the frame: same-box runs of builds that differ by N extra
# instructions of one class in every step (vrt_march.h take_step under -DVRT_AB_PAD_{VALU,HALF,SALU,NOP}=N, tools/ab/build_variant.sh),
# against the tree's build.  C2's headline, lone launch, and C3's shape.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
LIBS=${@:-voxelraytracing_amd/libvrt.so tools/ab/libvrt_pad_valu3.so tools/ab/libvrt_pad_valu6.so tools/ab/libvrt_pad_half3.so tools/ab/libvrt_pad_salu3.so tools/ab/libvrt_pad_salu6.so tools/ab/libvrt_pad_nop6.so}
for rep in 1 2; do for lib in $LIBS; do
a=$(VRT_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --steps 3000 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('C2 %.0f period %.2f us' % (d['value'], d['ms_per_step']*1e3), 'standing %.0f' % d['value_fixed_camera'], '1-in-flight %.0f' % d['value_1_in_flight'], 'lone launch %.2f us' % (d['avg_launch_ms_1_in_flight']*1e3))")
b=$(VRT_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 2000 --chunks 16 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('C3shape %.0f' % d['value'])")
echo "$lib $a | $b"
done; done | tee $O/pad_ab.txt
