#!/bin/bash
# tools/r6/gpu_waves_ab.sh [libs...] — same-box A/B of the headline kernel's workgroup size (tools/ab/build_variant.sh wavesN "-DVRT_AB_WAVES=N": N tiles =
# N waves per workgroup; the tree's default against the variants): C2 in every regime, the client's frame, C3's shape, 4K over C5's world, primary rays only.
cd $GRAFT_REPO_ROOT
LIBS=${@:-voxelraytracing_amd/libvrt.so tools/ab/libvrt_waves1.so}
for rep in 1 2; do for lib in $LIBS; do
a=$(VRT_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --steps 3000 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); op=d['operating_point']
print('C2 %.0f' % d['value'], 'standing %.0f' % d['value_fixed_camera'], '1-in-flight %.0f orbit %.0f' % (d['value_1_in_flight'], d['value_1_in_flight_orbit']), 'lone launch %.2f us' % (d['avg_launch_ms_1_in_flight']*1e3), '| client us:', ' '.join('%s=%.1f' % (k, v['ms_per_frame']*1e3) for k, v in op.items() if isinstance(v, dict)))")
b=$(for w in "--chunks 16" "--chunks 32 --width 3840 --height 2160" "--mode primary"; do VRT_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 2000 $w 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('[$w] %.0f' % d['value'], end=' ')"; done)
echo "$lib $a | $b"
done; done
