#!/bin/bash
# tools/r6/gpu_selector_ab.sh — round 6's last cut of the march step: the cell grid's air-leaf entry as the insert's bit selector
# (vrt_device.h kAirLeaf; three integer subtractions per step gone).  The -m gpu suite on the new build, then the same-box A/B against
# tools/ab/libvrt_base.so (the build before the change): C2 in every regime, C3's shape, 4K over C5's world, primary only, C4.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
if [ -z "$SKIP_TESTS" ]; then
timeout -k 10 900 env VRT_LIB=${TEST_LIB:-voxelraytracing_amd/libvrt.so} python -m pytest tests -x -q -m gpu > $O/selector_gpu_tests.log 2>&1; tail -2 $O/selector_gpu_tests.log
grep -q " failed\| error" $O/selector_gpu_tests.log && exit 1
fi
for rep in 1 2; do for lib in ${LIBS:-tools/ab/libvrt_base.so voxelraytracing_amd/libvrt.so}; do
a=$(VRT_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --steps 3000 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); op=d['operating_point']
print('C2 %.0f' % d['value'], 'standing %.0f' % d['value_fixed_camera'], '1-in-flight %.0f orbit %.0f' % (d['value_1_in_flight'], d['value_1_in_flight_orbit']), 'lone launch %.2f us' % (d['avg_launch_ms_1_in_flight']*1e3), '| client us:', ' '.join('%s=%.1f' % (k, v['ms_per_frame']*1e3) for k, v in op.items() if isinstance(v, dict)))")
b=$(for w in "--chunks 16" "--chunks 32 --width 3840 --height 2160" "--mode primary" "--mode path --steps 500"; do VRT_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 2000 $w 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('[$w] %.0f' % d['value'], end=' ')"; done)
echo "$lib $a | $b"
done; done | tee $O/selector_ab.txt
