#!/bin/bash
# tools/r6/gpu_nt_ab.sh — same-box A/B: the texel (and window-pixel) stores of the march kernel as plain stores against non-temporal
# ones (tools/ab/build_variant.sh nt_texels "-DVRT_AB_NT_TEXELS", nt_both "... -DVRT_AB_NT_SCREEN") -> gpurun_out/r06/nt_ab.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
for rep in 1 2; do
  for lib in voxelraytracing_amd/libvrt.so tools/ab/libvrt_nt_texels.so tools/ab/libvrt_nt_both.so; do
    VRT_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); op=d['operating_point']
print('$lib', 'C2 %.0f' % d['value'], 'fixed %.0f' % d['value_fixed_camera'], '1-in-flight %.0f orbit %.0f' % (d['value_1_in_flight'], d['value_1_in_flight_orbit']), 'lone launch %.2f us' % (d['avg_launch_ms_1_in_flight']*1e3),
      'client frame us:', ' '.join('%s=%.1f' % (k, v['ms_per_frame']*1e3) for k, v in op.items() if isinstance(v, dict)))"
  done
done | tee $O/nt_ab.txt
