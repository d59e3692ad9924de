#!/bin/bash
# A/B: the bounce launch's hand-outs asked for a march step ahead (-DVRT_CELLS_PREFETCH) against the tree's build
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04p; mkdir -p $O; cd $R
VRT_LIB=$R/tools/ab/libvrt_prefetch.so timeout -k 10 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -x -q -k "path or c4 or c5 or bounce or sample" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
for rep in 1 2; do
  for lib in $R/voxelraytracing_amd/libvrt.so $R/tools/ab/libvrt_prefetch.so; do
    for fif in 2 1; do
      VRT_LIB=$lib python bench.py --mode path --steps 500 --no-cpu-baseline --no-extras --frames-in-flight $fif 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$(basename $lib)', 'C4 in_flight=$fif', '%.0f Mrays/s' % d['value'], 'ms_per_step=%.4f' % d['ms_per_step'])"
    done
  done
done | tee $O/ab.txt
for lib in $R/voxelraytracing_amd/libvrt.so $R/tools/ab/libvrt_prefetch.so; do
  VRT_LIB=$lib python bench.py --mode path --chunks 32 --width 3840 --height 2160 --spp 16 --steps 10 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$(basename $lib)', 'C5', '%.0f Mrays/s' % d['value'])"
done | tee -a $O/ab.txt
