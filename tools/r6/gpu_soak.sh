#!/bin/bash
# tools/r6/gpu_soak.sh — round 6's soaks against the oracle: the client's draw + present loop with the presentation declared /
# redeclared / off (tools/soak_edits.py ... present), the plain session, and ONE context over three device contexts with issuing
# threads forced on (they make their own message waits since round 6).  -> gpurun_out/r06/soak.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
{ timeout -k 10 400 python tools/soak_edits.py 170 3 "" present 2>&1 | grep -v amdgpu.ids | tail -4
  timeout -k 10 300 python tools/soak_edits.py 100 5 2>&1 | grep -v amdgpu.ids | tail -2
  VRT_GROUP_THREADS=1 timeout -k 10 300 python tools/soak_edits.py 100 7 0,0,0 poison 2>&1 | grep -v amdgpu.ids | tail -2
  VRT_GROUP_THREADS=1 timeout -k 10 300 python tools/soak_edits.py 80 9 0,0 texel 2>&1 | grep -v amdgpu.ids | tail -2; } | tee $O/soak.txt
