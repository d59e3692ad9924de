#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04c; mkdir -p $O; cd $R
for rep in 1 2; do for v in 0 1; do
  echo "== VRT_REBUILD_EVENT_IN_LAUNCH=$v"; VRT_REBUILD_EVENT_IN_LAUNCH=$v python tools/edit_cost.py 8 2>&1 | grep -v amdgpu.ids | grep "8^3\|lone edit"
done; done | tee $O/event_in_launch_ab.txt
