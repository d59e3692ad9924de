#!/bin/bash
# tools/pmc.sh <tag> [bench args...] — rocprofv3 PMC passes (one counter group per run; no tracing mixed in,
# as the pool requires) over a short bench run of one and the same frame (standing camera, no extra legs).
# Output under gpurun_out/pmc_<tag>/; tools/traffic_from_pmc.py turns the summary into profiles/traffic_latest.json.
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU" \
           "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  if [ -n "$PMC_GROUPS" ] && ! echo " $PMC_GROUPS " | grep -q " $i "; then continue; fi   # PMC_GROUPS="1 2": only those passes
  timeout -k 10 150 rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --fixed-camera --no-extras --settle-seconds 0 --exact-steps "$@" > $OUT/g$i.log 2>&1 || echo "group $i failed: $grp"
done
# PMC_EXTRA="A B;C D": further counter groups, one run each
if [ -n "$PMC_EXTRA" ]; then
  IFS=';' read -ra EXTRA <<< "$PMC_EXTRA"
  for grp in "${EXTRA[@]}"; do
    i=$((i+1))
    timeout -k 10 150 rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --fixed-camera --no-extras --settle-seconds 0 --exact-steps "$@" > $OUT/g$i.log 2>&1 || echo "group $i failed: $grp"
  done
fi
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT > $OUT/summary.txt
cat $OUT/summary.txt
