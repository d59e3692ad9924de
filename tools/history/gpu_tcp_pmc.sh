#!/bin/bash
# tools/gpu_tcp_pmc.sh — is the bounce launch waiting for its vector L1?  TA / TCP busy and stall counters of a C4 path frame
mkdir -p gpurun_out
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --list-avail 2>/dev/null | grep -E "^\s*(Name|Counter_Name)?\s*:?\s*(TA_|TCP_|TD_)" | sort -u > $GRAFT_REPO_ROOT/gpurun_out/avail_ta_tcp.txt ) 
wc -l gpurun_out/avail_ta_tcp.txt
PMC_GROUPS="none" PMC_EXTRA="TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum;TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_TCR_TCP_STALL_CYCLES_sum;TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum;TA_BUFFER_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum TA_BUFFER_COALESCED_READ_CYCLES_sum;GRBM_GUI_ACTIVE TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum" bash tools/pmc.sh tcp --mode path > /dev/null 2>&1
grep -A40 "path_bounce_cells" gpurun_out/pmc_tcp/summary.txt | head -45
grep -l "failed\|rror" gpurun_out/pmc_tcp/*.log | head
