#!/bin/bash
# tools/gpu_r4_edit_trace.sh — kernel timeline of tools/edit_cost.py's lone edits (rocprofv3 --kernel-trace): where the device's share goes
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04_edit_kt -- python3 $GRAFT_REPO_ROOT/tools/edit_cost.py 8 > $GRAFT_REPO_ROOT/gpurun_out/r04_edit_kt.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/r04_edit_kt -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the lone-edit phase: find accel_chunks dispatches that are preceded by a gap (a synchronise) — print 6 of them with their neighbours
idx = [i for i, r in enumerate(rows) if "accel_chunks" in r["Kernel_Name"]]
shown = 0
for i in idx[5:]:
    if shown >= 5: break
    lo = max(0, i - 2); hi = min(len(rows), i + 3)
    t0 = int(rows[lo]["Start_Timestamp"])
    prev_end = int(rows[lo - 1]["End_Timestamp"]) if lo else t0
    print(f"--- idle before: {(t0 - prev_end) / 1e3:.1f} us")
    for r in rows[lo:hi]:
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-60:]
        print(f"  {n:60s} start {(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} us  dur {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f} us  stream {r.get('Stream_Id', r.get('Queue_Id', '?'))}")
    shown += 1
PY
