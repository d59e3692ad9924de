#!/bin/bash
# tools/gpu_path_trace.sh — kernel trace of the path trace at 2 frames in flight (start / end of every launch)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/kt_path2 -- python3 $R/bench.py --mode path --steps 60 --warmup 10 --no-cpu-baseline --no-extras --settle-seconds 0 ${BENCH_ARGS} > $R/gpurun_out/kt_path2.log 2>&1
f=$(find $R/gpurun_out/kt_path2 -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows if "vrt::path" in r["Kernel_Name"]]
ks.sort()
ks = ks[-8*40:]   # the last 40 frames
t0, t1 = ks[0][0], max(k[1] for k in ks)
ev = []
for s, e, n in ks:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
act, last, hist = 0, t0, {}
for t, d in ev:
    hist[act] = hist.get(act, 0) + (t - last)
    act += d; last = t
tot = sum(hist.values())
print("concurrency histogram (fraction of time with k path launches active):", {k: round(v / tot, 3) for k, v in sorted(hist.items())})
import collections
d = collections.defaultdict(list)
for s, e, n in ks:
    d["primary" if "primary" in n else "bounce"].append((e - s) / 1e3)
for k, v in d.items():
    print(k, "launches", len(v), "mean us %.1f" % (sum(v) / len(v)), "min %.1f max %.1f" % (min(v), max(v)))
print("span per frame us: %.1f" % ((t1 - t0) / 1e3 / 40))
# the sequence of one stretch
for s, e, n in ks[40:56]:
    print("%9.1f %9.1f %s" % ((s - t0) / 1e3, (e - t0) / 1e3, "primary" if "primary" in n else "bounce"))
PY
