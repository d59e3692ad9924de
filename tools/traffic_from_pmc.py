#!/usr/bin/env python3
"""profiles/traffic_latest.json from a tools/pmc.sh summary: HBM bytes per launch of the two default march kernels =
FETCH_SIZE x 2 (gfx950 tallies 128-B read requests at 64 B: MI355X_MICROARCH.md §HBM) + WRITE_SIZE, both in KB."""
import json
import re
import sys

summary, out = sys.argv[1], sys.argv[2]
cur, vals = None, {}
for line in open(summary):
    if line.startswith("== "):
        cur = line[3:].strip()
        vals[cur] = {}
    else:
        m = re.match(r"\s+(\S+)\s+dispatches=\s*\d+ mean/dispatch=(\S+)", line)
        if m and cur:
            vals[cur][m.group(1)] = float(m.group(2))
pick = {"primary_shadow_march": "primary_shadow_wave_kernel<0, false, false, 4>",
        "primary_march": "primary_tile_kernel<0, false, false, true>", "shadow_march": "shadow_kernel<0, false, false>"}
res = {k: (vals[v]["FETCH_SIZE"] * 2 + vals[v]["WRITE_SIZE"]) * 1024.0 for k, v in pick.items() if v in vals and "FETCH_SIZE" in vals[v]}
res["valu_wave_instructions"] = {k: vals[v]["SQ_INSTS_VALU"] for k, v in pick.items() if v in vals and "SQ_INSTS_VALU" in vals[v]}
res["_source"] = f"{summary} (FETCH_SIZE*2 + WRITE_SIZE, KB -> bytes per launch; separate --pmc passes, tools/pmc.sh)"
json.dump(res, open(out, "w"), indent=1)
print(res)
