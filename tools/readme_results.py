#!/usr/bin/env python3
"""tools/readme_results.py [rNN] — README.md's results table from the committed bench lines (profiles/rNN_final_bench*.json):
rewrites the rows between the table's header and the line that starts with 'CPU baseline'."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r06"


def J(n):
    return json.load(open(os.path.join(ROOT, "profiles", f"{R}_final_bench{n}.json")))


f, f20, p, c5, c3, c5s, pr, c44 = J(""), J("_20_steps"), J("_path"), J("_c5"), J("_c3shape"), J("_c5shape"), J("_primary"), J("_c4_4spp")
fr = lambda d: d["roofline"].get("frac")   # noqa: E731
rows = [
    ("C2 headline: 1920x1080, 8^3 chunks, primary + shadow, orbiting camera, per-frame seam, 2 frames in flight", f["value"], f["ms_per_step"], fr(f), f"`{R}_final_bench.json`"),
    (f"... the driver's command (`--steps 20 --warmup 5`): value over {f20['steps_timed']} frames; the 20 steps alone {f20['value_requested_steps']:.0f}",
     f20["value"], f20["ms_per_step"], fr(f20), f"`{R}_final_bench_20_steps.json`"),
    ("... standing camera", f["value_fixed_camera"], f["ms_per_step_fixed_camera"], None, f"`{R}_final_bench.json`"),
    ("... one frame at a time, view at rest", f["value_1_in_flight"], f["ms_per_step_1_in_flight"], None, f"`{R}_final_bench.json`"),
    ("... one frame at a time, orbit", f["value_1_in_flight_orbit"], f["ms_per_step_1_in_flight_orbit"], None, f"`{R}_final_bench.json`"),
    ("... the client's real frame: 30^3-chunk grid, untagged `chunk_roots` rewrite, render + present (declared: the frame's launch stores the window's image)", f["operating_point"]["2_in_flight"]["value"],
     f["operating_point"]["2_in_flight"]["ms_per_frame"], None, f"`{R}_final_bench.json` (`operating_point`)"),
    ("... the client's real frame, one frame at a time", f["operating_point"]["1_in_flight"]["value"],
     f["operating_point"]["1_in_flight"]["ms_per_frame"], None, f"`{R}_final_bench.json` (`operating_point`)"),
    ("... the same with the blit as a launch of its own (undeclared), two in flight / one at a time: " + " / ".join("%.0f" % f["operating_point"][k]["value"] for k in ("2_in_flight_blit_launch", "1_in_flight_blit_launch")),
     f["operating_point"]["2_in_flight_blit_launch"]["value"], f["operating_point"]["2_in_flight_blit_launch"]["ms_per_frame"], None, f"`{R}_final_bench.json` (`operating_point`)"),
    ("primary rays only", pr["value"], pr["ms_per_step"], fr(pr), f"`{R}_final_bench_primary.json`"),
    ("C3's shape on one GPU: 16^3 chunks", c3["value"], c3["ms_per_step"], fr(c3), f"`{R}_final_bench_c3shape.json`"),
    ("3840x2160 over C5's 32^3 world, primary + shadow", c5s["value"], c5s["ms_per_step"], fr(c5s), f"`{R}_final_bench_c5shape.json`"),
    ("C4: 4-bounce path trace, 1 spp", p["value"], p["ms_per_step"], fr(p), f"`{R}_final_bench_path.json`"),
    ("C4's scene at 4 spp", c44["value"], c44["ms_per_step"], fr(c44), f"`{R}_final_bench_c4_4spp.json`"),
    ("C5: 3840x2160, 16 spp, 32^3 chunks, one GPU", c5["value"], c5["ms_per_step"], fr(c5), f"`{R}_final_bench_c5.json`"),
]
tab = "| config | Mrays/s | ms per frame | `roofline.frac` | file (`profiles/`) |\n|---|---|---|---|---|\n"
for n, v, ms, fc, fl in rows:
    tab += f"| {n} | {format(v, ',.0f').replace(',', ' ')} | {ms:.4f} | {('%.2f' % fc) if fc else '-'} | {fl} |\n"
cpu = f["cpu_baseline"]
readme = open(os.path.join(ROOT, "README.md")).read()
a = readme.index("| config | Mrays/s |")
b = readme.index("CPU baseline (the oracle")
e = readme.index("\n", b)
line = f"CPU baseline (the oracle, a port: `cpu_baseline` of the same line): **{cpu['value']:.1f} Mrays/s on {cpu['cores']} cores**, {cpu['value_1_thread']:.2f} on one thread."
open(os.path.join(ROOT, "README.md"), "w").write(readme[:a] + tab + "\n" + line + readme[e:])
print(tab)
