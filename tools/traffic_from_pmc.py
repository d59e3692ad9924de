#!/usr/bin/env python3
"""tools/traffic_from_pmc.py <pmc summary> <isa mix json> <valu issue rates txt> <out json> [workload key]

profiles/traffic_latest.json from tools/pmc.sh summaries, for bench.py's `roofline` object: a MAP keyed by workload
(bench.py's workload_key: "shadow:8:1920x1080:v0", "shadow:16:1920x1080:v0", "path:8:1920x1080:v0:1spp:4b",
"path:32:3840x2160:v0:16spp:4b", ...), every entry from one tools/pmc.sh run of that workload, the file stamped with the code
object (sha256 of libvrt.so's device code) all of them were collected on.  A call adds or replaces ONE workload; a file of
another build is started over.  bench.py prints the PMC-derived fields only for that very build and for a workload the map
holds.  Per kernel of a workload:
  hbm_bytes                 FETCH_SIZE x 2 (gfx950 tallies 128-B read requests at 64 B: MI355X_MICROARCH.md §HBM) + WRITE_SIZE,
                            both in KB, separate --pmc passes
  valu_ / salu_wave_instructions    SQ_INSTS_VALU, SQ_INSTS_SALU per launch
  issue_cycles_by_class     (primary / shadow kernels only) SIMD cycles those instructions need at the issue costs
                            tools/valu_rates.hip measured, the VALU split into classes in the proportions of the march loop's
                            fast path (tools/isa_mix.py) — an estimate of the mix outside the loop, exact counts overall.  The
                            path trace's kernels carry their class counters only (bench.py brackets their issue cycles).
  launches_per_frame        dispatches of the kernel per rendered frame of the PMC run (a 16-spp frame is several chains)
"""
import json
import os
import re
import sys

summary, mix_path, rates_path, out = sys.argv[1:5]
workload = sys.argv[5] if len(sys.argv) > 5 else "shadow:8:1920x1080:v0"
cur, vals, disp = None, {}, {}
for line in open(summary):
    if line.startswith("== "):
        cur = line[3:].strip()
        vals[cur] = {}
        disp[cur] = 0
    else:
        m = re.match(r"\s+(\S+)\s+dispatches=\s*(\d+) mean/dispatch=(\S+)", line)
        if m and cur:
            vals[cur][m.group(1)] = float(m.group(3))
            disp[cur] = max(disp[cur], int(m.group(2)))
mix = json.load(open(mix_path))
rates = {}
for line in open(rates_path):
    m = re.match(r"(k_\w+)\s+\S+ ms\s+in-kernel clock (\S+) GHz\s+(\S+) cycles", line)
    if m:
        rates[m.group(1)] = float(m.group(3))
mean = lambda ks: sum(rates[k] for k in ks) / len(ks)   # noqa: E731
cost = {"valu_simple": mean(["k_add_f32", "k_mul_f32", "k_mul_abs", "k_fma_f32", "k_and_b32", "k_add_u32", "k_lshr_b32", "k_mov_b32"]),
        "valu_half": mean(["k_min3_f32", "k_cndmask_e64", "k_cvt_flr", "k_cvt_i32_f32", "k_cvt_f32_i32", "k_cmp_u32", "k_cmp_f32", "k_bfi_b32",
                           "k_mad_i24", "k_max3_u32"]),
        "valu_pk": rates["k_pk_add_f32"], "valu_trans": mean(["k_sqrt_f32", "k_rcp_f32"]), "salu": rates["k_salu"]}
fp = {}
for lp in mix["loops"].values():
    for k, v in lp["fast_path"].items():
        fp[k] = fp.get(k, 0) + v
n_v = sum(fp.get(k, 0) for k in ("valu_simple", "valu_half", "valu_pk", "valu_trans"))
# bench.py's kernel names -> the kernels of the summary (the first that was dispatched); `mix`: the march loop's class mix applies
pick = {"primary_shadow_march": (["primary_shadow_wave_kernel<0, false, false, 1, false>"], True),
        "primary_march": (["primary_tile_kernel<0, false, false, true>", "primary_tile_kernel<0, false, false, false>"], True),
        "shadow_march": (["shadow_kernel<0, false, false>"], True),
        "path_primary_march": (["path_primary_kernel<0, false, false, false, false>", "path_primary_kernel<0, false, false, true, false>"], False),
        "path_bounce_marches": (["path_bounce_cells_kernel<true, 5u>(vrt::CellsLaunch)", "path_bounce_cells_kernel<true, 4u>(vrt::CellsLaunch)", "path_bounce_cells_kernel<false, 4u>(vrt::CellsLaunch)"], False)}
frames = 8.0   # tools/pmc.sh: bench.py --steps 6 --warmup 2, standing camera, no extra legs
kernels = {}
for k, (names, use_mix) in pick.items():
    v = next((n for n in names if n in vals and "SQ_INSTS_VALU" in vals[n]), None)
    if v is None:
        continue
    c = vals[v]
    entry = {"hbm_bytes": (c["FETCH_SIZE"] * 2 + c["WRITE_SIZE"]) * 1024.0 if "FETCH_SIZE" in c and "WRITE_SIZE" in c else None,
             "valu_wave_instructions": c["SQ_INSTS_VALU"], "salu_wave_instructions": c.get("SQ_INSTS_SALU"),
             "launches_per_frame": disp[v] / frames, "kernel": v, "counters": c}
    if use_mix:
        by = {cl: c["SQ_INSTS_VALU"] * fp.get(cl, 0) / n_v * cost[cl] for cl in ("valu_simple", "valu_half", "valu_pk", "valu_trans")}
        by["salu"] = c.get("SQ_INSTS_SALU", 0.0) * cost["salu"]
        # the same at the architectural rates of a SIMD-32: a full-rate wave64 instruction holds it 2 cycles, half rate 4,
        # transcendental 8 (the microbenchmark's costs are these plus its own loop's scalar instructions and VOP2 operand reads)
        nominal = {"valu_simple": 2.0, "valu_half": 4.0, "valu_pk": 4.0, "valu_trans": 8.0}
        entry["issue_cycles_by_class"] = by
        entry["valu_issue_cycles_by_class_nominal"] = {cl: c["SQ_INSTS_VALU"] * fp.get(cl, 0) / n_v * nominal[cl] for cl in nominal}
    kernels[k] = entry
code = mix.get("code_object_sha256")
res = {"code_object_sha256": code, "workloads": {}}
if os.path.exists(out):
    try:
        old = json.load(open(out))
        if old.get("code_object_sha256") == code and isinstance(old.get("workloads"), dict):
            res = old
    except Exception:
        pass
for k_, n_ in (("mix_half_plain", "k_mix_half_plain"), ("mix_pk_plain", "k_mix_pk_plain"), ("mix_step_like", "k_mix_step_like")):
    if n_ in rates:
        cost[k_] = rates[n_]   # (per instruction of the alternating body: tools/valu_rates.hip k_mix_*)
res["issue_cost_cycles"] = cost
res["fast_path_mix"] = fp
res["workloads"][workload] = {"kernels": kernels, "_source": f"{summary} (tools/pmc.sh: separate --pmc passes), {mix_path}, {rates_path}"}
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
print(workload, json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "counters"} for k, v in kernels.items()}, indent=1))
