"""profiles/traffic_latest.json — the PMC-derived inputs of bench.py's `roofline` object — is a map keyed by workload
(tools/traffic_from_pmc.py).  Every workload README.md quotes a bench line for must be in it with what bench.py reads, so that
no quoted line prints `roofline.frac: null` (round 3's review: the path-trace, C3 and C5 lines did)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# bench.py's workload_key of every line README.md quotes: C2 (headline), C3-shaped, 4K over C5's world, primary only, C4, C4 at 4 spp, C5
QUOTED = {
    "shadow:8:1920x1080:v0": ["primary_shadow_march"],
    "shadow:16:1920x1080:v0": ["primary_shadow_march"],
    "shadow:32:3840x2160:v0": ["primary_shadow_march"],
    "primary:8:1920x1080:v0": ["primary_march"],
    "path:8:1920x1080:v0:1spp:4b": ["path_primary_march", "path_bounce_marches"],
    "path:8:1920x1080:v0:4spp:4b": ["path_primary_march", "path_bounce_marches"],
    "path:32:3840x2160:v0:16spp:4b": ["path_primary_march", "path_bounce_marches"],
}


def test_every_quoted_workload_has_its_counters():
    tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))
    assert isinstance(tj.get("code_object_sha256"), str) and len(tj["code_object_sha256"]) == 64
    assert isinstance(tj.get("workloads"), dict), "traffic_latest.json is a map keyed by workload"
    for key, kernels in QUOTED.items():
        assert key in tj["workloads"], f"no PMC data for workload {key}: bench.py would print roofline.frac = null for it"
        have = tj["workloads"][key]["kernels"]
        for k in kernels:
            assert k in have, f"{key}: kernel {k} missing"
            e = have[k]
            assert e["valu_wave_instructions"] > 0 and e["hbm_bytes"] and e["hbm_bytes"] > 0, (key, k)
            assert e["launches_per_frame"] >= 1.0, (key, k, e["launches_per_frame"])
            for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU_CVT", "SQ_INSTS_VALU_TRANS_F32"):
                assert c in e["counters"], (key, k, c)
            if k in ("primary_shadow_march", "primary_march"):
                assert e.get("valu_issue_cycles_by_class_nominal"), (key, k)   # the march loop's class mix applies to these


def test_bench_reads_the_map_by_its_own_workload_key():
    """bench.py builds the key from its arguments; the quoted keys must be the ones its formula gives."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'f"{args.mode}:{args.chunks}:{args.width}x{args.height}:v{args.variant}"' in src
    assert 'f":{args.spp}spp:{args.bounces}b"' in src and 'tj["workloads"][workload_key]' in src
