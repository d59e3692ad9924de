#!/usr/bin/env python3
"""bench.py — Mrays/s of the SVO ray-march on BASELINE.json's headline config.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one frame: 1920x1080, 8x8x8-chunk procedural SVO world, 1 primary ray per pixel + 1 shadow ray
per solid hit (config C2; inputs resident in HBM before the timed region).  With N > 1 the frame is
sharded by interleaved 8x8 screen tiles over N processes (one per GPU) and gathered to rank 0 with one RCCL
gather per frame plus a de-interleave kernel — total work fixed, so scaling is "strong".
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes(st, width, height):
    """SURVEY.md §8d / DESIGN.md §Algorithmic bytes, per kernel, from exact step / node-visit counts.

    march step: 4 B chunk_roots entry + 2 B x L node words + 4 B material is_liquid  = 8 + 2L
    primary kernel: + 16 B output per pixel (f32x3 + id word) + 16 B hit record per secondary ray launched
    shadow kernel:  + 16 B hit record read per ray (+ 32 B rgb/id read-modify-write per occluded ray, not
                    counted: occlusion count is not part of vrt_stats)"""
    p_steps, p_vis = st.primary_steps, st.primary_node_visits
    s_steps, s_vis = st.steps - p_steps, st.node_visits - p_vis
    primary = 8 * p_steps + 2 * p_vis + 16 * st.primary_rays + 16 * st.secondary_rays
    shadow = 8 * s_steps + 2 * s_vis + 16 * st.secondary_rays
    # the fused launch (default): both marches, one texel store per pixel; its hit records never leave LDS
    fused = 8 * st.steps + 2 * st.node_visits + 16 * st.primary_rays
    return primary, shadow, fused


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--chunks", type=int, default=8, help="world size in chunks (8 = config C2)")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--mode", choices=["shadow", "primary", "path"], default="shadow",
                    help="shadow = the headline metric (config C2); path = the C4/C5 kernel family (not the headline)")
    ap.add_argument("--spp", type=int, default=1)
    ap.add_argument("--bounces", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--frames-in-flight", type=int, default=2,
                    help="N = 1: frames the backend keeps in flight (vrt_set_frames_in_flight; 1 = one launch at a time)")
    ap.add_argument("--root-weight", type=int, default=0,
                    help="N > 1: tiles per period dealt to the gather root (vrt_config.shard_root_weight); 0 = measure "
                         "a few candidates off the clock and keep the fastest")
    ap.add_argument("--gather-batch", type=int, default=0,
                    help="N > 1: frames per gather (FrameGather.batch); 0 = measure {1, 2, 4, 8} off the clock and keep the fastest")
    ap.add_argument("--force-gather", action="store_true",
                    help="development only: with --gpus 1, still run the pipelined RCCL gather + assemble path (one-rank group)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="development only: run the N > 1 code path with every rank on cuda:0 and a gloo gather staged "
                         "through host memory (RCCL refuses two ranks on one device); never used for reported numbers")
    args = ap.parse_args()

    import torch
    from voxelraytracing_amd import Gpu, MODE_PATH, MODE_PRIMARY, MODE_PRIMARY_SHADOW, scenes
    from voxelraytracing_amd.shard import FrameGather

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP backend has no CPU path")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    sharded = world > 1 or args.force_gather
    if sharded:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    def all_reduce(t, op=None):
        """all_reduce of a small cuda tensor (through the host when rehearsing over gloo)."""
        if world == 1:
            return t
        kw = {} if op is None else {"op": op}
        if args.rehearse_on_one_gpu:
            c = t.cpu()
            dist.all_reduce(c, **kw)
            return c.to(t.device)
        dist.all_reduce(t, **kw)
        return t

    # ---- scene (deterministic, built by every rank) and upload: off the clock ----
    MODE = {"shadow": MODE_PRIMARY_SHADOW, "primary": MODE_PRIMARY, "path": MODE_PATH}[args.mode]
    rkw = dict(spp=args.spp, seed=0) if MODE == MODE_PATH else dict(variant=args.variant)
    sc = scenes.procedural(args.chunks, (args.width, args.height), MODE)
    if MODE == MODE_PATH:
        sc.settings.max_ray_bounces = args.bounces
        scenes._diffuse(sc.materials)
    side = None
    if sharded:
        # One non-default torch stream carries everything of this rank: the backend's kernels (vrt_set_stream), the
        # tensors' initialisation and the RCCL calls' stream dependencies.  (The default stream's handle is 0, which
        # vrt_set_stream reads as "use the context's own stream" — kernels there would not be ordered with RCCL.)
        side = torch.cuda.Stream(device=local_rank)
        torch.cuda.set_stream(side)
        assert side.cuda_stream != 0

    def make_pipeline(root_weight, batch=1):
        """(backend context, FrameGather) for this rank.  N > 1: the root renders its own tiles straight into the
        row-major frame (VRT_FLAG_ROW_MAJOR) and takes root_weight tiles of every root_weight + N - 1."""
        in_place = world > 1
        compact = in_place and MODE != MODE_PATH and args.variant == 0   # 8 B/pixel over the links, shaded at the root
        g = Gpu(sc.world.max_nodes(), sc.world.size_in_chunks(), sc.size, device=local_rank, shard_rank=rank, shard_count=world,
                tile_major=sharded and not (in_place and rank == 0), root_weight=root_weight, row_major=in_place and rank == 0,
                compact=compact and rank != 0)
        g.upload_world(sc.world, sc.materials)
        g.write_cam_data(sc.cam)
        g.write_settings(sc.settings)
        g.set_frames_in_flight(args.frames_in_flight)
        f = None
        if sharded:
            g.set_stream(side.cuda_stream)
            f = FrameGather(torch, dist, rank, world, args.width, args.height, torch.device("cuda", local_rank),
                            root_weight=root_weight, in_place=in_place, compact=compact, batch=batch)
            if args.rehearse_on_one_gpu:
                def staged_gather(which=0, async_op=False, nframes=None, f=f):
                    torch.cuda.synchronize()
                    parts = [torch.empty(f.msg.numel(), dtype=torch.int32) for _ in range(world)] if rank == 0 else None
                    dist.gather(f.msgs[which].cpu(), parts, dst=0)
                    if rank == 0:
                        f.recv[which].copy_(torch.stack(parts))
                f.gather = staged_gather
        return g, f

    def run_frames(g, f, n):
        for _ in range(n if (f is None or args.rehearse_on_one_gpu) else 0):
            if f is None:
                g.render(MODE, **rkw)
            else:
                f.bind(g, 0)
                g.render(MODE, **rkw)
                f.gather()
                if rank == 0:
                    f.assemble(g, 0)
        if f is not None and not args.rehearse_on_one_gpu:
            # the in-place root's own tiles never feed the collective: they run on the backend's in-flight streams
            own = dict(own_streams=True) if (f.in_place and rank == 0 and MODE != MODE_PATH) else {}
            left = n
            while left > 0:   # the gather of a batch overlaps the render of the next
                f.submit(g, lambda: g.render(MODE, **rkw, **own), min(left, f.batch))
                left -= f.batch
            f.drain(g)

    # ---- N > 1: how much of the frame the gather root should trace itself (off the clock) ----
    root_weight, batch, tuning, batch_tuning = 1, 1, None, None
    if world > 1:
        def trial(w0, b):
            g, f = make_pipeline(w0, b)
            run_frames(g, f, 8)
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run_frames(g, f, 48)
            dist.barrier()
            torch.cuda.synchronize()
            dt = all_reduce(torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda"), dist.ReduceOp.MAX)
            g.close()
            return float(dt[0]) / 48 * 1e3      # the all-reduced times are identical on every rank
        rehearsal = args.rehearse_on_one_gpu   # (its staged gather moves whole message buffers: one frame per gather)
        batch = args.gather_batch if args.gather_batch > 0 else (1 if rehearsal else 4)
        if args.root_weight > 0:
            root_weight = args.root_weight
        else:
            tuning = {w0: trial(w0, batch) for w0 in (1, 2, 3, 4, 6, 8, 12, 16)}
            root_weight = min(tuning, key=lambda k: (tuning[k], k))
        if args.gather_batch == 0 and not rehearsal:
            batch_tuning = {b: trial(root_weight, b) for b in (1, 2, 4, 8)}
            batch = min(batch_tuning, key=lambda k: (batch_tuning[k], k))
    elif sharded and args.gather_batch > 0:
        batch = args.gather_batch   # --force-gather: the one-rank pipeline with batched gathers
    gpu, fg = make_pipeline(root_weight, batch)

    # exact ray / step / node-visit counts of this frame (deterministic; a stats frame is never timed)
    if fg is not None:
        fg.bind(gpu, 0)
    gpu.render(MODE, stats=True, **(dict(rkw, variant=0) if rkw.get("variant") == 4 else rkw))   # (the persistent grid has no stats form)
    st = gpu.stats()
    counts = all_reduce(torch.tensor([st.primary_rays, st.secondary_rays], dtype=torch.int64, device="cuda"))
    rays_per_frame = int(counts[0] + counts[1])

    # set-up, off the clock like the scene build and the share tuning: let the clocks settle (a cold chip runs the first
    # ~hundred frames ~4 % slower), then the W warm-up frames the caller asked for
    if not args.rehearse_on_one_gpu:
        torch.cuda.synchronize()
        t_probe = time.perf_counter()
        run_frames(gpu, fg, 2)
        torch.cuda.synchronize()
        per_frame = (time.perf_counter() - t_probe) / 2
        run_frames(gpu, fg, int(min(300, max(2, 0.05 / max(per_frame, 1e-6)))))   # ~50 ms, at most 300 frames
    run_frames(gpu, fg, args.warmup)
    gpu.stats()  # drop the warm-up frames' kernel timings
    if sharded:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_frames(gpu, fg, args.steps)   # every timed frame is gathered and assembled on rank 0 before the clock stops
    if sharded:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dt = float(all_reduce(torch.tensor([dt], dtype=torch.float64, device="cuda"), dist.ReduceOp.MAX if world > 1 else None)[0])

    # per-kernel durations over exactly the timed frames: HIP events on the stream the kernels ran on
    kst = gpu.stats()
    if sharded and rank == 0 and os.environ.get("VRT_BENCH_VERIFY", "1") == "1":
        # off the clock: the assembled frame must equal an unsharded render of the same frame on this GPU
        ref = Gpu(sc.world.max_nodes(), sc.world.size_in_chunks(), sc.size, device=local_rank)
        ref.upload_world(sc.world, sc.materials)
        ref.write_cam_data(sc.cam)
        ref.write_settings(sc.settings)
        ref.render(MODE, **rkw)
        r_rgb, r_ids, _ = ref.read_output()
        from voxelraytracing_amd.shard import texels_to_frame
        import numpy as np
        a_rgb, a_ids = texels_to_frame(fg.frame.cpu().numpy().view(np.uint32))
        if not (np.array_equal(a_ids, r_ids) and np.array_equal(a_rgb, r_rgb)):
            raise SystemExit("gathered frame differs from the unsharded render")
        ref.close()
    if rank != 0:
        dist.destroy_process_group()
        return
    ai = gpu.accel_info()
    derived = {"available": bool(ai.available), "cells": int(ai.cells), "bricks": int(ai.bricks), "bytes": int(ai.bytes),
               "builds": int(ai.builds), "last_build_ms": round(float(ai.last_build_ms), 4)}
    b_primary, b_shadow, b_fused = algorithmic_bytes(st, args.width, args.height)  # rank 0's own launches (its shard when N > 1)
    ms_p = kst.sum_ms_primary / max(kst.frames, 1)
    ms_s = kst.sum_ms_secondary / max(kst.frames, 1)
    fused = args.mode == "shadow" and args.variant in (0, 4) and ms_s == 0.0  # one launch: no second kernel was timed
    if fused:
        dom_name, dom_bytes, dom_ms = "primary_shadow_march", b_fused, ms_p
    elif args.mode == "path":   # not the headline: first launch vs all bounce launches + finish; bytes as for shadow rays
        dom_name, dom_bytes, dom_ms = ("path_primary_march", b_primary, ms_p) if ms_p >= ms_s else ("path_bounce_marches", b_shadow, ms_s)
    else:
        dom_name, dom_bytes, dom_ms = ("primary_march", b_primary, ms_p) if ms_p >= ms_s else ("shadow_march", b_shadow, ms_s)
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    # frames in flight: consecutive launches overlap on the chip, so a launch's own begin-to-end time (what the events
    # and rocprofv3 report) is longer than the frame period; `achieved` stays bytes per launch / that duration,
    # `achieved_aggregate` is what the in-flight launches deliver together
    in_flight = args.frames_in_flight if (world == 1 and not sharded and (fused or args.mode == "primary")) else 1
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(dom_name)
        except Exception:
            traffic = None

    # what actually binds the kernel: VALU issue.  SIMD cycles available per VALU wave-instruction at the
    # measured frame period (1 024 SIMDs at the nominal 2.4 GHz); the simple class issues in ~2.5 cycles, the half-rate
    # class in ~4.2 (tools/valu_rates.hip), so a value between the two means the SIMDs issue VALU back to back
    valu = None
    try:
        n_valu = json.load(open(tpath)).get("valu_wave_instructions", {}).get(dom_name)
        if n_valu and world == 1 and not sharded:
            valu = {"wave_instructions_per_launch": n_valu, "simds": 1024, "clock_ghz": 2.4,
                    "simd_cycles_per_instruction": dt / args.steps * 2.4e9 * 1024 / n_valu,
                    "issue_cost_cycles": {"simple": 2.5, "half_rate": 4.2, "transcendental": 8.1}}
    except Exception:
        valu = None

    out = {
        "metric": "Mrays/s at 1920x1080, 1 primary + 1 shadow ray",
        "value": rays_per_frame * args.steps / dt / 1e6,
        "unit": "Mrays/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"{'C2' if (args.chunks, args.width, args.height) == (8, 1920, 1080) else 'C2-family'}: {args.width}x{args.height} frame, {args.chunks}x{args.chunks}x{args.chunks}-chunk procedural SVO "
                               f"world (seed 1), 1 primary + 1 shadow ray per solid hit",
                   "rays_per_frame_actual": rays_per_frame, "rays_per_frame_nominal": 2 * args.width * args.height,
                   "sharding": ("whole frame" if not sharded else "whole frame through the one-rank gather pipeline") if world == 1 else
                               f"8x8 tiles interleaved over {world} ranks ({root_weight} of every {root_weight + world - 1} to the gather root, "
                               f"which renders them in place) + RCCL gather of the other ranks' tile buffers "
                               f"({'8-byte records shaded at the root' if MODE != MODE_PATH and args.variant == 0 else '16-byte texels'}) to rank 0",
                   "root_weight": root_weight, "root_weight_tuning_ms_per_frame": tuning,
                   "frames_per_gather": batch, "frames_per_gather_tuning_ms_per_frame": batch_tuning,
                   "kernel_variant": args.variant, "derived_tables": derived},
        "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "launches_in_flight": in_flight,
                     "achieved_aggregate": achieved * in_flight, "frame_period_ms": dt / args.steps * 1e3, "traffic": traffic,
                     "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_ms": dom_ms,
                     "kernels_ms": ({"primary_shadow_march": ms_p} if fused else
                                    {"path_primary_march": ms_p, "path_bounce_marches": ms_s} if args.mode == "path" else
                                    {"primary_march": ms_p, "shadow_march": ms_s}),
                     "frames_timed": kst.frames},
    }
    if valu is not None:
        out["valu_issue"] = valu
    if args.mode != "shadow":
        out["metric"] = f"Mrays/s at {args.width}x{args.height}, mode {args.mode}" + (f" {args.bounces} bounces {args.spp} spp" if args.mode == "path" else "")
        out["config"]["workload"] = out["config"]["workload"].replace("C2:", "non-headline:").replace("1 primary + 1 shadow ray per solid hit", f"mode {args.mode}")
    if world == 1 and not args.no_cpu_baseline and args.mode == "shadow":
        out["cpu_baseline"] = cpu_baseline(sc, args, rays_per_frame)
    print(json.dumps(out), flush=True)
    if sharded:
        dist.destroy_process_group()


def usable_cores():
    """CPU share of this process: the affinity mask, capped by the cgroup CPU quota when there is one."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return min(n, int(os.environ.get("VRT_CPU_BASELINE_THREADS", "64")))


def cpu_baseline(sc, args, rays_per_frame):
    """The oracle (a port, not the reference: the reference is WGSL on wgpu and has no CPU tracer) timed on
    this box's host cores over a bounded sample of the same workload: whole frames of the same scene and
    camera, repeated for about 3 s of wall time (>= 2 frames)."""
    from oracle import orc
    o = orc.from_package_scene(sc)
    cores = usable_cores()
    times = []
    t_all = time.perf_counter()
    while len(times) < 2 or (time.perf_counter() - t_all < 3.0 and len(times) < 50):
        t0 = time.perf_counter()
        _, _, _, cst = o.render(orc.MODE_PRIMARY_SHADOW, args.width, args.height, threads=cores)
        times.append(time.perf_counter() - t0)
    rays = cst.primary_rays + cst.secondary_rays
    assert rays == rays_per_frame, "oracle and GPU disagree on the number of rays launched"
    dt = sorted(times)[len(times) // 2]
    t0 = time.perf_counter()   # SURVEY.md §8d also asks for a 1-thread figure: one frame
    o.render(orc.MODE_PRIMARY_SHADOW, args.width, args.height, threads=1)
    dt1 = time.perf_counter() - t0
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port", "value_1_thread": rays / dt1 / 1e6,
            "sample": f"{len(times)} full {args.width}x{args.height} frames of the same scene (median {dt:.3f} s/frame, "
                      f"{sum(times):.1f} s total), OpenMP dynamic over 8-row bands"}


if __name__ == "__main__":
    main()
