#!/usr/bin/env python3
"""bench.py — Mrays/s of the SVO ray-march on BASELINE.json's headline config.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one frame: 1920x1080, 8x8x8-chunk procedural SVO world, 1 primary ray per pixel + 1 shadow ray per solid
hit (config C2; scene, derived tables and ndc tables resident in HBM before the timed region).  A timed frame issues
the reference frame loop's whole seam (clientdesktop/src/main.rs:426-453: settings, camera, chunk_roots, world data,
dispatch) with a camera that orbits slowly, so the frames in flight are different frames; `value_fixed_camera` is the
same loop with the camera standing still and `value_1_in_flight` the standing camera with one launch at a time (a view
at rest: its tiles are launched longest first; `--frames-in-flight 1` gives the orbiting one-at-a-time loop, a kept block order).

N > 1: the frame is sharded by interleaved 8x8 screen tiles (total work fixed: "strong" scaling), one process per GPU
over RCCL (under torch.distributed.run, or started by this script itself when WORLD_SIZE is not set), or — with
--single-process — one process driving N devices through the C ABI's own multi-device context (vrt_config.device_ids).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
N_SIMD = 1024               # 256 CUs x 4 SIMD-32
NOMINAL_GHZ = 2.4
MIN_TIMED_SECONDS = 0.05    # a timed window shorter than this is followed by a second, longer one (see `steps_timed`)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--chunks", type=int, default=8, help="world size in chunks (8 = config C2, 16 = C3)")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--mode", choices=["shadow", "primary", "path"], default="shadow",
                    help="shadow = the headline metric (config C2); path = the C4/C5 kernel family (not the headline)")
    ap.add_argument("--spp", type=int, default=1)
    ap.add_argument("--bounces", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--frames-in-flight", type=int, default=2,
                    help="frames the backend keeps in flight (vrt_set_frames_in_flight; 1 = one launch at a time)")
    ap.add_argument("--fixed-camera", action="store_true", help="headline loop with a standing camera and no per-frame seam calls")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra legs (fixed camera, 1 in flight, clock probe)")
    ap.add_argument("--exact-steps", action="store_true", help="time the requested steps only, however short (no second, >= 50 ms window)")
    ap.add_argument("--settle-seconds", type=float, default=2.0, help="upper bound of the clock-settling frames before the warm-up (0 = none)")
    ap.add_argument("--root-weight", type=int, default=0,
                    help="N > 1: tiles per period dealt to the gather root (vrt_config.shard_root_weight); 0 = measure "
                         "a few candidates off the clock and keep the fastest")
    ap.add_argument("--gather-batch", type=int, default=0,
                    help="N > 1: frames per gather (FrameGather.batch); 0 = measure {1, 2, 4, 8} off the clock and keep the fastest")
    ap.add_argument("--single-process", action="store_true",
                    help="N > 1: one process, N devices behind one vrt context (vrt_config.device_ids; peer stores over xGMI)")
    ap.add_argument("--force-gather", action="store_true",
                    help="development only: with --gpus 1, still run the pipelined RCCL gather + assemble path (one-rank group)")
    ap.add_argument("--init-timeout", type=float, default=120.0,
                    help="N > 1: seconds the process group's rendezvous and the first collective may take before the rank gives "
                         "up (exit status 3, the failing rank named on stderr)")
    ap.add_argument("--launch-retries", type=int, default=1,
                    help="N > 1 started by this script: how often ranks that never got through the rendezvous / first collective "
                         "(exit status 3) are started again — always as fresh processes")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="development only: run the N > 1 code path with every rank / device on cuda:0 (multi-process: a gloo "
                         "gather staged through host memory, RCCL refuses two ranks on one device); never for reported numbers")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------------------
# N > 1 without a launcher: start one process per GPU ourselves — before anything touches torch or the GPU
# ---------------------------------------------------------------------------------------------------------------------
def self_launch(args) -> int:
    """One process per GPU, started here (the parent never touches torch or the GPU).  A rank that dies takes the others
    with it and is named; ranks that never got through the rendezvous or the first collective (exit status 3) are started
    again, as fresh processes on a fresh port, --launch-retries times."""
    for attempt in range(args.launch_retries + 1):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        procs = []
        for r in range(args.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            # rank 0's stdout carries the JSON line (into a file: nothing here blocks on a pipe); the other ranks' stdout joins stderr
            out_f = tempfile.TemporaryFile() if r == 0 else None
            procs.append((subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                           stdout=out_f if r == 0 else sys.stderr), out_f))
        rcs = [None] * args.gpus
        deadline = time.time() + 3300.0
        while any(rc is None for rc in rcs):
            for r, (pr, _) in enumerate(procs):
                if rcs[r] is None:
                    rcs[r] = pr.poll()
            failed = [r for r, rc in enumerate(rcs) if rc not in (None, 0)]
            if failed or time.time() > deadline:
                time.sleep(2.0)   # (the others may be on their way out with a status of their own)
                for r, (pr, _) in enumerate(procs):
                    if rcs[r] is None:
                        rcs[r] = pr.poll()
                    if rcs[r] is None:   # a rank that lost its peers would wait for them for ever: ended here, by its own PID
                        pr.kill()
                        rcs[r] = pr.wait()
                        rcs[r] = "killed (a peer failed)" if failed else "killed (overall time limit)"
                break
            time.sleep(0.05)
        bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
        if not bad:
            procs[0][1].seek(0)
            sys.stdout.write(procs[0][1].read().decode())
            sys.stdout.flush()
            return 0
        print(f"bench.py: attempt {attempt + 1}: ranks failed (rank, exit status): {bad}", file=sys.stderr)
        if not any(rc == 3 for _, rc in bad) or attempt == args.launch_retries:
            return 1
        print("bench.py: the rendezvous / first collective did not complete: starting fresh processes", file=sys.stderr)
    return 1


def algorithmic_bytes(st):
    """SURVEY.md §8d / DESIGN.md §Algorithmic bytes, from exact step / node-visit counts: what the *reference algorithm*
    moves.  march step: 4 B chunk_roots entry + 2 B x L node words + 4 B material is_liquid = 8 + 2L; + 16 B texel per
    primary ray; two-launch variants: + 16 B hit record written and read per secondary ray."""
    p_steps, p_vis = st.primary_steps, st.primary_node_visits
    s_steps, s_vis = st.steps - p_steps, st.node_visits - p_vis
    primary = 8 * p_steps + 2 * p_vis + 16 * st.primary_rays + 16 * st.secondary_rays
    shadow = 8 * s_steps + 2 * s_vis + 16 * st.secondary_rays
    fused = 8 * st.steps + 2 * st.node_visits + 16 * st.primary_rays
    return primary, shadow, fused


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.single_process:
        sys.exit(self_launch(args))

    # stdout carries ONE JSON line and nothing else: RCCL prints a version banner to file descriptor 1 when a communicator is
    # created (and other native code might), so descriptor 1 points at stderr for the whole run and the line goes to the
    # saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    from voxelraytracing_amd import Gpu, MODE_PATH, MODE_PRIMARY, MODE_PRIMARY_SHADOW, _ffi, scenes
    from voxelraytracing_amd import graphics as g
    from voxelraytracing_amd.shard import FrameGather

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = 1 if args.single_process else int(os.environ.get("WORLD_SIZE", "1"))
    if not args.single_process and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
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
        # the rendezvous and the first collective under a clock: a rank that cannot reach its peers says so and leaves with
        # status 3 instead of waiting for ever (the launcher — this script's, or torch's — then ends the others)
        import datetime

        def give_up(what):
            print(f"bench.py: rank {rank} of {world}: {what} did not complete within {args.init_timeout:.0f} s "
                  f"(MASTER {os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')}, device {local_rank})", file=sys.stderr, flush=True)
            os._exit(3)
        watchdog = threading.Timer(args.init_timeout, give_up, ("the process group's rendezvous",))
        watchdog.daemon = True
        watchdog.start()
        try:
            if args.rehearse_on_one_gpu:
                dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=args.init_timeout))
            else:
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(seconds=args.init_timeout))
        except Exception as e:
            print(f"bench.py: rank {rank} of {world}: init_process_group failed: {e}", file=sys.stderr, flush=True)
            os._exit(3)
        watchdog.cancel()

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

    # ---- who is here: every rank adds a one (RCCL's first collective, under the same clock), and what the devices can reach ----
    ranks_seen = 1
    if sharded:
        watchdog = threading.Timer(args.init_timeout, give_up, ("the first collective (an all-reduce of ones)",))
        watchdog.daemon = True
        watchdog.start()
        ranks_seen = int(all_reduce(torch.ones(1, dtype=torch.int64, device="cuda"))[0])
        torch.cuda.synchronize()
        watchdog.cancel()
        if ranks_seen != world:
            print(f"bench.py: rank {rank}: the all-reduce of ones gave {ranks_seen}, WORLD_SIZE is {world}", file=sys.stderr, flush=True)
            os._exit(3)
    n_vis = torch.cuda.device_count()
    peer = [[bool(torch.cuda.can_device_access_peer(i, j)) for j in range(n_vis)] for i in range(n_vis)] if n_vis > 1 else []
    links = {"devices_visible": n_vis,   # hipDeviceCanAccessPeer over every ordered pair of the devices this process sees
             "peer_access_pairs": sum(peer[i][j] for i in range(n_vis) for j in range(n_vis) if i != j) if n_vis > 1 else 0,
             "of_pairs": n_vis * (n_vis - 1),
             "to_device_0": [i for i in range(1, n_vis) if peer[i][0]] if n_vis > 1 else []}

    # ---- scene (deterministic, built by every rank) and upload: off the clock ----
    MODE = {"shadow": MODE_PRIMARY_SHADOW, "primary": MODE_PRIMARY, "path": MODE_PATH}[args.mode]
    rkw = dict(spp=args.spp, seed=0) if MODE == MODE_PATH else dict(variant=args.variant)
    sc = scenes.procedural(args.chunks, (args.width, args.height), MODE)
    if MODE == MODE_PATH:
        sc.settings.max_ray_bounces = args.bounces
        scenes._diffuse(sc.materials)
    # the orbit: ORBIT camera positions on a 6-voxel circle around the scene's eye, the view swinging +-8 degrees
    ORBIT = 48
    cams = []
    for k in range(ORBIT):
        a = 2.0 * math.pi * k / ORBIT
        eye = (sc.eye[0] + 6.0 * math.cos(a), sc.eye[1] + 1.5 * math.sin(2 * a), sc.eye[2] + 6.0 * math.sin(a))
        rot = (sc.rot[0] + 3.0 * math.sin(a), sc.rot[1] + 8.0 * math.sin(a), sc.rot[2])
        cams.append((rot, eye))
    world_data = sc.world.world_data()
    side = None
    if sharded:
        # One non-default torch stream carries everything of this rank: the backend's kernels (vrt_set_stream), the
        # tensors' initialisation and the RCCL calls' stream dependencies.  (The default stream's handle is 0, which
        # vrt_set_stream reads as "use the context's own stream" — kernels there would not be ordered with RCCL.)
        side = torch.cuda.Stream(device=local_rank)
        torch.cuda.set_stream(side)
        assert side.cuda_stream != 0

    devices = None
    if args.single_process and args.gpus > 1:
        devices = [0] * args.gpus if args.rehearse_on_one_gpu else list(range(args.gpus))

    def make_pipeline(root_weight, batch=1, in_flight=None):
        """(backend context, FrameGather) for this rank.  N > 1: the root renders its own tiles straight into the
        row-major frame (VRT_FLAG_ROW_MAJOR) and takes root_weight tiles of every root_weight + N - 1."""
        in_place = world > 1
        compact = in_place and MODE != MODE_PATH and args.variant == 0   # 8 B/pixel over the links, shaded at the root
        # (one context over N devices: 8-byte records for primary(+shadow) frames of the default march, 16-byte texels otherwise)
        kw = dict(devices=devices, root_weight=root_weight, texel_messages=MODE == MODE_PATH or args.variant != 0) if devices else \
            dict(shard_rank=rank, shard_count=world, tile_major=sharded and not (in_place and rank == 0), root_weight=root_weight,
                 row_major=in_place and rank == 0, compact=compact and rank != 0)
        gp = Gpu(sc.world.max_nodes(), sc.world.size_in_chunks(), sc.size, device=local_rank, **kw)
        gp.upload_world(sc.world, sc.materials)
        gp.write_cam_data(sc.cam)
        gp.write_settings(sc.settings)
        gp.set_frames_in_flight(args.frames_in_flight if in_flight is None else in_flight)
        f = None
        if sharded:
            gp.set_stream(side.cuda_stream)
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
        return gp, f

    frame_no = [0]

    def seam(gp, fixed):
        """What draw_frame does before the dispatch (main.rs:426-449): settings, camera, chunk_roots, world data."""
        if fixed:
            return
        rot, eye = cams[frame_no[0] % ORBIT]
        frame_no[0] += 1
        gp.write_settings(sc.settings)
        gp.write_cam_data(g.cam_data_create(rot, eye, 70.0, (float(args.width), float(args.height))))
        gp.write_chunk_roots(sc.world.chunk_roots(), tag=sc.world.roots_generation())   # a fresh S^3 table every frame, as the reference does (world.rs:154-159)
        gp.write_world_data(world_data)

    def run_frames(gp, f, n, fixed=False):
        for _ in range(n if (f is None or args.rehearse_on_one_gpu) else 0):
            seam(gp, fixed)
            if f is None:
                gp.render(MODE, **rkw)
            else:
                f.bind(gp, 0)
                gp.render(MODE, **rkw)
                f.gather()
                if rank == 0:
                    f.assemble(gp, 0)
        if f is not None and not args.rehearse_on_one_gpu:
            # the in-place root's own tiles never feed the collective: they run on the backend's in-flight streams
            own = dict(own_streams=True) if (f.in_place and rank == 0 and MODE != MODE_PATH) else {}

            def one():
                seam(gp, fixed)
                gp.render(MODE, **rkw, **own)
            left = n
            while left > 0:   # the gather of a batch overlaps the render of the next
                f.submit(gp, one, min(left, f.batch))
                left -= f.batch
            f.drain(gp)

    def sync():
        if sharded:
            dist.barrier()
        torch.cuda.synchronize()

    submit = [0.0]

    def timed(gp, f, n, fixed=False):
        sync()
        t0 = time.perf_counter()
        run_frames(gp, f, n, fixed)
        submit[0] = time.perf_counter() - t0     # the host's share: every call of the loop has returned, the GPU may lag behind
        sync()
        dt = time.perf_counter() - t0
        return float(all_reduce(torch.tensor([dt], dtype=torch.float64, device="cuda"), dist.ReduceOp.MAX if world > 1 else None)[0])

    # The interpreter's cyclic garbage collector is off from here (the share tuning's trials included) to the end of the
    # measurements, as `timeit` runs its statements: with torch loaded a full collection is a 40 ms pause of the submitting
    # thread, it comes every ~600 frames of the gather pipeline, and after the settling frames it used to fall into a
    # 20-step timed region — 2.5 ms per frame instead of 0.15 (found with `--force-gather --steps 20 --warmup 5`;
    # `--warmup 50` moved it out of the region).
    import gc
    gc.collect()
    gc.disable()
    # ---- N > 1 (one process per GPU): how much of the frame the gather root should trace itself (off the clock) ----
    root_weight, batch, tuning, batch_tuning = 1, 1, None, None
    if world > 1:
        def trial(w0, b):
            gp, f = make_pipeline(w0, b)
            run_frames(gp, f, 8, True)
            dt = timed(gp, f, 48, True)
            gp.close()
            return dt / 48 * 1e3      # the all-reduced times are identical on every rank
        rehearsal = args.rehearse_on_one_gpu   # (its staged gather moves whole message buffers: one frame per gather)
        batch = args.gather_batch if args.gather_batch > 0 else (1 if rehearsal else 4)
        if args.root_weight > 0:
            root_weight = args.root_weight
        else:
            # (32 and 128: the root keeps nearly the whole frame — the floor if the links turn out slower than the model's 60 GB/s,
            # so that a sharded run is never much slower than one GPU)
            tuning = {w0: trial(w0, batch) for w0 in (1, 2, 3, 4, 6, 8, 12, 16, 32, 128)}
            root_weight = min(tuning, key=lambda k: (tuning[k], k))
        if args.gather_batch == 0 and not rehearsal:
            batch_tuning = {b: trial(root_weight, b) for b in (1, 2, 4, 8)}
            batch = min(batch_tuning, key=lambda k: (batch_tuning[k], k))
    elif sharded and args.gather_batch > 0:
        batch = args.gather_batch   # --force-gather: the one-rank pipeline with batched gathers
    elif devices:
        # one context over N devices: the share of the root from the balance of DESIGN.md section 7 (shard.root_weight_model) on
        # the unsharded frame's own time, measured here off the clock — a 34 ms path-trace frame is bound by its render and
        # wants even shares, a 0.1 ms primary + shadow frame by its messages' links and wants the root to keep more
        if args.root_weight > 0:
            root_weight = args.root_weight
        else:
            from voxelraytracing_amd.shard import root_weight_model
            probe = Gpu(sc.world.max_nodes(), sc.world.size_in_chunks(), sc.size, device=0)
            probe.upload_world(sc.world, sc.materials)
            probe.write_cam_data(sc.cam)
            probe.write_settings(sc.settings)
            for _ in range(30 if MODE != MODE_PATH else 2):   # (first frames create streams, event pools and the derived tables)
                probe.render(MODE, **rkw)
            probe.synchronize()
            n_p, t0 = 0, time.perf_counter()
            while n_p < 50 and (n_p < 2 or time.perf_counter() - t0 < 0.1):
                probe.render(MODE, **rkw)
                n_p += 1
            probe.synchronize()
            t1_ms = (time.perf_counter() - t0) / n_p * 1e3
            probe.close()
            per_px = 16 if (MODE == MODE_PATH or args.variant != 0) else 8
            root_weight = root_weight_model(args.gpus, t1_ms, args.width * args.height * per_px)
    gpu, fg = make_pipeline(root_weight, batch)

    # ---- exact ray / step / node-visit counts (deterministic; a stats frame is never timed): the scene's own camera,
    # and the ray count of every camera of the orbit ----
    def stats_frame(cam):
        gpu.write_cam_data(cam)
        if fg is not None:
            fg.bind(gpu, 0)
        gpu.render(MODE, stats=True, **rkw)
        return gpu.stats()
    orbit_rays = []
    for rot, eye in cams:
        s_ = stats_frame(g.cam_data_create(rot, eye, 70.0, (float(args.width), float(args.height))))
        orbit_rays.append([s_.primary_rays, s_.secondary_rays])
    st = stats_frame(sc.cam)
    steps_img = None
    if not sharded and not devices and MODE != MODE_PATH:
        steps_img = gpu.read_steps()    # per pixel: primary | shadow << 16 march steps of the fixed-camera frame
    counts = all_reduce(torch.tensor(orbit_rays + [[st.primary_rays, st.secondary_rays]], dtype=torch.int64, device="cuda")).cpu().numpy()
    orbit_rays = counts[:-1].sum(axis=1)
    rays_fixed = int(counts[-1].sum())

    # ---- set-up, off the clock like the scene build and the share tuning: let the machine settle — the clocks, and the
    # backend's pool of timing events (512 frames' worth; the host creates them on first use, a few microseconds each) —
    # until consecutive 200-frame windows are within 1 % of each other, at least 3 windows, at most 2 s; then the W
    # warm-up frames the caller asked for ----
    fixed = args.fixed_camera
    settle = {"windows": 0, "frames": 0, "ms_per_frame": None, "converged": False}
    if not args.rehearse_on_one_gpu and args.settle_seconds > 0:
        t_end = time.perf_counter() + args.settle_seconds
        prev = None
        while time.perf_counter() < t_end:
            cur = timed(gpu, fg, 200, fixed) / 200
            settle["windows"] += 1
            settle["frames"] += 200
            settle["ms_per_frame"] = cur * 1e3
            if prev is not None and settle["windows"] >= 3 and abs(cur - prev) <= 0.01 * prev:
                settle["converged"] = True
                break
            prev = cur
    run_frames(gpu, fg, args.warmup, fixed)
    gpu.stats()  # drop the warm-up frames' kernel timings
    gpu.issue_profile()   # ... and their host times (vrt_get_issue_profile / FrameGather.host_profile: the timed frames' own)
    if fg is not None:
        fg.reset_host_profile()
    frame_no[0] = 0
    dt = timed(gpu, fg, args.steps, fixed)   # every timed frame is gathered and assembled on rank 0 before the clock stops
    host_submit_ms = submit[0] / args.steps * 1e3
    rays_total = rays_fixed * args.steps if fixed else int(sum(int(orbit_rays[i % ORBIT]) for i in range(args.steps)))
    # A short request (the driver's 20 steps are 1.9 ms of frames: two of them pipeline fill and drain, and one scheduling hiccup
    # of the host is a tenth of it) is followed by a second timed window of the same loop, long enough for >= 50 ms of frames,
    # bracketed like the first (barrier + synchronise on both sides, maximum over ranks).  `value` / `ms_per_step` are then the
    # LONG window's, `steps_timed` its length; the requested K steps alone stay in `value_requested_steps` /
    # `ms_per_step_requested_steps`.  A request that is itself >= 50 ms of frames is timed once, as before.
    requested = {"steps": args.steps, "value": rays_total / dt / 1e6, "ms_per_step": dt / args.steps * 1e3, "host_submit_ms_per_step": host_submit_ms}
    steps_timed = args.steps
    if dt < MIN_TIMED_SECONDS and not args.exact_steps:
        steps_timed = max(args.steps, int(np.ceil(MIN_TIMED_SECONDS / (dt / args.steps))))
        first = frame_no[0]
        dt = timed(gpu, fg, steps_timed, fixed)
        host_submit_ms = submit[0] / steps_timed * 1e3
        rays_total = rays_fixed * steps_timed if fixed else int(sum(int(orbit_rays[(first + i) % ORBIT]) for i in range(steps_timed)))
    kst = gpu.stats()   # per-kernel durations over the timed frames (both windows): HIP events on the streams the kernels ran on
    issue = gpu.issue_profile()                                   # what issuing those frames cost this rank's host thread
    gather_host = fg.host_profile() if fg is not None else None   # ... and its collectives

    # ---- extra legs, off the headline clock (N = 1): standing camera, one launch at a time, the clock the march runs at ----
    extras = {}
    xs = steps_timed   # (the extra legs' windows are as long as the headline's)
    if world == 1 and not sharded and not devices and not args.no_extras:
        if not fixed:
            gpu.write_cam_data(sc.cam)
            run_frames(gpu, fg, 50, True)
            d2 = timed(gpu, fg, xs, True)
            gpu.stats()
            extras["value_fixed_camera"] = rays_fixed * xs / d2 / 1e6
            extras["ms_per_step_fixed_camera"] = d2 / xs * 1e3
        if args.frames_in_flight != 1:
            gpu.set_frames_in_flight(1)
            gpu.write_cam_data(sc.cam)
            run_frames(gpu, fg, 50, True)
            d1 = timed(gpu, fg, xs, True)
            gpu.stats()
            # the lone launch's own duration: every frame timed (VRT_RENDER_TIMED) — between untimed neighbours a timed
            # launch's begin stamp falls into its predecessor's tail
            for _ in range(300):
                gpu.render(MODE, timed=True, **rkw)
            k1 = gpu.stats()
            extras["value_1_in_flight"] = rays_fixed * xs / d1 / 1e6
            extras["ms_per_step_1_in_flight"] = d1 / xs * 1e3
            extras["avg_launch_ms_1_in_flight"] = k1.sum_ms_primary / max(k1.frames, 1)
            if MODE == MODE_PATH:   # per frame: the primary launch(es) and the bounce launch(es), each alone on the GPU
                extras["avg_bounce_launches_ms_1_in_flight"] = k1.sum_ms_secondary / max(k1.frames, 1)
            if not fixed:
                # ... and what a host that renders one moving-camera frame at a time gets: the orbit, the whole seam per frame,
                # a block order kept while the camera stays within what its dilation covers (vrt.h: vrt_set_frames_in_flight)
                frame_no[0] = 0
                run_frames(gpu, fg, 50, False)
                frame_no[0] = 0
                d1o = timed(gpu, fg, xs, False)
                gpu.stats()
                extras["value_1_in_flight_orbit"] = int(sum(int(orbit_rays[i % ORBIT]) for i in range(xs))) / d1o / 1e6
                extras["ms_per_step_1_in_flight_orbit"] = d1o / xs * 1e3
            gpu.set_frames_in_flight(args.frames_in_flight)
        if MODE == MODE_PRIMARY_SHADOW and args.variant == 0:
            # the clock: right behind the timed load, frames of the probe build (the same kernel + two stamps per wave)
            run_frames(gpu, fg, 100, True)
            gpu.stats()
            for _ in range(200):
                gpu.render(MODE, stats=2)
            cs = gpu.stats()
            if cs.clock_ref_ticks:
                extras["shader_clock_ghz"] = cs.clock_shader_ticks / cs.clock_ref_ticks * 0.1   # x 100 MHz

    # ---- the client's real frame (main.rs:199, 426-454), off the headline clock: a 30^3-chunk grid around the player with a
    # world.min of mixed sign, the UNTAGGED chunk_roots rewrite of a fresh table every frame (main.rs:446: 27 000 entries), the
    # dispatch and the blit into the window's image (vrt_present_device, main.rs:454) — one and two frames in flight ----
    operating_point = None
    if world == 1 and not sharded and not devices and not args.no_extras and args.mode == "shadow" and args.variant == 0 and not fixed:
        from voxelraytracing_amd.world import ClientWorld, gen_height
        S_OP, player = 30, (7, -3, 11)
        w_op = ClientWorld(player, 1 << 27, S_OP)
        w_op.generate(0, 1)
        px, pz = player[0] * 32 + 16, player[2] * 32 + 16
        eye_op = (px + 0.5, float(gen_height(1, px, pz)) + 24.5, pz + 0.5)
        gp = Gpu(w_op.max_nodes(), S_OP, (args.width, args.height), device=local_rank)
        gp.upload_world(w_op, sc.materials)
        gp.write_settings(sc.settings)
        wd_op = w_op.world_data()
        cams_op = []
        for k in range(ORBIT):
            a = 2.0 * math.pi * k / ORBIT
            cams_op.append(g.cam_data_create((20.0 + 3.0 * math.sin(a), 35.0 + 8.0 * math.sin(a), 0.0),
                                             (eye_op[0] + 6.0 * math.cos(a), eye_op[1] + 1.5 * math.sin(2 * a), eye_op[2] + 6.0 * math.sin(a)),
                                             70.0, (float(args.width), float(args.height))))
        rays_op = []
        for cam in cams_op:
            gp.write_cam_data(cam)
            gp.render(MODE, stats=True)
            s_ = gp.stats()
            rays_op.append(s_.primary_rays + s_.secondary_rays)

        decl = [None]

        def client_frames(n):
            for i in range(n):
                gp.write_settings(sc.settings)                       # main.rs:428
                if decl[0] is not None:
                    gp.set_presentation((args.width, args.height), **decl[0])   # :429-432 — screen_size + crosshair, every frame as the reference writes them
                gp.write_cam_data(cams_op[i % ORBIT])                # :439
                gp.write_chunk_roots(w_op.chunk_roots())             # :446 — a fresh table, no tag: the backend compares 27 000 roots
                gp.write_world_data(wd_op)                           # :447-449
                gp.render(MODE)                                      # :452-453
                gp.present_device((args.width, args.height))         # :454
        operating_point = {"world": f"{S_OP}^3 chunks around player chunk {player} (main.rs:199), world.min {tuple(w_op.min_voxel())}, "
                                    f"{w_op.populated_count()} chunks with nodes", "frames": 400,
                           "per_frame": "vrt_set_settings, vrt_set_presentation (the blit's screen_size + crosshair uniforms), vrt_set_camera, vrt_write_chunk_roots (untagged, a fresh 27 000-entry table), vrt_set_world, "
                                        "vrt_render (primary + shadow), vrt_present_device at the frame's size"}
        # the window has the texture's size (the reference keeps its texture at 1080 rows and the window's aspect, main.rs:255-262):
        # declared (vrt_set_presentation), the frame's own launch stores the window's image and vrt_present_device launches nothing —
        # `*_in_flight`; `*_blit_launch`: undeclared, the blit as a launch of its own (rounds 4-5); `*_window_only`: declared with
        # VRT_PRESENT_SKIP_TEXELS (no 16-byte texel: a client that never reads back)
        operating_point["per_frame"] += " (declared with vrt_set_presentation: the frame's launch stores the window's pixels, the present call launches nothing)"
        for tag, d_ in (("", dict()), ("_blit_launch", None), ("_window_only", dict(skip_texels=True))):
            decl[0] = d_
            gp.set_presentation((args.width, args.height), **(d_ if d_ is not None else dict(off=True)))
            for nf in (1, 2):
                gp.set_frames_in_flight(nf)
                client_frames(100)
                gp.synchronize()
                t0 = time.perf_counter()
                client_frames(400)
                t_host = time.perf_counter() - t0
                gp.synchronize()
                t_all = time.perf_counter() - t0
                operating_point[f"{nf}_in_flight{tag}"] = {"ms_per_frame": t_all / 400 * 1e3, "host_us_per_frame": t_host / 400 * 1e6,
                                                           "value": sum(rays_op[i % ORBIT] for i in range(400)) / t_all / 1e6, "unit": "Mrays/s"}
        gp.close()

    one_gpu_ms = [None]   # N > 1: the unsharded frame on this rank's GPU, timed off the clock (config.expected_scaling's input)

    def time_unsharded(ref):
        # (other ranks may still be busy on a rehearsal's one GPU: the figure is then an upper bound, and says so)
        for _ in range(30 if MODE != MODE_PATH else 2):
            ref.render(MODE, **rkw)
        ref.synchronize()
        n_done, t0 = 0, time.perf_counter()
        while n_done < 200 and (n_done < 3 or time.perf_counter() - t0 < 0.15):
            ref.render(MODE, **rkw)
            n_done += 1
        ref.synchronize()
        one_gpu_ms[0] = (time.perf_counter() - t0) / n_done * 1e3

    if sharded and rank == 0 and os.environ.get("VRT_BENCH_VERIFY", "1") == "1":
        # off the clock: the assembled frame must equal an unsharded render of the same frame on this GPU
        last = cams[(frame_no[0] - 1) % ORBIT] if not fixed else None
        cam = sc.cam if fixed else g.cam_data_create(last[0], last[1], 70.0, (float(args.width), float(args.height)))
        ref = Gpu(sc.world.max_nodes(), sc.world.size_in_chunks(), sc.size, device=local_rank)
        ref.upload_world(sc.world, sc.materials)
        ref.write_cam_data(cam)
        ref.write_settings(sc.settings)
        ref.render(MODE, **rkw)
        r_rgb, r_ids, _ = ref.read_output()
        from voxelraytracing_amd.shard import texels_to_frame
        a_rgb, a_ids = texels_to_frame(fg.frame.cpu().numpy().view(np.uint32))
        if not (np.array_equal(a_ids, r_ids) and np.array_equal(a_rgb, r_rgb)):
            raise SystemExit("gathered frame differs from the unsharded render")
        time_unsharded(ref)
        ref.close()
    if devices and os.environ.get("VRT_BENCH_VERIFY", "1") == "1":
        ref = Gpu(sc.world.max_nodes(), sc.world.size_in_chunks(), sc.size, device=0)
        ref.upload_world(sc.world, sc.materials)
        last = cams[(frame_no[0] - 1) % ORBIT] if not fixed else None
        ref.write_cam_data(sc.cam if fixed else g.cam_data_create(last[0], last[1], 70.0, (float(args.width), float(args.height))))
        ref.write_settings(sc.settings)
        ref.render(MODE, **rkw)
        r_rgb, r_ids, _ = ref.read_output()
        a_rgb, a_ids, _ = gpu.read_output()
        if not (np.array_equal(a_ids, r_ids) and np.array_equal(a_rgb, r_rgb)):
            raise SystemExit("the multi-device frame differs from the single-device render")
        time_unsharded(ref)
        ref.close()
    if rank != 0:
        dist.destroy_process_group()
        return

    ai = gpu.accel_info()
    derived = {"available": bool(ai.available), "cells": int(ai.cells), "bricks": int(ai.bricks), "bytes": int(ai.bytes),
               "builds": int(ai.builds), "chunk_builds": int(ai.chunk_builds), "last_build_ms": round(float(ai.last_build_ms), 4)}
    b_primary, b_shadow, b_fused = algorithmic_bytes(st)  # rank 0's own launches (its shard when N > 1), fixed-camera frame
    ms_p = kst.sum_ms_primary / max(kst.frames, 1)
    ms_s = kst.sum_ms_secondary / max(kst.frames, 1)
    fused = args.mode == "shadow" and args.variant == 0 and ms_s == 0.0  # one launch: no second kernel was timed
    if fused:
        dom_name, dom_bytes, dom_ms = "primary_shadow_march", b_fused, ms_p
    elif args.mode == "path":   # not the headline: first launch vs all bounce launches + finish; bytes as for shadow rays
        dom_name, dom_bytes, dom_ms = ("path_primary_march", b_primary, ms_p) if ms_p >= ms_s else ("path_bounce_marches", b_shadow, ms_s)
    else:
        dom_name, dom_bytes, dom_ms = ("primary_march", b_primary, ms_p) if ms_p >= ms_s else ("shadow_march", b_shadow, ms_s)
    period_s = dt / steps_timed
    in_flight = args.frames_in_flight if (world == 1 and not sharded and not devices and (fused or args.mode in ("primary", "path"))) else 1

    # ---- what bounds the dominant kernel: instruction issue.  Counters come from profiles/traffic_latest.json, which the
    # PMC tool stamps with the code object and the workload it measured; they are printed only for that very build and workload
    code = _ffi.code_object_sha256()
    workload_key = f"{args.mode}:{args.chunks}:{args.width}x{args.height}:v{args.variant}" + (f":{args.spp}spp:{args.bounces}b" if args.mode == "path" else "")
    pmc, pmc_all, pmc_note, issue_cost = None, {}, None, {}
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))
        if tj.get("code_object_sha256") != code:
            pmc_note = "profiles/traffic_latest.json was collected on another build of the kernels (code-object hash differs): not printed"
        elif workload_key not in tj.get("workloads", {}):
            pmc_note = f"profiles/traffic_latest.json holds the workloads {sorted(tj.get('workloads', {}))}, this run is {workload_key}: not printed"
        else:
            pmc_all = tj["workloads"][workload_key].get("kernels", {})
            pmc = pmc_all.get(dom_name)
            issue_cost = tj.get("issue_cost_cycles", {})
    except Exception as e:
        pmc_note = f"profiles/traffic_latest.json unreadable: {e}"
    clock_ghz = extras.get("shader_clock_ghz")
    peak_clock = clock_ghz or NOMINAL_GHZ
    roof = {"bound": "valu_issue", "kernel": dom_name, "unit": "G wave-instructions/s",
            # one VALU wave64 instruction holds a SIMD-32 for 2 cycles: 1024 SIMDs x clock / 2
            "peak": N_SIMD * peak_clock / 2.0, "peak_clock_ghz": peak_clock,
            "peak_clock_source": "measured in-kernel (s_memtime / s_memrealtime, probe build, right behind the timed frames)" if clock_ghz
                                 else "nominal (no probe in this mode)",
            "achieved": None, "frac": None, "traffic": None,
            "frame_period_ms": period_s * 1e3, "launches_in_flight": in_flight, "avg_launch_ms": dom_ms,
            "kernels_ms": ({"primary_shadow_march": ms_p} if fused else
                           {"path_primary_march": ms_p, "path_bounce_marches": ms_s} if args.mode == "path" else
                           {"primary_march": ms_p, "shadow_march": ms_s}),
            "frames_timed": kst.frames, "code_object_sha256": code, "workload_key": workload_key}
    if pmc_note:
        roof["pmc_note"] = pmc_note

    def class_bracket(cnt, n_valu):
        """SIMD cycles a launch's VALU instructions need, bracketed by its own class counters (tools/pmc.sh group 3): f32 add /
        mul / fma are full rate (2 cycles), conversions half rate (4), transcendentals 8; the integer class and what no counter
        names (compares, selects, min / max, moves) hold full- and half-rate instructions alike: priced at 2 and at 4."""
        need = ("SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_CVT")
        if not all(k in cnt for k in need):
            return None
        full = cnt["SQ_INSTS_VALU_ADD_F32"] + cnt["SQ_INSTS_VALU_MUL_F32"] + cnt["SQ_INSTS_VALU_FMA_F32"]
        known = 2.0 * full + 4.0 * cnt["SQ_INSTS_VALU_CVT"] + 8.0 * cnt["SQ_INSTS_VALU_TRANS_F32"]
        rest = n_valu - full - cnt["SQ_INSTS_VALU_CVT"] - cnt["SQ_INSTS_VALU_TRANS_F32"]
        return known + 2.0 * rest, known + 4.0 * rest, rest

    if pmc and world == 1 and not sharded and not devices and args.mode != "path":
        n_valu, n_salu = pmc["valu_wave_instructions"], pmc.get("salu_wave_instructions")
        # The counters are of the STANDING camera's frame (tools/pmc.sh), so `achieved` / `frac` divide by that frame's period
        # when this run measured it (the fixed-camera leg, or --fixed-camera); the headline leg's own period — an orbit frame
        # launches 0.5 % more rays on average: config.rays_per_frame_* — gives the *_headline_period pair.
        period_fixed = (extras.get("ms_per_step_fixed_camera", period_s * 1e3) if not fixed else period_s * 1e3) * 1e-3
        roof["achieved"] = n_valu / period_fixed / 1e9
        roof["frac"] = roof["achieved"] / roof["peak"]
        roof["achieved_headline_period"] = n_valu / period_s / 1e9
        roof["frac_headline_period"] = roof["achieved_headline_period"] / roof["peak"]
        if issue_cost.get("valu_simple"):
            # the same at what a plain instruction really takes to issue (tools/valu_rates.hip; profiles/rNN_valu_issue_rates.txt) — which is
            # also what a half-rate instruction takes between plain ones (k_mix_half_plain): the count at this price is the VALU's time
            roof["measured_issue_cycles_per_plain_instruction"] = issue_cost["valu_simple"]
            roof["frac_at_measured_issue_time"] = roof["frac"] * issue_cost["valu_simple"] / 2.0
            if issue_cost.get("mix_half_plain"):
                roof["measured_issue_cycles_per_instruction_half_rate_and_plain_alternating"] = issue_cost["mix_half_plain"]
            if issue_cost.get("mix_pk_plain"):
                roof["measured_issue_cycles_per_instruction_packed_and_plain_alternating"] = issue_cost["mix_pk_plain"]
        roof["frac_note"] = ("instructions per second over the peak at 2 cycles per wave-instruction, on the period of the frame the counters "
                             "are of (standing camera); removing instructions lowers it; class_weighted.frac (each class at its issue time) "
                             "says how full the VALU pipes are")
        roof["traffic"] = pmc["hbm_bytes"]
        roof["valu_wave_instructions_per_launch"] = n_valu
        roof["salu_wave_instructions_per_launch"] = n_salu
        roof["period_used_ms"] = period_fixed * 1e3
        roof["period_headline_ms"] = period_s * 1e3
        # the same with every class at its measured issue cost (tools/valu_rates.hip): SIMD cycles the launch's
        # instructions need / SIMD cycles the frame period offers
        nominal, cls = pmc.get("valu_issue_cycles_by_class_nominal"), pmc.get("issue_cycles_by_class")
        if nominal and cls:
            avail = N_SIMD * peak_clock * 1e9 * period_fixed
            valu_measured = sum(v for k, v in cls.items() if k.startswith("valu"))
            br = class_bracket(pmc.get("counters", {}), n_valu)
            check = None
            if br:
                lo_c, hi_c, rest = br
                model_c = sum(nominal.values())
                check = {"cycles_lower": lo_c, "cycles_upper": hi_c, "cycles_model": model_c,
                         "model_inside_bracket": bool(lo_c <= model_c <= hi_c),
                         "residual_vs_bracket_midpoint": model_c / (0.5 * (lo_c + hi_c)) - 1.0,
                         "instructions_no_counter_classifies": rest,
                         "note": "whole-launch class counters; the model prices the whole launch at the march loop's class mix"}
            roof["class_weighted"] = {
                "counter_check": check,
                # VALU wave-instructions x their class's architectural issue time on a SIMD-32 (2 / 4 / 8 cycles for full rate /
                # half rate + packed / transcendental), classes in the proportions of the march loop's fast path (profiles/*isa_mix*)
                "valu_issue_cycles_per_launch_nominal": nominal, "simd_cycles_available": avail,
                "frac": sum(nominal.values()) / avail,
                # the same at the costs tools/valu_rates.hip measured (upper bounds: they carry the microbenchmark's own loop)
                "valu_issue_cycles_per_launch_at_measured_costs": valu_measured, "frac_at_measured_costs": valu_measured / avail,
                "salu_unit_cycles_per_launch_at_measured_cost": cls.get("salu"),
                "note": "the scalar unit is shared by a CU's four SIMDs and issues beside the VALU; its cycles are not additive.  A frac above 1 "
                        "says the classes' architectural issue times (2 / 4 / 8 cycles) overestimate this mix — beside plain instructions a "
                        "'half-rate' one cost ~ 3.3 cycles in the same-box A/Bs of profiles/r06_step_asm.txt, a plain one ~ 2.5 — i.e. the "
                        "VALU has no issue slots left: `roofline.frac` (2 cycles for every wave-instruction) is the conservative figure"}
    elif pmc and world == 1 and not sharded and not devices:
        # The path trace: a frame is the primary launch(es) + the bounce launch(es) (a chain per 8 samples).  `achieved` is the
        # dominant kernel's VALU wave-instructions of a frame over the time its launches take ALONE on the GPU — the one-frame-
        # at-a-time leg of this run when it ran (with two frames in flight a launch's own begin-to-end time holds the other
        # frame's work too) — and `frame` the same for everything a frame launches over the frame period.
        lone_ms = (extras.get("avg_bounce_launches_ms_1_in_flight") if dom_name == "path_bounce_marches" else extras.get("avg_launch_ms_1_in_flight")) or dom_ms
        per_frame = lambda k, what: (pmc_all[k][what] or 0.0) * pmc_all[k].get("launches_per_frame", 1.0)   # noqa: E731
        n_valu = per_frame(dom_name, "valu_wave_instructions")
        roof["achieved"] = n_valu / (lone_ms * 1e-3) / 1e9
        roof["frac"] = roof["achieved"] / roof["peak"]
        roof["lone_launches_ms_per_frame"] = lone_ms
        roof["lone_launch_source"] = "the one-frame-at-a-time leg of this run" if lone_ms != dom_ms else "the timed frames' own launch durations (no one-at-a-time leg in this run)"
        roof["launches_per_frame"] = {k: v.get("launches_per_frame", 1.0) for k, v in pmc_all.items()}
        roof["valu_wave_instructions_per_launch"] = pmc["valu_wave_instructions"]
        roof["salu_wave_instructions_per_launch"] = pmc.get("salu_wave_instructions")
        roof["traffic"] = pmc["hbm_bytes"]
        roof["traffic_per_frame"] = sum(per_frame(k, "hbm_bytes") for k in pmc_all)
        all_valu = sum(per_frame(k, "valu_wave_instructions") for k in pmc_all)
        roof["frame"] = {"valu_wave_instructions": all_valu, "achieved": all_valu / period_s / 1e9, "frac": all_valu / period_s / 1e9 / roof["peak"],
                         "note": "every launch of a frame over the frame period (frames in flight overlap)"}
        br = class_bracket(pmc.get("counters", {}), pmc["valu_wave_instructions"])
        if br:
            avail = N_SIMD * peak_clock * 1e9 * lone_ms * 1e-3 / max(pmc.get("launches_per_frame", 1.0), 1e-9)
            roof["class_weighted"] = {"cycles_lower": br[0], "cycles_upper": br[1], "simd_cycles_available": avail,
                                      "frac_lower": br[0] / avail, "frac_upper": br[1] / avail, "frac": 0.5 * (br[0] + br[1]) / avail,
                                      "note": "the launch's own class counters: f32 add / mul / fma at 2 cycles, conversions 4, transcendentals 8, the rest "
                                              "(integer class, compares, selects, moves) at 2 and at 4: a bracket, `frac` its midpoint"}
        c = pmc.get("counters", {})
        if c.get("TCP_TOTAL_CACHE_ACCESSES_sum"):
            roof["l1"] = {"line_accesses": c["TCP_TOTAL_CACHE_ACCESSES_sum"], "requests_to_l2": c.get("TCP_TCC_READ_REQ_sum"),
                          "hit_rate": 1.0 - c.get("TCP_TCC_READ_REQ_sum", 0.0) / c["TCP_TOTAL_CACHE_ACCESSES_sum"],
                          "lane_utilisation_valu": (c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"] / 64.0) if c.get("SQ_ACTIVE_INST_VALU") else None}
    # §8(d)'s algorithmic bytes stay as a secondary object; a fraction above 1 says the kernel does not move those bytes
    # (the derived tables answer from L1 / L2); `measured_hbm` is what the PMC passes saw
    hbm = {"algorithmic_bytes_per_launch": dom_bytes, "algorithmic_gbs_at_frame_period": dom_bytes / period_s / 1e9,
           "frac_of_hbm_peak": dom_bytes / period_s / 1e9 / HBM_PEAK_GBS, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "note": "bytes of the reference algorithm (SURVEY 8d); frac > 1 = not moved: the grid march reads derived tables that hit in L1/L2"}
    if roof["traffic"]:
        hbm["measured_hbm_bytes_per_launch"] = roof["traffic"]
        hbm["measured_hbm_gbs"] = roof["traffic"] / period_s / 1e9
        hbm["measured_frac_of_hbm_peak"] = hbm["measured_hbm_gbs"] / HBM_PEAK_GBS
    # wave-level march trips of the fixed-camera frame, from the per-pixel step counts: a tile's wave runs max-over-lanes trips
    if steps_img is not None:
        t = steps_img.reshape(args.height // 8, 8, args.width // 8, 8)
        prim = (t & 0xFFFF).max(axis=(1, 3)).astype(np.int64)
        shad = (t >> 16).max(axis=(1, 3)).astype(np.int64)
        lane_steps = int((steps_img & 0xFFFF).sum() + (steps_img >> 16).sum())
        roof["wave_march_trips_per_launch"] = {"primary": int(prim.sum()), "shadow": int(shad.sum()),
                                               "lane_steps": lane_steps, "lane_utilisation_of_trips": lane_steps / (64.0 * (prim.sum() + shad.sum()))}

    out = {
        "metric": "Mrays/s at 1920x1080, 1 primary + 1 shadow ray",
        "value": rays_total / dt / 1e6,
        "unit": "Mrays/s",
        "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
        # frames inside the timed window `value` comes from (> steps when the request was under 50 ms of frames: see
        # `value_requested_steps` for the K steps alone)
        "steps_timed": steps_timed,
        "value_requested_steps": requested["value"], "ms_per_step_requested_steps": requested["ms_per_step"],
        "value_is": ("the rate over exactly the requested steps" if steps_timed == args.steps else
                     f"the rate over a second timed window of {steps_timed} frames of the same loop (the requested {args.steps} steps are "
                     f"{requested['ms_per_step'] * args.steps:.2f} ms of frames, under the {MIN_TIMED_SECONDS * 1e3:.0f} ms a window needs to be more than pipeline fill and drain); "
                     "the requested steps alone: value_requested_steps"),
        "ranks_seen": ranks_seen,   # an all-reduce of ones over the process group (1: no group)
        "links": links,
        "ms_per_step": period_s * 1e3,
        "host_submit_ms_per_step": host_submit_ms,   # time until the loop's last call returned / steps: the host's share of a step
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"{'C2' if (args.chunks, args.width, args.height) == (8, 1920, 1080) else 'C3' if (args.chunks, args.width, args.height) == (16, 1920, 1080) else 'C2-family'}: "
                               f"{args.width}x{args.height} frame, {args.chunks}x{args.chunks}x{args.chunks}-chunk procedural SVO "
                               f"world (seed 1), 1 primary + 1 shadow ray per solid hit",
                   "camera": "standing" if fixed else f"orbit of {ORBIT} positions (6-voxel circle, +-8 degrees), the per-frame seam of main.rs:426-453 issued every frame "
                             "(settings, camera, chunk_roots, world data, dispatch) with the chunk_roots rewrite through vrt_write_chunk_roots_tagged — the mirror's "
                             "table generation as the tag; the untagged rewrite at the client's own 30^3 chunks, with the blit, is the `operating_point` object",
                   "rays_per_frame_actual": rays_fixed if fixed else float(np.mean(orbit_rays)), "rays_per_frame_nominal": 2 * args.width * args.height,
                   "rays_per_frame_fixed_camera": rays_fixed,
                   "sharding": ("whole frame" if not sharded else "whole frame through the one-rank gather pipeline") if (world == 1 and not devices) else
                               (f"one process, {args.gpus} devices behind one vrt context (vrt_config.device_ids): 8x8 tiles interleaved, "
                                f"{root_weight} of every {root_weight + args.gpus - 1} to device 0, the others store 8-byte records straight into device 0's memory over xGMI"
                                if devices else
                                f"8x8 tiles interleaved over {world} ranks ({root_weight} of every {root_weight + world - 1} to the gather root, "
                                f"which renders them in place) + RCCL gather of the other ranks' tile buffers "
                                f"({'8-byte records shaded at the root' if MODE != MODE_PATH and args.variant == 0 else '16-byte texels'}) to rank 0"),
                   "root_weight": root_weight, "root_weight_tuning_ms_per_frame": tuning,
                   "frames_per_gather": batch, "frames_per_gather_tuning_ms_per_frame": batch_tuning,
                   "frames_in_flight": args.frames_in_flight, "clock_settle": settle,
                   # (scalars, so that they survive in the driver's `parsed.config`: which window `value` / `ms_per_step` are of)
                   "steps_requested": args.steps, "steps_timed": steps_timed, "value_requested_steps": requested["value"],
                   "ms_per_step_requested_steps": requested["ms_per_step"], "value_window": "requested steps" if steps_timed == args.steps else "second window of steps_timed frames",
                   "kernel_variant": args.variant, "derived_tables": derived,
                   # N > 1: what this split should give, term by term, from the unsharded frame timed on this GPU in this run —
                   # stated so that the measured line can be read against it (shard.expected_scaling)
                   "expected_scaling": None},
        "roofline": roof,
        "hbm": hbm,
    }
    if (world > 1 or devices) and one_gpu_ms[0]:
        from voxelraytracing_amd.shard import expected_scaling
        texel_msgs = MODE == MODE_PATH or args.variant != 0
        n_ = args.gpus
        # The host's share per frame, MEASURED in this run (round 6): not the loop's own clock — a host that is ahead of its GPUs reads
        # the GPUs' time there — but the time spent inside the calls that issue a frame (vrt_get_issue_profile, FrameGather.host_profile).
        #   one process per GPU: rank 0's render call + (gather call + wait + assembly launches) / frames per collective;
        #   one context over N devices: with issuing threads (distinct devices) the whole vrt_render call as it was timed; without
        #   (a one-GPU rehearsal issues for every device in turn) the call as it would be with them: the slower of the root's issue
        #   and one shard device's issue + a hand-over, + the tail (waits, assembly launch, event) — and the measured serial call
        #   beside it, which must be the sum of those parts.
        HAND_OVER_US = 3.0   # post + wake of a spinning issuing thread, and the join: measured 2-4 us (profiles/r04_group_host_profile.txt)
        host_term = {"source": "measured in this run", "render_call_us": issue["render_us"]}
        if world > 1:
            per_collective = gather_host["gather_call_us"] + gather_host["wait_us"] + gather_host["assemble_us"]
            fpc = max(gather_host["frames_per_collective"], 1.0)
            host_term.update(collective=gather_host, collective_us_per_frame=per_collective / fpc)
            if args.rehearse_on_one_gpu:
                # (the rehearsal's collective is a gloo gather staged through host memory: its cost is not RCCL's; the model keeps
                # the 35 us per RCCL collective that --force-gather measures on one rank, and says so)
                host_term["collective_us_per_frame_used"] = 35.0 / max(batch, 1)
                host_term["source"] = "render call measured in this run; collective: 35 us assumed (the rehearsal timed a gloo stand-in)"
            else:
                host_term["collective_us_per_frame_used"] = per_collective / fpc
            host_us = issue["render_us"] + host_term["collective_us_per_frame_used"]
        else:
            host_term.update(issue_profile=issue)
            serial_sum = issue["root_issue_us"] + (n_ - 1) * issue["shard_issue_us_mean"] + issue["tail_us"]
            # (with issuing threads every thread makes its own message's stream wait — vrt_group.hip: frame_stream — so the caller's tail
            # keeps only the assembly's launch and the event record)
            one_wait = issue["message_waits_us"] / max(n_ - 1, 1)
            with_threads = max(issue["root_issue_us"], issue["shard_issue_us_mean"] + one_wait + HAND_OVER_US) + HAND_OVER_US + (issue["tail_us"] - issue["message_waits_us"])
            host_term.update(serial_sum_of_parts_us=serial_sum, with_issuing_threads_us=with_threads, hand_over_us_assumed=HAND_OVER_US)
            if issue["issuing_threads"]:
                host_us = issue["render_us"]
            else:
                host_us = with_threads
                host_term["source"] = ("parts measured in this run (one thread issued for every device in turn: render_call_us is their sum); "
                                       "the term used is the call with one issuing thread per device")
        host_term["host_us_per_frame_used"] = host_us
        bpp = 16 if texel_msgs else 8
        es = expected_scaling(n_, one_gpu_ms[0], args.width * args.height, bpp, root_weight,
                              host_us_per_frame=host_us, fixed_ms_per_launch=min(0.016, one_gpu_ms[0]))
        es["host_term"] = host_term
        # both ways of driving N GPUs, side by side: this run's mode with its measured host term, the other with the host term its
        # own runs measured (profiles/r06_bench_modes_rehearsal.txt) — which one the model expects to be faster for this workload
        other_host = (30.0 + 10.2) if world > 1 else (35.0 / 4.0 + 8.0)
        from voxelraytracing_amd.shard import root_weight_model as rwm
        w_other = rwm(n_, one_gpu_ms[0], args.width * args.height * bpp)
        other = expected_scaling(n_, one_gpu_ms[0], args.width * args.height, bpp, w_other, host_us_per_frame=other_host,
                                 fixed_ms_per_launch=min(0.016, one_gpu_ms[0]))
        this_mode, other_mode = ("one process per GPU + RCCL gather", "one context over N devices") if world > 1 else ("one context over N devices", "one process per GPU + RCCL gather")
        es["modes"] = {this_mode: {"predicted_ms": es["predicted_ms"], "speedup": es["speedup"], "bound": es["bound"], "host_us_per_frame": host_us, "host_term": "this run's"},
                       other_mode: {"predicted_ms": other["predicted_ms"], "speedup": other["speedup"], "bound": other["bound"], "host_us_per_frame": other_host,
                                    "host_term": "that mode's own measurement on one GPU (profiles/r06_bench_modes_rehearsal.txt)", "root_weight": w_other},
                       "predicted_faster": this_mode if es["predicted_ms"] <= other["predicted_ms"] else other_mode}
        # a root that keeps (nearly) the whole frame is one GPU with spectators: such a line is not a scaling point
        root_share = root_weight / float(root_weight + n_ - 1)
        es["root_share_of_tiles"] = root_share
        es["degenerate_scaling_point"] = bool(root_share > 0.8)
        es["host_submit_us_per_frame_this_run"] = host_submit_ms * 1e3
        # ... and at the root weight the same model would choose (this run's was measured, or — a rehearsal on one GPU — tuned
        # for a machine on which sharing buys nothing)
        from voxelraytracing_amd.shard import root_weight_model
        w_model = root_weight_model(n_, one_gpu_ms[0], args.width * args.height * (16 if texel_msgs else 8))
        if w_model != root_weight:
            best = expected_scaling(n_, one_gpu_ms[0], args.width * args.height, 16 if texel_msgs else 8, w_model,
                                    host_us_per_frame=host_us, fixed_ms_per_launch=min(0.016, one_gpu_ms[0]))
            es["at_model_root_weight"] = {k: best[k] for k in ("root_weight", "predicted_ms", "speedup", "bound")}
        es["frame_ms_1gpu_measured_in_this_run"] = one_gpu_ms[0]
        es["measured_ms"] = period_s * 1e3
        es["measured_speedup"] = one_gpu_ms[0] / (period_s * 1e3)
        es["measured_over_predicted"] = (period_s * 1e3) / es["predicted_ms"] if es.get("predicted_ms") else None
        if args.rehearse_on_one_gpu:
            es["rehearsal"] = "every rank / device on ONE GPU: the measured line is not a scaling point; the prediction is for N distinct devices"
        out["config"]["expected_scaling"] = es
        # (scalars beside the object: what survives in a record that keeps only a config's scalar fields)
        out["config"].update(expected_ms=es["predicted_ms"], expected_speedup=es["speedup"], expected_bound=es["bound"], host_us_per_frame_measured=host_us,
                             predicted_faster_mode=es["modes"]["predicted_faster"], root_share_of_tiles=root_share, degenerate_scaling_point=bool(root_share > 0.8),
                             frame_ms_1gpu_same_run=one_gpu_ms[0])
    if gather_host is not None:
        out["gather_host_profile"] = dict(gather_host, render_call_us=issue["render_us"])   # (this rank's host time per collective / per render call)
    out.update(extras)
    if operating_point:
        out["operating_point"] = operating_point
    if args.mode != "shadow":
        out["metric"] = f"Mrays/s at {args.width}x{args.height}, mode {args.mode}" + (f" {args.bounces} bounces {args.spp} spp" if args.mode == "path" else "")
        out["config"]["workload"] = out["config"]["workload"].replace("C2:", "non-headline:").replace("1 primary + 1 shadow ray per solid hit", f"mode {args.mode}")
    if world == 1 and not devices and not args.no_cpu_baseline and args.mode == "shadow":
        out["cpu_baseline"] = cpu_baseline(sc, args, rays_fixed)
    sys.stdout.flush()
    with os.fdopen(json_fd, "w") as real_stdout:
        real_stdout.write(json.dumps(out) + "\n")
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
    standing camera, repeated for about 8 s of wall time (>= 2 frames), plus one frame on one thread."""
    from oracle import orc
    o = orc.from_package_scene(sc)
    cores = usable_cores()
    times = []
    t_all = time.perf_counter()
    while len(times) < 2 or (time.perf_counter() - t_all < 8.0 and len(times) < 60):
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
            "sample": f"{len(times)} full {args.width}x{args.height} frames of the same scene, standing camera (median {dt:.3f} s/frame, "
                      f"{sum(times):.1f} s total), OpenMP dynamic over 8-row bands; + 1 frame on 1 thread ({dt1:.1f} s)"}


if __name__ == "__main__":
    main()
