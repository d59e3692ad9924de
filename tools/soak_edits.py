"""tools/soak_edits.py [seconds] [seed] [devices, e.g. 0,0,0] [texel|staged|poison|present ...] — a long random session against the oracle: edits (0-3 before a frame, now and then a cluster of 8-40 in one place), camera moves,
chunk_roots rewrites, the grid recentred by a chunk (center_chunks: the table shifts, world.min changes, the chunks that came
into the grid arrive over the next frames), quiet stretches, changes of the number of frames in flight, whole-world rebuilds, the primary and the
primary + shadow mode, variants 0 and 2 — and every few dozen frames the last frame is compared with the oracle's frame of
the world as it is.  With a device list the context is ONE context over those devices (the same device several times
rehearses it on one GPU): 8-byte records by default — then only the default march's frames — or texel messages.  Exercises the upload stream / per-frame-set table machinery for races that a short
test would not meet.  `present` (round 6): the session is the client's draw + present loop — the blit's uniforms declared before every frame
(vrt_set_presentation: now the default crosshair, now another, now off), vrt_present_device behind every frame — and every check also holds
the window's image at the texture's size to the oracle's blit of the frame: a frame that stored its own window pixels, or a blit launched
behind it, whichever the declaration and the frame's kind made it.  Exit status 1 on the first mismatch."""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voxelraytracing_amd import Gpu, MODE_PRIMARY, MODE_PRIMARY_SHADOW, graphics as g, scenes
from oracle import orc

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
W, H = 160, 96
sc = scenes.c2((W, H))
devices = [int(d) for d in sys.argv[3].split(",")] if len(sys.argv) > 3 and sys.argv[3] else None
flags = set(sys.argv[4:])
records_only = devices is not None and "texel" not in flags   # 8-byte records: plain frames of the default march only
max_in_flight = 2 if devices else 3
gpu = Gpu(sc.world.max_nodes(), sc.world.size_in_chunks(), sc.size, devices=devices, texel_messages="texel" in flags,
          staged_messages="staged" in flags, poison_messages="poison" in flags)
gpu.upload_world(sc.world, sc.materials); gpu.write_settings(sc.settings)
ex, ey, ez = (float(v) for v in sc.eye)
cam = sc.cam
prev_cam = cam
mode = MODE_PRIMARY_SHADOW
variant = 0
frames = checks = edits = recentres = 0
center0 = tuple(int(v) // 32 + 4 for v in sc.world.min_voxel())   # the grid's centre chunk (8^3 chunks)
center = center0
pending = []            # chunk ranges that arrived (generate_missing) and are not uploaded yet: a few per frame
import collections
ops = collections.deque(maxlen=80)   # what was done lately, for the report of a mismatch
t_end = time.time() + seconds
next_report = time.time() + 20
presenting = "present" in flags and not devices
CROSSHAIRS = [dict(), dict(style=1, size=7.5, color=(1.0, 0.3, 0.2, 0.6)), dict(style=0), None]   # (None: undeclared)
decl = CROSSHAIRS[0]
while time.time() < t_end:
    burst = int(rng.integers(5, 60))
    for _ in range(burst):
        r = rng.random()
        if r < 0.5:
            # (one time in twelve a cluster of 8-40 edits within a few voxels: several uploads of one chunk's range before a frame — the
            # staged ranges are merged in the pinned ring, csrc/vrt_uploads.hip)
            cluster = rng.random() < 1.0 / 12.0
            c0 = (int(ex) + int(rng.integers(-20, 21)), int(ey) + int(rng.integers(-24, 2)), int(ez) + int(rng.integers(-20, 21)))
            for _ in range(int(rng.integers(8, 41)) if cluster else int(rng.integers(1, 4))):
                p = (int(ex) + int(rng.integers(-24, 25)), int(ey) + int(rng.integers(-28, 6)), int(ez) + int(rng.integers(-24, 25)))
                if cluster:
                    p = (c0[0] + int(rng.integers(-3, 4)), c0[1] + int(rng.integers(-3, 4)), c0[2] + int(rng.integers(-3, 4)))
                try:
                    start, n = sc.world.set_voxel(p, int(rng.choice([0, 0, 0, 3, 4, 40, 47, 62])))
                except Exception as e:
                    if getattr(e, "kind", "") not in ("NoChange", "NoChunk", "OutOfMemory"):
                        raise
                    if getattr(e, "kind", "") != "OutOfMemory":
                        continue
                    start, n = e.range
                gpu.write_nodes(sc.world.nodes_ptr(), start, start + n)
                ops.append(f"f{frames} edit {p} nodes [{start},{start + n})")
                edits += 1
        elif r < 0.55:
            gpu.write_nodes(sc.world.nodes_ptr(), 0, 2)          # node 0: a whole-world rebuild
            ops.append(f"f{frames} write node 0")
        elif r < 0.6:
            nf = int(rng.integers(1, max_in_flight + 1))
            gpu.set_frames_in_flight(nf)
            ops.append(f"f{frames} frames in flight {nf}")
        elif r < 0.65:
            for _ in range(int(rng.integers(60, 90))):           # a quiet stretch: the table sets merge
                gpu.render(mode, variant=variant)
                frames += 1
            ops.append(f"f{frames} quiet stretch done")
        elif r < 0.69:
            # the player crosses into another chunk (client/src/lib.rs:55-65): the grid follows, within two chunks of where it began
            ax = int(rng.integers(0, 3))
            step = int(rng.choice([-1, 1]))
            nc = list(center)
            nc[ax] = min(max(nc[ax] + step, center0[ax] - 2), center0[ax] + 2)
            if tuple(nc) != center:
                center = tuple(nc)
                sc.world.center_chunks(center)
                pending.extend(sc.world.generate_missing(0, 1).tolist())
                gpu.write_world_data(sc.world.world_data())
                ops.append(f"f{frames} recentre to {center}, {len(pending)} chunk ranges pending")
                recentres += 1
        for root, n in pending[:6]:                                # main.rs:289-295: the chunk ranges that arrived
            gpu.write_nodes(sc.world.nodes_ptr(), int(root), int(root) + int(n))
            ops.append(f"f{frames} chunk range [{int(root)},{int(root) + int(n)})")
        del pending[:6]
        if rng.random() < 0.3:
            rot = (float(rng.uniform(-40, 10)), float(rng.uniform(0, 360)), 0.0)
            eye = (ex + float(rng.uniform(-6, 6)), ey + float(rng.uniform(-3, 3)), ez + float(rng.uniform(-6, 6)))
            prev_cam = cam
            cam = g.cam_data_create(rot, eye, float(rng.uniform(50, 100)), (float(W), float(H)))
            gpu.write_cam_data(cam)
            ops.append(f"f{frames} camera")
        if rng.random() < 0.1:
            mode = MODE_PRIMARY if rng.random() < 0.3 else MODE_PRIMARY_SHADOW
            variant = 2 if rng.random() < 0.2 and not records_only else 0
            ops.append(f"f{frames} mode {mode} variant {variant}")
        gpu.write_chunk_roots(sc.world.chunk_roots(), tag=sc.world.roots_generation())
        if presenting:
            if rng.random() < 0.05:
                decl = CROSSHAIRS[int(rng.integers(0, len(CROSSHAIRS)))]
                ops.append(f"f{frames} presentation {decl}")
            if decl is None:
                gpu.set_presentation(off=True)
            else:
                gpu.set_presentation((W, H), **decl)       # main.rs:429-432, every frame
        gpu.render(mode, variant=variant)
        if presenting:
            gpu.present_device((W, H), **(decl or {}))     # main.rs:454
        frames += 1
        r2 = rng.random()
        if r2 < 0.02:
            gpu.stats()                 # folds the timing events: waits for the frames in flight
        elif r2 < 0.03:
            gpu.accel_info()            # brings the first table set up to date
        elif r2 < 0.035:
            gpu.synchronize()
    if pending:                                                    # the comparison is of the world as the host has it: all of it uploaded
        for root, n in pending:
            gpu.write_nodes(sc.world.nodes_ptr(), int(root), int(root) + int(n))
        pending.clear()
        gpu.render(mode, variant=variant)
        frames += 1
    r3 = rng.random()
    if r3 < 0.1:
        gpu.stats()
    elif r3 < 0.15:
        gpu.accel_info()
    shown = gpu.present((W + 32, H + 18)) if r3 > 0.9 and not devices and not presenting else None
    window = gpu.present((W, H), **(decl or {})) if presenting else None
    rgb, ids, _ = gpu.read_output()
    o = orc.from_package_scene(sc)
    o.set_cam(cam)
    r_rgb, r_ids, _, _ = o.render(mode, W, H)
    checks += 1
    if shown is not None and np.array_equal(ids, r_ids):
        want = orc.present(rgb, (W + 32, H + 18))
        if not np.array_equal(shown, want):
            print(f"PRESENT MISMATCH at check {checks}: {int((shown != want).any(axis=2).sum())} pixels differ", flush=True)
            sys.exit(1)
    if window is not None and not np.array_equal(window, orc.present(rgb, (W, H), **(decl or {}))):
        print(f"WINDOW MISMATCH at check {checks}, frame {frames}: {int((window != orc.present(rgb, (W, H), **(decl or {}))).any(axis=2).sum())} pixels differ (declared {decl}, mode {mode}, variant {variant})", flush=True)
        print("the last operations:\n  " + "\n  ".join(list(ops)[-12:]), flush=True)
        sys.exit(1)
    if not np.array_equal(ids, r_ids) or float(np.nanmax(np.abs(rgb - r_rgb))) > 1e-4:
        print(f"MISMATCH at check {checks}, frame {frames}: {int((ids != r_ids).sum())} id words differ (mode {mode}, variant {variant})", flush=True)
        _, ids_again, _ = gpu.read_output()
        print(f"the same output read once more, nothing rendered in between: {int((ids_again != r_ids).sum())} id words differ", flush=True)
        print("the last operations:\n  " + "\n  ".join(list(ops)[-12:]), flush=True)
        os.makedirs("gpurun_out", exist_ok=True)
        np.savez_compressed("gpurun_out/soak_mismatch.npz", ids=ids, r_ids=r_ids, rgb=rgb, r_rgb=r_rgb)
        diff = ids != r_ids
        ys, xs = np.nonzero(diff)
        print(f"differing pixels: rows {ys.min()}..{ys.max()}, columns {xs.min()}..{xs.max()}; per 8-row band: {[int(diff[y:y + 8].sum()) for y in range(0, H, 8)]}", flush=True)
        o.set_cam(prev_cam)
        _, p_ids, _, _ = o.render(mode, W, H)
        print(f"of the {int(diff.sum())} differing pixels {int((ids[diff] == p_ids[diff]).sum())} are the oracle's frame of this world through the camera BEFORE the last change", flush=True)
        vals, cnt = np.unique(ids[diff] & 0x7FFF, return_counts=True)
        print(f"voxels the frame has there: {dict(zip(vals.tolist()[:8], cnt.tolist()[:8]))}; the oracle: {dict(zip(*[a.tolist()[:8] for a in np.unique(r_ids[diff] & 0x7FFF, return_counts=True)]))}", flush=True)
        a = gpu.accel_info()
        print(f"tables: builds {a.builds}, chunk builds {a.chunk_builds}; world min {sc.world.min_voxel()}, generation {sc.world.roots_generation()}", flush=True)

        def again(what):
            gpu.render(mode, variant=variant)
            _, ids2, _ = gpu.read_output()
            print(f"  {what}: {int((ids2 != r_ids).sum())} id words differ", flush=True)
        again("the same frame once more")
        if not records_only:
            gpu.render(mode, variant=2); _, ids2, _ = gpu.read_output()
            print(f"  through the octree walk (variant 2): {int((ids2 != r_ids).sum())} id words differ", flush=True)
        gpu.write_cam_data(cam); again("camera written again")
        gpu.write_world_data(sc.world.world_data()); again("world data written again")
        gpu.write_chunk_roots(sc.world.chunk_roots()); again("chunk roots written again, untagged")
        gpu.write_nodes(sc.world.nodes_ptr(), 0, sc.world.max_nodes()); again("every node written again")
        sys.exit(1)
    if checks % 8 == 0 and not records_only:   # and a path-traced frame of the same world: random samples per pixel, seed and bounce count
        from voxelraytracing_amd import MODE_PATH
        spp, pseed = int(rng.integers(1, 7)), int(rng.integers(0, 1000))
        for _ in range(int(rng.integers(1, 4))):
            gpu.render(MODE_PATH, spp=spp, seed=pseed)
        rgb, ids, _ = gpu.read_output()
        r_rgb, r_ids, _, _ = o.render(orc.MODE_PATH, W, H, spp=spp, seed=pseed)
        path_checks = globals().get("path_checks", 0) + 1
        globals()["path_checks"] = path_checks
        if not np.array_equal(ids, r_ids) or float(np.nanmax(np.abs(rgb - r_rgb))) > 1e-4:
            print(f"PATH MISMATCH at check {checks}: {int((ids != r_ids).sum())} id words differ, max radiance error {float(np.nanmax(np.abs(rgb - r_rgb))):.3g} (spp {spp}, seed {pseed})", flush=True)
            sys.exit(1)
    if time.time() > next_report:
        a = gpu.accel_info()
        print(f"{frames} frames, {edits} edits, {recentres} recentres, {checks} checks ok; whole-world builds {a.builds}, chunks rebuilt alone {a.chunk_builds}", flush=True)
        next_report = time.time() + 20
a = gpu.accel_info()
print(f"soak ok: {frames} frames, {edits} edits, {recentres} recentres of the grid, {checks} checks against the oracle (+ {globals().get('path_checks', 0)} of path-traced frames); whole-world builds {a.builds}, chunks rebuilt alone {a.chunk_builds}")
