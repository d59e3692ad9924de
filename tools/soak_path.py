"""tools/soak_path.py [seconds] [seed] [devices] [texel] — the path trace under the same random session as tools/soak_edits.py: bursts
of path-traced frames (1-3 samples, a seed of their own each, 1-3 frames in flight) with edits, camera moves, recentred grids and
arriving chunks between them; every burst's last frame against the oracle's (ids bit-exact, radiance within 1e-4).  The march
cells of every frame set are rebuilt chunk by chunk under the bounce launches' feet here."""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voxelraytracing_amd import Gpu, MODE_PATH, graphics as g, scenes
from oracle import orc

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
devices = [int(d) for d in sys.argv[3].split(",")] if len(sys.argv) > 3 and sys.argv[3] else None
rng = np.random.default_rng(seed)
W, H = 160, 96
sc = scenes.c4((W, H))
gpu = Gpu(sc.world.max_nodes(), sc.world.size_in_chunks(), sc.size, devices=devices, texel_messages=devices is not None)
gpu.upload_world(sc.world, sc.materials); gpu.write_settings(sc.settings); gpu.write_cam_data(sc.cam)
ex, ey, ez = (float(v) for v in sc.eye)
cam = sc.cam
center0 = tuple(int(v) // 32 + sc.world.size_in_chunks() // 2 for v in sc.world.min_voxel())
center = center0
frames = checks = edits = recentres = 0
pending = []
t_end = time.time() + seconds
next_report = time.time() + 20
while time.time() < t_end:
    if rng.random() < 0.3:
        gpu.set_frames_in_flight(int(rng.integers(1, (2 if devices else 3) + 1)))
    for _ in range(int(rng.integers(2, 12))):
        r = rng.random()
        if r < 0.6:
            for _ in range(int(rng.integers(1, 4))):
                p = (int(ex) + int(rng.integers(-24, 25)), int(ey) + int(rng.integers(-28, 6)), int(ez) + int(rng.integers(-24, 25)))
                try:
                    start, n = sc.world.set_voxel(p, int(rng.choice([0, 0, 0, 3, 4, 40, 47, 62])))
                except Exception as e:
                    if getattr(e, "kind", "") != "OutOfMemory":
                        continue
                    start, n = e.range
                gpu.write_nodes(sc.world.nodes_ptr(), start, start + n)
                edits += 1
        elif r < 0.7:
            ax = int(rng.integers(0, 3))
            nc = list(center)
            nc[ax] = min(max(nc[ax] + int(rng.choice([-1, 1])), center0[ax] - 2), center0[ax] + 2)
            if tuple(nc) != center:
                center = tuple(nc)
                sc.world.center_chunks(center)
                pending.extend(sc.world.generate_missing(0, 1).tolist())
                gpu.write_world_data(sc.world.world_data())
                recentres += 1
        for root, n in pending[:8]:
            gpu.write_nodes(sc.world.nodes_ptr(), int(root), int(root) + int(n))
        del pending[:8]
        if rng.random() < 0.4:
            rot = (float(rng.uniform(-40, 10)), float(rng.uniform(0, 360)), 0.0)
            eye = (ex + float(rng.uniform(-6, 6)), ey + float(rng.uniform(-3, 3)), ez + float(rng.uniform(-6, 6)))
            cam = g.cam_data_create(rot, eye, float(rng.uniform(50, 100)), (float(W), float(H)))
            gpu.write_cam_data(cam)
        gpu.write_chunk_roots(sc.world.chunk_roots(), tag=sc.world.roots_generation())
        spp, pseed = int(rng.integers(1, 4)), int(rng.integers(0, 1000))
        gpu.render(MODE_PATH, spp=spp, seed=pseed)
        frames += 1
    for root, n in pending:
        gpu.write_nodes(sc.world.nodes_ptr(), int(root), int(root) + int(n))
    if pending:
        pending.clear()
        gpu.render(MODE_PATH, spp=spp, seed=pseed)
        frames += 1
    rgb, ids, _ = gpu.read_output()
    o = orc.from_package_scene(sc)
    o.set_cam(cam)
    r_rgb, r_ids, _, _ = o.render(orc.MODE_PATH, W, H, spp=spp, seed=pseed)
    checks += 1
    err = float(np.nanmax(np.abs(rgb - r_rgb)))
    if not np.array_equal(ids, r_ids) or err > 1e-4:
        print(f"MISMATCH at check {checks}, frame {frames}: {int((ids != r_ids).sum())} id words differ, max radiance error {err:.3g} (spp {spp}, seed {pseed})", flush=True)
        gpu.render(MODE_PATH, spp=spp, seed=pseed)
        rgb2, ids2, _ = gpu.read_output()
        print(f"  the same frame once more: {int((ids2 != r_ids).sum())} id words differ, max radiance error {float(np.nanmax(np.abs(rgb2 - r_rgb))):.3g}", flush=True)
        sys.exit(1)
    if time.time() > next_report:
        print(f"{frames} path-traced frames, {edits} edits, {recentres} recentres, {checks} checks ok", flush=True)
        next_report = time.time() + 20
a = gpu.accel_info()
print(f"path soak ok: {frames} path-traced frames, {edits} edits, {recentres} recentres of the grid, {checks} checks against the oracle; whole-world builds {a.builds}, chunks rebuilt alone {a.chunk_builds}")
