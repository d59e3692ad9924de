"""recentre + chunk arrival + edits under 1-3 frames in flight, checked after every burst (the scenario of the soak's mismatch)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from voxelraytracing_amd import Gpu, MODE_PRIMARY, MODE_PRIMARY_SHADOW, graphics as g, scenes
from oracle import orc
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
rng = np.random.default_rng(seed)
W, H = 160, 96
sc = scenes.c2((W, H))
gpu = Gpu(sc.world.max_nodes(), sc.world.size_in_chunks(), sc.size)
gpu.upload_world(sc.world, sc.materials); gpu.write_settings(sc.settings); gpu.write_cam_data(sc.cam)
ex, ey, ez = (float(v) for v in sc.eye)
cam = sc.cam
center0 = tuple(int(v) // 32 + 4 for v in sc.world.min_voxel())
center = center0
bad = 0
for it in range(iters):
    nf = int(rng.integers(1, 4))
    gpu.set_frames_in_flight(nf)
    mode = MODE_PRIMARY if rng.random() < 0.5 else MODE_PRIMARY_SHADOW
    pending = []
    pre = int(rng.integers(0, 8))
    for _ in range(pre):   # frames with edits before
        for _ in range(int(rng.integers(0, 3))):
            p = (int(ex) + int(rng.integers(-24, 25)), int(ey) + int(rng.integers(-28, 6)), int(ez) + int(rng.integers(-24, 25)))
            try:
                start, n = sc.world.set_voxel(p, int(rng.choice([0, 0, 3, 4, 40])))
            except Exception as e:
                if getattr(e, "kind", "") == "OutOfMemory":
                    start, n = e.range
                else:
                    continue
            gpu.write_nodes(sc.world.nodes_ptr(), start, start + n)
        gpu.write_chunk_roots(sc.world.chunk_roots(), tag=sc.world.roots_generation())
        gpu.render(mode)
    did = False
    if rng.random() < 0.8:
        ax = int(rng.integers(0, 3)); step = int(rng.choice([-1, 1]))
        nc = list(center); nc[ax] = min(max(nc[ax] + step, center0[ax] - 2), center0[ax] + 2)
        if tuple(nc) != center:
            center = tuple(nc)
            sc.world.center_chunks(center)
            pending.extend(sc.world.generate_missing(0, 1).tolist())
            gpu.write_world_data(sc.world.world_data())
            did = True
    per = int(rng.integers(1, 12))
    while pending:
        for root, n in pending[:per]:
            gpu.write_nodes(sc.world.nodes_ptr(), int(root), int(root) + int(n))
        del pending[:per]
        if pending or rng.random() < 0.5:
            if rng.random() < 0.5:
                p = (int(ex) + int(rng.integers(-24, 25)), int(ey) + int(rng.integers(-28, 6)), int(ez) + int(rng.integers(-24, 25)))
                try:
                    start, n = sc.world.set_voxel(p, int(rng.choice([0, 0, 3, 4, 40])))
                    gpu.write_nodes(sc.world.nodes_ptr(), start, start + n)
                except Exception as e:
                    if getattr(e, "kind", "") == "OutOfMemory":
                        gpu.write_nodes(sc.world.nodes_ptr(), e.range[0], e.range[0] + e.range[1])
            gpu.write_chunk_roots(sc.world.chunk_roots(), tag=sc.world.roots_generation())
            gpu.render(mode)
    gpu.write_chunk_roots(sc.world.chunk_roots(), tag=sc.world.roots_generation())
    gpu.render(mode)
    rgb, ids, _ = gpu.read_output()
    o = orc.from_package_scene(sc); o.set_cam(cam)
    r_rgb, r_ids, _, _ = o.render(mode, W, H)
    d = int((ids != r_ids).sum())
    if d:
        bad += 1
        gpu.render(mode); _, ids2, _ = gpu.read_output()
        print(f"iteration {it}: {d} id words differ (in flight {nf}, {pre} frames before, recentred {did}, {per} ranges per frame); once more: {int((ids2 != r_ids).sum())}", flush=True)
        if bad >= 5: break
print(f"{it + 1} iterations, {bad} bad", flush=True)
