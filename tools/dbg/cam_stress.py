"""camera changes under 1-3 frames in flight, every frame after a change checked (is a frame ever traced with the camera before?)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from voxelraytracing_amd import Gpu, MODE_PRIMARY, MODE_PRIMARY_SHADOW, graphics as g, scenes
from oracle import orc
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
rng = np.random.default_rng(seed)
W, H = 160, 96
sc = scenes.c2((W, H))
gpu = Gpu(sc.world.max_nodes(), sc.world.size_in_chunks(), sc.size)
gpu.upload_world(sc.world, sc.materials); gpu.write_settings(sc.settings); gpu.write_cam_data(sc.cam)
ex, ey, ez = (float(v) for v in sc.eye)
o = orc.from_package_scene(sc)
bad = 0
for it in range(iters):
    if rng.random() < 0.2:
        gpu.set_frames_in_flight(int(rng.integers(1, 4)))
    mode = MODE_PRIMARY if rng.random() < 0.5 else MODE_PRIMARY_SHADOW
    for _ in range(int(rng.integers(0, 4))):   # frames with the old camera, still in flight
        gpu.render(mode)
    rot = (float(rng.uniform(-40, 10)), float(rng.uniform(0, 360)), 0.0)
    eye = (ex + float(rng.uniform(-6, 6)), ey + float(rng.uniform(-3, 3)), ez + float(rng.uniform(-6, 6)))
    cam = g.cam_data_create(rot, eye, float(rng.uniform(50, 100)), (float(W), float(H)))
    gpu.write_cam_data(cam)
    o.set_cam(cam)
    r_rgb, r_ids, _, _ = o.render(mode, W, H)
    for k in range(int(rng.integers(1, 4))):
        gpu.render(mode)
        rgb, ids, _ = gpu.read_output()
        d = int((ids != r_ids).sum())
        if d:
            bad += 1
            print(f"iteration {it}, frame {k} after the change: {d} id words differ", flush=True)
    if bad >= 5: break
print(f"{it + 1} iterations, {bad} bad", flush=True)
