"""Where the orbit loop's per-frame time goes: the same 2000 frames with more and more of the per-frame seam issued."""
import math, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voxelraytracing_amd import Gpu, MODE_PRIMARY_SHADOW, graphics as g, scenes
sc = scenes.c2()
gpu = Gpu(sc.world.max_nodes(), sc.world.size_in_chunks(), sc.size)
gpu.upload_world(sc.world, sc.materials); gpu.write_settings(sc.settings)
ORBIT = 48
params = []
for k in range(ORBIT):
    a = 2.0 * math.pi * k / ORBIT
    eye = (sc.eye[0] + 6.0 * math.cos(a), sc.eye[1] + 1.5 * math.sin(2 * a), sc.eye[2] + 6.0 * math.sin(a))
    rot = (sc.rot[0] + 3.0 * math.sin(a), sc.rot[1] + 8.0 * math.sin(a), sc.rot[2])
    params.append((rot, eye))
cams = [g.cam_data_create(r, e, 70.0, (1920.0, 1080.0)) for r, e in params]
wd = sc.world.world_data()
roots = sc.world.chunk_roots()
def loop(kind, n=2000):
    gpu.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        if kind >= 1: gpu.write_cam_data(cams[i % ORBIT] if kind < 4 else g.cam_data_create(*params[i % ORBIT], 70.0, (1920.0, 1080.0)))
        if kind >= 2: gpu.write_settings(sc.settings); gpu.write_world_data(wd)
        if kind >= 3: gpu.write_chunk_roots(roots if kind < 5 else sc.world.chunk_roots())
        gpu.render(MODE_PRIMARY_SHADOW)
    th = time.perf_counter() - t0
    gpu.synchronize(); return (time.perf_counter() - t0) / n * 1e6, th / n * 1e6
for kind, name in enumerate(["render only, standing camera", "+ camera (precomputed CamData)", "+ settings, world data", "+ chunk_roots (same array)",
                             "+ CamData::create per frame", "+ a fresh chunk_roots() per frame"]):
    loop(kind, 300)
    t, h = loop(kind)
    print(f"{name:36s} {t:6.1f} us per frame, host {h:5.1f} us")
