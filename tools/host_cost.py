"""Host time of one vrt_render call (trivial frames: the camera is outside the world, the GPU keeps up easily)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voxelraytracing_amd import Gpu, MODE_PRIMARY_SHADOW, graphics as g, scenes
sc = scenes.c2()
gpu = Gpu(sc.world.max_nodes(), sc.world.size_in_chunks(), sc.size)
gpu.upload_world(sc.world, sc.materials); gpu.write_settings(sc.settings)
gpu.write_cam_data(g.cam_data_create(sc.rot, (-10.0, 300.0, -10.0), 70.0, (1920.0, 1080.0)))
for _ in range(1200): gpu.render(MODE_PRIMARY_SHADOW)
gpu.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(2000): gpu.render(MODE_PRIMARY_SHADOW)
    th = (time.perf_counter() - t0) / 2000 * 1e6
    gpu.synchronize()
    print("host %.1f us per vrt_render, with the GPU %.1f us" % (th, (time.perf_counter() - t0) / 2000 * 1e6))
