"""Fixed (non-march) cost of the default primary + shadow kernel: frames whose rays stop at once, one launch at a time.
   outside       the camera is outside the world: every ray is a miss before its first lookup (no shadow rays)
   inside_solid  the camera sits in solid rock: every primary ray hits at its first lookup and launches a shadow ray that
                 does the same
   normal        the bench's frame"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voxelraytracing_amd import Gpu, MODE_PRIMARY_SHADOW, graphics as g, scenes

sc = scenes.c2()
gpu = Gpu(sc.world.max_nodes(), sc.world.size_in_chunks(), sc.size)
gpu.upload_world(sc.world, sc.materials)
gpu.write_settings(sc.settings)
gpu.set_frames_in_flight(1)
cases = {"outside": (-10.0, 300.0, -10.0), "inside_solid": (128.5, 20.5, 128.5), "normal": sc.eye}
for name, eye in cases.items():
    gpu.write_cam_data(g.cam_data_create(sc.rot, eye, 70.0, (1920.0, 1080.0)))
    for _ in range(300):
        gpu.render(MODE_PRIMARY_SHADOW)
    gpu.stats()
    for _ in range(500):
        gpu.render(MODE_PRIMARY_SHADOW)
    s = gpu.stats()
    print(f"{name:13s} {s.sum_ms_primary / s.frames * 1e3:7.1f} us per launch (1 in flight, {s.frames} frames)")
