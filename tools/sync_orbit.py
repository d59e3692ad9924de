"""tools/sync_orbit.py — a host that WAITS for every frame (render, synchronise; one frame at a time) under the bench's orbit: frame period with
the tile order made from the frame before (VRT_TILE_ORDER_MOVING=1) and without."""
import math
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voxelraytracing_amd import Gpu, MODE_PRIMARY_SHADOW, graphics as g, scenes
W, H, ORBIT = 1920, 1080, 48
sc = scenes.c2((W, H))
gpu = Gpu(sc.world.max_nodes(), sc.world.size_in_chunks(), sc.size)
gpu.upload_world(sc.world, sc.materials); gpu.write_settings(sc.settings)
gpu.set_frames_in_flight(1)
cams = []
for k in range(ORBIT):
    a = 2.0 * math.pi * k / ORBIT
    cams.append(g.cam_data_create((sc.rot[0] + 3.0 * math.sin(a), sc.rot[1] + 8.0 * math.sin(a), 0.0),
                                  (sc.eye[0] + 6.0 * math.cos(a), sc.eye[1] + 1.5 * math.sin(2 * a), sc.eye[2] + 6.0 * math.sin(a)), 70.0, (float(W), float(H))))
for mode in ("synchronise", "present to host"):
    def frames(n):
        for i in range(n):
            gpu.write_cam_data(cams[i % ORBIT])
            gpu.render(MODE_PRIMARY_SHADOW)
            if mode == "synchronise": gpu.synchronize()
            else: gpu.present((W, H))
    frames(100)
    t0 = time.perf_counter()
    frames(1000)
    print(f"VRT_TILE_ORDER_MOVING={os.environ.get('VRT_TILE_ORDER_MOVING', '0')}  render + {mode}: {(time.perf_counter() - t0) / 1000 * 1e6:.1f} us per frame; ordered frames {gpu.accel_info().ordered_frames}")
