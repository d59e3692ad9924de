"""tools/resize_cost.py — a window being dragged: vrt_resize_output to a new size before every frame (main.rs:257-262 recreates the result texture
on every Resized event), render, present at that size."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voxelraytracing_amd import Gpu, MODE_PRIMARY_SHADOW, graphics as g, scenes
sc = scenes.c2((1920, 1080))
gpu = Gpu(sc.world.max_nodes(), sc.world.size_in_chunks(), sc.size)
gpu.upload_world(sc.world, sc.materials); gpu.write_cam_data(sc.cam); gpu.write_settings(sc.settings)
for _ in range(20): gpu.render(MODE_PRIMARY_SHADOW)
gpu.synchronize()
for name, sizes in (("shrinking by 8 columns a frame", [(1920 - 8 * k, 1080) for k in range(1, 41)]),
                    ("growing by 8 columns a frame", [(1600 + 8 * k, 1080) for k in range(1, 41)]),
                    ("jittering +-16 columns", [(1760 + (16 if k % 2 else -16), 1080) for k in range(40)])):
    ts = []
    for (w, h) in sizes:
        t0 = time.perf_counter()
        gpu.resize_result_texture((w, h))
        gpu.write_cam_data(g.cam_data_create(sc.rot, sc.eye, 70.0, (float(w), float(h))))
        gpu.render(MODE_PRIMARY_SHADOW)
        gpu.present_device((w, h))
        gpu.synchronize()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    print(f"{name}: resize + render + present + synchronise: median {ts[len(ts) // 2] * 1e6:.0f} us, worst {ts[-1] * 1e6:.0f} us (a plain frame + present + synchronise: ~ 130 us)")
