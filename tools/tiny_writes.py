"""tools/tiny_writes.py — a host that writes the pool a few nodes at a time: N two-node vrt_write_nodes calls before a frame."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voxelraytracing_amd import Gpu, MODE_PRIMARY_SHADOW, scenes
sc = scenes.c2()
gpu = Gpu(sc.world.max_nodes(), sc.world.size_in_chunks(), sc.size)
gpu.upload_world(sc.world, sc.materials); gpu.write_cam_data(sc.cam); gpu.write_settings(sc.settings)
for _ in range(50): gpu.render(MODE_PRIMARY_SHADOW)
gpu.synchronize()
import gc; gc.disable()
for n in (100, 1000, 10000, 50000):
    t0 = time.perf_counter()
    for f in range(5):
        for i in range(n):
            a = 2 + 2 * ((i * 7919) % 200000)
            gpu.write_nodes(sc.world.nodes_ptr(), a, a + 2)
        gpu.render(MODE_PRIMARY_SHADOW)
    gpu.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"{n:6d} two-node writes per frame: {dt * 1e3:8.2f} ms per frame, {dt / n * 1e6:6.2f} us per write")
