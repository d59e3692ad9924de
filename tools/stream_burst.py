"""tools/stream_burst.py — the join burst (main.rs:289-295 drains every pending GiveChunkData per frame; a 30^3 join is 27 000 of them): N chunk
ranges re-uploaded before every frame for N up to several hundred, and the per-chunk cost."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voxelraytracing_amd import Gpu, MODE_PRIMARY_SHADOW, scenes
sc = scenes.c2()
gpu = Gpu(sc.world.max_nodes(), sc.world.size_in_chunks(), sc.size)
gpu.upload_world(sc.world, sc.materials); gpu.write_cam_data(sc.cam); gpu.write_settings(sc.settings)
for _ in range(100): gpu.render(MODE_PRIMARY_SHADOW)
gpu.synchronize()
roots = sorted(int(r) for r in sc.world.chunk_roots() if r)
ranges = [(roots[i], roots[i + 1]) for i in range(len(roots) - 1)]
print(f"{len(ranges)} chunks, mean range {sum(b - a for a, b in ranges) / len(ranges) * 2 / 1024:.0f} KiB")
k = 0
def frames(n, per_frame):
    global k
    host = 0.0
    for _ in range(n):
        t0 = time.perf_counter()
        for _ in range(per_frame):
            a, b = ranges[k % len(ranges)]; k += 7
            gpu.write_nodes(sc.world.nodes_ptr(), a, b)
        host += time.perf_counter() - t0
        gpu.write_chunk_roots(sc.world.chunk_roots(), tag=sc.world.roots_generation())
        gpu.render(MODE_PRIMARY_SHADOW)
    return host / n
import gc
gc.collect(); gc.disable()
frames(50, 4); gpu.synchronize()
base = None
for per_frame in (0, 16, 32, 64, 128, 256, 400):
    n = 100
    t0 = time.perf_counter()
    h = frames(n, per_frame)
    gpu.synchronize()
    dt = (time.perf_counter() - t0) / n
    if base is None: base = dt
    a = gpu.accel_info()
    per = (dt - base) / per_frame * 1e6 if per_frame else 0.0
    print(f"{per_frame:3d} chunk uploads per frame: {dt * 1e6:8.1f} us per frame, {per:5.2f} us per chunk beyond the plain frame (vrt_write_nodes calls {h * 1e6:7.1f} us of it)   whole-world builds {a.builds}, chunks rebuilt alone {a.chunk_builds}")
