"""Frame period of a pipelined frame loop while chunks arrive (main.rs:289-295: create_chunk, the chunk's range re-uploaded,
chunk_roots rewritten): N chunk ranges re-uploaded before every frame, nothing waiting for the device."""
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
ranges = [r for r in sc.world.chunk_ranges() if r[1] > r[0]] if hasattr(sc.world, "chunk_ranges") else None
if ranges is None:
    roots = sorted(int(r) for r in sc.world.chunk_roots() if r)
    ranges = [(roots[i], roots[i + 1]) for i in range(len(roots) - 1)]
print(f"{len(ranges)} chunks, mean range {sum(b - a for a, b in ranges) / len(ranges) * 2 / 1024:.0f} KiB")
k = 0
def frames(n, per_frame):
    global k
    for _ in range(n):
        for _ in range(per_frame):
            a, b = ranges[k % len(ranges)]; k += 7
            gpu.write_nodes(sc.world.nodes_ptr(), a, b)
        gpu.write_chunk_roots(sc.world.chunk_roots(), tag=sc.world.roots_generation())
        gpu.render(MODE_PRIMARY_SHADOW)

import gc
gc.collect(); gc.disable()   # (a full collection is a ~30-40 ms pause of this thread around its 500th frame: the interpreter's, not the library's)
frames(300, 1); gpu.synchronize(); frames(300, 0); gpu.synchronize()
for per_frame in (0, 1, 2, 3, 4, 8, 16):
    t0 = time.perf_counter()
    frames(300, per_frame)
    gpu.synchronize()
    dt = (time.perf_counter() - t0) / 300
    a = gpu.accel_info()
    print(f"{per_frame:2d} chunk uploads per frame: {dt * 1e6:6.1f} us per frame   (whole-world builds {a.builds}, chunks rebuilt alone {a.chunk_builds})")
