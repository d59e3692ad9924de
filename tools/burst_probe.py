import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from voxelraytracing_amd import Gpu, MODE_PRIMARY_SHADOW, scenes
import numpy as np
sc = scenes.c2()
gpu = Gpu(sc.world.max_nodes(), sc.world.size_in_chunks(), sc.size)
gpu.upload_world(sc.world, sc.materials); gpu.write_cam_data(sc.cam); gpu.write_settings(sc.settings)
for _ in range(100): gpu.render(MODE_PRIMARY_SHADOW)
gpu.synchronize()
roots = sorted(int(r) for r in sc.world.chunk_roots() if r)
ranges = [(roots[i], roots[i + 1]) for i in range(len(roots) - 1)]
import gc; gc.disable()
k = 0
N = 256
acc = np.zeros(N); rend = []
for f in range(60):
    for i in range(N):
        a, b = ranges[k % len(ranges)]; k += 7
        t0 = time.perf_counter(); gpu.write_nodes(sc.world.nodes_ptr(), a, b); acc[i] += time.perf_counter() - t0
    gpu.write_chunk_roots(sc.world.chunk_roots(), tag=sc.world.roots_generation())
    t0 = time.perf_counter(); gpu.render(MODE_PRIMARY_SHADOW); rend.append(time.perf_counter() - t0)
gpu.synchronize()
acc /= 60
print("us per call by position in the frame:", " ".join(f"{acc[i:i+16].mean()*1e6:.1f}" for i in range(0, N, 16)))
print("render us:", np.median(rend) * 1e6)
