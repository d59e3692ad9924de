"""The first seconds of load: frame period per 100-frame window from a cold start (what bench.py's settling has to sit out)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voxelraytracing_amd import Gpu, MODE_PRIMARY_SHADOW, scenes
sc = scenes.c2()
gpu = Gpu(sc.world.max_nodes(), sc.world.size_in_chunks(), sc.size, device=0)
gpu.upload_world(sc.world, sc.materials); gpu.write_settings(sc.settings); gpu.write_cam_data(sc.cam)
gpu.render(MODE_PRIMARY_SHADOW); gpu.synchronize()
t_start = time.perf_counter()
out = []
for w in range(80):
    t0 = time.perf_counter()
    for i in range(100): gpu.render(MODE_PRIMARY_SHADOW)
    gpu.synchronize()
    out.append((time.perf_counter() - t_start, (time.perf_counter() - t0) / 100 * 1e6))
print(" ".join(f"{t*1e3:.0f}ms:{p:.1f}" for t, p in out))
