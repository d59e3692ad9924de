"""Cost of a voxel edit between two frames: range upload + identical-table check + table rebuild + the frame."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voxelraytracing_amd import Gpu, MODE_PRIMARY_SHADOW, scenes
sc = scenes.c2()
gpu = Gpu(sc.world.max_nodes(), sc.world.size_in_chunks(), sc.size)
gpu.upload_world(sc.world, sc.materials); gpu.write_cam_data(sc.cam); gpu.write_settings(sc.settings)
for _ in range(50): gpu.render(MODE_PRIMARY_SHADOW)
gpu.synchronize()
t0=time.perf_counter()
for _ in range(200): gpu.render(MODE_PRIMARY_SHADOW)
gpu.synchronize(); base=(time.perf_counter()-t0)/200
ex,ey,ez=(int(v) for v in sc.eye)
ts=[]; bs=[]
for k in range(40):
    p=(ex+(k%7)-3, ey-8-(k%5), ez+(k%9)-4)
    try: start,n=sc.world.set_voxel(p, 4 if k%2 else 0)
    except Exception: continue
    t0=time.perf_counter()
    gpu.write_nodes(sc.world.nodes_ptr(), start, start+n)
    gpu.write_chunk_roots(sc.world.chunk_roots())
    gpu.render(MODE_PRIMARY_SHADOW); gpu.synchronize()
    ts.append(time.perf_counter()-t0); bs.append(gpu.accel_info().last_build_ms)
ts.sort(); bs.sort()
print("frame %.1f us; edit+upload+rebuild+frame median %.1f us; rebuild (events) median %.1f us, n=%d" % (base*1e6, ts[len(ts)//2]*1e6, bs[len(bs)//2]*1e3, len(ts)))
