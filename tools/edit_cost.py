"""Cost of a voxel edit between two frames (main.rs:352-362): range upload of the chunk (incl. its 2048-node slack), the
per-frame chunk_roots rewrite, the table update of that chunk, the frame.  Three figures: a lone frame with a synchronise
behind it, the same with an edit in front, and the frame period of a pipelined loop with an edit before every frame."""
import os
import sys
import time
import gc
gc.disable()   # (a full collection would be a 30-40 ms pause inside a measurement)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voxelraytracing_amd import Gpu, MODE_PRIMARY_SHADOW, scenes
S = int(sys.argv[1]) if len(sys.argv) > 1 else 8     # world size in chunks (8 = C2, 32 = C5's world)
sc = scenes.procedural(S, (1920, 1080), MODE_PRIMARY_SHADOW)
gpu = Gpu(sc.world.max_nodes(), sc.world.size_in_chunks(), sc.size)
gpu.upload_world(sc.world, sc.materials); gpu.write_cam_data(sc.cam); gpu.write_settings(sc.settings)
for _ in range(50): gpu.render(MODE_PRIMARY_SHADOW)
gpu.synchronize()
t0=time.perf_counter()
for _ in range(200): gpu.render(MODE_PRIMARY_SHADOW)
gpu.synchronize(); base=(time.perf_counter()-t0)/200
lone=[]
for _ in range(30):
    t0=time.perf_counter(); gpu.render(MODE_PRIMARY_SHADOW); gpu.synchronize(); lone.append(time.perf_counter()-t0)
lone.sort()
ex,ey,ez=(int(v) for v in sc.eye)
ts=[]; bs=[]
for k in range(40):
    p=(ex+(k%7)-3, ey-8-(k%5), ez+(k%9)-4)
    try: start,n=sc.world.set_voxel(p, 4 if k%2 else 0)
    except Exception: continue
    t0=time.perf_counter()
    gpu.write_nodes(sc.world.nodes_ptr(), start, start+n)
    t1=time.perf_counter()
    gpu.write_chunk_roots(sc.world.chunk_roots())
    t2=time.perf_counter()
    gpu.render(MODE_PRIMARY_SHADOW)
    t3=time.perf_counter()
    gpu.synchronize()
    t4=time.perf_counter()
    ts.append(t4-t0); bs.append((t1-t0,t2-t1,t3-t2,t4-t3))
ts.sort()
print("lone edit, host us: write_nodes %.1f, chunk_roots + write_chunk_roots (untagged) %.1f, render %.1f, synchronise (the device's share) %.1f" %
      tuple(sorted(b[i] for b in bs)[len(bs)//2]*1e6 for i in range(4)))
edits=[]
for k in range(200):
    p=(ex+(k%7)-3, ey-8-(k%5), ez+(k%9)-4)
    try: edits.append(sc.world.set_voxel(p, 5 if k%2 else 0) and p)
    except Exception: pass
gpu.synchronize()
# the per-frame seam's table rewrite alone (main.rs:446), table unchanged: a fresh copy + the backend's compare, and tagged
seam=[]
for tagged in (False, True):
    t0=time.perf_counter()
    for _ in range(300):
        gpu.write_chunk_roots(sc.world.chunk_roots(), tag=sc.world.roots_generation() if tagged else 0)
    seam.append((time.perf_counter()-t0)/300*1e6)
print("unchanged-table rewrite, host us per frame: chunk_roots() + vrt_write_chunk_roots %.1f; + tagged with the world's generation %.1f" % tuple(seam))
n_pipe=0; host=[0.0,0.0,0.0,0.0]
t0=time.perf_counter()
for k in range(200):   # an edit before every frame, nothing waits for the device
    p=(ex+(k%7)-3, ey-8-(k%5), ez+(k%9)-4)
    h0=time.perf_counter()
    try: start,n=sc.world.set_voxel(p, 4 if k%2 else 0)
    except Exception: continue
    h1=time.perf_counter()
    gpu.write_nodes(sc.world.nodes_ptr(), start, start+n)
    h2=time.perf_counter()
    gpu.write_chunk_roots(sc.world.chunk_roots(), tag=sc.world.roots_generation())
    h3=time.perf_counter()
    gpu.render(MODE_PRIMARY_SHADOW); n_pipe+=1
    h4=time.perf_counter()
    for i,d in enumerate((h1-h0,h2-h1,h3-h2,h4-h3)): host[i]+=d
gpu.synchronize(); pipe=(time.perf_counter()-t0)/max(n_pipe,1)
print("host us per edit frame: set_voxel %.1f, write_nodes %.1f, chunk_roots + write_chunk_roots %.1f, render (incl. the table update's launch) %.1f" % tuple(h/max(n_pipe,1)*1e6 for h in host))
a = gpu.accel_info()
print("%d^3 world: pipelined frame period %.1f us, with an edit before every frame %.1f us (n=%d); lone frame + synchronise %.1f us, "
      "edit + range upload + chunk_roots rewrite + table update + frame + synchronise %.1f us (n=%d); whole-world builds %d, chunks rebuilt alone %d" %
      (S, base*1e6, pipe*1e6, n_pipe, lone[len(lone)//2]*1e6, ts[len(ts)//2]*1e6, len(ts), a.builds, a.chunk_builds))
