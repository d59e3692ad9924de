"""tools/chunk_probe.py — where a single chunk's rebuild (accel_chunks_kernel) spends its time: the phases of workgroup 0, from
stamps of an experiment build (tools/ab/build_variant.sh chunkdbg "-DVRT_EXP_CHUNKDBG"; run with VRT_LIB=tools/ab/libvrt_chunkdbg.so).
Lone voxel edits as tools/edit_cost.py makes them: edit, range upload, chunk_roots, frame, synchronise."""
import ctypes as C
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voxelraytracing_amd import Gpu, MODE_PRIMARY_SHADOW, scenes, _ffi
S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sc = scenes.procedural(S, (1920, 1080), MODE_PRIMARY_SHADOW)
gpu = Gpu(sc.world.max_nodes(), sc.world.size_in_chunks(), sc.size)
gpu.upload_world(sc.world, sc.materials); gpu.write_cam_data(sc.cam); gpu.write_settings(sc.settings)
lib = _ffi.vrt()
lib.vrt_exp_chunk_dbg.argtypes = [C.c_void_p]
for _ in range(20): gpu.render(MODE_PRIMARY_SHADOW)
gpu.synchronize()
buf = np.zeros(16, dtype=np.uint64)
lib.vrt_exp_chunk_dbg(buf.ctypes.data)
ex, ey, ez = (int(v) for v in sc.eye)
rows = []
for k in range(40):
    p = (ex + (k % 7) - 3, ey - 8 - (k % 5), ez + (k % 9) - 4)
    try: start, n = sc.world.set_voxel(p, 4 if k % 2 else 0)
    except Exception: continue
    gpu.write_nodes(sc.world.nodes_ptr(), start, start + n)
    gpu.write_chunk_roots(sc.world.chunk_roots())
    gpu.render(MODE_PRIMARY_SHADOW)
    gpu.synchronize()
    lib.vrt_exp_chunk_dbg(buf.ctypes.data)
    if buf[0] and buf[6]:
        rows.append((buf.copy(), n))
names = ["start", "staging loads stored to LDS", "barrier", "descent, rank, thread 0's bookkeeping", "barrier", "-", "the split cells' bricks, entries and march cells stored (a wave per cell)", "leaf cells stored"]
a = np.array([r[0] for r in rows]).astype(np.float64)
ghz = np.median((a[:, 6] - a[:, 0]) / ((a[:, 14] - a[:, 8]) * 10.0))   # shader clocks per ns
print(f"{len(rows)} lone edits, nodes uploaded per edit {np.median([r[1] for r in rows]):.0f}; shader clock {ghz:.2f} GHz (against the 100 MHz clock)")
for i in (1, 2, 3, 4, 7, 6):
    print(f"  until the last wave is past '{names[i]}': {np.median(a[:, i] - a[:, 0]) / ghz / 1e3:6.2f} us")

