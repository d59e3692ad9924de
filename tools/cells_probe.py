"""tools/cells_probe.py — the bounce launch's waves one by one (experiment build: -DVRT_EXP_CELLDBG -DVRT_CELLS_NO_WAVES_ATTR,
VRT_LIB=tools/ab/libvrt_celldbg.so): lane occupancy of the march while the wave's pool has rays and after it ran dry, and when
the waves start and end — is the launch waiting for work or for its slowest waves?"""
import ctypes as C
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voxelraytracing_amd import Gpu, MODE_PATH, _ffi, scenes

sc = scenes.c4()
gpu = Gpu(sc.world.max_nodes(), sc.world.size_in_chunks(), sc.size)
gpu.upload_world(sc.world, sc.materials)
gpu.write_settings(sc.settings)
gpu.write_cam_data(sc.cam)
gpu.set_frames_in_flight(1)
lib = _ffi.vrt()
buf = np.zeros((16384, 8), dtype=np.uint64)
lib.vrt_exp_cells_dbg.argtypes = [C.c_void_p]
for _ in range(20):
    gpu.render(MODE_PATH)
gpu.synchronize()
lib.vrt_exp_cells_dbg(buf.ctypes.data)
gpu.render(MODE_PATH)
gpu.synchronize()
lib.vrt_exp_cells_dbg(buf.ctypes.data)
raw = buf[buf[:, 0] > 0]
lo = np.uint64(0xFFFFFFFF)
n = raw[:, 0].astype(float)
t0, t1 = raw[:, 1].astype(float), raw[:, 2].astype(float)
wet, dry = (raw[:, 3] & lo).astype(float), (raw[:, 3] >> np.uint64(32)).astype(float)
wl, dl = (raw[:, 4] & lo).astype(float), (raw[:, 4] >> np.uint64(32)).astype(float)
start = t0.min()
s, e = (t0 - start) / 100, (t1 - start) / 100
print(f"waves {len(raw)}, rays per wave {n.mean():.0f}; wave-steps per wave {(wet + dry).mean():.0f} (p50 {np.percentile(wet + dry, 50):.0f}, p90 {np.percentile(wet + dry, 90):.0f}, "
      f"max {(wet + dry).max():.0f}); pool has rays: {wet.sum():.0f} at {wl.sum() / wet.sum():.1f} lanes; pool dry: {dry.sum():.0f} ({100 * dry.sum() / (wet + dry).sum():.0f} %) at {dl.sum() / dry.sum():.1f} lanes")
print(f"launch span {e.max():.1f} us; wave start p50 {np.percentile(s, 50):.1f} p99 {np.percentile(s, 99):.1f} max {s.max():.1f}; "
      f"wave end p10 {np.percentile(e, 10):.1f} p50 {np.percentile(e, 50):.1f} p90 {np.percentile(e, 90):.1f} p99 {np.percentile(e, 99):.1f} max {e.max():.1f}")
life = e - s
print(f"us per wave-step (wave life / its wave-steps): mean {(life / (wet + dry)).mean():.3f}, of the 1 % slowest waves {(life / (wet + dry))[e >= np.percentile(e, 99)].mean():.3f}")
air4, air8, air16 = (raw[:, 6] & lo).astype(float).sum(), (raw[:, 6] >> np.uint64(32)).astype(float).sum(), raw[:, 7].astype(float).sum()
print(f"lookups {wl.sum() + dl.sum():.0f}: in an air leaf of the cell grid (4 voxels or more) {100 * air4 / (wl.sum() + dl.sum()):.1f} %, of 8 or more {100 * air8 / (wl.sum() + dl.sum()):.1f} %, of 16 or more {100 * air16 / (wl.sum() + dl.sum()):.1f} %")
for t in np.linspace(0, e.max(), 16):
    alive = (s <= t) & (e > t)
    print(f"  t={t:6.1f} us: {int(alive.sum()):5d} waves alive")
