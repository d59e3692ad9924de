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
if len(sys.argv) > 1:   # another world size (chunks): how the wave-step's time depends on the table's size against the 4 MB L2s
    sc = scenes.c5((1920, 1080), chunks=int(sys.argv[1]))
gpu = Gpu(sc.world.max_nodes(), sc.world.size_in_chunks(), sc.size)
gpu.upload_world(sc.world, sc.materials)
gpu.write_settings(sc.settings)
gpu.write_cam_data(sc.cam)
gpu.set_frames_in_flight(1)
lib = _ffi.vrt()
buf = np.zeros((16384, 8), dtype=np.uint64)
lib.vrt_exp_cells_dbg.argtypes = [C.c_void_p]
tot = np.zeros(32, dtype=np.uint64)
lib.vrt_exp_cells_tot.argtypes = [C.c_void_p]
for _ in range(20):
    gpu.render(MODE_PATH)
gpu.synchronize()
lib.vrt_exp_cells_dbg(buf.ctypes.data)
lib.vrt_exp_cells_tot(tot.ctypes.data)
gpu.render(MODE_PATH)
gpu.synchronize()
lib.vrt_exp_cells_dbg(buf.ctypes.data)
lib.vrt_exp_cells_tot(tot.ctypes.data)
raw = buf[buf[:, 0] > 0]
life_sum = ((raw[:, 2].astype(float) - raw[:, 1].astype(float)) / 100).sum()
lo = np.uint64(0xFFFFFFFF)
n = (raw[:, 0] & np.uint64(0xFFFFFFFF)).astype(float)
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
air4, air8, air16 = (raw[:, 6] & lo).astype(float).sum(), (raw[:, 6] >> np.uint64(32)).astype(float).sum(), (raw[:, 7] & lo).astype(float).sum()
refills, refill_us = (raw[:, 5] >> np.uint64(32)).astype(float), (raw[:, 7] >> np.uint64(32)).astype(float) * 16.0 / 2400.0   # (shader clock ~ 2.4 GHz)
park_us = (raw[:, 0] >> np.uint64(32)).astype(float) * 16.0 / 2400.0
load_us = ((raw[:, 5] >> np.uint64(8)) & np.uint64(0xFFFFFF)).astype(float) * 16.0 / 2400.0
print(f"  of a round: parking {park_us.sum() / max(refills.sum(), 1):.2f} us, until the records are there {load_us.sum() / max(refills.sum(), 1):.2f} us, the set-up behind them {(refill_us.sum() - park_us.sum() - load_us.sum()) / max(refills.sum(), 1):.2f} us")
print(f"hand-out rounds that took rays: {refills.mean():.1f} per wave, {refill_us.sum() / max(refills.sum(), 1):.2f} us each (park + the records' loads + the set-up, waited for): {100 * refill_us.sum() / life_sum:.1f} % of the waves' time")
print(f"lookups {wl.sum() + dl.sum():.0f}: in an air leaf of the cell grid (4 voxels or more) {100 * air4 / (wl.sum() + dl.sum()):.1f} %, of 8 or more {100 * air8 / (wl.sum() + dl.sum()):.1f} %, of 16 or more {100 * air16 / (wl.sum() + dl.sum()):.1f} %")
for t in np.linspace(0, e.max(), 16):
    alive = (s <= t) & (e > t)
    print(f"  t={t:6.1f} us: {int(alive.sum()):5d} waves alive")

# what a wave-step costs with the pool still handing out rays (57 lanes marching) and after it ran dry (15): least squares over
# the waves of life = a wet + b dry + c (phases A and C, the hand-outs)
A_ = np.stack([wet, dry, np.ones_like(wet)], axis=1)
coef, *_ = np.linalg.lstsq(A_, life, rcond=None)
res = life - A_ @ coef
print(f"wave life ~ {coef[0]:.3f} us x wave-steps while the pool has rays + {coef[1]:.3f} us x wave-steps after it ran dry + {coef[2]:.1f} us "
      f"(rms residual {np.sqrt((res ** 2).mean()):.1f} us of a mean life of {life.mean():.1f}); the dry steps are {100 * coef[1] * dry.sum() / life.sum():.0f} % of the waves' time")

# what a lookup finds and whether the lane's previous lookup was in the same line — the misses a ray cannot avoid — under
# today's layout (16-byte cells, a line = 8^3 voxels) and under denser ones
t = tot.astype(float)
L = t[0]
if L == 0:   # (the light probe build: no lookup classes)
    sys.exit(0)
pc = lambda a, b=None: f"{100 * a / (L if b is None else b):5.1f} %"
print(f"lookups {L:.0f}; into a line the lane's previous lookup was not in: 8x8x8 {pc(t[1])}, 16x8x8 (8-byte cells) {pc(t[2])}, 16x8x16 (4-byte) {pc(t[3])}")
print(f"  air leaves of 8 voxels or more {pc(t[4])} of the lookups, new line {pc(t[5], t[4])} of them; a byte per 8^3 voxels in lines of 64x32x32: new line {pc(t[6], t[4])}")
print(f"  air leaves of 4 voxels {pc(t[7])}, new line {pc(t[8], t[7])}, with 16x8x8 lines {pc(t[9], t[7])}")
print(f"  split cells {pc(t[10])}, new line {pc(t[11], t[10])}, with 16x8x8 lines {pc(t[12], t[10])}")
print(f"  other (solid leaves, the border) {pc(t[16])}, new line {pc(t[17], max(t[16], 1))}")
print(f"  a lane that was in a leaf of 8 or more finds something finer: {pc(t[13])} of the lookups; stays in such leaves {pc(t[18])}, of which into another coarse line {pc(t[19], max(t[18], 1))}")
print(f"  within 32 voxels of the ray's origin {pc(t[14])}, new line {pc(t[15], max(t[14], 1))} of them")
print(f"  distinct 128-byte lines per wave-step {t[20] / max(t[21], 1):.1f} (lanes marching {L / max(t[21], 1):.1f})")
