"""tools/pool_probe.py — per-phase cycle counts of the pool bounce kernel's waves (experiment build: -DVRT_EXP_POOLDBG,
VRT_LIB=tools/ab/libvrt_pooldbg.so).  For the last bounce launch of a C4 frame: waves, cycles in phases A / B / C,
wave-steps and refills per wave, and when the waves start and end on the 100 MHz clock."""
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
lib.vrt_exp_pool_dbg.argtypes = [C.c_void_p]
for _ in range(50):
    gpu.render(MODE_PATH)
gpu.synchronize()
lib.vrt_exp_pool_dbg(buf.ctypes.data)
gpu.render(MODE_PATH)     # the records hold the frame's last bounce launch (each launch overwrites them)
gpu.synchronize()
lib.vrt_exp_pool_dbg(buf.ctypes.data)
import sys as _s
want_cont = len(_s.argv) > 1 and _s.argv[1] == "cont"
raw = buf[buf[:, 0] > 0]
raw = raw[((raw[:, 0] >> np.uint64(32)) != 0) == want_cont]
print("straggler-chain launches" if want_cont else "bounce launches")
raw[:, 0] &= np.uint64(0xFFFFFFFF)
d = raw.astype(np.float64)
n, a, b, c, steps, refills, r0, r1 = d.T
t0 = r0.min()
steps = (raw[:, 4] & np.uint64(0xFFFFFFFF)).astype(np.float64)
dry_steps = (raw[:, 4] >> np.uint64(32)).astype(np.float64)
wet_lanes = (raw[:, 5] & np.uint64(0xFFFFFFFF)).astype(np.float64)
dry_lanes = (raw[:, 5] >> np.uint64(32)).astype(np.float64)
wet_steps = steps - dry_steps
print(f"wave-steps with rays left in the pool: {wet_steps.mean():.1f} per wave at {wet_lanes.sum() / wet_steps.sum():.1f} lanes marching after the step; "
      f"after the pool ran dry: {dry_steps.mean():.1f} at {dry_lanes.sum() / max(dry_steps.sum(), 1):.1f} lanes")
print(f"ray-steps per ray {(wet_lanes.sum() + dry_lanes.sum() + n.sum()) / n.sum():.1f} (approx.)")
print(f"waves {len(d)}  rays/wave {n.mean():.1f}  launch span {(r1.max() - t0) / 100:.1f} us")
print(f"cycles per wave, mean (p10 / p50 / p90 / max): A {a.mean():.0f} ({np.percentile(a, 10):.0f} / {np.percentile(a, 50):.0f} / {np.percentile(a, 90):.0f} / {a.max():.0f})")
print(f"   B {b.mean():.0f} ({np.percentile(b, 10):.0f} / {np.percentile(b, 50):.0f} / {np.percentile(b, 90):.0f} / {b.max():.0f})")
print(f"   C {c.mean():.0f} ({np.percentile(c, 10):.0f} / {np.percentile(c, 50):.0f} / {np.percentile(c, 90):.0f} / {c.max():.0f})")
print(f"wave-steps per wave {steps.mean():.1f} (p50 {np.percentile(steps, 50):.0f}, p90 {np.percentile(steps, 90):.0f}, max {steps.max():.0f}); ideal {(n.sum() and 0) or 0}")
print(f"cycles per wave-step {b.sum() / steps.sum():.0f}")
s = (r0 - t0) / 100
e = (r1 - t0) / 100
print(f"wave start us: p10 {np.percentile(s, 10):.1f} p50 {np.percentile(s, 50):.1f} p90 {np.percentile(s, 90):.1f} max {s.max():.1f}")
print(f"wave end   us: p10 {np.percentile(e, 10):.1f} p50 {np.percentile(e, 50):.1f} p90 {np.percentile(e, 90):.1f} max {e.max():.1f}")
print(f"wave life  us: mean {(e - s).mean():.1f} p50 {np.percentile(e - s, 50):.1f} p90 {np.percentile(e - s, 90):.1f} max {(e - s).max():.1f}")
# how many waves are alive over time
for t in np.linspace(0, e.max(), 12):
    print(f"  t={t:6.1f} us alive {int(((s <= t) & (e > t)).sum())}")
