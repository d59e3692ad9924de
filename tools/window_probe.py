"""tools/window_probe.py — the window bounce launch's waves one by one (experiment build: tools/ab/build_variant.sh windbg
"-DVRT_EXP_WINDBG", VRT_LIB=tools/ab/libvrt_windbg.so): where a wave's time goes (staging the window, phase A, the march inside
the window, the march of the rays that left it, phase C), how many wave-steps and lane-steps each march takes."""
import ctypes as C
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voxelraytracing_amd import Gpu, MODE_PATH, _ffi, scenes

sc = scenes.c4()
if len(sys.argv) > 1:
    sc = scenes.c5((1920, 1080), chunks=int(sys.argv[1]))
gpu = Gpu(sc.world.max_nodes(), sc.world.size_in_chunks(), sc.size)
gpu.upload_world(sc.world, sc.materials)
gpu.write_settings(sc.settings)
gpu.write_cam_data(sc.cam)
gpu.set_frames_in_flight(1)
lib = _ffi.vrt()
buf = np.zeros((16384, 8), dtype=np.uint64)
lib.vrt_exp_win_dbg.argtypes = [C.c_void_p]
for _ in range(20):
    gpu.render(MODE_PATH)
gpu.synchronize()
lib.vrt_exp_win_dbg(buf.ctypes.data)
gpu.render(MODE_PATH)
gpu.synchronize()
lib.vrt_exp_win_dbg(buf.ctypes.data)
raw = buf[buf[:, 1] > 0]
lo = np.uint64(0xFFFFFFFF)
f = lambda col, hi=False: ((raw[:, col] >> np.uint64(32)) if hi else (raw[:, col] & lo)).astype(float)
rays, left = f(0), f(0, True)
t0, t1 = raw[:, 1].astype(float), raw[:, 2].astype(float)
us = lambda ticks: ticks * 16.0 / 2400.0   # (shader clock ~ 2.4 GHz)
m16 = np.uint64(0xFFFF)
stage, a, b0, b1, c = us(f(3)), us(f(3, True)), us(f(4)), us(f(4, True)), us(f(5))
rf0, rf1 = us(((raw[:, 5] >> np.uint64(32)) & m16).astype(float)), us(((raw[:, 5] >> np.uint64(48)) & m16).astype(float))
ws0, rn0 = (raw[:, 6] & m16).astype(float), ((raw[:, 6] >> np.uint64(16)) & m16).astype(float)
ws1, rn1 = ((raw[:, 6] >> np.uint64(32)) & m16).astype(float), ((raw[:, 6] >> np.uint64(48)) & m16).astype(float)
ls0, ls1 = f(7), f(7, True)
s, e = (t0 - t0.min()) / 100, (t1 - t0.min()) / 100
life = e - s
print(f"waves {len(raw)}, ray-segments per wave {rays.mean():.0f}, of which left the window {100 * left.sum() / rays.sum():.1f} %")
print(f"launch span {e.max():.1f} us; wave start p50 {np.percentile(s, 50):.1f} p99 {np.percentile(s, 99):.1f}; wave end p10 {np.percentile(e, 10):.1f} p50 {np.percentile(e, 50):.1f} p90 {np.percentile(e, 90):.1f} max {e.max():.1f}; mean life {life.mean():.1f} us")
print(f"a wave's time (us, mean): staging {stage.mean():.1f}, A {a.mean():.1f}, march in the window {b0.mean():.1f}, march outside {b1.mean():.1f}, C {c.mean():.1f}")
print(f"wave-steps per wave: in the window {ws0.mean():.1f} at {ls0.sum() / max(ws0.sum(), 1):.1f} lanes, {b0.sum() / max(ws0.sum(), 1):.3f} us each; outside {ws1.mean():.1f} at {ls1.sum() / max(ws1.sum(), 1):.1f} lanes, {b1.sum() / max(ws1.sum(), 1):.3f} us each")
print(f"hand-out rounds that took rays: in the window {rn0.mean():.1f} per wave, {rf0.sum() / max(rn0.sum(), 1):.2f} us each = {rf0.mean():.1f} us of a wave's {b0.mean():.1f}: a wave-step without them {(b0.sum() - rf0.sum()) / max(ws0.sum(), 1):.3f} us; outside {rn1.mean():.1f}, {rf1.sum() / max(rn1.sum(), 1):.2f} us each = {rf1.mean():.1f} us: a wave-step without them {(b1.sum() - rf1.sum()) / max(ws1.sum(), 1):.3f} us")
print(f"lane-steps: in the window {ls0.sum():.0f} ({100 * ls0.sum() / (ls0.sum() + ls1.sum()):.1f} %), outside {ls1.sum():.0f}")
print(f"ray-segments per wave: p10 {np.percentile(rays, 10):.0f} p50 {np.percentile(rays, 50):.0f} p90 {np.percentile(rays, 90):.0f} max {rays.max():.0f}; wave life p10 {np.percentile(life, 10):.0f} p50 {np.percentile(life, 50):.0f} p90 {np.percentile(life, 90):.0f} p99 {np.percentile(life, 99):.0f} max {life.max():.0f} us; "
      f"life ~ rays: corr {np.corrcoef(rays, life)[0, 1]:.2f}; us per ray-segment p10 {np.percentile(life / np.maximum(rays, 1), 10):.3f} p50 {np.percentile(life / np.maximum(rays, 1), 50):.3f} p90 {np.percentile(life / np.maximum(rays, 1), 90):.3f}")
steps = ws0 + ws1
print(f"wave-steps per wave p50 {np.percentile(steps, 50):.0f} p90 {np.percentile(steps, 90):.0f} p99 {np.percentile(steps, 99):.0f} max {steps.max():.0f}; life ~ wave-steps: corr {np.corrcoef(steps, life)[0, 1]:.2f}")
for t in np.linspace(0, e.max(), 12):
    alive = (s <= t) & (e > t)
    print(f"  t={t:6.1f} us: {int(alive.sum()):5d} waves alive")
