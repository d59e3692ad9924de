"""tools/present_cost.py — vrt_present_device alone (behind a frame, synchronised) for windows over a 1920x1080 result texture: the reference keeps the
texture at 1080 rows and the window's aspect (main.rs:255-262), so only a window 1080 rows tall is the texture's size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voxelraytracing_amd import Gpu, MODE_PRIMARY_SHADOW, scenes
sc = scenes.c2((1920, 1080))
gpu = Gpu(sc.world.max_nodes(), sc.world.size_in_chunks(), sc.size)
gpu.upload_world(sc.world, sc.materials); gpu.write_cam_data(sc.cam); gpu.write_settings(sc.settings)
gpu.set_frames_in_flight(1)
for _ in range(20): gpu.render(MODE_PRIMARY_SHADOW)
gpu.synchronize()
def loop(n, screen):
    t0 = time.perf_counter()
    for _ in range(n):
        gpu.render(MODE_PRIMARY_SHADOW)
        if screen: gpu.present_device(screen)
        gpu.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
loop(50, None)
base = loop(300, None)
for screen in ((1920, 1080), (1280, 720), (2560, 1440), (3840, 2160), (1600, 900)):
    loop(20, screen)
    t = loop(300, screen)
    print(f"window {screen[0]}x{screen[1]}: render + present + synchronise {t:.1f} us, the blit's share {t - base:.1f} us (render + synchronise {base:.1f})")
