import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from voxelraytracing_amd import Gpu, MODE_PATH, scenes
for bounces in (2, 3, 4):
    sc = scenes.c4((320, 184), bounces=bounces)
    gpu = Gpu(sc.world.max_nodes(), sc.world.size_in_chunks(), sc.size)
    gpu.upload_world(sc.world, sc.materials); gpu.write_settings(sc.settings); gpu.write_cam_data(sc.cam)
    gpu.render(MODE_PATH, stats=True, spp=1, seed=0)
    rgb, ids, _ = gpu.read_output()
    for rep in range(2):
        gpu.render(MODE_PATH, spp=1, seed=0)
        rgb2, ids2, _ = gpu.read_output()
        d = np.abs(rgb2 - rgb).max(axis=2)
        bad = np.argwhere(d > 0)
        print(f"bounces {bounces} rep {rep}: {len(bad)} pixels differ, max {d.max():.4g}; first {bad[:6].tolist()}", flush=True)
        if len(bad):
            y, x = bad[0]
            print("   ", rgb[y, x], rgb2[y, x])
