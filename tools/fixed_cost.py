"""Fixed (non-march) cost of the primary kernel: frames where every ray stops immediately.
Run under rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES to read VALU instructions per wave for each case."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voxelraytracing_amd import Gpu, MODE_PRIMARY_SHADOW, MODE_PRIMARY, graphics as g, scenes

sc = scenes.c2()
gpu = Gpu(sc.world.max_nodes(), sc.world.size_in_chunks(), sc.size)
gpu.upload_world(sc.world, sc.materials)
gpu.write_settings(sc.settings)
cases = {"outside": (-10.0, 300.0, -10.0), "inside_solid": (128.5, 20.5, 128.5), "normal": sc.eye}
for name, eye in cases.items():
    gpu.write_cam_data(g.cam_data_create(sc.rot, eye, 70.0, (1920.0, 1080.0)))
    for mode in (MODE_PRIMARY, MODE_PRIMARY_SHADOW):
        gpu.render(mode)
        gpu.synchronize()
    s = gpu.stats()
    print(name, "primary us", s.ms_primary * 1e3, "shadow us", s.ms_secondary * 1e3)
