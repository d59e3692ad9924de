#!/usr/bin/env python3
"""tools/r6/present_trace.py — the client's draw + present loop (main.rs:452-454) under a kernel trace:
    rocprofv3 --kernel-trace --stats -- python3 tools/r6/present_trace.py [declared|undeclared]
`declared` (vrt_set_presentation): the frames' own launches store the window's image and vrt_present_device launches nothing — the
trace holds N march kernels and no blit; `undeclared`: N march kernels + N present_plain_kernel."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from voxelraytracing_amd import MODE_PRIMARY_SHADOW, scenes   # noqa: E402
from voxelraytracing_amd import graphics as g   # noqa: E402

how = sys.argv[1] if len(sys.argv) > 1 else "declared"
sc = scenes.c2()
gpu = g.Gpu(sc.world.max_nodes(), sc.world.size_in_chunks(), sc.size, device=0)
gpu.upload_world(sc.world, sc.materials)
gpu.write_cam_data(sc.cam)
gpu.write_settings(sc.settings)
if how == "declared":
    gpu.set_presentation()
N = 200
for k in range(N):
    gpu.write_cam_data(g.cam_data_create((sc.rot[0], sc.rot[1] + 0.05 * k, 0.0), sc.eye, 70.0, (float(sc.size[0]), float(sc.size[1]))))
    gpu.render(MODE_PRIMARY_SHADOW)
    gpu.present_device()
gpu.synchronize()
print(f"{how}: {N} frames drawn and presented")
