"""tools/op_point.py [frames in flight] [present: 1|0] — bench.py's `operating_point` leg alone (the client's real frame: 30^3 chunks, untagged
chunk_roots rewrite, render + blit per frame), for profilers: rocprofv3 --kernel-trace --stats -- python3 tools/op_point.py 2"""
import math
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voxelraytracing_amd import Gpu, MODE_PRIMARY_SHADOW, graphics as g, scenes
from voxelraytracing_amd.world import ClientWorld, gen_height
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 2
present = (sys.argv[2] if len(sys.argv) > 2 else "1") == "1"
W, H, ORBIT = 1920, 1080, 48
sc = scenes.c2((W, H))
S_OP, player = 30, (7, -3, 11)
w_op = ClientWorld(player, 1 << 27, S_OP)
w_op.generate(0, 1)
px, pz = player[0] * 32 + 16, player[2] * 32 + 16
eye = (px + 0.5, float(gen_height(1, px, pz)) + 24.5, pz + 0.5)
gp = Gpu(w_op.max_nodes(), S_OP, (W, H))
gp.upload_world(w_op, sc.materials)
gp.write_settings(sc.settings)
wd = w_op.world_data()
cams = []
for k in range(ORBIT):
    a = 2.0 * math.pi * k / ORBIT
    cams.append(g.cam_data_create((20.0 + 3.0 * math.sin(a), 35.0 + 8.0 * math.sin(a), 0.0),
                                  (eye[0] + 6.0 * math.cos(a), eye[1] + 1.5 * math.sin(2 * a), eye[2] + 6.0 * math.sin(a)), 70.0, (float(W), float(H))))


def frames(n):
    for i in range(n):
        gp.write_settings(sc.settings)
        gp.write_cam_data(cams[i % ORBIT])
        gp.write_chunk_roots(w_op.chunk_roots())
        gp.write_world_data(wd)
        gp.render(MODE_PRIMARY_SHADOW)
        if present:
            gp.present_device((W, H))


gp.set_frames_in_flight(nf)
frames(100)
gp.synchronize()
t0 = time.perf_counter()
frames(1000)
gp.synchronize()
print(f"{nf} in flight, present {present}: {(time.perf_counter() - t0) / 1000 * 1e6:.1f} us per frame")
