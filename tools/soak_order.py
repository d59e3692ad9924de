"""tools/soak_order.py [seconds] [seed] — a long random session of a one-frame-at-a-time context whose tile order follows the view
(the exact order of a view at rest, the KEPT block order of a moving one: csrc/vrt_order.hip, hold_limits) against a context that
always launches in screen order (VRT_TILE_ORDER=0): camera steps of every size (the bench's orbit step, strides, leaps), rests,
voxel edits in front of frames, changes of settings, of the mode, of the number of frames in flight, resizes of the result texture —
the same calls to both, and EVERY frame compared word for word.  Any order of the tiles is the same frame; what this looks for is an
order that is not one (a stale buffer after a resize, an order read while it is written, a permutation of another frame's tiles).
Exit status 1 on the first mismatch."""
import math
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voxelraytracing_amd import Gpu, MODE_PRIMARY, MODE_PRIMARY_SHADOW, graphics as g, scenes

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
SIZES = [(320, 184), (256, 144), (640, 360), (200, 104), (1920, 1080)]
size = SIZES[0]
sc = scenes.c2(size)


def make(order_on):
    if not order_on:
        os.environ["VRT_TILE_ORDER"] = "0"
    gp = Gpu(sc.world.max_nodes(), sc.world.size_in_chunks(), size)
    os.environ.pop("VRT_TILE_ORDER", None)
    gp.upload_world(sc.world, sc.materials)
    gp.write_settings(sc.settings)
    gp.set_frames_in_flight(1)
    return gp


mov, ref = make(True), make(False)
rot = [float(sc.rot[0]), float(sc.rot[1])]
eye = [float(v) for v in sc.eye]
mode = MODE_PRIMARY_SHADOW
in_flight = 1
frames = edits = resizes = 0
pace = 0
t_end = time.time() + seconds
next_report = time.time() + 20
import collections
ops = collections.deque(maxlen=60)
settings = sc.settings
while time.time() < t_end:
    # a stretch of one pace: rest, the orbit's step, a stride, a leap
    pace = int(rng.choice([0, 1, 1, 1, 2, 2, 3]))
    for _ in range(int(rng.integers(3, 40))):
        r = rng.random()
        if r < 0.08:
            p = (int(eye[0]) + int(rng.integers(-24, 25)), int(eye[1]) + int(rng.integers(-28, 6)), int(eye[2]) + int(rng.integers(-24, 25)))
            try:
                start, n = sc.world.set_voxel(p, int(rng.choice([0, 0, 3, 4, 40, 47, 62])))
                for gp in (mov, ref):
                    gp.write_nodes(sc.world.nodes_ptr(), start, start + n)
                edits += 1
                ops.append(("edit", p))
            except Exception as e:
                if getattr(e, "kind", "") not in ("NoChange", "NoChunk", "OutOfMemory"):
                    raise
        elif r < 0.10:
            settings.sun_intensity = float(rng.choice([1.0, 2.0, 4.0]))
            for gp in (mov, ref):
                gp.write_settings(settings)
            ops.append(("settings",))
        elif r < 0.13:
            mode = MODE_PRIMARY if mode == MODE_PRIMARY_SHADOW else MODE_PRIMARY_SHADOW
            ops.append(("mode", mode))
        elif r < 0.15:
            in_flight = int(rng.choice([1, 1, 1, 2]))
            for gp in (mov, ref):
                gp.set_frames_in_flight(in_flight)
            ops.append(("in flight", in_flight))
        elif r < 0.16:
            size = SIZES[int(rng.integers(0, len(SIZES) - (0 if rng.random() < 0.1 else 1)))]   # (full HD now and then)
            for gp in (mov, ref):
                gp.resize_result_texture(size)
            resizes += 1
            ops.append(("resize", size))
        step = [(0.0, 0.0, 0.0), (0.3, 1.0, 0.8), (1.0, 3.5, 2.0), (5.0, 35.0, 12.0)][pace]
        rot[0] = max(-80.0, min(80.0, rot[0] + step[0] * float(rng.uniform(-1, 1))))
        rot[1] += step[1] * float(rng.uniform(0.3, 1.0)) * (1 if frames % 97 < 60 else -1)
        a = float(rng.uniform(0, 2 * math.pi))
        eye[0] += step[2] * math.cos(a) * 0.7
        eye[2] += step[2] * math.sin(a) * 0.7
        eye[1] += step[2] * 0.1 * float(rng.uniform(-1, 1))
        for k in (0, 2):   # (stay over the terrain)
            eye[k] = min(max(eye[k], float(sc.eye[k]) - 60.0), float(sc.eye[k]) + 60.0)
        eye[1] = min(max(eye[1], float(sc.eye[1]) - 10.0), float(sc.eye[1]) + 30.0)
        cam = g.cam_data_create((rot[0], rot[1], 0.0), tuple(eye), 70.0, (float(size[0]), float(size[1])))
        for gp in (mov, ref):
            gp.write_cam_data(cam)
            gp.encode_pass(mode)
        ops.append(("frame", pace, mode, size))
        a_rgb, a_ids, _ = mov.read_output()
        b_rgb, b_ids, _ = ref.read_output()
        frames += 1
        if not (np.array_equal(a_ids, b_ids) and np.array_equal(a_rgb.view(np.uint32), b_rgb.view(np.uint32))):
            bad = np.argwhere(a_ids != b_ids)
            print(f"MISMATCH at frame {frames}: {len(bad)} id words differ (first {tuple(bad[0]) if len(bad) else '-'}); last operations:", flush=True)
            for o in ops:
                print("   ", o)
            sys.exit(1)
    if time.time() > next_report:
        print(f"... {frames} frames, {mov.accel_info().ordered_frames} launched in an order, {edits} edits, {resizes} resizes", flush=True)
        next_report = time.time() + 20
print(f"soak_order: {frames} frames equal to the screen-order context's ({mov.accel_info().ordered_frames} of them launched in an order; "
      f"{edits} edits, {resizes} resizes; seed {seed}); the screen-order context ordered {ref.accel_info().ordered_frames}")
mov.close(); ref.close()
