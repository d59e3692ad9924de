"""One context over several devices (vrt_config.device_ids, include/vrt.h) — SURVEY.md §8(b)'s boundary for a multi-GPU
node: the caller keeps the reference's shape (one thread, one GpuResources, main.rs:398-455) and the backend shards the
frame behind vrt_render.  The pool has one GPU per box, so every device id here is 0: the same code path (N contexts,
messages stored into device 0's receive buffer, event-ordered assembly) with the peer stores staying on one device.
The frame must be, bit for bit, the single-device frame."""
import os
import subprocess

import numpy as np
import pytest

from voxelraytracing_amd import MODE_PATH, MODE_PRIMARY, MODE_PRIMARY_SHADOW, scenes
from voxelraytracing_amd import graphics as g

from util import gpu_for_scene

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def c2_small():
    return scenes.c2((640, 360))


def reference_frames(sc, cams, mode, **kw):
    one = gpu_for_scene(sc)
    out = []
    for cam in cams:
        one.write_cam_data(cam)
        one.render(mode, **kw)
        out.append(one.read_output()[:2])
    st = None
    one.render(mode, stats=True, **kw)
    st = one.stats()
    one.close()
    return out, st


@pytest.mark.parametrize("threads", ["0", "1"])
@pytest.mark.parametrize("n", [2, 3, 8])
def test_group_frames_equal_single_device_frames(c2_small, n, threads, monkeypatch):
    # VRT_GROUP_THREADS: one issuing thread per device besides the caller's (the default when the device ids differ)
    monkeypatch.setenv("VRT_GROUP_THREADS", threads)
    w, h = c2_small.size
    cams = [g.cam_data_create((18.0 + 7 * k, 30.0 + 33 * k, 0.0), (c2_small.eye[0] + k, c2_small.eye[1], c2_small.eye[2] - k), 70.0,
                              (float(w), float(h))) for k in range(5)]
    want, st1 = reference_frames(c2_small, cams, MODE_PRIMARY_SHADOW)
    grp = gpu_for_scene(c2_small, devices=[0] * n)
    assert grp.shard_info()[2] == (w // 8) * (h // 8)
    # frames enqueued back to back (two in flight, alternating message slots): read-backs return the last one
    for upto in (1, 2, 5):
        for cam in cams[:upto]:
            grp.write_cam_data(cam)
            grp.render(MODE_PRIMARY_SHADOW)
        rgb, ids, _ = grp.read_output()
        assert np.array_equal(ids, want[upto - 1][1]) and np.array_equal(rgb, want[upto - 1][0]), f"{n} devices, after {upto} frames"
    # primary-only frames, one at a time
    grp.set_frames_in_flight(1)
    one = gpu_for_scene(c2_small)
    one.write_cam_data(cams[2])
    one.render(MODE_PRIMARY)
    grp.write_cam_data(cams[2])
    grp.render(MODE_PRIMARY)
    a, b = grp.read_output(), one.read_output()
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0])
    # the counters of a stats frame are the sum over the devices = the single device's
    grp.write_cam_data(cams[-1])
    grp.render(MODE_PRIMARY_SHADOW, stats=True)
    s = grp.stats()
    assert (s.primary_rays, s.secondary_rays, s.hits, s.steps, s.node_visits) == \
           (st1.primary_rays, st1.secondary_rays, st1.hits, st1.steps, st1.node_visits)
    rgb, ids, _ = grp.read_output()
    assert np.array_equal(ids, want[-1][1]) and np.array_equal(rgb, want[-1][0])
    # presentation comes from device 0's frame
    one.write_cam_data(cams[-1])
    one.render(MODE_PRIMARY_SHADOW)
    assert np.array_equal(grp.present(), one.present())
    grp.close()
    one.close()


def test_group_follows_edits_uploads_and_resizes(orc):
    sc = scenes.c1_flat((128, 128))
    grp = gpu_for_scene(sc, devices=[0, 0, 0])
    one = gpu_for_scene(sc)
    for pos, v in [((32, 12, 40), 0), ((30, 13, 44), 4), ((34, 13, 44), 3)]:
        start, n = sc.world.set_voxel(pos, v)
        for gp in (grp, one):
            gp.write_nodes(sc.world.nodes_ptr(), start, start + n)
            gp.write_chunk_roots(sc.world.chunk_roots())
            gp.render(MODE_PRIMARY_SHADOW)
        a, b = grp.read_output(), one.read_output()
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0])
    r_rgb, r_ids, _, _ = orc.from_package_scene(sc).render(orc.MODE_PRIMARY_SHADOW, 128, 128)
    assert np.array_equal(a[1], r_ids)
    ai = grp.accel_info()
    assert ai.available and ai.chunk_builds > 0
    for gp in (grp, one):
        gp.resize_result_texture((192, 96))
        gp.write_cam_data(g.cam_data_create(sc.rot, sc.eye, 70.0, (192.0, 96.0)))
        gp.render(MODE_PRIMARY_SHADOW)
    a, b = grp.read_output(), one.read_output()
    assert a[1].shape == (96, 192) and np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0])
    ptr, nbytes = grp.device_output()
    assert ptr and nbytes == 192 * 96 * 16          # device 0's row-major frame: what a host with GPU interop presents
    # what belongs to a caller that runs its own collective is refused here
    for call in (lambda: grp.set_stream(0), lambda: grp.bind_output(0), lambda: grp.assemble(1, 1), lambda: grp.read_steps()):
        with pytest.raises(g.VrtError):
            call()
    with pytest.raises(g.VrtError):
        grp.render(MODE_PATH)                       # 8-byte records cannot carry a path-traced pixel
    grp.close()
    one.close()


@pytest.mark.parametrize("threads", ["0", "1"])
def test_group_is_consistent_after_a_frame_that_failed_on_a_shard(c2_small, threads, monkeypatch):
    """A frame that one device refuses — here the step-count view, which 8-byte records cannot carry — has taken a message
    slot and may have been enqueued on other devices: the group drains and starts over, and the frames after the error
    are the single-device frames again (round 3's advisor finding: the slot's `consumed` event was never recorded)."""
    monkeypatch.setenv("VRT_GROUP_THREADS", threads)
    sc = c2_small
    w, h = sc.size
    cams = [g.cam_data_create((18.0 + 9 * k, 30.0 + 41 * k, 0.0), (sc.eye[0] + k, sc.eye[1], sc.eye[2] - k), 70.0, (float(w), float(h))) for k in range(4)]
    want, _ = reference_frames(sc, cams, MODE_PRIMARY_SHADOW)
    grp = gpu_for_scene(sc, devices=[0, 0, 0])
    grp.write_cam_data(cams[0])
    grp.render(MODE_PRIMARY_SHADOW)                # slot 0 in use
    debug = g.make_settings(sun_pos=scenes.SUN_POS)
    debug.show_step_count = 1
    grp.write_settings(debug)
    for _ in range(3):                             # an odd number of failures: the slot sequence would be out of step
        with pytest.raises(g.VrtError):
            grp.render(MODE_PRIMARY_SHADOW)
    grp.write_settings(sc.settings)
    for k, cam in enumerate(cams):                 # back to back, both slots
        grp.write_cam_data(cam)
        grp.render(MODE_PRIMARY_SHADOW)
    rgb, ids, _ = grp.read_output()
    assert np.array_equal(ids, want[3][1]) and np.array_equal(rgb, want[3][0])
    for k in (1, 2):
        grp.write_cam_data(cams[k])
        grp.render(MODE_PRIMARY_SHADOW)
        rgb, ids, _ = grp.read_output()
        assert np.array_equal(ids, want[k][1]) and np.array_equal(rgb, want[k][0]), f"frame {k} after the failed frames"
    grp.close()


def test_group_with_texel_messages_runs_every_kind_of_frame(orc):
    sc = scenes.c4((256, 144), bounces=3)
    grp = gpu_for_scene(sc, devices=[0, 0, 0, 0], texel_messages=True)
    one = gpu_for_scene(sc)
    for mode, kw in ((MODE_PATH, dict(spp=2, seed=5)), (MODE_PRIMARY_SHADOW, dict(variant=1)), (MODE_PRIMARY_SHADOW, dict(variant=3)),
                     (MODE_PRIMARY_SHADOW, {})):
        for gp in (grp, one):
            gp.render(mode, **kw)
            gp.render(mode, **kw)
        a, b = grp.read_output(), one.read_output()
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0]), f"mode {mode} {kw}"
    grp.close()
    one.close()


def test_c3_frame_through_a_group_of_8(orc):
    """Config C3 as a host of the C ABI sees it: one context, 8 devices, 1920x1080 over the 16^3-chunk world."""
    sc = scenes.c3()
    grp = gpu_for_scene(sc, devices=[0] * 8)
    grp.render(MODE_PRIMARY_SHADOW)
    grp.render(MODE_PRIMARY_SHADOW)
    rgb, ids, _ = grp.read_output()
    r_rgb, r_ids, _, _ = orc.from_package_scene(sc).render(orc.MODE_PRIMARY_SHADOW, 1920, 1080)
    assert np.array_equal(ids, r_ids) and float(np.abs(rgb - r_rgb).max()) <= 1e-4
    grp.close()


def test_cpp_host_renders_through_a_multi_device_context(tmp_path):
    """csrc/host/example_frame_loop.cpp — the reference's frame loop on the C++ mirror, vrt_* calls only — with its Gpu
    created over three devices: the file it writes equals the single-device run's."""
    exe = os.path.join(ROOT, "voxelraytracing_amd", "vrt_frame_loop")
    outs = []
    for extra in ([], ["0,0,0"]):
        out = tmp_path / f"frame{len(extra)}.bin"
        r = subprocess.run([exe, str(out), "256", "256"] + extra, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        outs.append(np.fromfile(out, dtype=np.uint32))
    assert np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("extra", [["--gpus", "2"], ["--gpus", "3", "--single-process"]])
def test_bench_starts_its_own_ranks_and_prints_one_line(extra):
    """`python bench.py --gpus N` without a launcher: the script starts its own N processes (one per GPU, here all on this
    one GPU with a gloo gather staged through the host) and relays rank 0's JSON line; `--single-process` drives N device
    contexts through one multi-device vrt context instead.  Both verify the assembled frame against an unsharded render."""
    import json
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *extra, "--rehearse-on-one-gpu", "--steps", "12", "--warmup", "2",
                        "--width", "640", "--height", "360"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == int(extra[1]) and d["value"] > 0 and d["steps"] == 12 and d["scaling"] == "strong"


@pytest.mark.parametrize("texels", [False, True])
@pytest.mark.parametrize("n", [2, 3])
def test_staged_messages_when_peer_stores_are_refused(c2_small, n, texels):
    """VRT_FLAG_STAGED_MESSAGES forces what a refused hipDeviceEnablePeerAccess falls back to: a device renders its message
    into a buffer of its own and copies it to device 0 (hipMemcpyPeerAsync on its own stream) before signalling `done`.
    With VRT_FLAG_POISON_MESSAGES on top — device 0 fills a slot with 0xFF once it has assembled it — a frame that read a
    slot before its senders had written it again could not equal the single-device frame."""
    w, h = c2_small.size
    cams = [g.cam_data_create((20.0 + 5 * k, 40.0 + 29 * k, 0.0), (c2_small.eye[0] - k, c2_small.eye[1] + 0.5 * k, c2_small.eye[2] + k), 70.0,
                              (float(w), float(h))) for k in range(6)]
    want, _ = reference_frames(c2_small, cams, MODE_PRIMARY_SHADOW)
    grp = gpu_for_scene(c2_small, devices=[0] * n, staged_messages=True, poison_messages=True, texel_messages=texels)
    for upto in (1, 2, 3, 6):
        for cam in cams[:upto]:
            grp.write_cam_data(cam)
            grp.render(MODE_PRIMARY_SHADOW)
        rgb, ids, _ = grp.read_output()
        assert np.array_equal(ids, want[upto - 1][1]) and np.array_equal(rgb, want[upto - 1][0]), f"{n} devices, after {upto} frames"
    grp.set_frames_in_flight(1)
    for cam in cams[:3]:
        grp.write_cam_data(cam)
        grp.render(MODE_PRIMARY_SHADOW)
    rgb, ids, _ = grp.read_output()
    assert np.array_equal(ids, want[2][1]) and np.array_equal(rgb, want[2][0])
    if texels:   # the path trace travels as texels
        sc = scenes.c4((320, 184), bounces=3)
        one = gpu_for_scene(sc)
        one.render(MODE_PATH, spp=2, seed=11)
        r1, i1, _ = one.read_output()
        one.close()
        pg = gpu_for_scene(sc, devices=[0] * n, staged_messages=True, poison_messages=True, texel_messages=True)
        for _ in range(3):
            pg.render(MODE_PATH, spp=2, seed=11)
        r2, i2, _ = pg.read_output()
        assert np.array_equal(i2, i1) and np.array_equal(r2, r1)
        pg.close()
    grp.close()


@pytest.mark.parametrize("n", [2, 8])
def test_poisoned_message_slots_never_reach_a_frame(c2_small, n):
    """Peer stores (the default), two frames in flight, a different camera every frame: device 0 poisons every slot it has
    consumed; every frame read back equals the single-device frame of its camera."""
    w, h = c2_small.size
    cams = [g.cam_data_create((15.0 + 3 * k, 11.0 * k, 0.0), (c2_small.eye[0] + 2 * k, c2_small.eye[1], c2_small.eye[2] - k), 70.0,
                              (float(w), float(h))) for k in range(8)]
    want, _ = reference_frames(c2_small, cams, MODE_PRIMARY_SHADOW)
    grp = gpu_for_scene(c2_small, devices=[0] * n, poison_messages=True)
    for k, cam in enumerate(cams):
        grp.write_cam_data(cam)
        grp.render(MODE_PRIMARY_SHADOW)
        if k % 3 == 2 or k == len(cams) - 1:
            rgb, ids, _ = grp.read_output()
            assert np.array_equal(ids, want[k][1]) and np.array_equal(rgb, want[k][0]), f"frame {k}"
    grp.close()
