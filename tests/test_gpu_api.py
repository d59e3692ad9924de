"""C-ABI behaviour of libvrt.so on a GPU: error codes instead of the reference's panics, write semantics of
NodeBuffer / ArrayBuffer / SimpleBuffer (clientdesktop/src/graphics/shader.rs:7-143), resizes, read-back."""
import ctypes as C

import numpy as np
import pytest

from voxelraytracing_amd import Gpu, MODE_PATH, MODE_PRIMARY, MODE_PRIMARY_SHADOW, VrtError, graphics as g, scenes

from util import assert_frame_parity, gpu_for_scene

pytestmark = pytest.mark.gpu


def test_state_and_range_errors():
    gpu = Gpu(1024, 2, (64, 64))
    with pytest.raises(VrtError) as e:
        gpu.read_output()
    assert e.value.code == -5 and "nothing rendered" in str(e.value)
    with pytest.raises(VrtError) as e:   # WorldData not written yet: size_in_chunks = 0
        gpu.render(MODE_PRIMARY)
    assert e.value.code == -5
    pool = np.zeros(4096, dtype=np.uint16)
    with pytest.raises(VrtError) as e:   # past the NodeBuffer's capacity
        gpu.write_nodes(pool, 1000, 1100)
    assert e.value.code == -2
    with pytest.raises(VrtError) as e:
        gpu.write_nodes(pool, 10, 5)
    assert e.value.code == -1
    with pytest.raises(VrtError) as e:   # [Material; 256]
        gpu.write_materials(g.std_materials(), first=10, n=250)
    assert e.value.code == -2
    with pytest.raises(VrtError) as e:
        gpu.resize_result_texture((0, 64))
    assert e.value.code == -1
    with pytest.raises(VrtError) as e:
        gpu.render(MODE_PRIMARY, variant=99)
    assert e.value.code == -1
    wd = g.WorldData()
    wd.size, wd.size_in_chunks = 96, 3   # 27 roots > the 8 allocated: resize_chunk_buffer first (main.rs:441-445)
    gpu.write_world_data(wd)
    with pytest.raises(VrtError) as e:
        gpu.render(MODE_PRIMARY)
    assert e.value.code == -5 and "vrt_resize_world" in str(e.value)
    gpu.resize_chunk_buffer(3)
    gpu.write_cam_data(g.cam_data_create((10.0, 20.0, 0.0), (48.5, 48.5, 48.5), 70.0, (64.0, 64.0)))
    gpu.write_settings(g.make_settings(sun_pos=scenes.SUN_POS))
    gpu.render(MODE_PRIMARY)            # fresh chunk_roots buffer: all 0 -> every chunk is the air leaf -> sky only
    rgb, ids, _ = gpu.read_output()
    assert not ids.any()


def test_node_writes_widen_to_even_bounds_and_chunk_roots_truncate(orc):
    """NodeBuffer::write (shader.rs:24-33) and ArrayBuffer::write (:134-135)."""
    sc = scenes.c1_flat((64, 64))
    gpu = gpu_for_scene(sc)
    gpu.render(MODE_PRIMARY)
    _, ids0, _ = gpu.read_output()
    # overwrite a copy of the pool with limestone leaves, upload only the odd range [root, root+1) of chunk 0:
    # the widened write covers node root-1 (the air leaf at 0 when root = 1) and root
    pool = sc.world.nodes().copy()
    root = int(sc.world.chunk_roots()[0])
    assert root == 1
    alt = pool.copy()
    alt[0], alt[1] = 0, 40           # chunk 0 becomes one solid grass leaf; node 0 stays air
    gpu.write_nodes(alt, 1, 2)
    gpu.render(MODE_PRIMARY)
    _, ids1, _ = gpu.read_output()
    o = orc.OracleScene(alt, sc.world.chunk_roots(), sc.materials, sc.cam, sc.settings, sc.world.world_data())
    _, r_ids, _, _ = o.render(orc.MODE_PRIMARY, 64, 64)
    assert np.array_equal(ids1, r_ids) and not np.array_equal(ids1, ids0)
    # more roots than the buffer holds: silently truncated; offset past the end is an error
    gpu.write_chunk_roots(np.concatenate([sc.world.chunk_roots(), np.full(100, 7, dtype=np.uint32)]))
    gpu.write_nodes(pool, 0, 2)
    gpu.render(MODE_PRIMARY)
    _, ids2, _ = gpu.read_output()
    assert np.array_equal(ids2, ids0)
    with pytest.raises(VrtError):
        gpu.write_chunk_roots(np.zeros(1, dtype=np.uint32), offset=9)


def test_resize_output_and_rgba8_readback(orc):
    sc = scenes.c2((128, 72))
    gpu = gpu_for_scene(sc)
    gpu.render(MODE_PRIMARY_SHADOW)
    a_rgb, a_ids, a_q = gpu.read_output(rgba8=True)
    # textureStore to rgba8unorm (ray_tracer.wgsl:179): clamp, x255, round to nearest even; alpha = 1
    want = np.rint(np.clip(a_rgb, 0.0, 1.0) * np.float32(255.0)).astype(np.uint8)
    assert np.array_equal(a_q[..., :3], want) and (a_q[..., 3] == 255).all()
    gpu.resize_result_texture((256, 144))     # GpuResources::resize_result_texture, mod.rs:201-211
    cam = g.cam_data_create(sc.rot, sc.eye, 70.0, (256.0, 144.0))
    gpu.write_cam_data(cam)
    with pytest.raises(VrtError):
        gpu.read_output()                     # a fresh texture has not been rendered to
    gpu.render(MODE_PRIMARY_SHADOW)
    b_rgb, b_ids, _ = gpu.read_output()
    o = orc.from_package_scene(sc)
    o.set_cam(cam)
    r_rgb, r_ids, _, _ = o.render(orc.MODE_PRIMARY_SHADOW, 256, 144)
    assert b_ids.shape == (144, 256) and np.array_equal(b_ids, r_ids)
    assert float(np.abs(b_rgb - r_rgb).max()) <= 1e-4


@pytest.mark.parametrize("size", [(360, 200), (364, 100), (1920, 1080), (16380, 8)])
def test_a_window_of_the_textures_size_is_blitted_by_the_short_kernel_to_the_same_bytes(orc, size):
    """Round 4: a window of the texture's size goes through present_plain_kernel — a texel quantised per pixel, the general
    routine only inside the crosshair's box — when the host has found every sample within 1e-4 of its own texel's centre.
    These sizes have columns and rows whose sample is NOT exactly the centre in binary32 (360: 40 of them, 364: 42 and a
    texture whose last four columns no workgroup stores to, 1920 x 1080: 51 and 42; 16 380 columns: one is 5e-4 off its texel,
    beyond what the short kernel's proof covers, so that window takes the general one): the bytes are the oracle's general
    bilinear blit, for every crosshair — none, the default cross, a dot, a cross as large as the window, sizes that are not numbers."""
    w, h = size
    sc = scenes.c2((w, h))
    gpu = gpu_for_scene(sc)
    gpu.render(MODE_PRIMARY_SHADOW)
    rgb, _, q = gpu.read_output(rgba8=True)
    kinds = [dict(style=0), dict(), dict(style=1, size=9.5, color=(1.0, 0.2, 0.1, 0.75)), dict(style=2, size=17.0, color=(0.0, 0.0, 0.0, 1.0))]
    if w <= 400:
        kinds += [dict(style=2, size=1.0e4, color=(0.3, 0.6, 0.9, 0.5)), dict(style=1, size=float("nan")), dict(style=2, size=-3.0), dict(style=1, size=0.4),
                  # colours that are not numbers reach every pixel (0 * NaN, 0 * inf): no shortcut may be taken for them
                  dict(style=2, size=7.0, color=(float("nan"), 0.5, 0.5, 0.5)), dict(style=1, size=7.0, color=(0.5, float("inf"), 0.5, 0.5)),
                  dict(style=2, size=7.0, color=(0.5, 0.5, 0.5, float("nan"))), dict(style=1, size=7.0, color=(0.5, 0.5, -float("inf"), float("inf")))]
    for kw in kinds:
        got = gpu.present((w, h), **kw)
        assert np.array_equal(got, orc.present(rgb, (w, h), **kw)), kw
        if w <= 400:   # ... and a window of another size (the general kernel skips the crosshair's arithmetic outside its box)
            other = (w + w // 3, h - h // 5)
            assert np.array_equal(gpu.present(other, **kw), orc.present(rgb, other, **kw)), (kw, other)
    assert np.array_equal(gpu.present((w, h), style=0), q)
    gpu.close()


@pytest.mark.parametrize("size", [(203, 77), (128, 75), (131, 72), (7, 40), (40, 5)])
def test_result_textures_of_any_size(orc, size):
    """The reference's result texture is 1080 rows at the window's aspect — any width (main.rs:257-262) — and its compute
    pass dispatches tex_size / 8 workgroups per axis (main.rs:452, no bounds test in the shader): whole 8x8 tiles are traced
    with the NDC of the full size, the columns and rows beyond them keep the fresh texture's zeros, alpha included.  Frames,
    step counts, the rgba8 texture and the presented image against the oracle; primary + shadow and the path trace; one
    and two frames in flight; a context created at that size and one resized to it."""
    w, h = size
    sc = scenes.c2((w, h))
    o = orc.from_package_scene(sc)
    r_rgb, r_ids, r_steps, _ = o.render(orc.MODE_PRIMARY_SHADOW, w, h, want_steps=True)
    cw, chh = w & ~7, h & ~7
    assert not r_ids[:, cw:].any() and not r_ids[chh:, :].any() and not r_rgb[:, cw:].any() and not r_rgb[chh:, :].any()
    made = gpu_for_scene(sc)
    resized = gpu_for_scene(scenes.c2((64, 64)))
    resized.resize_result_texture((w, h))
    resized.write_cam_data(sc.cam)
    for gpu in (made, resized):
        for _ in range(3):                                 # (every frame set of the frames in flight)
            gpu.render(MODE_PRIMARY_SHADOW)
        a_rgb, a_ids, a_q = gpu.read_output(rgba8=True)
        assert a_ids.shape == (h, w)
        assert_frame_parity(a_rgb, a_ids, r_rgb, r_ids, f"{w}x{h}")
        assert not a_rgb[:, cw:].any() and not a_rgb[chh:, :].any()
        want = np.zeros((h, w, 4), dtype=np.uint8)
        want[..., :3] = np.rint(np.clip(a_rgb, 0.0, 1.0) * np.float32(255.0)).astype(np.uint8)
        want[:chh, :cw, 3] = 255
        assert np.array_equal(a_q, want)
        for screen in ((w, h), (2 * w + 1, h + 3), (max(w // 2, 1), max(h // 2, 1))):
            assert np.array_equal(gpu.present(screen), orc.present(a_rgb, screen)), screen
        gpu.render(MODE_PRIMARY_SHADOW, stats=True)
        assert np.array_equal(gpu.read_steps(), r_steps)
        st = gpu.stats()
        assert st.primary_rays == cw * chh
    # tile shards and one context over two (here: the same) devices cover the same whole tiles
    acc_rgb, acc_ids = np.zeros_like(r_rgb), np.zeros_like(r_ids)
    for r in range(3):
        sh = gpu_for_scene(sc, shard_rank=r, shard_count=3)
        sh.render(MODE_PRIMARY_SHADOW)
        s_rgb, s_ids, _ = sh.read_output()
        acc_rgb += s_rgb
        acc_ids |= s_ids
        sh.close()
    assert_frame_parity(acc_rgb, acc_ids, r_rgb, r_ids, f"{w}x{h} in three shards")
    both = gpu_for_scene(sc, devices=[0, 0])
    both.render(MODE_PRIMARY_SHADOW)
    d_rgb, d_ids, _ = both.read_output()
    assert_frame_parity(d_rgb, d_ids, r_rgb, r_ids, f"{w}x{h} over two devices")
    both.close()
    # one frame at a time: the view rests, so from the third frame on the tiles are launched longest first (>= 128 tiles)
    made.set_frames_in_flight(1)
    for _ in range(4):
        made.render(MODE_PRIMARY_SHADOW)
    l_rgb, l_ids, _ = made.read_output()
    assert_frame_parity(l_rgb, l_ids, r_rgb, r_ids, f"{w}x{h} one frame at a time")
    made.set_frames_in_flight(2)
    settings = g.make_settings(sun_pos=scenes.SUN_POS, max_ray_bounces=2)
    made.write_settings(settings)
    o.set_settings(settings)
    for spp in (1, 3):
        p_rgb, p_ids, _, _ = o.render(orc.MODE_PATH, w, h, spp=spp, seed=5)
        for _ in range(2):
            made.render(MODE_PATH, spp=spp, seed=5)
        g_rgb, g_ids, _ = made.read_output()
        assert np.array_equal(g_ids, p_ids) and float(np.abs(g_rgb - p_rgb).max()) <= 1e-4, spp
        assert not g_rgb[:, cw:].any() and not g_rgb[chh:, :].any()
    made.close()
    resized.close()


def test_two_contexts_are_independent():
    a, b = scenes.c1_flat((64, 64)), scenes.c2((64, 64))
    ga, gb = gpu_for_scene(a), gpu_for_scene(b)
    ga.render(MODE_PRIMARY)
    gb.render(MODE_PRIMARY)
    ia, ib = ga.read_output()[1], gb.read_output()[1]
    ga.render(MODE_PRIMARY)
    assert np.array_equal(ga.read_output()[1], ia) and not np.array_equal(ia, ib)
    s = ga.stats()
    assert s.frames >= 1 and s.ms_total > 0 and s.primary_rays == 64 * 64


def test_present_quantise_and_crosshair_blit(orc):
    """vrt_present = textureStore to rgba8unorm + fs_main of screen_shader.wgsl, byte for byte what the oracle's
    restatement gives from the same f32 frame: default cross, a dot, no crosshair; windows at 1:1, magnified, minified to
    0.75x and 0.5x and squeezed on one axis only (the reference renders 1080 rows into whatever the window is)."""
    sc = scenes.c2((128, 72))
    gpu = gpu_for_scene(sc)
    gpu.render(MODE_PRIMARY_SHADOW)
    rgb, _, q = gpu.read_output(rgba8=True)
    for kw in (dict(), dict(style=1, size=9.5, color=(1.0, 0.2, 0.1, 0.75)), dict(style=0), dict(style=2, size=17.0, color=(0.0, 0.0, 0.0, 1.0))):
        for screen in ((128, 72), (256, 144), (200, 100), (96, 54), (64, 36), (150, 40), (33, 7)):
            got = gpu.present(screen, **kw)
            want = orc.present(rgb, screen, **kw)
            assert got.shape == (screen[1], screen[0], 4) and np.array_equal(got, want), (kw, screen)
    plain = gpu.present(style=0)
    assert np.array_equal(plain, q)                     # without a crosshair the blit is the rgba8 texture itself
    cross = gpu.present()                                 # Crosshair::default(): white, alpha 0.33, size 5
    changed = np.argwhere((cross != q).any(axis=2))
    assert 0 < len(changed) <= 2 * (10 * 3) and np.abs(changed - np.array([36, 64])).max() <= 5
    # the image can stay on the device (a host with GPU interop): same bytes, nothing copied by the library
    import ctypes
    from voxelraytracing_amd import _ffi
    ptr, nbytes = gpu.present_device((96, 54))
    assert ptr and nbytes == 96 * 54 * 4
    gpu.synchronize()
    host = np.empty(nbytes, dtype=np.uint8)
    hip_memcpy = _ffi.vrt().hipMemcpy                      # the HIP runtime libvrt.so itself is linked against
    hip_memcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    assert hip_memcpy(host.ctypes.data, ptr, nbytes, 2) == 0    # hipMemcpyDeviceToHost
    assert np.array_equal(host.reshape(54, 96, 4), gpu.present((96, 54)))
    with pytest.raises(VrtError):
        gpu.present((0, 36))


def test_issue_profile_counts_the_render_calls_and_their_parts(monkeypatch):
    """vrt_get_issue_profile: what issuing frames cost the calling thread since the previous call — for one context the whole
    vrt_render call, for a context over several devices also its parts, which add up to the call when one thread issues for
    every device in turn (the default when the ordinals repeat) and overlap when issuing threads do it."""
    sc = scenes.c2((256, 144))
    gpu = gpu_for_scene(sc)
    for _ in range(20):
        gpu.render(MODE_PRIMARY_SHADOW)
    p = gpu.issue_profile()
    assert p["frames"] == 20 and p["devices"] == 1 and p["issuing_threads"] == 0 and 0.5 < p["render_us"] < 5000.0
    assert p["root_issue_us"] == 0.0 and p["tail_us"] == 0.0
    assert gpu.issue_profile()["frames"] == 0          # read and reset
    gpu.close()
    for threads in ("0", "1"):
        monkeypatch.setenv("VRT_GROUP_THREADS", threads)
        grp = Gpu(sc.world.max_nodes(), sc.world.size_in_chunks(), sc.size, devices=[0, 0, 0])
        grp.upload_world(sc.world, sc.materials)
        grp.write_cam_data(sc.cam)
        grp.write_settings(sc.settings)
        for _ in range(30):
            grp.render(MODE_PRIMARY_SHADOW)
        grp.issue_profile()
        for _ in range(50):
            grp.render(MODE_PRIMARY_SHADOW)
        p = grp.issue_profile()
        assert p["frames"] == 50 and p["devices"] == 3 and p["issuing_threads"] == (2 if threads == "1" else 0)
        assert p["root_issue_us"] > 0 and p["shard_issue_us_mean"] > 0 and p["shard_issue_us_max"] >= p["shard_issue_us_mean"] and p["tail_us"] > 0
        if threads == "0":   # one thread, in turn: the call is the sum of its parts; the caller makes the message waits
            parts = p["root_issue_us"] + 2 * p["shard_issue_us_mean"] + p["tail_us"]
            assert p["join_wait_us"] == 0.0 and 0 < p["message_waits_us"] < p["tail_us"] and abs(parts - p["render_us"]) < 0.15 * p["render_us"] + 3.0
        else:                # every issuing thread makes its own message's wait
            assert p["message_waits_us"] == 0.0
        grp.close()


def _poison_texels(gpu):
    """Overwrite the texel buffer of the last frame with 0xFF bytes (behind everything enqueued): a blit launched from here on
    would show it."""
    from voxelraytracing_amd import _ffi
    gpu.synchronize()
    ptr, nbytes = gpu.device_output()
    hip_memset = _ffi.vrt().hipMemset
    hip_memset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    assert hip_memset(ptr, 0xFF, nbytes) == 0


@pytest.mark.parametrize("size", [(256, 144), (360, 120), (360, 200), (1920, 1080), (364, 100), (16380, 8)])
def test_frames_presented_texel_for_texel_store_their_own_window_pixels(orc, size):
    """Round 6 (vrt_set_presentation): the reference's compute pass stores rgba8unorm and its blit follows in the same submission
    (ray_tracer.wgsl:179, main.rs:452-454); here, declared, the march kernel stores the window's pixel itself — the texel
    quantised, fs_main's crosshair blended in from the lanes of the pixel's tile — and vrt_present* launches nothing.  The bytes are
    the oracle's general bilinear blit for every crosshair; the texel buffer is overwritten before the present, so a blit that
    did launch would show.  360 x 120 and 1920 x 1080 have samples 2e-6 beside their texel's centre INSIDE the crosshair's box
    (columns 182, 185, rows 61, 63; columns 962-964): their taps come from the neighbouring lanes.  At 360 x 200 one such row's
    tap lies in the tile above (row 104 takes 2e-6 of row 103), 364 columns are a ragged texture and 16 380 have a sample 5e-4
    off: those frames cannot store their window pixels — the declaration changes nothing and the blit's own launch gives the
    same bytes."""
    w, h = size
    sc = scenes.c2((w, h))
    ref = gpu_for_scene(sc)
    stores_its_pixels = size in ((256, 144), (360, 120), (1920, 1080))   # (for the crosshairs whose box is the small one around the centre)
    kinds = [dict(style=0), dict(), dict(style=1, size=9.5, color=(1.0, 0.2, 0.1, 0.75)), dict(style=2, size=17.0, color=(0.0, 0.0, 0.0, 1.0))]
    if w <= 400:
        kinds += [dict(style=1, size=0.4), dict(style=2, size=-3.0), dict(style=1, size=float("nan")),
                  dict(style=2, size=7.0, color=(float("nan"), 0.5, 0.5, 0.5)), dict(style=2, size=1.0e4, color=(0.3, 0.6, 0.9, 0.5))]
    for mode, variant in ((MODE_PRIMARY_SHADOW, 0), (MODE_PRIMARY, 0), (MODE_PRIMARY, 1)):
        if w > 400 and (mode, variant) != (MODE_PRIMARY_SHADOW, 0):
            continue
        ref.render(mode, variant=variant)
        rgb, ids, _ = ref.read_output()
        gpu = gpu_for_scene(sc)
        for kw in kinds:
            gpu.set_presentation((w, h), **kw)
            for _ in range(3):   # (frames in flight alternate between the frame sets: each has its own screen buffer)
                gpu.render(mode, variant=variant)
            a_rgb, a_ids, _ = gpu.read_output()              # the texels are stored as ever
            assert np.array_equal(a_ids, ids) and np.array_equal(a_rgb, rgb)
            want = orc.present(rgb, (w, h), **kw)
            small_box = kw.get("style", 2) == 0 or (0.0 < kw.get("size", 5.0) <= 17.0 and all(c == c for c in kw.get("color", (1.0,))))
            if stores_its_pixels and small_box:
                _poison_texels(gpu)
            assert np.array_equal(gpu.present((w, h), **kw), want), (mode, variant, kw)
        # another crosshair or size than the declared one: the blit's own launch, from the texels
        gpu.set_presentation((w, h))
        gpu.render(mode, variant=variant)
        other = dict(style=1, size=6.0, color=(0.2, 0.9, 0.1, 0.5))
        assert np.array_equal(gpu.present((w, h), **other), orc.present(rgb, (w, h), **other))
        if w <= 400:
            assert np.array_equal(gpu.present((w + 40, h + 8)), orc.present(rgb, (w + 40, h + 8)))
        assert np.array_equal(gpu.present((w, h)), orc.present(rgb, (w, h)))   # (and the declared one again, blitted this time)
        gpu.close()
    ref.close()


def test_presentation_only_frames_and_frames_that_cannot_store_their_pixels(orc):
    """VRT_PRESENT_SKIP_TEXELS: a frame that stores its window pixels stores nothing else — it can be presented with the declared
    crosshair, and everything that would need its texels says so.  Frames that cannot store window pixels (the path trace, stats
    frames of the two-launch variants, sharded contexts) are untouched by the declaration.  A draw + present loop with frames in
    flight gets every frame's own image."""
    from voxelraytracing_amd import _ffi
    sc = scenes.c2((320, 184))
    ref = gpu_for_scene(sc)
    ref.render(MODE_PRIMARY_SHADOW)
    rgb, ids, _ = ref.read_output()
    gpu = gpu_for_scene(sc)
    gpu.set_presentation(skip_texels=True)
    gpu.render(MODE_PRIMARY_SHADOW)
    assert np.array_equal(gpu.present(), orc.present(rgb, sc.size))
    with pytest.raises(VrtError):
        gpu.read_output()
    with pytest.raises(VrtError):
        gpu.present(style=0)
    with pytest.raises(VrtError):
        gpu.present((400, 200))
    gpu.render(MODE_PRIMARY_SHADOW, variant=3)               # two launches: texels, and the blit's own launch
    a_rgb, a_ids, _ = gpu.read_output()
    assert np.array_equal(a_ids, ids) and np.array_equal(a_rgb, rgb)
    assert np.array_equal(gpu.present(), orc.present(rgb, sc.size))
    gpu.render(MODE_PATH, spp=2, seed=5)                     # the path trace accumulates in its texels
    ref.render(MODE_PATH, spp=2, seed=5)
    p_rgb, p_ids, _ = ref.read_output()
    a_rgb, a_ids, _ = gpu.read_output()
    assert np.array_equal(a_ids, p_ids) and np.array_equal(a_rgb, p_rgb)
    assert np.array_equal(gpu.present(), orc.present(p_rgb, sc.size))
    gpu.set_presentation(off=True)
    gpu.render(MODE_PRIMARY_SHADOW)
    a_rgb, a_ids, _ = gpu.read_output()
    assert np.array_equal(a_ids, ids) and np.array_equal(a_rgb, rgb)
    gpu.close()
    sh = gpu_for_scene(sc, shard_rank=1, shard_count=2)      # a shard's tiles are a message, not a window
    sh.set_presentation()
    sh.render(MODE_PRIMARY_SHADOW)
    s_rgb, s_ids, _ = sh.read_output()
    mine = s_ids != 0
    assert mine.any() and np.array_equal(s_ids[mine], ids[mine]) and np.array_equal(s_rgb[mine], rgb[mine])
    with pytest.raises(VrtError):
        sh.present()
    sh.close()
    # the client's loop: draw, present, next frame — no wait in between; each of the frames in flight keeps its own image
    for in_flight in (1, 2, 3):
        gpu = gpu_for_scene(sc)
        gpu.set_frames_in_flight(in_flight)
        gpu.set_presentation(skip_texels=in_flight == 2)
        cams = [g.cam_data_create((20.0 + 9 * k, 35.0 + 50 * k, 0.0), (sc.eye[0] + 2 * k, sc.eye[1] + k, sc.eye[2] - k), 70.0, (320.0, 184.0)) for k in range(5)]
        want = []
        for cam in cams:
            ref.write_cam_data(cam)
            ref.render(MODE_PRIMARY_SHADOW)
            want.append(ref.present())
        ptrs = []
        for cam in cams:
            gpu.write_cam_data(cam)
            gpu.render(MODE_PRIMARY_SHADOW)
            ptrs.append(gpu.present_device())
        gpu.synchronize()
        assert len({p for p, _ in ptrs[-in_flight:]}) == in_flight
        hip_memcpy = _ffi.vrt().hipMemcpy
        hip_memcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        for k in range(len(cams) - in_flight, len(cams)):
            ptr, nbytes = ptrs[k]
            host = np.empty(nbytes, dtype=np.uint8)
            assert nbytes == 320 * 184 * 4 and hip_memcpy(host.ctypes.data, ptr, nbytes, 2) == 0
            assert np.array_equal(host.reshape(184, 320, 4), want[k]), f"frame {k}, {in_flight} in flight"
        gpu.close()
    ref.close()


@pytest.mark.parametrize("in_flight", [1, 2, 3])
def test_present_device_follows_its_frame_without_waiting(orc, in_flight):
    """vrt_present_device enqueues the blit behind the frame it presents, on that frame's stream, into the screen buffer of
    the frame's set: a draw + present loop (main.rs:452-454) keeps its frames in flight, and each of the last `in_flight`
    images is still there after a synchronise."""
    from voxelraytracing_amd import _ffi
    sc = scenes.c2((160, 96))
    gpu = gpu_for_scene(sc)
    gpu.set_frames_in_flight(in_flight)
    cams = [g.cam_data_create((20.0 + 9 * k, 35.0 + 50 * k, 0.0), (sc.eye[0] + 2 * k, sc.eye[1] + k, sc.eye[2] - k), 70.0, (160.0, 96.0)) for k in range(5)]
    want = []
    for cam in cams:                          # the synchronous present of every frame
        gpu.write_cam_data(cam)
        gpu.render(MODE_PRIMARY_SHADOW)
        want.append(gpu.present((200, 120)))
    assert any(not np.array_equal(want[0], w) for w in want[1:])
    ptrs = []
    for cam in cams:                          # the loop of the client: draw, present, next frame — no wait in between
        gpu.write_cam_data(cam)
        gpu.render(MODE_PRIMARY_SHADOW)
        ptrs.append(gpu.present_device((200, 120)))
    gpu.synchronize()
    assert len({p for p, _ in ptrs[-in_flight:]}) == in_flight, "the frames in flight share a screen buffer"
    hip_memcpy = _ffi.vrt().hipMemcpy
    hip_memcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    for k in range(len(cams) - in_flight, len(cams)):
        ptr, nbytes = ptrs[k]
        host = np.empty(nbytes, dtype=np.uint8)
        assert nbytes == 200 * 120 * 4 and hip_memcpy(host.ctypes.data, ptr, nbytes, 2) == 0
        assert np.array_equal(host.reshape(120, 200, 4), want[k]), f"frame {k} of {len(cams)}, {in_flight} in flight"


@pytest.mark.parametrize("in_flight", [1, 2, 3, 4])
def test_frames_in_flight_keep_frames_apart(orc, in_flight):
    """Back-to-back vrt_render calls with different cameras and no synchronisation in between: every read-back is the
    frame of the most recent call, whatever number of frames the context keeps in flight; uploads wait for them."""
    sc = scenes.c2((160, 96))
    gpu = gpu_for_scene(sc)
    gpu.set_frames_in_flight(in_flight)
    o = orc.from_package_scene(sc)
    cams = [g.cam_data_create((20.0 + 7 * k, 35.0 + 50 * k, 0.0), (sc.eye[0] + 3 * k, sc.eye[1] + k, sc.eye[2] - 2 * k), 70.0, (160.0, 96.0))
            for k in range(5)]
    refs = []
    for cam in cams:
        o.set_cam(cam)
        refs.append(o.render(orc.MODE_PRIMARY_SHADOW, 160, 96)[:2])
    for last in range(len(cams)):
        for k in range(last + 1):              # k + 1 frames enqueued with nothing waiting in between
            gpu.write_cam_data(cams[k])
            gpu.render(MODE_PRIMARY_SHADOW)
        rgb, ids, _ = gpu.read_output()
        assert_frame_parity(rgb, ids, refs[last][0], refs[last][1], f"{in_flight} in flight, frame {last}")
        assert gpu.stats().secondary_rays == int(((refs[last][1] >> 21) & 1).sum())
    # an edit between two frames: the upload (and the table rebuild) must not overtake the frame before it
    gpu.write_cam_data(cams[0])
    gpu.render(MODE_PRIMARY_SHADOW)
    start, n = sc.world.set_voxel((int(sc.eye[0]) + 4, int(sc.eye[1]) - 6, int(sc.eye[2]) + 9), 4)
    gpu.write_nodes(sc.world.nodes_ptr(), start, start + n)
    gpu.render(MODE_PRIMARY_SHADOW)
    gpu.render(MODE_PRIMARY)                    # and a primary-only frame right behind it
    rgb, ids, _ = gpu.read_output()
    o2 = orc.from_package_scene(sc)
    o2.set_cam(cams[0])
    r_rgb, r_ids, _, _ = o2.render(orc.MODE_PRIMARY, 160, 96)
    assert_frame_parity(rgb, ids, r_rgb, r_ids, "after an edit")
    with pytest.raises(VrtError):
        gpu.set_frames_in_flight(5)


@pytest.mark.parametrize("in_flight", [2, 3])
def test_a_frame_stays_announced_when_the_event_pool_is_folded(orc, in_flight, monkeypatch):
    """A context folds its pool of timing events every 512 timed frames and waits for the frames in flight to do so — between
    picking the next frame's stream and enqueueing the frame.  The frame must still count as in flight afterwards: a
    read-back right behind it once copied the slot's previous frame (tools/soak_edits.py, seed 7: once in 180 000 frames).
    Every frame timed (VRT_TIMING_EVERY=1), the camera alternating, every frame read back."""
    monkeypatch.setenv("VRT_TIMING_EVERY", "1")
    sc = scenes.c2((160, 96))
    gpu = gpu_for_scene(sc)
    gpu.set_frames_in_flight(in_flight)
    o = orc.from_package_scene(sc)
    cams = [g.cam_data_create((20.0 + 30 * k, 35.0 + 120 * k, 0.0), (sc.eye[0] + 3 * k, sc.eye[1] + k, sc.eye[2] - 2 * k), 70.0, (160.0, 96.0))
            for k in range(2)]
    refs = []
    for cam in cams:
        o.set_cam(cam)
        refs.append(o.render(orc.MODE_PRIMARY, 160, 96)[1])
    assert int((refs[0] != refs[1]).sum()) > 5000
    for i in range(1100):        # two folds
        k = (i // 2) & 1 if in_flight == 2 else i & 1   # the slot's previous frame (in_flight frames ago) had the other camera
        gpu.write_cam_data(cams[k])
        gpu.render(MODE_PRIMARY)
        if i >= 500:
            _, ids, _ = gpu.read_output()
            assert np.array_equal(ids, refs[k]), f"frame {i}: {int((ids != refs[k]).sum())} id words of another frame"


def test_an_edit_behind_the_fold_frame_waits_for_the_frame_that_reads_the_shared_tables(orc, monkeypatch):
    """With two frames in flight and no edit for a while both frame sets read table set 0.  A frame of the second set that
    falls on the fold of the timing-event pool (every 512 timed frames: a drain between picking the frame's stream and
    enqueueing it) must still count as a reader of set 0 afterwards: an edit right behind it makes the next frame rebuild
    the edited chunk IN set 0, and without the reader's flag that rebuild ran beside the frame still marching over the
    tables (round 3's advisor finding).  The fold frame here is a long one — a 16-sample path trace — and is read out of its
    own buffer after the edit's frame: it must be the oracle's frame of the world BEFORE the edit."""
    from voxelraytracing_amd import _ffi
    monkeypatch.setenv("VRT_TIMING_EVERY", "1")
    W, H = 640, 360
    sc = scenes.c4((W, H))
    gpu = gpu_for_scene(sc)
    gpu.set_frames_in_flight(2)
    o = orc.from_package_scene(sc)
    r_rgb, r_ids, _, _ = o.render(orc.MODE_PATH, W, H, spp=16, seed=5)     # the world before the edit
    gpu.render(MODE_PRIMARY)
    gpu.stats()                                  # folds the pool here, so that the next fold falls on a frame of the second set
    for i in range(512):
        gpu.render(MODE_PRIMARY)                 # (no edits: the table sets stay merged)
    gpu.render(MODE_PATH, spp=16, seed=5)        # timed frame 513 since the fold: the pool is folded inside this call
    ptr, nbytes = gpu.device_output()
    # a plate of limestone a few voxels in front of the camera (a chunk's range uploaded once), then the next frame
    ex, ey, ez = (int(v) for v in sc.eye)
    ranges = {}
    for dx in range(-5, 2):
        for dy in range(-4, 3):
            start, n = sc.world.set_voxel((ex + dx, ey + dy, ez - 3), 4)
            ranges[start] = n
    for start, n in ranges.items():
        gpu.write_nodes(sc.world.nodes_ptr(), start, start + n)
    gpu.render(MODE_PRIMARY)
    p2, _ = gpu.device_output()
    assert p2 != ptr, "the edit's frame went to the fold frame's buffer: the frames did not alternate as this test assumes"
    gpu.synchronize()
    host = np.zeros(nbytes // 4, dtype=np.uint32)
    hip_memcpy = _ffi.vrt().hipMemcpy
    hip_memcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    assert hip_memcpy(host.ctypes.data, ptr, nbytes, 2) == 0    # hipMemcpyDeviceToHost
    t = host.reshape(H, W, 4)
    assert_frame_parity(t[..., :3].copy().view(np.float32), t[..., 3], r_rgb, r_ids, "the fold frame, read after the edit's frame")
    # and the edit itself arrived
    _, ids, _ = gpu.read_output()
    o2 = orc.from_package_scene(sc)
    _, e_ids, _, _ = o2.render(orc.MODE_PRIMARY, W, H)
    assert np.array_equal(ids, e_ids) and int((e_ids != r_ids).sum()) > 2000


def test_a_pool_uploaded_in_ragged_pieces_is_the_pool(orc):
    """NodeBuffer::write for ranges of every size and alignment (shader.rs:24-33: widened to even nodes, nothing else): the
    batched upload kernel moves 16 bytes per load out of the pinned ring and stores words, pieces of up to 16 KiB — a pool
    written in a few thousand random ranges, in random order, traces like the pool written once."""
    sc = scenes.c2((160, 96))
    used = int(np.nonzero(np.frombuffer((C.c_uint16 * sc.world.max_nodes()).from_address(sc.world.nodes_ptr()), dtype=np.uint16))[0].max()) + 2
    gpu = Gpu(sc.world.max_nodes(), sc.world.size_in_chunks(), sc.size)
    gpu.write_chunk_roots(sc.world.chunk_roots())
    gpu.write_world_data(sc.world.world_data())
    gpu.write_materials(sc.materials)
    gpu.write_cam_data(sc.cam)
    gpu.write_settings(sc.settings)
    rng = np.random.default_rng(4)
    cuts = np.unique(np.concatenate([[0, used], rng.integers(1, used, 3000)]))
    # (a few long ranges among them: several pieces, and one beyond a ring segment — the synchronous route)
    pieces = list(zip(cuts[:-1].tolist(), cuts[1:].tolist()))
    rng.shuffle(pieces)
    for k, (a, b) in enumerate(pieces):
        gpu.write_nodes(sc.world.nodes_ptr(), a, b)
        if k % 700 == 350:
            gpu.render(MODE_PRIMARY)          # frames in between: the staged ranges are flushed in batches of every size
    gpu.write_nodes(sc.world.nodes_ptr(), 0, min(used, 700001))   # one write of > 1 MiB over what is there already
    gpu.render(MODE_PRIMARY_SHADOW)
    rgb, ids, _ = gpu.read_output()
    r_rgb, r_ids, _, _ = orc.from_package_scene(sc).render(orc.MODE_PRIMARY_SHADOW, 160, 96)
    assert_frame_parity(rgb, ids, r_rgb, r_ids, "a pool uploaded in ragged pieces")
    # the derived tables agree too: every cell of the grid is what a walk of the host's pool finds
    a = gpu.accel_info()
    assert a.available and a.builds >= 1


def test_staged_node_writes_survive_a_lap_of_the_upload_ring(orc):
    """vrt_write_nodes stages its bytes in the pinned ring until the next frame; uploads that are not staged (the material
    table: 8 KB a call) share the ring.  More than a lap of them (8 MiB) between the edit and its frame must not write over
    the staged range: the ring flushes what a segment still holds before it reuses the segment."""
    sc = scenes.c2((160, 96))
    gpu = gpu_for_scene(sc)
    gpu.render(MODE_PRIMARY)
    ex, ey, ez = (int(v) for v in sc.eye)
    start, n = sc.world.set_voxel((ex - 2, ey - 1, ez - 3), 4)
    gpu.write_nodes(sc.world.nodes_ptr(), start, start + n)
    for _ in range(1100):                      # 1100 x 8 KB: the ring's eight segments and a bit
        gpu.write_materials(sc.materials)
    gpu.render(MODE_PRIMARY)
    rgb, ids, _ = gpu.read_output()
    r_rgb, r_ids, _, _ = orc.from_package_scene(sc).render(orc.MODE_PRIMARY, 160, 96)
    assert_frame_parity(rgb, ids, r_rgb, r_ids, "the edit's frame after a lap of the ring")


def test_render_on_own_streams_with_a_caller_stream_and_bound_outputs(orc):
    """VRT_RENDER_OWN_STREAMS — what the in-place gather root does with its own tiles: the caller has set its stream and
    binds a different buffer per frame, the frames still overlap on the context's streams; after a device-wide
    synchronise every buffer holds its own frame."""
    import torch
    from voxelraytracing_amd.shard import texels_to_frame
    sc = scenes.c2((160, 96))
    gpu = gpu_for_scene(sc)
    side = torch.cuda.Stream()
    gpu.set_stream(side.cuda_stream)
    o = orc.from_package_scene(sc)
    cams = [g.cam_data_create((20.0 + 5 * k, 35.0 + 60 * k, 0.0), (sc.eye[0] + k, sc.eye[1] + 2 * k, sc.eye[2]), 70.0, (160.0, 96.0)) for k in range(6)]
    bufs = [torch.zeros((96, 160, 4), dtype=torch.int32, device="cuda") for _ in cams]
    torch.cuda.synchronize()
    for cam, buf in zip(cams, bufs):
        gpu.write_cam_data(cam)
        gpu.bind_output(buf.data_ptr())
        gpu.render(MODE_PRIMARY_SHADOW, own_streams=True)
    torch.cuda.synchronize()
    for k, (cam, buf) in enumerate(zip(cams, bufs)):
        o.set_cam(cam)
        r_rgb, r_ids, _, _ = o.render(orc.MODE_PRIMARY_SHADOW, 160, 96)
        rgb, ids = texels_to_frame(buf.cpu().numpy().view(np.uint32))
        assert_frame_parity(rgb, ids, r_rgb, r_ids, f"frame {k}")
    assert 1 <= gpu.stats().frames <= 6           # frames *timed*: the first, then every 8th (include/vrt.h)
    gpu.bind_output(0)
    gpu.render(MODE_PRIMARY_SHADOW)          # back to the context's own buffer and the caller's stream
    rgb, ids, _ = gpu.read_output()
    assert_frame_parity(rgb, ids, r_rgb, r_ids, "own buffer again")
