"""GPU parity proper: the HIP path (through the C ABI) against the CPU oracle on the same inputs.

Bar (BASELINE.json north_star): id words bit-exact, f32 radiance within 1e-4.
"""
import math
import numpy as np
import pytest

from voxelraytracing_amd import MODE_PATH, MODE_PRIMARY, MODE_PRIMARY_SHADOW, scenes
from voxelraytracing_amd import graphics as g

from util import assert_frame_parity, gpu_for_scene

pytestmark = pytest.mark.gpu


def needs_experiments():
    """The structures that were measured and not chosen compile into tools/ab/libvrt_exp.so only (make -C
    voxelraytracing_amd/csrc experiments; run the tests with VRT_LIB=tools/ab/libvrt_exp.so): their tests skip on the default build."""
    from voxelraytracing_amd import _ffi
    if not hasattr(_ffi.vrt(), "vrt_experiments_build"):
        pytest.skip("the experiments build only (VRT_LIB=tools/ab/libvrt_exp.so)")


@pytest.fixture(scope="module")
def c1():
    return scenes.c1_flat()


@pytest.fixture(scope="module")
def c2_small():
    return scenes.c2((640, 360))


VARIANTS = [0, 1, 2, 3]  # 0 = grid march, primary+shadow fused (default); 1 = literal octree walk; 2 = ancestor-cache octree walk; 3 = grid march, two launches


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("mode", [MODE_PRIMARY, MODE_PRIMARY_SHADOW])
def test_c1_flat_matches_oracle(c1, orc, mode, variant):
    gpu = gpu_for_scene(c1)
    gpu.render(mode, variant=variant, stats=True)
    rgb, ids, _ = gpu.read_output()
    o = orc.from_package_scene(c1)
    r_rgb, r_ids, r_steps, st = o.render(mode, *c1.size, want_steps=True)
    assert_frame_parity(rgb, ids, r_rgb, r_ids, f"C1 mode {mode}")
    assert np.array_equal(gpu.read_steps(), r_steps)
    s = gpu.stats()
    assert (s.primary_rays, s.secondary_rays, s.hits, s.steps, s.node_visits) == \
           (st.primary_rays, st.secondary_rays, st.hits, st.steps, st.node_visits)
    assert (s.primary_steps, s.primary_node_visits) == (st.primary_steps, st.primary_node_visits)


@pytest.mark.parametrize("variant", VARIANTS)
def test_a_world_no_chunk_of_which_has_arrived(orc, variant):
    """The client's first frames (client/src/world.rs:259-295: the grid exists, `chunk_roots()` is all zeros until GiveChunkData
    arrives — every chunk resolves to pool[0], the permanent air leaf, a free 32^3 cell): primary, primary + shadow and a
    path-traced frame of a 4^3-chunk world without a single chunk, the camera inside and looking down across it — ids, per-pixel
    step counts and the ray counts against the oracle (every ray crosses the grid in 32-voxel steps and leaves: sky everywhere)."""
    from voxelraytracing_amd.world import ClientWorld
    world = ClientWorld((2, 2, 2), 1 << 14, 4)
    assert not np.any(world.chunk_roots())
    sc = scenes._scene("empty 4^3", world, (160, 96), (64.5, 100.5, 64.5), (35.0, 20.0, 0.0), MODE_PRIMARY_SHADOW)
    gpu = gpu_for_scene(sc)
    o = orc.from_package_scene(sc)
    for mode in (MODE_PRIMARY, MODE_PRIMARY_SHADOW):
        gpu.render(mode, variant=variant, stats=True)
        rgb, ids, _ = gpu.read_output()
        r_rgb, r_ids, r_steps, st = o.render(mode, *sc.size, want_steps=True)
        assert_frame_parity(rgb, ids, r_rgb, r_ids, f"empty world, mode {mode}")
        assert np.array_equal(gpu.read_steps(), r_steps)
        s = gpu.stats()
        assert (s.primary_rays, s.secondary_rays, s.hits) == (st.primary_rays, st.secondary_rays, st.hits) == (160 * 96, 0, 0)
        assert r_steps.max() <= 8 and r_steps.min() >= 1
    if variant == 0:
        gpu.render(MODE_PATH, spp=2, seed=5)
        rgb, ids, _ = gpu.read_output()
        r_rgb, r_ids, _, _ = o.render(MODE_PATH, *sc.size, spp=2, seed=5)
        assert_frame_parity(rgb, ids, r_rgb, r_ids, "empty world, path trace")
    gpu.close()


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("mode", [MODE_PRIMARY, MODE_PRIMARY_SHADOW])
def test_c2_procedural_matches_oracle(c2_small, orc, mode, variant):
    gpu = gpu_for_scene(c2_small)
    gpu.render(mode, variant=variant, stats=True)
    rgb, ids, _ = gpu.read_output()
    o = orc.from_package_scene(c2_small)
    r_rgb, r_ids, r_steps, st = o.render(mode, *c2_small.size, want_steps=True)
    assert_frame_parity(rgb, ids, r_rgb, r_ids, f"C2 mode {mode}")
    assert np.array_equal(gpu.read_steps(), r_steps)
    s = gpu.stats()
    assert (s.steps, s.node_visits, s.secondary_rays) == (st.steps, st.node_visits, st.secondary_rays)
    # the timed (stats-free) kernels must give the same frame
    gpu.render(mode, variant=variant, stats=False)
    rgb2, ids2, _ = gpu.read_output()
    assert np.array_equal(ids2, ids) and np.array_equal(rgb2, rgb)


@pytest.mark.parametrize("rot,eye_dy", [((0.0, 0.0, 0.0), 30.0), ((89.0, 10.0, 0.0), 60.0), ((-60.0, 200.0, 0.0), 10.0),
                                        ((35.0, 135.0, 20.0), 3.0)])
@pytest.mark.parametrize("variant", VARIANTS)
def test_c2_other_cameras(c2_small, orc, rot, eye_dy, variant):
    """Axis-parallel centre ray (NaN unit steps at rot 0), straight-down, sky-only and rolled views."""
    w, h = 320, 200
    eye = (c2_small.eye[0], c2_small.eye[1] - 24.0 + eye_dy, c2_small.eye[2])
    cam = g.cam_data_create(rot, eye, 70.0, (float(w), float(h)))
    gpu = gpu_for_scene(c2_small, (w, h))
    gpu.write_cam_data(cam)
    gpu.render(MODE_PRIMARY_SHADOW, variant=variant, stats=True)
    rgb, ids, _ = gpu.read_output()
    o = orc.from_package_scene(c2_small)
    o.set_cam(cam)
    r_rgb, r_ids, r_steps, _ = o.render(MODE_PRIMARY_SHADOW, w, h, want_steps=True)
    assert_frame_parity(rgb, ids, r_rgb, r_ids, f"rot {rot}")
    assert np.array_equal(gpu.read_steps(), r_steps)


def test_camera_outside_world_sees_only_sky(c1, orc):
    cam = g.cam_data_create((10.0, 30.0, 0.0), (-5.0, 20.0, 10.0), 70.0, (64.0, 64.0))
    gpu = gpu_for_scene(c1, (64, 64))
    gpu.write_cam_data(cam)
    gpu.render(MODE_PRIMARY_SHADOW)
    rgb, ids, _ = gpu.read_output()
    assert not ids.any()
    o = orc.from_package_scene(c1)
    o.set_cam(cam)
    r_rgb, r_ids, _, _ = o.render(MODE_PRIMARY_SHADOW, 64, 64)
    assert_frame_parity(rgb, ids, r_rgb, r_ids, "outside")


@pytest.mark.parametrize("variant", VARIANTS)
def test_nan_and_degenerate_cameras(c1, orc, variant):
    """NaN eye components march 500 steps through chunk 0 (i32(NaN) = 0); exactly axis-parallel rays have
    NaN / inf unit steps; an eye on an integer lattice point triggers the start nudge and zero axis distances."""
    w, h = 64, 64
    gpu = gpu_for_scene(c1, (w, h))
    o = orc.from_package_scene(c1)
    cases = [((0.0, 0.0, 0.0), (32.0, 20.0, 60.0)), ((90.0, 0.0, 0.0), (32.5, 20.0, 32.5)),
             ((0.0, 90.0, 0.0), (40.0, 13.0, 40.0)), ((15.0, 0.0, 0.0), (float("nan"), 20.5, 60.5)),
             ((15.0, 30.0, 0.0), (32.5, float("nan"), float("nan")))]
    for rot, eye in cases:
        cam = g.cam_data_create(rot, eye, 70.0, (float(w), float(h)))
        gpu.write_cam_data(cam)
        gpu.render(MODE_PRIMARY_SHADOW, variant=variant, stats=True)
        rgb, ids, _ = gpu.read_output()
        o.set_cam(cam)
        r_rgb, r_ids, r_steps, _ = o.render(MODE_PRIMARY_SHADOW, w, h, want_steps=True)
        bad = ids != r_ids
        assert not bad.any(), f"rot {rot} eye {eye}: {int(bad.sum())} id words differ"
        assert np.array_equal(gpu.read_steps(), r_steps)
        both = np.isfinite(rgb) & np.isfinite(r_rgb)
        assert np.array_equal(np.isnan(rgb), np.isnan(r_rgb))
        assert float(np.abs(rgb[both] - r_rgb[both]).max(initial=0.0)) <= 1e-4


def test_show_step_count_debug_view(c2_small, orc):
    st = g.make_settings(sun_pos=scenes.SUN_POS, show_step_count=1)
    gpu = gpu_for_scene(c2_small, (320, 200))
    cam = g.cam_data_create(c2_small.rot, c2_small.eye, 70.0, (320.0, 200.0))
    gpu.write_cam_data(cam)
    gpu.write_settings(st)
    gpu.render(MODE_PRIMARY)
    rgb, ids, _ = gpu.read_output()
    o = orc.from_package_scene(c2_small)
    o.set_cam(cam)
    o.set_settings(st)
    r_rgb, r_ids, _, _ = o.render(MODE_PRIMARY, 320, 200)
    assert_frame_parity(rgb, ids, r_rgb, r_ids, "step-count view")


def test_edit_then_reupload_range(orc):
    """Voxel edit -> re-upload of the chunk's range (main.rs:352-362) changes exactly what the oracle says."""
    sc = scenes.c1_flat((128, 128))
    gpu = gpu_for_scene(sc)
    # dig a hole and build a pillar with water on top, in front of the camera
    edits = [((32, 12, 40), 0), ((32, 11, 40), 0), ((30, 13, 44), 4), ((30, 14, 44), 4), ((34, 13, 44), 3), ((34, 14, 44), 3)]
    for pos, v in edits:
        start, n = sc.world.set_voxel(pos, v)
        gpu.write_nodes(sc.world.nodes_ptr(), start, start + n)
    gpu.write_chunk_roots(sc.world.chunk_roots())
    gpu.render(MODE_PRIMARY_SHADOW)
    rgb, ids, _ = gpu.read_output()
    o = orc.from_package_scene(sc)
    r_rgb, r_ids, _, _ = o.render(MODE_PRIMARY_SHADOW, 128, 128)
    assert_frame_parity(rgb, ids, r_rgb, r_ids, "after edits")
    assert (ids & 0x7FFF == 4).any() and (ids & g._ffi.ID_WATER).any()


def test_sharded_contexts_assemble_to_the_unsharded_frame(c2_small):
    """N tile-interleaved shard contexts on one GPU: their tiles together are the whole frame."""
    full = gpu_for_scene(c2_small)
    full.render(MODE_PRIMARY_SHADOW)
    rgb, ids, _ = full.read_output()
    for n in (2, 3, 8):
        acc_rgb = np.zeros_like(rgb)
        acc_ids = np.zeros_like(ids)
        owned = np.zeros(ids.shape, dtype=np.int32)
        for r in range(n):
            sh = gpu_for_scene(c2_small, shard_rank=r, shard_count=n)
            sh.render(MODE_PRIMARY_SHADOW)
            s_rgb, s_ids, _ = sh.read_output()
            tl, tp, tt = sh.shard_info()
            assert tp == -(-tt // n) and tl == len(range(r, tt, n))
            acc_rgb += s_rgb
            acc_ids |= s_ids
            ty, tx = np.divmod(np.arange(tt), c2_small.size[0] // 8)
            mine = np.zeros((c2_small.size[1] // 8, c2_small.size[0] // 8), dtype=np.int32)
            mine[ty[r::n], tx[r::n]] = 1
            owned += np.kron(mine, np.ones((8, 8), dtype=np.int32))
            sh.close()
        assert (owned == 1).all()
        assert np.array_equal(acc_ids, ids) and np.array_equal(acc_rgb, rgb)


def test_device_side_gather_and_assemble(c2_small):
    """The multi-GPU data path on one GPU: N shard contexts render straight into slices of one torch tensor
    (what RCCL's gather fills on rank 0), vrt_assemble de-interleaves it into the row-major texel frame."""
    import torch
    from voxelraytracing_amd.shard import FrameGather, texels_to_frame
    w, h = c2_small.size
    full = gpu_for_scene(c2_small)
    full.render(MODE_PRIMARY_SHADOW)
    rgb, ids, _ = full.read_output()
    for n in (2, 5):
        fg0 = FrameGather(torch, None, 0, n, w, h, torch.device("cuda", 0))
        ctxs = []
        for r in range(n):
            sh = gpu_for_scene(c2_small, shard_rank=r, shard_count=n)
            sh.bind_output(fg0.gathered[r].data_ptr())
            sh.render(MODE_PRIMARY_SHADOW)
            sh.synchronize()
            ctxs.append(sh)
        fg0.assemble(ctxs[0])
        ctxs[0].synchronize()
        a_rgb, a_ids = texels_to_frame(fg0.frame.cpu().numpy().view(np.uint32))
        assert np.array_equal(a_ids, ids) and np.array_equal(a_rgb, rgb)
        for c in ctxs:
            c.close()


@pytest.mark.parametrize("compact", [False, True])
@pytest.mark.parametrize("n,w0", [(2, 4), (4, 3), (8, 2), (3, 7)])
def test_weighted_in_place_root_gather_and_assemble(c2_small, n, w0, compact):
    """The bench's N > 1 data path on one GPU with a weighted root: rank 0 takes w0 tiles of every w0 + n - 1 and
    renders them straight into the row-major frame (VRT_FLAG_ROW_MAJOR); the other ranks render tile-major messages
    into slices of the receive tensor; vrt_assemble scatters those and leaves the root's tiles alone."""
    import torch
    from voxelraytracing_amd import shard
    from voxelraytracing_amd.shard import FrameGather, texels_to_frame
    w, h = c2_small.size
    full = gpu_for_scene(c2_small)
    full.render(MODE_PRIMARY_SHADOW)
    rgb, ids, _ = full.read_output()
    fg0 = FrameGather(torch, None, 0, n, w, h, torch.device("cuda", 0), root_weight=w0, in_place=True, compact=compact)
    ctxs = []
    total = 0
    for r in range(n):
        sh = gpu_for_scene(c2_small, shard_rank=r, shard_count=n, root_weight=w0, row_major=(r == 0), compact=compact and r != 0)
        tl, tp, tt = sh.shard_info()
        mine, padded, tot = shard.tiles_of_rank(w, h, r, n, w0)
        assert (tl, tp, tt) == (len(mine), padded, tot)
        total += tl
        if r == 0:
            fg0.bind(sh, 1)                      # frame buffer 1, as frame k = 1 of the pipeline would
            assert sh.device_output()[1] == w * h * 16
        else:
            assert sh.device_output()[1] == padded * 64 * (8 if compact else 16)
            sh.bind_output(fg0.recv[1][r].data_ptr())
        sh.render(MODE_PRIMARY_SHADOW)
        sh.synchronize()
        ctxs.append(sh)
    assert total == (w // 8) * (h // 8)
    fg0.assemble(ctxs[0], 1)
    ctxs[0].synchronize()
    a_rgb, a_ids = texels_to_frame(fg0.frame.cpu().numpy().view(np.uint32))
    assert np.array_equal(a_ids, ids) and np.array_equal(a_rgb, rgb)
    if compact:   # the records are shaded on the device only; what was checked above is the whole point
        with pytest.raises(g.VrtError):
            ctxs[1].read_output()
        for c in ctxs:
            c.close()
        return
    # the host twin used by the gloo tests agrees with the device scatter
    host = shard.assemble_numpy(fg0.recv[1].cpu().numpy().view(np.uint32), w, h, n, w0,
                                frame=np.zeros((h, w, 4), dtype=np.uint32))
    r_rgb, r_ids = texels_to_frame(host)
    others = np.ones((h // 8, w // 8), dtype=bool)
    t0 = shard.tiles_of_rank(w, h, 0, n, w0)[0]
    others[np.divmod(t0, w // 8)] = False
    px = np.kron(others, np.ones((8, 8), dtype=bool))
    assert np.array_equal(r_ids[px], ids[px]) and np.array_equal(r_rgb[px], rgb[px])
    for c in ctxs:
        c.close()


def test_batched_gather_frames_stay_apart(c2_small, orc):
    """FrameGather with 3 frames per gather: three different cameras rendered by 4 shard contexts (weighted, in-place
    root, compact records) into the three slices of one message set; every assembled frame equals its unsharded render."""
    import torch
    from voxelraytracing_amd.shard import FrameGather, texels_to_frame
    w, h = c2_small.size
    n, w0, batch = 4, 3, 3
    cams = [g.cam_data_create((18.0 + 9 * k, 30.0 + 40 * k, 0.0), (c2_small.eye[0] + 2 * k, c2_small.eye[1] + k, c2_small.eye[2]), 70.0,
                              (float(w), float(h))) for k in range(batch)]
    full = gpu_for_scene(c2_small)
    want = []
    for cam in cams:
        full.write_cam_data(cam)
        full.render(MODE_PRIMARY_SHADOW)
        want.append(full.read_output()[:2])
    fg0 = FrameGather(torch, None, 0, n, w, h, torch.device("cuda", 0), root_weight=w0, in_place=True, compact=True, batch=batch)
    ctxs = [gpu_for_scene(c2_small, shard_rank=r, shard_count=n, root_weight=w0, row_major=(r == 0), compact=(r != 0)) for r in range(n)]
    which = 1
    for j, cam in enumerate(cams):
        for r, sh in enumerate(ctxs):
            sh.write_cam_data(cam)
            if r == 0:
                fg0.bind(sh, which, j)
            else:   # what RCCL's gather would have put into rank r's row of the receive buffer
                sh.bind_output(fg0.recv[which][r].data_ptr() + j * fg0.frame_words * 4)
            sh.render(MODE_PRIMARY_SHADOW)
    for sh in ctxs:
        sh.synchronize()
    for j, cam in enumerate(cams):
        ctxs[0].write_cam_data(cam)          # the root shades with the uniforms of the frame it assembles
        fg0.assemble(ctxs[0], which, j)
        ctxs[0].synchronize()
        a_rgb, a_ids = texels_to_frame(fg0.frame.cpu().numpy().view(np.uint32))
        assert np.array_equal(a_ids, want[j][1]) and np.array_equal(a_rgb, want[j][0]), f"frame {j}"
    for c in ctxs + [full]:
        c.close()


def test_full_size_properties():
    """At BASELINE's full size (1920x1080, 8^3 world) check size-independent properties instead of the oracle:
    determinism, shadow pass only darkens launched pixels by exactly the factor, stats add up."""
    sc = scenes.c2()
    gpu = gpu_for_scene(sc)
    gpu.render(MODE_PRIMARY)
    p_rgb, p_ids, _ = gpu.read_output()
    gpu.render(MODE_PRIMARY_SHADOW, stats=True)
    s_rgb, s_ids, _ = gpu.read_output()
    st = gpu.stats()
    launched = (s_ids & g._ffi.ID_SHADOW_RAY) != 0
    shadowed = (s_ids & g._ffi.ID_SHADOWED) != 0
    assert st.primary_rays == 1920 * 1080 and st.secondary_rays == int(launched.sum())
    assert not (shadowed & ~launched).any()
    assert np.array_equal(s_ids & ~np.uint32(g._ffi.ID_SHADOW_RAY | g._ffi.ID_SHADOWED), p_ids)
    assert np.array_equal(s_rgb[~shadowed], p_rgb[~shadowed])
    assert np.array_equal(s_rgb[shadowed], p_rgb[shadowed] * np.float32(0.35))
    solid = ((p_ids & g._ffi.ID_HIT) != 0) & ((p_ids & 0x7FFF) != 0) & ((p_ids & 0x7FFF) != 3)
    assert np.array_equal(launched, solid)
    assert st.hits == int(((p_ids & g._ffi.ID_HIT) != 0).sum())
    gpu.render(MODE_PRIMARY_SHADOW)
    r2, i2, _ = gpu.read_output()
    assert np.array_equal(i2, s_ids) and np.array_equal(r2, s_rgb)


def test_full_size_c2_matches_oracle(orc):
    """The bench's own frame — 1920x1080, 8^3 world, primary + shadow — against the oracle, every pixel: the oracle's
    OpenMP build finishes it in a fraction of a second on the GPU box's cores."""
    sc = scenes.c2()
    gpu = gpu_for_scene(sc)
    gpu.render(MODE_PRIMARY_SHADOW, stats=True)
    rgb, ids, _ = gpu.read_output()
    r_rgb, r_ids, r_steps, st = orc.from_package_scene(sc).render(MODE_PRIMARY_SHADOW, 1920, 1080, want_steps=True)
    assert_frame_parity(rgb, ids, r_rgb, r_ids, "C2 at full size")
    assert np.array_equal(gpu.read_steps(), r_steps)
    s = gpu.stats()
    assert (s.steps, s.node_visits, s.secondary_rays, s.hits) == (st.steps, st.node_visits, st.secondary_rays, st.hits)
    gpu.render(MODE_PRIMARY_SHADOW)          # the timed (stats-free, two-in-flight) kernels give the same frame
    gpu.render(MODE_PRIMARY_SHADOW)
    rgb2, ids2, _ = gpu.read_output()
    assert np.array_equal(ids2, ids) and np.array_equal(rgb2, rgb)


@pytest.mark.parametrize("bounces,spp,seed", [(1, 1, 0), (2, 1, 0), (4, 1, 7), (4, 3, 0), (3, 2, 123)])
def test_path_trace_matches_oracle(orc, bounces, spp, seed):
    """Config C4's kernel family at a size the oracle finishes quickly: wavefront path trace, ids bit-exact,
    radiance within 1e-4, exact segment / step / node-visit counts."""
    sc = scenes.c4((320, 184), bounces=bounces)
    gpu = gpu_for_scene(sc)
    gpu.render(MODE_PATH, stats=True, spp=spp, seed=seed)
    rgb, ids, _ = gpu.read_output()
    o = orc.from_package_scene(sc)
    r_rgb, r_ids, r_steps, st = o.render(orc.MODE_PATH, *sc.size, want_steps=True, spp=spp, seed=seed)
    assert_frame_parity(rgb, ids, r_rgb, r_ids, f"path b{bounces} spp{spp}")
    assert np.array_equal(gpu.read_steps(), r_steps)
    s = gpu.stats()
    assert (s.primary_rays, s.secondary_rays, s.hits, s.steps, s.node_visits, s.primary_steps, s.primary_node_visits) == \
           (st.primary_rays, st.secondary_rays, st.hits, st.steps, st.node_visits, st.primary_steps, st.primary_node_visits)
    gpu.render(MODE_PATH, spp=spp, seed=seed)
    rgb2, ids2, _ = gpu.read_output()
    assert np.array_equal(ids2, ids) and np.array_equal(rgb2, rgb)


@pytest.mark.parametrize("in_flight", [1, 2, 3])
def test_path_trace_frames_in_flight(orc, in_flight):
    """Path-trace frames enqueued back to back with different seeds (each on its own stream, path buffers and segment
    cursors when more than one is in flight): the read-back is the last one, bit for bit what a lone frame gives."""
    sc = scenes.c4((160, 96), bounces=3)
    gpu = gpu_for_scene(sc)
    gpu.set_frames_in_flight(in_flight)
    for seed in (11, 12, 13, 14):
        gpu.render(MODE_PATH, spp=2, seed=seed)
    rgb, ids, _ = gpu.read_output()
    r_rgb, r_ids, _, _ = orc.from_package_scene(sc).render(orc.MODE_PATH, *sc.size, spp=2, seed=14)
    assert_frame_parity(rgb, ids, r_rgb, r_ids, f"path, {in_flight} in flight")
    gpu.render(MODE_PATH, spp=2, seed=12)
    gpu.render(MODE_PRIMARY_SHADOW)                 # a different kind of frame right behind a path frame
    gpu.render(MODE_PATH, spp=2, seed=14)
    rgb2, ids2, _ = gpu.read_output()
    assert np.array_equal(ids2, ids) and np.array_equal(rgb2, rgb)


def test_path_trace_mirror_materials_and_sharding(orc):
    """scatter = 0 (what Material::construct produces, graphics/mod.rs:44) makes every voxel a mirror; and a
    sharded path-traced frame is the union of its shards."""
    sc = scenes.c4((256, 144), bounces=3)
    for i in range(256):
        sc.materials[i].scatter = 0.0 if i % 2 else 0.5
    gpu = gpu_for_scene(sc)
    gpu.render(MODE_PATH, spp=2, seed=5)
    rgb, ids, _ = gpu.read_output()
    o = orc.from_package_scene(sc)
    r_rgb, r_ids, _, _ = o.render(orc.MODE_PATH, *sc.size, spp=2, seed=5)
    assert_frame_parity(rgb, ids, r_rgb, r_ids, "mirror path")
    acc_rgb, acc_ids = np.zeros_like(rgb), np.zeros_like(ids)
    for r in range(3):
        sh = gpu_for_scene(sc, shard_rank=r, shard_count=3)
        sh.render(MODE_PATH, spp=2, seed=5)
        s_rgb, s_ids, _ = sh.read_output()
        acc_rgb += s_rgb
        acc_ids |= s_ids
        sh.close()
    assert np.array_equal(acc_ids, ids) and np.array_equal(acc_rgb, rgb)


def test_large_world_uses_global_chunk_table(orc):
    """24^3 chunks = 13 824 chunk roots: more than the 8 192 entries staged in LDS, so the kernels read the
    table from global memory (LDS_ROOTS = false).  (ClientWorld((12,12,12), ., 24) has min chunk (0,0,0): world.min is the
    origin here; worlds whose min is not are tests/test_gpu_operating_point.py.)"""
    from voxelraytracing_amd.world import ClientWorld, gen_height
    w = ClientWorld((12, 12, 12), 1 << 25, 24)
    w.generate(0, 1)
    c = 24 * 16
    eye = (c + 0.5, float(gen_height(1, c, c) + 30) + 0.5, c + 0.5)
    sc = scenes._scene("24^3", w, (320, 184), eye, (25.0, 60.0, 0.0), MODE_PRIMARY_SHADOW)
    gpu = gpu_for_scene(sc)
    o = orc.from_package_scene(sc)
    for mode, kw in ((MODE_PRIMARY_SHADOW, {}), (MODE_PATH, dict(spp=1, seed=3))):
        gpu.render(mode, stats=True, **kw)
        rgb, ids, _ = gpu.read_output()
        r_rgb, r_ids, r_steps, st = o.render(mode, *sc.size, want_steps=True, **kw)
        assert_frame_parity(rgb, ids, r_rgb, r_ids, f"24^3 mode {mode}")
        assert np.array_equal(gpu.read_steps(), r_steps)
        assert gpu.stats().steps == st.steps
    gpu.render(MODE_PRIMARY_SHADOW, variant=1)
    rgb1, ids1, _ = gpu.read_output()
    r_rgb, r_ids, _, _ = o.render(MODE_PRIMARY_SHADOW, *sc.size)
    assert_frame_parity(rgb1, ids1, r_rgb, r_ids, "24^3 literal")


def test_random_cameras_stress(orc):
    """24 seeded random cameras (positions anywhere inside the 8^3 world incl. inside terrain and under water,
    any yaw/pitch, some roll, fov 40..110) at 160x96, primary + shadow: id words and step counts bit-exact."""
    sc = scenes.c2((160, 96))
    gpu = gpu_for_scene(sc)
    o = orc.from_package_scene(sc)
    rng = np.random.default_rng(2026)
    worst = 0.0
    for i in range(24):
        eye = tuple(float(v) for v in rng.uniform(1.0, 255.0, 3))
        if i % 3 == 0:   # just above the terrain
            from voxelraytracing_amd.world import gen_height
            eye = (eye[0], float(gen_height(1, int(eye[0]), int(eye[2])) + rng.uniform(0.5, 12.0)), eye[2])
        rot = (float(rng.uniform(-89, 89)), float(rng.uniform(0, 360)), float(rng.choice([0.0, 0.0, rng.uniform(-30, 30)])))
        fov = float(rng.uniform(40, 110))
        cam = g.cam_data_create(rot, eye, fov, (160.0, 96.0))
        gpu.write_cam_data(cam)
        gpu.render(MODE_PRIMARY_SHADOW, stats=True)
        rgb, ids, _ = gpu.read_output()
        o.set_cam(cam)
        r_rgb, r_ids, r_steps, _ = o.render(MODE_PRIMARY_SHADOW, 160, 96, want_steps=True)
        bad = np.argwhere(ids != r_ids)
        assert bad.size == 0, f"camera {i} eye {eye} rot {rot} fov {fov}: {len(bad)} id words differ, first {tuple(bad[0])}"
        assert np.array_equal(gpu.read_steps(), r_steps), f"camera {i}: step counts differ"
        worst = max(worst, float(np.abs(rgb - r_rgb).max()))
    assert worst <= 1e-4


def test_random_edits_stress(orc):
    """200 seeded random voxel edits (air / solids / water) near the camera, re-uploading each edited chunk's range."""
    sc = scenes.c2((160, 96))
    gpu = gpu_for_scene(sc)
    rng = np.random.default_rng(7)
    ex, ey, ez = (int(v) for v in sc.eye)
    n_ok = 0
    for _ in range(200):
        p = (ex + int(rng.integers(-20, 21)), ey + int(rng.integers(-24, 6)), ez + int(rng.integers(-20, 21)))
        v = int(rng.choice([0, 0, 3, 4, 40, 47, 62]))
        try:
            start, n = sc.world.set_voxel(p, v)
        except Exception as e:   # NoChange / NoChunk (an all-air chunk has no storage) / OutOfMemory (slack used up)
            assert getattr(e, "kind", "") in ("NoChange", "NoChunk", "OutOfMemory")
            continue
        gpu.write_nodes(sc.world.nodes_ptr(), start, start + n)
        n_ok += 1
    assert n_ok > 100
    gpu.write_chunk_roots(sc.world.chunk_roots())
    gpu.render(MODE_PRIMARY_SHADOW, stats=True)
    rgb, ids, _ = gpu.read_output()
    r_rgb, r_ids, r_steps, _ = orc.from_package_scene(sc).render(MODE_PRIMARY_SHADOW, 160, 96, want_steps=True)
    assert_frame_parity(rgb, ids, r_rgb, r_ids, "after 200 edits")
    assert np.array_equal(gpu.read_steps(), r_steps)


def test_fuzz_all_marches_agree_on_many_cameras():
    """400 seeded cameras — random, axis-aligned, on integer coordinates and chunk faces, inside terrain, under water,
    outside the world, extreme fov — rendered by the default march and by the literal restatement of the shader (which
    the oracle tests pin): id words, radiance and per-pixel step counts identical, frame by frame; every tenth camera also
    by the other two variants; every camera also by the default kernel without its counters (the hand-written march loop)."""
    from voxelraytracing_amd.world import gen_height
    sc = scenes.c2((96, 64))
    gpu = gpu_for_scene(sc)
    rng = np.random.default_rng(424242)
    n_checked = 0
    for i in range(400):
        kind = i % 8
        eye = [float(v) for v in rng.uniform(0.5, 255.5, 3)]
        rot = [float(rng.uniform(-89, 89)), float(rng.uniform(0, 360)), 0.0]
        fov = float(rng.uniform(30, 120))
        if kind == 1:    # exactly axis-aligned view directions (NaN unit steps on the centre column / row)
            rot = [float(rng.choice([0.0, 90.0, -90.0])), float(rng.choice([0.0, 90.0, 180.0, 270.0])), 0.0]
        elif kind == 2:  # integer coordinates: the start nudge (:188-190) and zero axis distances
            eye = [float(int(v)) for v in eye]
        elif kind == 3:  # on chunk faces
            eye = [float(32 * int(rng.integers(1, 8))), eye[1], float(32 * int(rng.integers(1, 8)))]
        elif kind == 4:  # just above / inside the terrain surface
            eye[1] = float(gen_height(1, int(eye[0]), int(eye[2]))) + float(rng.choice([-2.5, -0.5, 0.0, 0.001, 0.5, 3.0]))
        elif kind == 5:  # under the water line, looking up
            eye[1] = float(rng.uniform(41.0, 70.0))
            rot[0] = float(rng.uniform(-89, -10))
        elif kind == 6:  # outside the world: nothing but sky
            eye[int(rng.integers(0, 3))] = float(rng.choice([-3.0, 0.0, 256.0, 300.0]))
        elif kind == 7:
            fov = float(rng.choice([1.0, 5.0, 150.0, 175.0]))
            rot[2] = float(rng.uniform(-180, 180))
        gpu.write_cam_data(g.cam_data_create(tuple(rot), tuple(eye), fov, (96.0, 64.0)))
        frames = {}
        for variant in ((0, 1, 2, 3) if i % 10 == 0 else (0, 1)):
            gpu.render(MODE_PRIMARY_SHADOW, variant=variant, stats=True)
            rgb, ids, _ = gpu.read_output()
            frames[variant] = (rgb, ids, gpu.read_steps(), gpu.stats().node_visits)
        for variant, (rgb, ids, steps, visits) in frames.items():
            what = f"camera {i} kind {kind} eye {eye} rot {rot} fov {fov} variant {variant}"
            assert np.array_equal(ids, frames[1][1]), what
            assert np.array_equal(steps, frames[1][2]) and visits == frames[1][3], what
            assert np.array_equal(rgb, frames[1][0], equal_nan=True), what
        # ... and by the kernel that does not count: its march loop is the hand-written one (vrt_march.h (r), (s): the zero-distance path, the
        # split cells' air voxels, parked lanes, the exits for water and non-finite rays), the counting kernels' is the compiler's
        gpu.render(MODE_PRIMARY_SHADOW, variant=0)
        rgb, ids, _ = gpu.read_output()
        what = f"camera {i} kind {kind} eye {eye} rot {rot} fov {fov} the frame without counters"
        assert np.array_equal(ids, frames[1][1]), what
        assert np.array_equal(rgb, frames[1][0], equal_nan=True), what
        n_checked += 1
    assert n_checked == 400


def test_fuzz_path_trace_kernels_agree_on_many_cameras():
    """96 seeded cameras of the kinds above, path-traced (3 bounces) by the counting kernels — a launch per bounce, the compiler's march
    loop, pinned to the oracle by test_path_trace_matches_oracle — and by the default ones, whose bounce segments are one launch with the
    hand-written march loop (vrt_path.hip): identical id words and radiance, frame by frame."""
    from voxelraytracing_amd.world import gen_height
    sc = scenes.c4((96, 64), bounces=3)
    gpu = gpu_for_scene(sc)
    rng = np.random.default_rng(20261005)
    for i in range(96):
        kind = i % 8
        eye = [float(v) for v in rng.uniform(0.5, 255.5, 3)]
        rot = [float(rng.uniform(-89, 89)), float(rng.uniform(0, 360)), 0.0]
        fov = float(rng.uniform(30, 120))
        if kind == 1:
            rot = [float(rng.choice([0.0, 90.0, -90.0])), float(rng.choice([0.0, 90.0, 180.0, 270.0])), 0.0]
        elif kind == 2:
            eye = [float(int(v)) for v in eye]
        elif kind == 3:
            eye = [float(32 * int(rng.integers(1, 8))), eye[1], float(32 * int(rng.integers(1, 8)))]
        elif kind == 4:
            eye[1] = float(gen_height(1, int(eye[0]), int(eye[2]))) + float(rng.choice([-2.5, -0.5, 0.0, 0.001, 0.5, 3.0]))
        elif kind == 5:
            eye[1] = float(rng.uniform(41.0, 70.0))
            rot[0] = float(rng.uniform(-89, -10))
        elif kind == 6:
            eye[int(rng.integers(0, 3))] = float(rng.choice([-3.0, 0.0, 256.0, 300.0]))
        elif kind == 7:
            fov = float(rng.choice([1.0, 5.0, 150.0, 175.0]))
            rot[2] = float(rng.uniform(-180, 180))
        gpu.write_cam_data(g.cam_data_create(tuple(rot), tuple(eye), fov, (96.0, 64.0)))
        gpu.render(MODE_PATH, stats=True, seed=i)
        rgb, ids, _ = gpu.read_output()
        gpu.render(MODE_PATH, seed=i)
        rgb2, ids2, _ = gpu.read_output()
        what = f"camera {i} kind {kind} eye {eye} rot {rot} fov {fov}"
        assert np.array_equal(ids2, ids), what
        assert np.array_equal(rgb2, rgb, equal_nan=True), what
    gpu.close()


def test_c5_world_32_cubed_matches_oracle(orc):
    """Config C5's world (32^3 chunks = 1024^3 voxels, 64 MiB cell grid) at a small frame: primary + shadow and one
    path-traced sample against the oracle, ids / step counts bit-exact; the table sizes are what DESIGN.md states."""
    sc = scenes.c5((256, 144), bounces=3)
    gpu = gpu_for_scene(sc)
    o = orc.from_package_scene(sc)
    for mode, kw in ((MODE_PRIMARY_SHADOW, {}), (MODE_PATH, dict(spp=1, seed=5))):
        gpu.render(mode, stats=True, **kw)
        rgb, ids, _ = gpu.read_output()
        r_rgb, r_ids, r_steps, st = o.render(mode, *sc.size, want_steps=True, **kw)
        assert_frame_parity(rgb, ids, r_rgb, r_ids, f"32^3 mode {mode}")
        assert np.array_equal(gpu.read_steps(), r_steps) and gpu.stats().node_visits == st.node_visits
    a = gpu.accel_info()
    assert a.available == 1 and a.cells == 256 ** 3 and a.bytes == 256 * 257 * 257 * 4 + a.bricks * 128   # (zero border)
    gpu.render(MODE_PRIMARY_SHADOW, variant=2)      # the octree walk on the same world (chunk table in global memory: 32 768 roots)
    _, ids2, _ = gpu.read_output()
    gpu.render(MODE_PRIMARY_SHADOW)
    assert np.array_equal(gpu.read_output()[1], ids2)


def test_pipelined_in_place_root_with_a_stand_in_collective(c2_small):
    """bench.py's N > 1 frame loop on the root, on one GPU: FrameGather.submit / drain with batched gathers, compact
    records, a weighted in-place root that renders on the backend's own in-flight streams (VRT_RENDER_OWN_STREAMS) — and
    a stand-in for RCCL's gather that copies the other ranks' (pre-rendered) messages into the receive buffer on the
    current stream.  Every frame of every batch must equal the unsharded render of its camera."""
    import torch
    from voxelraytracing_amd.shard import FrameGather, texels_to_frame
    w, h = c2_small.size
    n, w0, batch, n_batches = 3, 2, 2, 3
    cams = [g.cam_data_create((15.0 + 6 * k, 20.0 + 55 * k, 0.0), (c2_small.eye[0] + k, c2_small.eye[1] + 0.5 * k, c2_small.eye[2] - k),
                              70.0, (float(w), float(h))) for k in range(batch * n_batches)]
    full = gpu_for_scene(c2_small)
    want = []
    for cam in cams:
        full.write_cam_data(cam)
        full.render(MODE_PRIMARY_SHADOW)
        want.append(full.read_output()[:2])
    dev = torch.device("cuda", 0)
    # what ranks 1..n-1 would send for every frame: rendered up front by their shard contexts
    others = [gpu_for_scene(c2_small, shard_rank=r, shard_count=n, root_weight=w0, compact=True) for r in range(1, n)]
    probe = FrameGather(torch, None, 1, n, w, h, dev, root_weight=w0, in_place=True, compact=True, batch=batch)
    sent = []   # sent[k][r-1] = rank r's message for frame k
    for cam in cams:
        row = []
        for sh in others:
            msg = torch.zeros(probe.frame_words, dtype=torch.int32, device=dev)
            sh.write_cam_data(cam)
            sh.bind_output(msg.data_ptr())
            sh.render(MODE_PRIMARY_SHADOW)
            sh.synchronize()
            row.append(msg)
        sent.append(row)

    class Work:
        def wait(self):
            pass

    class StandInDist:
        def __init__(self):
            self.calls = 0
        def gather(self, msg, gather_list, dst=0, async_op=False):
            first = self.calls * batch          # the frames of this call's batch
            nframes = msg.numel() // probe.frame_words
            for j in range(nframes):
                for r in range(1, n):
                    gather_list[r][j * probe.frame_words:(j + 1) * probe.frame_words].copy_(sent[first + j][r - 1], non_blocking=True)
            self.calls += 1
            return Work()

    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        root = gpu_for_scene(c2_small, shard_rank=0, shard_count=n, root_weight=w0, row_major=True)
        root.set_stream(side.cuda_stream)
        fg = FrameGather(torch, StandInDist(), 0, n, w, h, dev, root_weight=w0, in_place=True, compact=True, batch=batch)
        k = [0]
        def render():
            root.write_cam_data(cams[k[0]])     # the uniforms of the frame being rendered ...
            root.render(MODE_PRIMARY_SHADOW, own_streams=True)
            k[0] += 1
        done = []
        orig_assemble = fg.assemble
        def assemble(gpu, which=0, j=0):
            frame_no = (len(done) // batch) * batch + j
            gpu.write_cam_data(cams[frame_no])  # ... and of the frame being shaded at assembly (bench.py's camera never changes)
            orig_assemble(gpu, which, j)
            gpu.synchronize()
            torch.cuda.synchronize()
            a_rgb, a_ids = texels_to_frame(fg.frame.cpu().numpy().view(np.uint32))
            assert np.array_equal(a_ids, want[frame_no][1]) and np.array_equal(a_rgb, want[frame_no][0]), f"frame {frame_no}"
            done.append(frame_no)
        fg.assemble = assemble
        for _ in range(n_batches):
            fg.submit(root, render, batch)
        fg.drain(root)
    torch.cuda.synchronize()
    assert done == list(range(batch * n_batches))
    for c in others + [root, full]:
        c.close()


@pytest.mark.parametrize("which", ["sun_intensity_inf", "sky_nan", "sun_pos_inf", "liquid_everything", "materials_shifted"])
def test_non_finite_settings_and_odd_materials(orc, which):
    """Settings the host would never send on purpose — an infinite sun, a NaN sky colour, a sun at infinity — make
    `vox*f32(hit) + sky*f32(!hit)` (:135) produce NaNs that the one-sided shortcut (§3 g) would hide, so the kernels must
    fall back to the blend; and material tables that flip which voxels are liquid.  All four marches against the oracle,
    NaN for NaN, and the compact-record path of the gather root as well."""
    import torch
    from voxelraytracing_amd.shard import FrameGather, texels_to_frame
    sc = scenes.c2((160, 96))
    if which == "sun_intensity_inf":
        sc.settings.sun_intensity = float("inf")
    elif which == "sky_nan":
        sc.settings.sky_color[1] = float("nan")
    elif which == "sun_pos_inf":
        sc.settings.sun_pos[0] = float("inf")
    elif which == "liquid_everything":
        for i in range(256):
            sc.materials[i].is_liquid = 1       # every ray runs to the world's edge or 500 steps through "water"
    elif which == "materials_shifted":
        for i in range(256):
            sc.materials[i].is_liquid = 1 if i in (40, 47) else 0   # grass and sand are liquid, water (3) is solid
    gpu = gpu_for_scene(sc)
    o = orc.from_package_scene(sc)
    r_rgb, r_ids, r_steps, _ = o.render(MODE_PRIMARY_SHADOW, 160, 96, want_steps=True)

    def same(rgb, ids, what):
        assert np.array_equal(ids, r_ids), what
        assert np.array_equal(np.isnan(rgb), np.isnan(r_rgb)), what
        both = np.isfinite(rgb) & np.isfinite(r_rgb)
        assert np.array_equal(np.isinf(rgb), np.isinf(r_rgb)) and float(np.abs(rgb[both] - r_rgb[both]).max(initial=0.0)) <= 1e-4, what

    for variant in VARIANTS:
        gpu.render(MODE_PRIMARY_SHADOW, variant=variant, stats=True)
        rgb, ids, _ = gpu.read_output()
        same(rgb, ids, f"{which} variant {variant}")
        assert np.array_equal(gpu.read_steps(), r_steps), f"{which} variant {variant}"
    if which == "liquid_everything":   # air flagged liquid: the backend hands the frame to the literal march; compact shards refuse it
        sh = gpu_for_scene(sc, shard_rank=1, shard_count=2, compact=True)
        with pytest.raises(g.VrtError):
            sh.render(MODE_PRIMARY_SHADOW)
        sh.close()
        scp = scenes.c4((160, 96), bounces=3)
        for i in range(256):
            scp.materials[i].is_liquid = 1
        gp = gpu_for_scene(scp)
        gp.render(MODE_PATH, spp=1, seed=2, stats=True)
        p_rgb, p_ids, _ = gp.read_output()
        q_rgb, q_ids, q_steps, _ = orc.from_package_scene(scp).render(MODE_PATH, 160, 96, want_steps=True, spp=1, seed=2)
        assert_frame_parity(p_rgb, p_ids, q_rgb, q_ids, "path trace with every material liquid")
        assert np.array_equal(gp.read_steps(), q_steps)
        return
    # the gather root shading compact records must reproduce the same NaNs
    n, w0 = 3, 2
    fg0 = FrameGather(torch, None, 0, n, 160, 96, torch.device("cuda", 0), root_weight=w0, in_place=True, compact=True)
    ctxs = []
    for r in range(n):
        sh = gpu_for_scene(sc, shard_rank=r, shard_count=n, root_weight=w0, row_major=(r == 0), compact=(r != 0))
        if r == 0:
            fg0.bind(sh, 0)
        else:
            sh.bind_output(fg0.recv[0][r].data_ptr())
        sh.render(MODE_PRIMARY_SHADOW)
        sh.synchronize()
        ctxs.append(sh)
    fg0.assemble(ctxs[0], 0)
    ctxs[0].synchronize()
    a_rgb, a_ids = texels_to_frame(fg0.frame.cpu().numpy().view(np.uint32))
    same(a_rgb, a_ids, f"{which} compact gather")
    for c in ctxs:
        c.close()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_garbage_node_pools_and_roots(orc, seed):
    """Node pools the host would never build — random words (random split bits, child indices pointing anywhere, past
    the end of the pool included) under random chunk roots (inside the pool, past it, zero) — are still a well-defined
    input of the shader: a walk is at most six reads, a read past the end is 0.  The derived tables must agree with the
    walks on them (every voxel), and all four marches with the oracle."""
    from voxelraytracing_amd import Gpu
    from test_gpu_accel import lookup_tables, walk_octree
    rng = np.random.default_rng(seed)
    S, n_nodes = 2, 4096
    nodes = rng.integers(0, 1 << 16, n_nodes, dtype=np.uint16)
    nodes[rng.random(n_nodes) < 0.5] &= 0x00FF        # half of them leaves with small voxel ids
    nodes[rng.random(n_nodes) < 0.3] = 0              # plenty of air
    nodes[0] = 0
    roots = rng.integers(0, n_nodes + 300, S ** 3).astype(np.uint32)   # some past the end
    roots[rng.integers(0, S ** 3)] = 0
    sc = scenes.c1_flat((96, 64))                      # for the camera, settings and materials of a 2^3 world
    gpu = Gpu(n_nodes, S, (96, 64))
    gpu.write_nodes(nodes, 0, n_nodes)
    gpu.write_chunk_roots(roots)
    gpu.write_world_data(sc.world.world_data())
    gpu.write_materials(sc.materials)
    gpu.write_settings(sc.settings)
    o = orc.OracleScene(nodes, roots, sc.materials, sc.cam, sc.settings, sc.world.world_data())
    for k, (rot, eye) in enumerate([((15.0, 0.0, 0.0), (32.5, 20.5, 60.5)), ((-30.0, 140.0, 0.0), (10.5, 50.5, 12.25)),
                                    ((80.0, 45.0, 10.0), (40.0, 63.5, 33.0))]):
        cam = g.cam_data_create(rot, eye, 80.0, (96.0, 64.0))
        gpu.write_cam_data(cam)
        o.set_cam(cam)
        r_rgb, r_ids, r_steps, st = o.render(MODE_PRIMARY_SHADOW, 96, 64, want_steps=True)
        for variant in VARIANTS:
            gpu.render(MODE_PRIMARY_SHADOW, variant=variant, stats=True)
            rgb, ids, _ = gpu.read_output()
            assert_frame_parity(rgb, ids, r_rgb, r_ids, f"seed {seed} camera {k} variant {variant}")
            assert np.array_equal(gpu.read_steps(), r_steps) and gpu.stats().node_visits == st.node_visits
    grid, bricks = gpu.read_accel()
    v_ref, d_ref = walk_octree(nodes, roots, S)
    v, size = lookup_tables(grid, bricks, S)
    assert np.array_equal(v, v_ref) and np.array_equal(size, 32 >> d_ref)


def test_compact_shards_without_the_derived_tables_stay_inside_their_records(c2_small, monkeypatch):
    """A VRT_FLAG_COMPACT context whose world is too large for the derived tables (forced here through VRT_ACCEL_MAX_S)
    falls back to the octree walk — still as one launch that stores 8-byte records: the two-launch kernels store
    16-byte texels, which a bound slots*8-byte message has no room for.  A guard region behind the message stays intact
    and the assembled frame equals the unsharded one; an explicit two-launch variant is refused."""
    import torch
    from voxelraytracing_amd.shard import FrameGather, texels_to_frame
    w, h = c2_small.size
    full = gpu_for_scene(c2_small)
    full.render(MODE_PRIMARY_SHADOW)
    rgb, ids, _ = full.read_output()
    monkeypatch.setenv("VRT_ACCEL_MAX_S", "1")
    n, w0 = 3, 2
    fg0 = FrameGather(torch, None, 0, n, w, h, torch.device("cuda", 0), root_weight=w0, in_place=True, compact=True)
    words = fg0.frame_words
    guard = torch.full((n, words + 4096), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
    ctxs = []
    for r in range(n):
        sh = gpu_for_scene(c2_small, shard_rank=r, shard_count=n, root_weight=w0, row_major=(r == 0), compact=(r != 0))
        if r == 0:
            fg0.bind(sh, 0)
        else:
            assert sh.device_output()[1] == words * 4
            sh.bind_output(guard[r].data_ptr())
            with pytest.raises(g.VrtError):
                sh.render(MODE_PRIMARY_SHADOW, variant=3)
        sh.render(MODE_PRIMARY_SHADOW)
        sh.synchronize()
        assert not sh.accel_info().available
        ctxs.append(sh)
    assert bool((guard[:, words:] == 0x5A5A5A5A).all()), "a shard wrote past its 8-byte records"
    fg0.recv[0][1:].copy_(guard[1:, :words])
    fg0.assemble(ctxs[0], 0)
    ctxs[0].synchronize()
    a_rgb, a_ids = texels_to_frame(fg0.frame.cpu().numpy().view(np.uint32))
    assert np.array_equal(a_ids, ids) and np.array_equal(a_rgb, rgb)
    for c in ctxs + [full]:
        c.close()


@pytest.mark.parametrize("k", ["4", "5"])
def test_pool_depth_of_the_bounce_waves_does_not_change_the_frame(orc, monkeypatch, k):
    """The bounce launch's waves keep 256 rays (VRT_PATH_POOL_K=4) or 320 (5: the default for small worlds with two frames in
    flight, profiles/r05_pool_k5.txt): the same frame, one frame at a time and two in flight, one and several samples."""
    sc = scenes.c4((320, 184), bounces=4)
    r_rgb, r_ids, _, _ = orc.from_package_scene(sc).render(orc.MODE_PATH, *sc.size, spp=3, seed=11)
    monkeypatch.setenv("VRT_PATH_POOL_K", k)
    for in_flight in (1, 2):
        gpu = gpu_for_scene(sc)
        gpu.set_frames_in_flight(in_flight)
        for _ in range(3):
            gpu.render(MODE_PATH, spp=3, seed=11)
        rgb, ids, _ = gpu.read_output()
        assert_frame_parity(rgb, ids, r_rgb, r_ids, f"pool of {k} batches, {in_flight} in flight")
        gpu.close()


@pytest.mark.parametrize("env", [{"VRT_PATH_POOL": "0"}, {"VRT_PATH_CELLS": "0"}] +
                                [{"VRT_PATH_WINDOW": "1", "VRT_PATH_WINDOW_SHAPE": str(k)} for k in range(5)])
def test_every_form_of_the_bounce_launch_gives_the_same_frames(orc, monkeypatch, env):
    """The default bounce launch is the pool kernel over the march cells (a wave refills its lanes from its own LDS pool of
    rays; one 16-byte load per step).  VRT_PATH_POOL=0 / VRT_PATH_CELLS=0: lane = path for the whole kernel, one launch per
    bounce, the round-1 structure (what worlds without march cells and stats frames run).
    VRT_PATH_WINDOW=1 (round 5; built and measured, not the default: profiles/r05_window_ab.txt; the experiments build): the rays
    grouped by screen block, the march cells around a group staged in LDS (shapes 0-3), or (shape 4) the pool kernel with its rays'
    state in global memory.  All bit for bit the same frame, with several samples, sharded, and with two frames in flight."""
    if "VRT_PATH_WINDOW" in env:
        needs_experiments()
    sc = scenes.c4((320, 184), bounces=4)
    ref = gpu_for_scene(sc)
    ref.render(MODE_PATH, spp=3, seed=11)
    rgb, ids, _ = ref.read_output()
    r_rgb, r_ids, _, _ = orc.from_package_scene(sc).render(orc.MODE_PATH, *sc.size, spp=3, seed=11)
    assert_frame_parity(rgb, ids, r_rgb, r_ids, "pool bounce kernel")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    gpu = gpu_for_scene(sc)
    for _ in range(4):   # frames in flight alternate between the context's frame sets
        gpu.render(MODE_PATH, spp=3, seed=11)
    rgb2, ids2, _ = gpu.read_output()
    assert np.array_equal(ids2, ids) and np.array_equal(rgb2, rgb)
    gpu.close()
    acc_rgb, acc_ids = np.zeros_like(rgb), np.zeros_like(ids)
    for r in range(2):
        sh = gpu_for_scene(sc, shard_rank=r, shard_count=2)
        sh.render(MODE_PATH, spp=3, seed=11)
        s_rgb, s_ids, _ = sh.read_output()
        acc_rgb += s_rgb
        acc_ids |= s_ids
        sh.close()
    assert np.array_equal(acc_ids, ids) and np.array_equal(acc_rgb, rgb)
    ref.close()


@pytest.mark.parametrize("per_chain", ["1", "2", "4", "16"])
def test_samples_per_launch_chain_do_not_change_the_frame(orc, monkeypatch, per_chain):
    """A path-traced frame with spp > 1 runs its samples several per launch chain (default 4; VRT_PATH_SAMPLES_PER_CHAIN):
    every sample accumulates into its own plane and the chain's finishing pass adds the planes in sample order — the frame
    is bit for bit the one that one sample per chain gives, for chains that divide spp and chains that do not."""
    sc = scenes.c4((320, 184), bounces=4)
    monkeypatch.setenv("VRT_PATH_SAMPLES_PER_CHAIN", "1")
    ref = gpu_for_scene(sc)
    ref.render(MODE_PATH, spp=5, seed=3)
    rgb, ids, _ = ref.read_output()
    ref.close()
    r_rgb, r_ids, _, _ = orc.from_package_scene(sc).render(orc.MODE_PATH, *sc.size, spp=5, seed=3)
    assert_frame_parity(rgb, ids, r_rgb, r_ids, "one sample per chain")
    monkeypatch.setenv("VRT_PATH_SAMPLES_PER_CHAIN", per_chain)
    gpu = gpu_for_scene(sc)
    for _ in range(3):
        gpu.render(MODE_PATH, spp=5, seed=3)
    rgb2, ids2, _ = gpu.read_output()
    assert np.array_equal(ids2, ids) and np.array_equal(rgb2, rgb)
    gpu.render(MODE_PATH, spp=1, seed=3)    # and a one-sample frame on the same context afterwards
    rgb1, ids1, _ = gpu.read_output()
    gpu.close()
    one = gpu_for_scene(sc)
    one.render(MODE_PATH, spp=1, seed=3)
    o_rgb, o_ids, _ = one.read_output()
    one.close()
    assert np.array_equal(ids1, o_ids) and np.array_equal(rgb1, o_rgb)


def test_tiles_ordered_under_a_moving_camera_are_the_same_frames(orc, monkeypatch):
    """A one-frame-at-a-time context keeps launching its tiles longest first while the camera MOVES (round 5):
    ONE launch sorts blocks of 4 x 4 tiles by their trips dilated over the image motion of several camera steps
    (vrt_kernels.hip: launch_tile_order_blocks), and the order is kept while the camera stays within what the dilation covers.
    (An order per frame, in two forms, was measured slower and is gone: profiles/r04_tile_order_moving.txt, r05_tile_order_moving.txt.)
    A walk of small steps (the bench's orbit step: < 1 voxel, ~ 1 degree), a rest, a jump (screen order
    again), more steps: every frame is the screen-order context's frame, the last one the oracle's; and the counter says which
    frames were ordered."""
    from voxelraytracing_amd import graphics as g
    sc = scenes.c2()
    mov = gpu_for_scene(sc)
    mov.set_frames_in_flight(1)
    monkeypatch.setenv("VRT_TILE_ORDER", "0")
    ref = gpu_for_scene(sc)
    ref.set_frames_in_flight(1)
    monkeypatch.delenv("VRT_TILE_ORDER")

    def cam_at(k, jump=0.0):
        return g.cam_data_create((sc.rot[0] + 0.3 * k, sc.rot[1] + 0.9 * k + jump, 0.0), (sc.eye[0] + 0.5 * k + jump, sc.eye[1] + 0.1 * (k % 3), sc.eye[2] - 0.4 * k), 70.0, (1920.0, 1080.0))
    #           steps 0..5 move, 6..8 rest at step 5's camera, 9 jumps 40 degrees / 40 voxels, 10..13 move on from there
    plan = [(k, 0.0) for k in range(6)] + [(5, 0.0)] * 3 + [(9 + k, 40.0) for k in range(5)]
    ordered = []
    cam = None
    for n, (k, jump) in enumerate(plan):
        cam = cam_at(k, jump)
        for gpu in (mov, ref):
            gpu.write_cam_data(cam)
            gpu.render(MODE_PRIMARY_SHADOW)
        a_rgb, a_ids, _ = mov.read_output()
        b_rgb, b_ids, _ = ref.read_output()
        assert np.array_equal(a_ids, b_ids) and np.array_equal(a_rgb, b_rgb), n
        ordered.append(mov.accel_info().ordered_frames)
    used = [b - a for a, b in zip([0] + ordered[:-1], ordered)]
    # the first frame has no order; every frame a step (or no step) from its predecessor has one; the frame after the jump has none
    want = [0, 1, 1, 1, 1, 1, 1, 1, 1, 0, 1, 1, 1, 1]
    assert used == want, used
    assert ref.accel_info().ordered_frames == 0
    o = orc.from_package_scene(sc)
    o.set_cam(cam)
    r_rgb, r_ids, _, _ = o.render(MODE_PRIMARY_SHADOW, 1920, 1080)
    assert_frame_parity(a_rgb, a_ids, r_rgb, r_ids, "tiles ordered under a moving camera")
    # the default is form 1; VRT_TILE_ORDER_MOVING=0 is round 3's rule: screen order whenever the view has changed
    dflt = gpu_for_scene(sc)
    monkeypatch.setenv("VRT_TILE_ORDER_MOVING", "0")
    old = gpu_for_scene(sc)
    monkeypatch.delenv("VRT_TILE_ORDER_MOVING")
    for gpu in (dflt, old):
        gpu.set_frames_in_flight(1)
        for k in range(4):
            gpu.write_cam_data(cam_at(k))
            gpu.render(MODE_PRIMARY_SHADOW)
    assert dflt.accel_info().ordered_frames == 3 and old.accel_info().ordered_frames == 0
    dflt.close()
    for gpu in (mov, ref, old):
        gpu.close()


@pytest.mark.parametrize("size", [(3840, 2160), (200, 104), (64, 8)])
def test_block_order_on_other_frame_shapes(monkeypatch, size):
    """The one-launch order of a moving view (launch_tile_order_blocks) on a 4K frame (8 160 blocks: 96 KiB of LDS, opted in), on one
    whose tile rows and columns are not multiples of four, and on a single row of tiles: the frames of a walking camera equal the
    screen-order context's, and the order is really used."""
    from voxelraytracing_amd import graphics as g
    sc = scenes.c2(size)
    monkeypatch.setenv("VRT_TILE_ORDER_MOVING", "1")
    mov = gpu_for_scene(sc)
    mov.set_frames_in_flight(1)
    monkeypatch.setenv("VRT_TILE_ORDER_MOVING", "0")
    monkeypatch.setenv("VRT_TILE_ORDER", "0")
    ref = gpu_for_scene(sc)
    ref.set_frames_in_flight(1)
    for k in range(5):
        cam = g.cam_data_create((sc.rot[0] + 0.3 * k, sc.rot[1] + 0.9 * k, 0.0), (sc.eye[0] + 0.5 * k, sc.eye[1], sc.eye[2] - 0.4 * k), 70.0, (float(size[0]), float(size[1])))
        for gpu in (mov, ref):
            gpu.write_cam_data(cam)
            gpu.render(MODE_PRIMARY_SHADOW)
        a_rgb, a_ids, _ = mov.read_output()
        b_rgb, b_ids, _ = ref.read_output()
        assert np.array_equal(a_ids, b_ids) and np.array_equal(a_rgb, b_rgb), k
    tiles = (size[0] // 8) * (size[1] // 8)
    assert mov.accel_info().ordered_frames == (4 if tiles >= 128 else 0)   # (a frame of fewer than 128 tiles is not ordered at all)
    assert ref.accel_info().ordered_frames == 0
    mov.close(); ref.close()


@pytest.mark.parametrize("pace", ["walk", "run", "leap"])
def test_a_kept_block_order_serves_the_frames_it_covers_and_no_others(monkeypatch, pace):
    """The moving view's order is KEPT (vrt_order.hip: hold_limits): a walk of the bench's orbit step for 30 frames is
    ordered from its second frame on — the order is made again before the camera leaves what its dilation covers, never after —;
    at 4 degrees a step an order serves a few frames and is made again; at 40 degrees and 10 voxels a step no order is ever used, and the
    context stops asking for them.  Every frame is the screen-order context's."""
    from voxelraytracing_amd import graphics as g
    sc = scenes.c2((640, 360))
    mov = gpu_for_scene(sc)
    mov.set_frames_in_flight(1)
    monkeypatch.setenv("VRT_TILE_ORDER", "0")
    ref = gpu_for_scene(sc)
    ref.set_frames_in_flight(1)
    monkeypatch.delenv("VRT_TILE_ORDER")
    step = {"walk": (0.3, 0.9, 0.6), "run": (0.5, 4.0, 1.5), "leap": (1.0, 40.0, 20.0)}[pace]
    used, before = [], 0
    for k in range(30):
        cam = g.cam_data_create((sc.rot[0] + step[0] * (k % 7), sc.rot[1] + step[1] * k, 0.0), (sc.eye[0] + step[2] * math.cos(0.1 * k) * k / 3.0, sc.eye[1], sc.eye[2] - step[2] * k / 2.0),
                                70.0, (640.0, 360.0))
        for gpu in (mov, ref):
            gpu.write_cam_data(cam)
            gpu.render(MODE_PRIMARY_SHADOW)
        a_rgb, a_ids, _ = mov.read_output()
        b_rgb, b_ids, _ = ref.read_output()
        assert np.array_equal(a_ids, b_ids) and np.array_equal(a_rgb, b_rgb), k
        n = mov.accel_info().ordered_frames
        used.append(n - before)
        before = n
    if pace == "walk":
        assert used == [0] + [1] * 29, used
    elif pace == "run":
        assert 10 <= sum(used) <= 29 and used[0] == 0, used
    else:
        assert sum(used) == 0, used
    assert ref.accel_info().ordered_frames == 0
    mov.close(); ref.close()


def test_longest_tiles_first_is_the_same_frame(orc, monkeypatch):
    """A context that renders one frame at a time launches the tiles of a view at rest longest first (the order from the
    trips the view's second frame noted; any change of the view goes back to screen order): any order of the tiles is the
    same frame.  Full size; a camera that rests for 1-4 frames at each stop, so that screen-order frames, the frames that
    note their trips and the ordered ones are all compared, against the screen-order context and the oracle."""
    sc = scenes.c2()
    lpt = gpu_for_scene(sc)
    lpt.set_frames_in_flight(1)
    monkeypatch.setenv("VRT_TILE_ORDER", "0")
    ref = gpu_for_scene(sc)
    ref.set_frames_in_flight(1)
    from voxelraytracing_amd import graphics as g
    cam = sc.cam
    for k in range(12):
        cam = g.cam_data_create((sc.rot[0] + 2.0 * k, sc.rot[1] + 17.0 * k, 0.0), (sc.eye[0] + 3.0 * k, sc.eye[1] + (k % 3), sc.eye[2] - 2.0 * k), 70.0, (1920.0, 1080.0))
        for gpu in (lpt, ref):
            gpu.write_cam_data(cam)
        for _ in range(1 + k % 4):
            lpt.render(MODE_PRIMARY_SHADOW)
        ref.render(MODE_PRIMARY_SHADOW)
        a_rgb, a_ids, _ = lpt.read_output()
        b_rgb, b_ids, _ = ref.read_output()
        assert np.array_equal(a_ids, b_ids) and np.array_equal(a_rgb, b_rgb), k
    o = orc.from_package_scene(sc)
    o.set_cam(cam)
    r_rgb, r_ids, _, _ = o.render(MODE_PRIMARY_SHADOW, 1920, 1080)
    assert_frame_parity(a_rgb, a_ids, r_rgb, r_ids, "longest tiles first")
    # sharded contexts order their own tiles
    monkeypatch.delenv("VRT_TILE_ORDER")
    acc_rgb, acc_ids = np.zeros_like(a_rgb), np.zeros_like(a_ids)
    for r in range(2):
        sh = gpu_for_scene(sc, shard_rank=r, shard_count=2)
        sh.set_frames_in_flight(1)
        sh.write_cam_data(cam)
        for _ in range(5):
            sh.render(MODE_PRIMARY_SHADOW)
        s_rgb, s_ids, _ = sh.read_output()
        acc_rgb += s_rgb
        acc_ids |= s_ids
        sh.close()
    assert np.array_equal(acc_ids, a_ids) and np.array_equal(acc_rgb, a_rgb)
    lpt.close()
    ref.close()

