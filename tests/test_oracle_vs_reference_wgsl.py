"""The oracle against the REFERENCE'S OWN SHADER TEXT.

tests/golden/wgsl_*.npz were made by executing `/root/reference/clientdesktop/src/graphics/ray_tracer.wgsl` as it stands,
invocation by invocation, with tests/wgsl_interp.py — a generic WGSL interpreter that knows nothing about octrees or ray
marching (tests/golden/make_wgsl_fixtures.py, which reads the shader from the reference at generation time; no shader text is
stored here).  oracle/vrt_oracle.c must reproduce what the shader computed: which voxel every ray stops on, the `hit` flag,
the normal's axes, whether the ray went through water — bit for bit — the per-pixel iteration count of `ray_world`, the hit
position and water distance as binary32 values, and the colour handed to textureStore to 1e-6 (`pow` in the sky).  That pins
the oracle to the reference's source up to the corners WGSL leaves to the implementation (the interpreter lists its choices).
Where /root/reference exists (the build container) a few pixels are traced again, so the fixtures cannot drift from the
generator."""
import os
import sys
import zlib

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_wgsl_fixtures as mk   # noqa: E402  (the scenes' definitions live with the generator)

GOLD = os.path.join(HERE, "golden")


def _load(case):
    return np.load(os.path.join(GOLD, f"wgsl_{case}.npz"))


@pytest.mark.parametrize("case", mk.CASES)
def test_scene_inputs_are_the_ones_the_shader_was_run_on(case):
    f = _load(case)
    sc, win = mk.case_scene(case)
    ck = mk.scene_checksums(sc)
    for k, v in ck.items():
        assert np.array_equal(v, f[k]), f"{case}: {k} differs — the fixture was made from another scene"
    assert tuple(f["window"]) == win and tuple(f["size"]) == tuple(sc.size)


@pytest.mark.parametrize("case", mk.CASES)
def test_oracle_computes_what_the_reference_shader_computes(case, orc):
    f = _load(case)
    sc, (x0, y0, x1, y1) = mk.case_scene(case)
    w, h = sc.size
    o = orc.from_package_scene(sc)
    rgb, ids, steps, _ = o.render(orc.MODE_PRIMARY, w, h, want_steps=True)
    rgb, ids, steps = rgb[y0:y1, x0:x1], ids[y0:y1, x0:x1], steps[y0:y1, x0:x1]
    hit = f["hit"].astype(bool)
    # the id word, composed as the kernels and the oracle compose it, from what the SHADER computed
    want = np.where(hit, f["voxel"] & orc.ID_VOXEL_MASK, 0).astype(np.uint32)
    want |= np.where(hit, orc.ID_HIT, 0).astype(np.uint32)
    for axis, bit in enumerate((orc.ID_NX, orc.ID_NY, orc.ID_NZ)):
        want |= np.where(f["norm"][..., axis] != 0.0, bit, 0).astype(np.uint32)
    want |= np.where(f["water_dist"] != 0.0, orc.ID_WATER, 0).astype(np.uint32)
    bad = np.argwhere(ids != want)
    assert bad.size == 0, f"{case}: {len(bad)} id words differ from the shader's, first at (y, x) = {tuple(bad[0])}: " \
                          f"oracle {ids[tuple(bad[0])]:#x}, shader {want[tuple(bad[0])]:#x}"
    assert np.array_equal(steps, f["iters"]), f"{case}: per-pixel iteration counts of ray_world differ from the shader's"
    assert (hit.any() and (~hit).any()) or case.startswith(("c1_axis", "nan_")), f"{case}: a fixture should hold hits and misses"
    err = np.abs(rgb - f["rgb"])
    assert np.isnan(rgb).sum() == np.isnan(f["rgb"]).sum()
    assert float(np.nanmax(err)) <= 1e-6, f"{case}: colour differs from the shader's textureStore argument by {np.nanmax(err)}"
    # every pixel through the oracle's single-ray entry point: the HitResult's position, normal and water distance are the
    # shader's binary32 values, bit for bit (a miss leaves position and normal at their zero initial values in both)
    def bits(a):   # the binary32 patterns, every NaN as one pattern (a NaN's payload is not something WGSL defines)
        a = np.ascontiguousarray(a, dtype=np.float32)
        return np.where(np.isnan(a), np.uint32(0x7FC00000), a.view(np.uint32))
    got = np.array([[o.trace_pixel(orc.MODE_PRIMARY, x, y)[3] for x in range(x0, x1)] for y in range(y0, y1)], dtype=np.float32)
    assert np.array_equal(bits(got[..., 6]), bits(f["water_dist"])), f"{case}: water distances differ from the shader's"
    assert np.array_equal(got[..., 7].astype(np.uint32), f["iters"])
    assert np.array_equal(bits(got[..., 0:3])[hit], bits(f["pos"])[hit]), f"{case}: hit positions differ from the shader's"
    # (the sign of a zero component is -sign(dir) * 0 in the shader: the normals compare as values)
    assert np.array_equal(got[..., 3:6][hit], f["norm"][hit], equal_nan=True), f"{case}: normals differ from the shader's"


@pytest.mark.skipif(not os.path.exists(mk.SHADER), reason="the reference tree is only in the build container")
@pytest.mark.parametrize("case", ["c1_48", "c2_64x40"])
def test_the_generator_still_makes_the_fixtures(case):
    """A few rows traced again from the reference's shader: the committed fixtures are what the generator produces today."""
    f = _load(case)
    assert int(f["shader_crc"][0]) == zlib.crc32(open(mk.SHADER, "rb").read()), "the reference's shader changed"
    mk._init(case)
    _, (x0, y0, x1, y1) = mk.case_scene(case)
    for y in (y0 + 3, (y0 + y1) // 2):
        _, row = mk._trace_row((y, x0, min(x0 + 12, x1)))
        for i, (c, hh, v, it, nn, wd, pp) in enumerate(row):
            assert np.array_equal(np.float32(c), f["rgb"][y - y0, i]) and bool(hh) == bool(f["hit"][y - y0, i])
            assert v == f["voxel"][y - y0, i] and it == f["iters"][y - y0, i]


def test_rng_is_the_reference_shaders(orc):
    """rng_next (path_tracer.wgsl:56-61): state sequence and uniforms bit for bit; rng_next_dir (:62-72) to 1e-6 (log, cos)."""
    import ctypes as C
    f = np.load(os.path.join(GOLD, "wgsl_rng.npz"))
    L = orc.lib()
    for k, seed in enumerate(f["seeds"]):
        st = C.c_uint32(int(seed))
        for j in range(16):
            u = L.orc_rng_next(C.byref(st))
            assert st.value == int(f["states"][k, j]), (seed, j)
            assert np.float32(u).view(np.uint32) == f["uniforms"][k, j].view(np.uint32), (seed, j, u, f["uniforms"][k, j])
        st = C.c_uint32(int(seed))
        d = (C.c_float * 3)()
        L.orc_test_rng_dir(C.byref(st), d)
        assert st.value == int(f["dir_states"][k])
        # (the oracle clamps the uniform under the logarithm to 1e-10, the shader would take log(0): none of these seeds draws a zero)
        assert float(np.abs(np.float32(list(d)) - f["dirs"][k]).max()) <= 1e-6, (seed, list(d), f["dirs"][k])


def test_present_is_the_reference_fragment_shader(orc):
    """fs_main of screen_shader.wgsl:43-65 at 1:1: crosshair off / dot / cross — the oracle's blit, byte for byte after the
    unorm8 conversion of the colour attachment."""
    f = np.load(os.path.join(GOLD, "wgsl_present.npz"))
    src = np.load(os.path.join(GOLD, "wgsl_c1_48.npz"))["rgb"]
    h, w, _ = src.shape
    for k, row in enumerate(f["styles"]):
        style, size, color = int(row[0]), float(row[1]), tuple(float(c) for c in row[2:6])
        got = orc.present(src, (w, h), color=color, style=style, size=size)
        want = np.rint(np.clip(f[f"img{k}"], 0.0, 1.0) * np.float32(255.0)).astype(np.uint8)
        assert np.array_equal(got, want), f"crosshair {row}: {int((got != want).any(axis=2).sum())} pixels differ from fs_main's"
        if style:
            assert (want != np.rint(np.clip(f["img0"], 0.0, 1.0) * 255.0).astype(np.uint8)).any()


def test_reads_past_an_arrays_end_follow_the_pinned_clamp_policy(orc):
    """The two reads of ray_tracer.wgsl that can go past their arrays (voxel_mats[voxel] for ids >= 256, :226; chunk_roots_[idx]
    past the table, :121-124) under the two things WGSL lets an implementation do: tests/golden/wgsl_oob.npz holds what the
    reference's shader text computes with the index CLAMPED and with a ZERO value.  The oracle implements the clamp (its
    header says so; the kernels do too: tests/test_gpu_reference_wgsl.py).  That is a PINNED CHOICE, NOT A DECISION: which of the two
    the reference's own runs get depends on its backend (Metal / DX12 clamp; Vulkan with robustBufferAccess2 may return zero), so the
    test holds the oracle to the clamped frame and only shows that the fixture can tell the two apart — one run of the same two
    scenes through a real wgpu (tools/wgpu_check prints which policy it sees) settles it."""
    f = np.load(os.path.join(GOLD, "wgsl_oob.npz"))
    sc = mk.oob_material_scene()
    sc2, wd2 = mk.oob_chunk_scene()
    for name, s_ in (("material", sc), ("chunk", sc2)):
        for k, v in mk.scene_checksums(s_).items():
            assert np.array_equal(v, f[f"{name}_{k}"]), f"{name}: {k} differs — the fixture was made from another scene"
    scenes_ = {"material": (orc.from_package_scene(sc), sc.size),
               "chunk": (orc.OracleScene(sc2.world.nodes(), sc2.world.chunk_roots(), sc2.materials, sc2.cam, sc2.settings, wd2), sc2.size)}
    for name, (o, (w, h)) in scenes_.items():
        rgb, ids, steps, _ = o.render(orc.MODE_PRIMARY, w, h, want_steps=True)
        hit = f[f"{name}_clamp_hit"].astype(bool)
        assert np.array_equal((ids & orc.ID_HIT) != 0, hit), f"{name}: hit flags differ from the shader's under the clamp policy"
        assert np.array_equal(np.where(hit, ids & orc.ID_VOXEL_MASK, 0), np.where(hit, f[f"{name}_clamp_voxel"] & orc.ID_VOXEL_MASK, 0))
        assert np.array_equal(steps, f[f"{name}_clamp_iters"]), f"{name}: iteration counts differ from the shader's under the clamp policy"
        assert float(np.abs(rgb - f[f"{name}_clamp_rgb"]).max()) <= 1e-6
        differ = np.abs(f[f"{name}_clamp_rgb"] - f[f"{name}_zero_rgb"]).max(axis=-1) > 0
        assert differ.any(), f"{name}: the fixture cannot tell the two policies apart"
    assert {255, 256, 300, 1000, 32767} <= set(f["material_clamp_voxel"].reshape(-1).tolist())
