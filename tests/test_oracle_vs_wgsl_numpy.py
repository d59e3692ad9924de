"""The C oracle against a second restatement of the live shader written separately, from the WGSL text, in vectorised
numpy (tests/wgsl_numpy.py): voxel ids, hit / normal / water flags and per-pixel iteration counts bit for bit, radiance to
1e-6 (numpy's float32 pow is not libm's powf).  Neither is the reference — nothing in this image runs WGSL — but they share
no code and no structure."""
import numpy as np
import pytest

from voxelraytracing_amd import graphics as g, scenes

import wgsl_numpy


def both(orc, sc, w, h):
    o = orc.from_package_scene(sc)
    r_rgb, r_ids, r_steps, _ = o.render(orc.MODE_PRIMARY, w, h, want_steps=True)
    wd = sc.world
    n_rgb, n_ids, n_it = wgsl_numpy.render_primary(wd.nodes(), wd.chunk_roots(), sc.materials, sc.cam, sc.settings, wd.world_data(), w, h)
    return (r_rgb, r_ids, r_steps & 0xFFFF), (n_rgb, n_ids, n_it)


def agree(a, b, what):
    bad = np.argwhere(a[1] != b[1])
    assert bad.size == 0, f"{what}: {len(bad)} id words differ, first at (y, x) = {tuple(bad[0])}: oracle {a[1][tuple(bad[0])]:#x}, numpy {b[1][tuple(bad[0])]:#x}"
    bad = np.argwhere(a[2] != b[2])
    assert bad.size == 0, f"{what}: {len(bad)} iteration counts differ, first at {tuple(bad[0])}: {a[2][tuple(bad[0])]} vs {b[2][tuple(bad[0])]}"
    assert float(np.abs(a[0] - b[0]).max()) <= 1e-6, what


def test_flat_world_c1(orc):
    sc = scenes.c1_flat((64, 64))
    agree(*both(orc, sc, 64, 64), "C1 64x64")


@pytest.mark.parametrize("size", [(96, 56), (75, 43)])
def test_procedural_world_c2(orc, size):
    sc = scenes.c2(size)
    a, b = both(orc, sc, *size)
    agree(a, b, f"C2 {size}")
    ids = a[1]
    assert (ids & orc.ID_HIT).any() and not (ids & orc.ID_HIT).all()          # terrain and sky
    assert (ids & orc.ID_WATER).any()                                           # and rays through water


def test_other_views_and_the_step_count_view(orc):
    sc = scenes.c2((64, 40))
    for rot, eye_dy in (((80.0, 10.0, 0.0), 0.0), ((-60.0, 200.0, 0.0), 6.0), ((0.0, 90.0, 0.0), -20.0), ((45.0, 45.0, 0.0), -30.0)):
        eye = (sc.eye[0], sc.eye[1] + eye_dy, sc.eye[2])
        sc.cam = g.cam_data_create(rot, eye, 70.0, (64.0, 40.0))
        agree(*both(orc, sc, 64, 40), f"rot {rot} eye {eye}")
    sc.settings = g.make_settings(sun_pos=scenes.SUN_POS, show_step_count=1)
    agree(*both(orc, sc, 64, 40), "step-count view")


def test_presentation(orc):
    """screen_shader.wgsl over the result texture: the oracle's orc_present against the numpy restatement, byte for byte —
    windows at 1:1, magnified, minified, squeezed; a texture that is not whole tiles; every crosshair style."""
    for size in ((64, 40), (75, 43)):
        sc = scenes.c2(size)
        rgb, _, _, _ = orc.from_package_scene(sc).render(orc.MODE_PRIMARY, *size)
        for kw in (dict(), dict(style=1, size=9.5, color=(1.0, 0.2, 0.1, 0.75)), dict(style=0)):
            for screen in (size, (2 * size[0] + 1, size[1] + 3), (size[0] // 2, size[1] // 2), (33, 7)):
                assert np.array_equal(orc.present(rgb, screen, **kw), wgsl_numpy.present(rgb, screen, **kw)), (size, screen, kw)
