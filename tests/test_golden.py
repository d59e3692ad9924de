"""Golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py): the oracle must keep
reproducing them bit for bit; the GPU must match them to the parity bar."""
import os

import numpy as np
import pytest

from voxelraytracing_amd import MODE_PATH, MODE_PRIMARY, MODE_PRIMARY_SHADOW, scenes

from util import assert_frame_parity, gpu_for_scene

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = {"c1_64": lambda: scenes.c1_flat((64, 64)), "c1_256": lambda: scenes.c1_flat((256, 256)),
         "c2_128x72": lambda: scenes.c2((128, 72)), "c4_128x72": lambda: scenes.c4((128, 72), bounces=4)}
PATH_SPP, PATH_SEED = 2, 11


def _modes(name, primary, shadow, path):
    return ((path, "path"),) if name.startswith("c4") else ((primary, "primary"), (shadow, "shadow"))


def _load(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


def _crc(sc):
    n = sc.world.nodes()
    return np.bitwise_xor.reduce(n.astype(np.uint64) * np.arange(1, n.size + 1, dtype=np.uint64))


@pytest.mark.parametrize("name", list(CASES))
def test_scene_inputs_are_the_ones_the_vectors_were_made_from(name):
    g, sc = _load(name), CASES[name]()
    assert np.array_equal(np.frombuffer(bytes(sc.cam), dtype=np.uint8), g["cam_bytes"])
    assert _crc(sc) == g["nodes_crc"][0]


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_reproduces_golden(name, orc):
    g, sc = _load(name), CASES[name]()
    o = orc.from_package_scene(sc)
    for mode, tag in _modes(name, orc.MODE_PRIMARY, orc.MODE_PRIMARY_SHADOW, orc.MODE_PATH):
        rgb, ids, steps, st = o.render(mode, *sc.size, want_steps=True, spp=PATH_SPP, seed=PATH_SEED)
        assert np.array_equal(ids, g[f"{tag}_ids"]) and np.array_equal(steps, g[f"{tag}_steps"])
        assert np.array_equal(rgb, g[f"{tag}_rgb"])
        assert [st.primary_rays, st.secondary_rays, st.hits, st.steps, st.node_visits, st.primary_steps,
                st.primary_node_visits] == g[f"{tag}_stats"].tolist()


def test_oracle_is_thread_count_independent(orc):
    sc = CASES["c2_128x72"]()
    o = orc.from_package_scene(sc)
    a = o.render(orc.MODE_PRIMARY_SHADOW, *sc.size, threads=1)
    b = o.render(orc.MODE_PRIMARY_SHADOW, *sc.size, threads=4)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[3].steps == b[3].steps
    # rendering a rectangle gives the same pixels as the full frame
    c = o.render(orc.MODE_PRIMARY_SHADOW, *sc.size, rect=(16, 8, 80, 40))
    assert np.array_equal(c[1][8:40, 16:80], a[1][8:40, 16:80]) and not c[1][:8].any()


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [0, 1, 2, 3])
@pytest.mark.parametrize("name", list(CASES))
def test_gpu_matches_golden(name, variant):
    g, sc = _load(name), CASES[name]()
    gpu = gpu_for_scene(sc)
    if name.startswith("c4") and variant:
        pytest.skip("the path trace has one kernel variant")
    for mode, tag in _modes(name, MODE_PRIMARY, MODE_PRIMARY_SHADOW, MODE_PATH):
        gpu.render(mode, variant=variant, stats=True, spp=PATH_SPP, seed=PATH_SEED)
        rgb, ids, _ = gpu.read_output()
        assert_frame_parity(rgb, ids, g[f"{tag}_rgb"], g[f"{tag}_ids"], f"{name} {tag}")
        assert np.array_equal(gpu.read_steps(), g[f"{tag}_steps"])
        s = gpu.stats()
        assert [s.primary_rays, s.secondary_rays, s.hits, s.steps, s.node_visits, s.primary_steps,
                s.primary_node_visits] == g[f"{tag}_stats"].tolist()
