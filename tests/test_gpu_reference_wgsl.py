"""The HIP path against the REFERENCE'S OWN SHADER TEXT: tests/golden/wgsl_*.npz hold what `ray_tracer.wgsl`, executed as it
stands by tests/wgsl_interp.py, computes per pixel (tests/golden/make_wgsl_fixtures.py; tests/test_oracle_vs_reference_wgsl.py
holds the oracle to the same files).  Through the C ABI, for every march of the backend: the id words and the per-pixel
iteration counts bit for bit, radiance within north_star's 1e-4."""
import os
import sys

import numpy as np
import pytest

from voxelraytracing_amd import MODE_PRIMARY

from util import RADIANCE_TOL, gpu_for_scene

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_wgsl_fixtures as mk   # noqa: E402

pytestmark = pytest.mark.gpu
ID_VOXEL_MASK, ID_HIT, ID_NX, ID_NY, ID_NZ, ID_WATER = 0x7FFF, 1 << 16, 1 << 17, 1 << 18, 1 << 19, 1 << 20


@pytest.mark.parametrize("variant", [0, 1, 2, 3])
@pytest.mark.parametrize("case", mk.CASES)
def test_kernels_compute_what_the_reference_shader_computes(case, variant):
    f = np.load(os.path.join(HERE, "golden", f"wgsl_{case}.npz"))
    sc, (x0, y0, x1, y1) = mk.case_scene(case)
    for k, v in mk.scene_checksums(sc).items():
        assert np.array_equal(v, f[k]), f"{case}: {k} differs — the fixture was made from another scene"
    gpu = gpu_for_scene(sc)
    gpu.render(MODE_PRIMARY, variant=variant, stats=True)
    rgb, ids, _ = gpu.read_output()
    steps = gpu.read_steps()
    rgb, ids, steps = rgb[y0:y1, x0:x1], ids[y0:y1, x0:x1], steps[y0:y1, x0:x1] & 0xFFFF
    hit = f["hit"].astype(bool)
    want = np.where(hit, (f["voxel"] & ID_VOXEL_MASK) | ID_HIT, 0).astype(np.uint32)
    for axis, bit in enumerate((ID_NX, ID_NY, ID_NZ)):
        want |= np.where(f["norm"][..., axis] != 0.0, bit, 0).astype(np.uint32)
    want |= np.where(f["water_dist"] != 0.0, ID_WATER, 0).astype(np.uint32)
    bad = np.argwhere(ids != want)
    assert bad.size == 0, f"{case} variant {variant}: {len(bad)} id words differ from the shader's, first at {tuple(bad[0])}"
    assert np.array_equal(steps, f["iters"]), f"{case} variant {variant}: iteration counts differ from the shader's"
    assert float(np.nanmax(np.abs(rgb - f["rgb"]))) <= RADIANCE_TOL
    gpu.close()


@pytest.mark.parametrize("variant", [0, 1, 2, 3])
def test_material_ids_past_the_table_follow_the_pinned_clamp_policy(variant):
    """Voxel ids >= 256 index voxel_mats past its 256 entries (ray_tracer.wgsl:226): tests/golden/wgsl_oob.npz holds the
    reference's shader text under a clamped index and under a zero value.  Every march of the backend gives the CLAMPED
    frame (material 255), as the oracle does: the pinned choice (undecided until a real wgpu run of the scene — tools/wgpu_check —
    says which policy the reference's backend has; the zero-valued frame is in the fixture for that day)."""
    f = np.load(os.path.join(HERE, "golden", "wgsl_oob.npz"))
    sc = mk.oob_material_scene()
    for k, v in mk.scene_checksums(sc).items():
        assert np.array_equal(v, f[f"material_{k}"])
    gpu = gpu_for_scene(sc)
    gpu.render(MODE_PRIMARY, variant=variant, stats=True)
    rgb, ids, _ = gpu.read_output()
    steps = gpu.read_steps() & 0xFFFF
    hit = f["material_clamp_hit"].astype(bool)
    assert np.array_equal((ids & ID_HIT) != 0, hit)
    assert np.array_equal(np.where(hit, ids & ID_VOXEL_MASK, 0), np.where(hit, f["material_clamp_voxel"] & ID_VOXEL_MASK, 0))
    assert np.array_equal(steps, f["material_clamp_iters"])
    assert float(np.abs(rgb - f["material_clamp_rgb"]).max()) <= RADIANCE_TOL
    assert (np.abs(f["material_clamp_rgb"] - f["material_zero_rgb"]).max(axis=-1) > 0).any()   # (the fixture can tell the policies apart)
    gpu.close()


def test_a_world_data_that_indexes_past_chunk_roots_is_refused():
    """The other read that can leave its array — chunk_roots_[idx] under a WorldData whose size is not 32 * size_in_chunks
    (tests/golden/wgsl_oob.npz, `chunk`) — cannot happen behind the C ABI: vrt_render refuses such a frame (VRT_ERR_STATE)."""
    from voxelraytracing_amd.graphics import VrtError
    sc, wd = mk.oob_chunk_scene()
    gpu = gpu_for_scene(sc)
    with pytest.raises(VrtError) as e:
        gpu.write_world_data(wd)
        gpu.render(MODE_PRIMARY)
    assert "size" in str(e.value)
    gpu.close()
