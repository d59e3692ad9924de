"""Shared helpers for the parity tests."""
import numpy as np

from voxelraytracing_amd import Gpu

RADIANCE_TOL = 1e-4  # BASELINE.json north_star: "within 1e-4 on float radiance"


def gpu_for_scene(scene, size=None, shard_rank=0, shard_count=1, **kw) -> Gpu:
    size = size or scene.size
    g = Gpu(scene.world.max_nodes(), scene.world.size_in_chunks(), size, shard_rank=shard_rank, shard_count=shard_count, **kw)
    g.upload_world(scene.world, scene.materials)
    g.write_cam_data(scene.cam)
    g.write_settings(scene.settings)
    return g


def assert_frame_parity(gpu_rgb, gpu_ids, ref_rgb, ref_ids, what=""):
    bad = np.argwhere(gpu_ids != ref_ids)
    assert bad.size == 0, f"{what}: {len(bad)} id words differ, first at (y,x)={tuple(bad[0])}: " \
                          f"gpu={gpu_ids[tuple(bad[0])]:#x} oracle={ref_ids[tuple(bad[0])]:#x}"
    err = np.abs(gpu_rgb - ref_rgb)
    assert np.isfinite(gpu_rgb).all() == np.isfinite(ref_rgb).all()
    assert float(np.nanmax(err)) <= RADIANCE_TOL, f"{what}: max radiance error {np.nanmax(err)}"
