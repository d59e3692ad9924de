"""The benchmark / parity scenes of BASELINE.json (SURVEY.md §8d "Synthetic inputs").

Every scene is (ClientWorld, CamData, Settings, materials, (W, H), mode); worlds come from the host
library's deterministic generator, cameras from CamData::create.  Nothing here touches the oracle.
"""
from __future__ import annotations

from dataclasses import dataclass

from . import graphics as g
from .world import ClientWorld, gen_height

SUN_POS = (10000.0, 20000.0, 5000.0)  # the reference default (0,0,0) is degenerate (main.rs:152-156)


@dataclass
class Scene:
    name: str
    world: ClientWorld
    cam: g.CamData
    settings: g.Settings
    materials: object
    size: tuple
    mode: int
    eye: tuple
    rot: tuple


def _world(size_chunks: int, kind: int, seed: int, max_nodes: int) -> ClientWorld:
    h = size_chunks // 2
    w = ClientWorld((h, h, h), max_nodes, size_chunks)  # min chunk = (0,0,0) -> world.min = (0,0,0)
    w.generate(kind, seed)
    return w


def _scene(name, world, size, eye, rot, mode, fov=70.0, **settings) -> Scene:
    cam = g.cam_data_create(rot, eye, fov, (float(size[0]), float(size[1])))
    st = g.make_settings(sun_pos=SUN_POS, **settings)
    return Scene(name, world, cam, st, g.std_materials(), size, mode, eye, rot)


def c1_flat(size=(256, 256)) -> Scene:
    """C1: 256x256, 2x2x2-chunk Superflat SVO built by set_node, primary rays only."""
    world = _world(2, 1, 0, 1 << 18)
    return _scene("C1 256x256 2^3 superflat primary", world, size, (32.5, 20.5, 60.5), (15.0, 0.0, 0.0), g.MODE_PRIMARY)


def procedural(size_chunks: int, size=(1920, 1080), mode=g.MODE_PRIMARY_SHADOW, seed=1, name=None) -> Scene:
    """C2/C3/C4/C5 family: S^3-chunk procedural world, eye above the terrain at the world centre."""
    per_chunk_budget = {8: 1 << 23, 16: 1 << 25, 32: 1 << 27}.get(size_chunks, 1 << 23)
    world = _world(size_chunks, 0, seed, per_chunk_budget)
    c = size_chunks * 16
    eye = (c + 0.5, float(gen_height(seed, c, c) + 24) + 0.5, c + 0.5)
    return _scene(name or f"{size[0]}x{size[1]} {size_chunks}^3 procedural", world, size, eye, (20.0, 35.0, 0.0), mode)


def c2(size=(1920, 1080)) -> Scene:
    """C2: 1920x1080, 8x8x8-chunk procedural world, primary + 1 shadow ray, 1 GPU — the headline config."""
    return procedural(8, size, g.MODE_PRIMARY_SHADOW, name=f"C2 {size[0]}x{size[1]} 8^3 procedural primary+shadow")


def c3(size=(1920, 1080)) -> Scene:
    return procedural(16, size, g.MODE_PRIMARY_SHADOW, name=f"C3 {size[0]}x{size[1]} 16^3 procedural primary+shadow")


def _diffuse(materials):
    """All solids fully scattering (Material.scatter = 1): the 'diffuse path trace' of configs C4/C5.
    Material::construct leaves scatter at 0.0 (graphics/mod.rs:44), which would make every voxel a mirror."""
    for i in range(256):
        materials[i].scatter = 1.0
    return materials


def c4(size=(1920, 1080), bounces=4) -> Scene:
    """C4: 1920x1080, 8^3-chunk world, 4-bounce diffuse path trace (1 spp), 1 GPU."""
    sc = procedural(8, size, g.MODE_PATH, name=f"C4 {size[0]}x{size[1]} 8^3 procedural {bounces}-bounce diffuse path trace")
    sc.settings.max_ray_bounces = bounces
    _diffuse(sc.materials)
    return sc


def c5(size=(3840, 2160), bounces=4, chunks=32) -> Scene:
    """C5: 3840x2160, 16 spp, 32^3-chunk world, sharded over 8 GPUs (spp is a render option)."""
    sc = procedural(chunks, size, g.MODE_PATH, name=f"C5 {size[0]}x{size[1]} {chunks}^3 procedural {bounces}-bounce path trace")
    sc.settings.max_ray_bounces = bounces
    _diffuse(sc.materials)
    return sc
