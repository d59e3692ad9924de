"""SURVEY.md §8f N3 on the GPU path: a world that went through the reference's on-disk and wire formats — region files
(servercli/src/main.rs:25-73) and `GiveChunkData` messages (common/src/net.rs:46-55) — into a fresh ClientWorld is traced
by the HIP backend and matches the oracle's frame of the world it came from, pixel for pixel."""
import os

import numpy as np
import pytest

from voxelraytracing_amd import MODE_PRIMARY_SHADOW, scenes
from voxelraytracing_amd import world as W
from voxelraytracing_amd.world import ClientWorld

from util import assert_frame_parity, gpu_for_scene

pytestmark = pytest.mark.gpu


def test_world_saved_to_region_files_and_loaded_back_traces_like_the_original(tmp_path, orc):
    sc = scenes.c3((640, 360))                       # 16^3 chunks: exactly region (0,0,0) (REGION_SIZE = 16, mod.rs:25)
    assert sc.world.size_in_chunks() == 16 and sc.world.min_voxel() == (0, 0, 0)
    r_rgb, r_ids, r_steps, st = orc.from_package_scene(sc).render(orc.MODE_PRIMARY_SHADOW, *sc.size, want_steps=True)
    # WorldFs::save writes regions/r_X_Y_Z_.data (servercli/src/main.rs:25-27, 106-133)
    name = W.region_file_name((0, 0, 0))
    path = tmp_path / name
    os.makedirs(path.parent, exist_ok=True)
    path.write_bytes(sc.world.save_region((0, 0, 0)))
    assert name == "regions/r_0_0_0_.data" and path.stat().st_size > 1 << 20
    # a client joins: fresh world, chunks from the file, whole pool uploaded (join_game, main.rs:199-223)
    loaded = ClientWorld((8, 8, 8), sc.world.max_nodes(), 16)
    assert loaded.load_region(path.read_bytes(), (0, 0, 0)) == sc.world.populated_count()
    import copy
    sc2 = copy.copy(sc)
    sc2.world = loaded
    gpu = gpu_for_scene(sc2)
    gpu.render(MODE_PRIMARY_SHADOW, stats=True)
    rgb, ids, _ = gpu.read_output()
    assert_frame_parity(rgb, ids, r_rgb, r_ids, "world loaded from a region file")
    assert np.array_equal(gpu.read_steps(), r_steps) and gpu.stats().steps == st.steps
    gpu.close()


def test_world_streamed_in_as_chunk_messages_traces_like_the_original(orc):
    """The frame loop's ingest (main.rs:278-297): messages arrive in pieces, every complete one becomes create_chunk + a
    range upload + a chunk_roots rewrite — here 40 chunks per frame, rendering in between — and the final frame is the
    oracle's frame of the source world; the derived tables followed chunk by chunk."""
    sc = scenes.c2((640, 360))
    src = sc.world
    S = src.size_in_chunks()
    have = [(x, y, z) for z in range(S) for y in range(S) for x in range(S) if src.chunk_roots().reshape(S, S, S)[z, y, x]]
    rng = np.random.default_rng(9)
    rng.shuffle(have)
    stream = b"".join(src.encode_chunk_msg(p) for p in have)
    dst = ClientWorld((S // 2,) * 3, src.max_nodes(), S)
    import copy
    sc2 = copy.copy(sc)
    sc2.world = dst
    gpu = gpu_for_scene(sc2)                         # an empty world: every chunk missing = air
    gpu.render(MODE_PRIMARY_SHADOW)
    assert not (gpu.read_output()[1] & 0x10000).any()
    buf, off, got_chunks, frames = b"", 0, 0, 0
    cuts = sorted(set(int(c) for c in rng.integers(0, len(stream), size=60))) + [len(stream)]
    for cut in cuts:                                 # the socket hands over arbitrary pieces (client/src/net.rs:44-60)
        buf += stream[off:cut]
        off = cut
        while True:
            got = dst.ingest_chunk_msg(buf)
            if got is None:
                break
            used, pos, root, n = got
            buf = buf[used:]
            gpu.write_nodes(dst.nodes_ptr(), root, root + n)        # main.rs:289-295
            got_chunks += 1
            if got_chunks % 40 == 0:
                gpu.write_chunk_roots(dst.chunk_roots())            # main.rs:446
                gpu.render(MODE_PRIMARY_SHADOW)
                frames += 1
    assert got_chunks == len(have) and buf == b""
    gpu.write_chunk_roots(dst.chunk_roots())
    gpu.render(MODE_PRIMARY_SHADOW)
    rgb, ids, _ = gpu.read_output()
    r_rgb, r_ids, _, _ = orc.from_package_scene(sc).render(orc.MODE_PRIMARY_SHADOW, *sc.size)
    assert_frame_parity(rgb, ids, r_rgb, r_ids, "world streamed in as GiveChunkData messages")
    ai = gpu.accel_info()
    assert ai.available and ai.chunk_builds >= 40 and frames >= 5
    gpu.close()
