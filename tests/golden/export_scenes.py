"""tests/golden/export_scenes.py [out_dir] [case ...] — flat dumps of the reference-shader fixtures' scenes and expected outputs, for
tools/wgpu_check (the one-command cross-check of tests/golden/wgsl_*.npz against a REAL wgpu / naga run of the reference's shader).

The fixtures were made by executing `clientdesktop/src/graphics/ray_tracer.wgsl` through tests/wgsl_interp.py — an interpreter
written for this repo, so they pin the oracle to the shader's TEXT but not to the reference's TOOLCHAIN.  A
maintainer with cargo + a GPU settles that with one run: this script writes, per case, what the Rust harness uploads and what it
compares against, in a format that needs no crate to read:

  <case>.vrtscene  (little endian)
      0    char[8]  "VRTSCN01"
      8    u32      width, height          result texture; the harness dispatches width / 8 x height / 8 workgroups (main.rs:452)
      16   u32      max_nodes              NodeBuffer capacity in nodes, even (shader.rs:9-16): the harness allocates all of it
      20   u32      n_node_words           u32 words of the pool that follow: up to its last non-zero word (the rest is zero)
      24   u32      n_roots                chunk_roots entries (S^3)
      28   u32      n_materials            256
      32   u8[160]  CamData                (graphics/mod.rs:82-91)
      192  u8[48]   Settings               (mod.rs:132-143)
      240  u8[32]   WorldData              (mod.rs:113-120)
      272  u8[32 * n_materials]            Material table (mod.rs:20-28)
      ...  u32[n_roots]                    chunk_roots
      ...  u32[n_node_words]               the node pool as NodeBuffer::write packs it: two nodes per u32 (shader.rs:22-40)
  <case>.<tag>.vrtexpect
      0    char[8]  "VRTEXP01"
      8    u32      width, height
      16   u32      fields                 bit 0 rgb f32x3, 1 hit u8 (rows padded to 4 bytes in total), 2 voxel u32, 3 iters u32,
                                           4 norm f32x3, 5 water_dist f32, 6 pos f32x3 — the arrays follow in that order, row-major
      20   u32      shader_crc             zlib.crc32 of the shader file the fixture was made from: the harness refuses another text

`tag` is "wgsl" for the thirteen cases, "clamp" / "zero" for the two out-of-range scenes of wgsl_oob.npz (whichever the run
matches is what the reference's backend does with an index past the end: oracle and kernels implement clamp).

Nothing here reads /root/reference: the scenes are rebuilt from this repo's host mirror (the same code that made the fixtures) and
checked against the checksums the fixtures hold; the expected arrays are copied out of the committed .npz files.
"""
import os
import struct
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, ROOT)

SCENE_MAGIC, EXPECT_MAGIC = b"VRTSCN01", b"VRTEXP01"
HEADER_BYTES, CAM_BYTES, SETTINGS_BYTES, WORLD_BYTES, MATERIAL_BYTES = 32, 160, 48, 32, 32
FIELDS = ("rgb", "hit", "voxel", "iters", "norm", "water_dist", "pos")
FIELD_DTYPE = {"rgb": np.float32, "hit": np.uint8, "voxel": np.uint32, "iters": np.uint32, "norm": np.float32, "water_dist": np.float32, "pos": np.float32}
OOB_CASES = ("oob_material", "oob_chunk")


def scene_for(case):
    """(scene, WorldData to upload) of a fixture case — rebuilt by the code that made the fixture."""
    import make_wgsl_fixtures as M
    if case == "oob_material":
        sc = M.oob_material_scene()
        return sc, sc.world.world_data()
    if case == "oob_chunk":
        return M.oob_chunk_scene()
    sc, _ = M.case_scene(case)
    return sc, sc.world.world_data()


def all_cases():
    import make_wgsl_fixtures as M
    return list(M.CASES) + list(OOB_CASES)


def write_scene(path, sc, world_data):
    nodes = np.ascontiguousarray(sc.world.nodes(), dtype=np.uint16)
    max_nodes = int(nodes.size) & ~1
    words = nodes[:max_nodes].view("<u4")
    used = int(np.flatnonzero(words)[-1]) + 1 if words.any() else 0
    roots = np.ascontiguousarray(sc.world.chunk_roots(), dtype=np.uint32)
    mats = b"".join(bytes(m) for m in sc.materials)
    assert len(bytes(sc.cam)) == CAM_BYTES and len(bytes(sc.settings)) == SETTINGS_BYTES and len(bytes(world_data)) == WORLD_BYTES
    assert len(mats) == MATERIAL_BYTES * len(sc.materials)
    with open(path, "wb") as f:
        f.write(SCENE_MAGIC)
        f.write(struct.pack("<6I", sc.size[0], sc.size[1], max_nodes, used, roots.size, len(sc.materials)))
        f.write(bytes(sc.cam))
        f.write(bytes(sc.settings))
        f.write(bytes(world_data))
        f.write(mats)
        f.write(roots.astype("<u4").tobytes())
        f.write(words[:used].astype("<u4").tobytes())


def read_scene(path):
    """The dump as a dict (what tools/wgpu_check/src/scene.rs parses): header fields, the three uniform blobs, materials, roots and the
    node pool padded back to max_nodes."""
    b = open(path, "rb").read()
    assert b[:8] == SCENE_MAGIC, "not a scene dump"
    w, h, max_nodes, used, n_roots, n_mats = struct.unpack_from("<6I", b, 8)
    at = HEADER_BYTES
    cam, at = b[at:at + CAM_BYTES], at + CAM_BYTES
    settings, at = b[at:at + SETTINGS_BYTES], at + SETTINGS_BYTES
    world, at = b[at:at + WORLD_BYTES], at + WORLD_BYTES
    mats, at = b[at:at + MATERIAL_BYTES * n_mats], at + MATERIAL_BYTES * n_mats
    roots = np.frombuffer(b, dtype="<u4", count=n_roots, offset=at)
    at += 4 * n_roots
    words = np.frombuffer(b, dtype="<u4", count=used, offset=at)
    assert at + 4 * used == len(b), "trailing bytes"
    pool = np.zeros(max_nodes // 2, dtype="<u4")
    pool[:used] = words
    return dict(width=w, height=h, max_nodes=max_nodes, n_node_words=used, cam=cam, settings=settings, world=world, materials=mats, roots=roots,
                nodes=pool.view("<u2"))


def scene_checksums_of_dump(d):
    """The checksums tests/golden/make_wgsl_fixtures.py: scene_checksums stored with every fixture, from a dump."""
    return dict(cam_bytes=np.frombuffer(d["cam"], dtype=np.uint8), settings_bytes=np.frombuffer(d["settings"], dtype=np.uint8),
                nodes_crc=np.uint32(zlib.crc32(d["nodes"].tobytes())), roots_crc=np.uint32(zlib.crc32(d["roots"].astype("<u4").tobytes())))


def write_expect(path, size, arrays, shader_crc):
    w, h = size
    fields = sum(1 << i for i, k in enumerate(FIELDS) if k in arrays)
    with open(path, "wb") as f:
        f.write(EXPECT_MAGIC)
        f.write(struct.pack("<4I", w, h, fields, int(shader_crc)))
        for k in FIELDS:
            if k not in arrays:
                continue
            a = np.ascontiguousarray(arrays[k], dtype=FIELD_DTYPE[k])
            assert a.shape[:2] == (h, w), (k, a.shape)
            raw = a.astype(a.dtype.newbyteorder("<")).tobytes()
            f.write(raw + b"\0" * (-len(raw) % 4))


def read_expect(path):
    b = open(path, "rb").read()
    assert b[:8] == EXPECT_MAGIC
    w, h, fields, crc = struct.unpack_from("<4I", b, 8)
    at, out = 24, {}
    for i, k in enumerate(FIELDS):
        if not fields >> i & 1:
            continue
        per = 3 if k in ("rgb", "norm", "pos") else 1
        n = w * h * per
        out[k] = np.frombuffer(b, dtype=np.dtype(FIELD_DTYPE[k]).newbyteorder("<"), count=n, offset=at).reshape((h, w, 3) if per == 3 else (h, w))
        at += n * np.dtype(FIELD_DTYPE[k]).itemsize
        at += -at % 4
    assert at == len(b)
    return dict(width=w, height=h, shader_crc=crc, **out)


def export(case, out_dir):
    """Write <case>.vrtscene and its .vrtexpect file(s); returns the paths."""
    sc, wd = scene_for(case)
    paths = [os.path.join(out_dir, f"{case}.vrtscene")]
    write_scene(paths[0], sc, wd)
    if case in OOB_CASES:
        z = np.load(os.path.join(HERE, "wgsl_oob.npz"))
        name = case[len("oob_"):]
        for pol in ("clamp", "zero"):
            p = os.path.join(out_dir, f"{case}.{pol}.vrtexpect")
            write_expect(p, sc.size, {k: z[f"{name}_{pol}_{k}"] for k in ("rgb", "hit", "voxel", "iters")}, z["shader_crc"][0])
            paths.append(p)
    else:
        z = np.load(os.path.join(HERE, f"wgsl_{case}.npz"))
        x0, y0, x1, y1 = (int(v) for v in z["window"])
        assert (x0, y0, x1, y1) == (0, 0, sc.size[0], sc.size[1]), "the fixtures trace whole frames"
        p = os.path.join(out_dir, f"{case}.wgsl.vrtexpect")
        write_expect(p, sc.size, {k: z[k] for k in FIELDS}, z["shader_crc"][0])
        paths.append(p)
    return paths


if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tools", "wgpu_check", "scenes")
    os.makedirs(out, exist_ok=True)
    total = 0
    for c in (sys.argv[2:] or all_cases()):
        ps = export(c, out)
        n = sum(os.path.getsize(p) for p in ps)
        total += n
        print(f"{c}: {', '.join(os.path.basename(p) for p in ps)} ({n / 1e6:.2f} MB)")
    print(f"{total / 1e6:.1f} MB in {out}")
