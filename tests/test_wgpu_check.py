"""tools/wgpu_check (the one-command cross-check of the reference-shader fixtures against a real wgpu run) on the CPU: the scene
dumps tests/golden/export_scenes.py writes round-trip into the checksums every fixture holds, the expectation files into the
fixtures' arrays, the committed dumps are what the exporter writes today, and the harness's struct sizes are the header's."""
import ctypes as C
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import export_scenes as X   # noqa: E402
from voxelraytracing_amd import _ffi   # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
CRATE = os.path.join(ROOT, "tools", "wgpu_check")


@pytest.mark.parametrize("case", X.all_cases())
def test_a_scene_dump_round_trips_into_its_fixtures_checksums(tmp_path, case):
    paths = X.export(case, str(tmp_path))
    d = X.read_scene(paths[0])
    got = X.scene_checksums_of_dump(d)
    if case in X.OOB_CASES:
        z, prefix = np.load(os.path.join(GOLDEN, "wgsl_oob.npz")), case[len("oob_"):] + "_"
    else:
        z, prefix = np.load(os.path.join(GOLDEN, f"wgsl_{case}.npz")), ""
    assert np.array_equal(got["cam_bytes"], z[prefix + "cam_bytes"]) and np.array_equal(got["settings_bytes"], z[prefix + "settings_bytes"])
    assert int(got["nodes_crc"]) == int(z[prefix + "nodes_crc"][0]) and int(got["roots_crc"]) == int(z[prefix + "roots_crc"][0])
    assert d["width"] % 8 == 0 and d["height"] % 8 == 0 and len(d["materials"]) == 256 * X.MATERIAL_BYTES
    # the expectations: the fixture's arrays, bit for bit (NaN positions included)
    for p in paths[1:]:
        e = X.read_expect(p)
        tag = os.path.basename(p).split(".")[-2]
        assert e["shader_crc"] == int(z["shader_crc"][0]) and (e["width"], e["height"]) == (d["width"], d["height"])
        for k in X.FIELDS:
            name = k if tag == "wgsl" else f"{prefix}{tag}_{k}"
            if name in z.files:
                a = np.ascontiguousarray(z[name], dtype=X.FIELD_DTYPE[k])
                assert a.tobytes() == np.ascontiguousarray(e[k]).tobytes(), (case, tag, k)
            else:
                assert k not in e


def test_the_committed_dumps_are_what_the_exporter_writes(tmp_path):
    committed = sorted(f for f in os.listdir(os.path.join(CRATE, "scenes")) if f.endswith((".vrtscene", ".vrtexpect")))
    assert committed, "tools/wgpu_check/scenes holds the small cases"
    for case in sorted({f.split(".")[0] for f in committed}):
        for p in X.export(case, str(tmp_path)):
            name = os.path.basename(p)
            assert name in committed and open(p, "rb").read() == open(os.path.join(CRATE, "scenes", name), "rb").read(), name


def test_the_harness_declares_the_headers_struct_sizes_and_the_dumps_layout():
    src = open(os.path.join(CRATE, "src", "scene.rs")).read()
    consts = {m.group(1): int(m.group(2)) for m in re.finditer(r"pub const (\w+): usize = (\d+);", src)}
    assert consts == {"SCENE_HEADER_BYTES": X.HEADER_BYTES, "CAM_DATA_BYTES": C.sizeof(_ffi.CamData), "SETTINGS_BYTES": C.sizeof(_ffi.Settings),
                      "WORLD_DATA_BYTES": C.sizeof(_ffi.WorldData), "MATERIAL_BYTES": C.sizeof(_ffi.Material), "EXPECT_HEADER_BYTES": 24}
    assert (X.CAM_BYTES, X.SETTINGS_BYTES, X.WORLD_BYTES, X.MATERIAL_BYTES) == (160, 48, 32, 32)
    assert f'b"{X.SCENE_MAGIC.decode()}"' in src and f'b"{X.EXPECT_MAGIC.decode()}"' in src
    # the debug record the instrumented shader writes and main.rs reads: four scalars, two vec4
    patch = open(os.path.join(CRATE, "src", "patch.rs")).read()
    assert "DEBUG_RECORD_BYTES: usize = 48" in patch and "hit: u32, voxel: u32, iters: u32, water_dist: f32, norm: vec4<f32>, pos: vec4<f32>" in patch
    main = open(os.path.join(CRATE, "src", "main.rs")).read()
    assert all(f"rec_u32(r, {o})" in main for o in (0, 4, 8)) and "f(12)" in main and "f(16 + 4 * k)" in main and "f(32 + 4 * k)" in main
    # the bindings of PixelShader::new (shader.rs:309-321) and one more for the records
    gpu = open(os.path.join(CRATE, "src", "gpu.rs")).read()
    assert [int(b) for b in re.findall(r"entry\((\d), ", gpu)] == [0, 1, 2, 3, 5, 6, 7]
    # nothing of the reference's shader is stored in the crate: it is read from the maintainer's checkout
    for f in os.listdir(os.path.join(CRATE, "src")):
        text = open(os.path.join(CRATE, "src", f)).read()
        assert "fn find_chunk_node" not in text and "fn ray_world" not in text and "fn ray_sky" not in text
