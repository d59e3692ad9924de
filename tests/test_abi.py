"""The C-ABI libraries load without a GPU and export exactly what include/*.h declares."""
import ctypes as C
import os
import re

import pytest

from voxelraytracing_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header, prefix):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"\w+)\s*\(", src)))


def test_vrt_exports_every_declared_symbol():
    lib = _ffi.vrt()
    names = _declared("vrt.h", "vrt_")
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_ffi.VRT_SYMBOLS) == names, "bindings drifted from include/vrt.h"


def test_vrt_host_exports_every_declared_symbol():
    lib = _ffi.host()
    names = _declared("vrt_host.h", "vrth_")
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_ffi.VRTH_SYMBOLS) == names, "bindings drifted from include/vrt_host.h"


def test_uniform_struct_layouts_match_the_reference():
    # clientdesktop/src/graphics/mod.rs:20-28, 82-91, 113-120, 132-143
    assert C.sizeof(_ffi.Material) == 32 and _ffi.Material.is_liquid.offset == 16 and _ffi.Material.scatter.offset == 20
    assert C.sizeof(_ffi.CamData) == 160
    assert (_ffi.CamData.inv_view_mat.offset, _ffi.CamData.inv_proj_mat.offset, _ffi.CamData.proj_size.offset) == (16, 80, 144)
    assert C.sizeof(_ffi.WorldData) == 32 and _ffi.WorldData.size.offset == 12 and _ffi.WorldData.size_in_chunks.offset == 16
    assert C.sizeof(_ffi.Settings) == 48
    assert (_ffi.Settings.sun_intensity.offset, _ffi.Settings.show_step_count.offset, _ffi.Settings.sky_color.offset,
            _ffi.Settings.sun_pos.offset) == (4, 8, 16, 32)


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    monkeypatch.setattr(_ffi, "_HERE", str(tmp_path))
    with pytest.raises(ImportError, match="no fallback"):
        _ffi._load("libvrt.so", _ffi.VRT_SYMBOLS)


def test_create_validates_arguments_without_touching_the_gpu():
    lib = _ffi.vrt()
    h = C.c_void_p()
    for cfg in (_ffi.Config(1024, 2, 60, 64, -1, 0, 1, 0),   # width not a multiple of 8 (main.rs:452)
                _ffi.Config(1, 2, 64, 64, -1, 0, 1, 0),      # max_nodes < 2
                _ffi.Config(1024, 2, 64, 64, -1, 3, 2, 0)):  # shard_rank >= shard_count
        assert lib.vrt_create(C.byref(cfg), C.byref(h)) == -1
        assert lib.vrt_last_error(None)
    assert lib.vrt_create(None, C.byref(h)) == -1
