"""ctypes bindings of the two in-tree native libraries.

* ``libvrt.so``       — include/vrt.h: the HIP backend (kernels + C ABI), the drop-in for the reference's
                        ``GpuResources``/``Buffers``/``PixelShader`` seam (clientdesktop/src/graphics).
* ``libvrt_host.so``  — include/vrt_host.h: the C++ host mirror of the reference's world / camera types.

Both are REQUIRED: there is no Python or CPU fallback.  A missing library raises ``ImportError`` with the
build command, so a GPU box can never silently run something else.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))


class Material(C.Structure):
    """clientdesktop/src/graphics/mod.rs:20-28"""
    _fields_ = [("color", C.c_float * 3), ("is_empty", C.c_uint32), ("is_liquid", C.c_uint32),
                ("scatter", C.c_float), ("_padding", C.c_uint32 * 2)]


class CamData(C.Structure):
    """clientdesktop/src/graphics/mod.rs:82-91"""
    _fields_ = [("pos", C.c_float * 3), ("_padding0", C.c_uint32), ("inv_view_mat", C.c_float * 16),
                ("inv_proj_mat", C.c_float * 16), ("proj_size", C.c_float * 2), ("_padding1", C.c_uint32 * 2)]


class WorldData(C.Structure):
    """clientdesktop/src/graphics/mod.rs:113-120"""
    _fields_ = [("min", C.c_int32 * 3), ("size", C.c_uint32), ("size_in_chunks", C.c_uint32),
                ("_padding", C.c_uint32 * 3)]


class Settings(C.Structure):
    """clientdesktop/src/graphics/mod.rs:132-143"""
    _fields_ = [("max_ray_bounces", C.c_uint32), ("sun_intensity", C.c_float), ("show_step_count", C.c_uint32),
                ("_padding0", C.c_uint32), ("sky_color", C.c_float * 3), ("_padding1", C.c_uint32),
                ("sun_pos", C.c_float * 3), ("_padding2", C.c_uint32)]


MAX_DEVICES = 16


class Config(C.Structure):
    _fields_ = [("max_nodes", C.c_uint32), ("world_size_chunks", C.c_uint32), ("width", C.c_uint32),
                ("height", C.c_uint32), ("device", C.c_int32), ("shard_rank", C.c_uint32),
                ("shard_count", C.c_uint32), ("flags", C.c_uint32), ("shard_root_weight", C.c_uint32),
                ("n_devices", C.c_uint32), ("device_ids", C.c_int32 * MAX_DEVICES)]


class RenderOpts(C.Structure):
    _fields_ = [("mode", C.c_uint32), ("variant", C.c_uint32), ("stats", C.c_uint32), ("spp", C.c_uint32),
                ("seed", C.c_uint32), ("flags", C.c_uint32), ("_reserved", C.c_uint32 * 2)]


class Stats(C.Structure):
    _fields_ = [("primary_rays", C.c_uint64), ("secondary_rays", C.c_uint64), ("hits", C.c_uint64),
                ("steps", C.c_uint64), ("node_visits", C.c_uint64), ("primary_steps", C.c_uint64),
                ("primary_node_visits", C.c_uint64), ("ms_total", C.c_float), ("ms_primary", C.c_float),
                ("ms_secondary", C.c_float), ("frames", C.c_uint32), ("sum_ms_primary", C.c_double),
                ("sum_ms_secondary", C.c_double), ("sum_ms_total", C.c_double),
                ("clock_shader_ticks", C.c_uint64), ("clock_ref_ticks", C.c_uint64)]


class Crosshair(C.Structure):
    """clientdesktop/src/graphics/mod.rs:63-80"""
    _fields_ = [("color", C.c_float * 4), ("style", C.c_uint32), ("size", C.c_float), ("_padding", C.c_uint32 * 2)]


class AccelInfo(C.Structure):
    _fields_ = [("available", C.c_uint32), ("world_size_chunks", C.c_uint32), ("cells", C.c_uint64),
                ("bricks", C.c_uint64), ("bytes", C.c_uint64), ("builds", C.c_uint32), ("last_build_ms", C.c_float),
                ("chunk_builds", C.c_uint32), ("ordered_frames", C.c_uint32)]


class IssueProfile(C.Structure):
    _fields_ = [("frames", C.c_uint32), ("devices", C.c_uint32), ("issuing_threads", C.c_uint32), ("_reserved", C.c_uint32),
                ("render_us", C.c_double), ("root_issue_us", C.c_double), ("shard_issue_us_mean", C.c_double), ("shard_issue_us_max", C.c_double),
                ("join_wait_us", C.c_double), ("tail_us", C.c_double), ("message_waits_us", C.c_double)]


assert C.sizeof(Material) == 32 and C.sizeof(CamData) == 160
assert C.sizeof(WorldData) == 32 and C.sizeof(Settings) == 48

MODE_PRIMARY, MODE_PRIMARY_SHADOW, MODE_PATH = 0, 1, 2

ID_VOXEL_MASK = 0x7FFF
ID_HIT, ID_NX, ID_NY, ID_NZ = 1 << 16, 1 << 17, 1 << 18, 1 << 19
ID_WATER, ID_SHADOW_RAY, ID_SHADOWED = 1 << 20, 1 << 21, 1 << 22

# every entry point include/vrt.h declares: (restype, argtypes)
_P = C.c_void_p
VRT_SYMBOLS = {
    "vrt_create": (C.c_int, [C.POINTER(Config), C.POINTER(_P)]),
    "vrt_destroy": (None, [_P]),
    "vrt_last_error": (C.c_char_p, [_P]),
    "vrt_write_nodes": (C.c_int, [_P, _P, C.c_uint32, C.c_uint32]),
    "vrt_write_chunk_roots": (C.c_int, [_P, C.c_uint32, _P, C.c_uint32]),
    "vrt_write_chunk_roots_tagged": (C.c_int, [_P, C.c_uint32, _P, C.c_uint32, C.c_uint64]),
    "vrt_resize_world": (C.c_int, [_P, C.c_uint32]),
    "vrt_write_materials": (C.c_int, [_P, C.c_uint32, _P, C.c_uint32]),
    "vrt_set_camera": (C.c_int, [_P, C.POINTER(CamData)]),
    "vrt_set_settings": (C.c_int, [_P, C.POINTER(Settings)]),
    "vrt_set_world": (C.c_int, [_P, C.POINTER(WorldData)]),
    "vrt_resize_output": (C.c_int, [_P, C.c_uint32, C.c_uint32]),
    "vrt_render": (C.c_int, [_P, C.POINTER(RenderOpts)]),
    "vrt_set_frames_in_flight": (C.c_int, [_P, C.c_uint32]),
    "vrt_synchronize": (C.c_int, [_P]),
    "vrt_read_output": (C.c_int, [_P, _P, _P, _P]),
    "vrt_present": (C.c_int, [_P, C.POINTER(Crosshair), C.c_uint32, C.c_uint32, _P]),
    "vrt_selftest_exact_math": (C.c_int, [C.c_int32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]),
    "vrt_present_device": (C.c_int, [_P, C.POINTER(Crosshair), C.c_uint32, C.c_uint32, C.POINTER(_P), C.POINTER(C.c_uint64)]),
    "vrt_set_presentation": (C.c_int, [_P, C.POINTER(Crosshair), C.c_uint32, C.c_uint32, C.c_uint32]),
    "vrt_get_stats": (C.c_int, [_P, C.POINTER(Stats)]),
    "vrt_get_issue_profile": (C.c_int, [_P, C.POINTER(IssueProfile)]),
    "vrt_get_accel_info": (C.c_int, [_P, C.POINTER(AccelInfo)]),
    "vrt_read_accel": (C.c_int, [_P, _P, _P]),
    "vrt_read_march_cells": (C.c_int, [_P, _P, _P]),
    "vrt_read_steps": (C.c_int, [_P, _P]),
    "vrt_set_stream": (C.c_int, [_P, _P]),
    "vrt_bind_output": (C.c_int, [_P, _P]),
    "vrt_device_output": (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_uint64)]),
    "vrt_shard_info": (C.c_int, [_P, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "vrt_assemble": (C.c_int, [_P, _P, C.c_uint64, _P]),
    "vrt_assemble_compact": (C.c_int, [_P, _P, C.c_uint64, _P]),
}


def _load(name: str, symbols: dict) -> C.CDLL:
    path = os.path.join(_HERE, name)
    if name == "libvrt.so" and os.environ.get("VRT_LIB"):   # development: A/B another build of the backend (tools/ab/)
        path = os.environ["VRT_LIB"]
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing — the native library is the product, there is no fallback. Build it with "
            f"`python -c 'import __graft_entry__ as g; g.build()'` or `make -C voxelraytracing_amd/csrc`.")
    lib = C.CDLL(path)
    for sym, (res, args) in symbols.items():
        fn = getattr(lib, sym)  # AttributeError if the ABI drifted from the header
        fn.restype = res
        fn.argtypes = args
    return lib


def code_object_sha256(path: str | None = None) -> str:
    """sha256 of the device code (the ELF section .hip_fatbin) of libvrt.so: names the kernels a measurement belongs to
    (bench.py prints PMC-derived figures only next to the build they were collected on)."""
    import hashlib
    import struct
    path = path or os.environ.get("VRT_LIB") or os.path.join(_HERE, "libvrt.so")
    data = open(path, "rb").read()
    if data[:4] != b"\x7fELF" or data[4] != 2:
        return hashlib.sha256(data).hexdigest()
    shoff, = struct.unpack_from("<Q", data, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", data, 0x3A)
    sec = [struct.unpack_from("<IIQQQQIIQQ", data, shoff + i * shentsize) for i in range(shnum)]
    names = sec[shstrndx]
    for s in sec:
        name = data[names[4] + s[0]:data.index(b"\0", names[4] + s[0])]
        if name == b".hip_fatbin":
            return hashlib.sha256(data[s[4]:s[4] + s[5]]).hexdigest()
    return hashlib.sha256(data).hexdigest()


def kernel_registers(path: str | None = None) -> dict:
    """{kernel symbol: {"vgprs", "sgprs", "vgpr_spills", "sgpr_spills", "scratch_bytes"}} of the gfx950 kernels in libvrt.so,
    read from the code objects' metadata (llvm-readelf; no GPU).  tests/test_abi.py holds every kernel to no spills and no
    scratch: a bounce kernel that spilled (SGPRs into VGPR lanes, three VGPRs into scratch) faulted on the card in round 3
    while the same source without the spills was bit-exact, so a spill is a build error here, not a slowdown."""
    import re
    import struct
    import subprocess
    import tempfile
    path = path or os.environ.get("VRT_LIB") or os.path.join(_HERE, "libvrt.so")
    data = open(path, "rb").read()
    shoff, = struct.unpack_from("<Q", data, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", data, 0x3A)
    sec = [struct.unpack_from("<IIQQQQIIQQ", data, shoff + i * shentsize) for i in range(shnum)]
    names = sec[shstrndx]
    fat = b""
    for s in sec:
        if data[names[4] + s[0]:data.index(b"\0", names[4] + s[0])] == b".hip_fatbin":
            fat = data[s[4]:s[4] + s[5]]
    magic = b"__CLANG_OFFLOAD_BUNDLE__"   # one bundle per translation unit: {count, {offset, size, triple}...}
    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    found = {}
    at = fat.find(magic)
    while at >= 0:
        count, = struct.unpack_from("<Q", fat, at + 24)
        q = at + 32
        for _ in range(count):
            off, size, tl = struct.unpack_from("<QQQ", fat, q)
            triple = fat[q + 24:q + 24 + tl].decode()
            q += 24 + tl
            if "gfx950" not in triple or not size:
                continue
            with tempfile.NamedTemporaryFile(suffix=".co") as f:
                f.write(fat[at + off:at + off + size])
                f.flush()
                notes = subprocess.run([readelf, "--notes", f.name], capture_output=True, text=True, check=True).stdout
            for m in re.finditer(r"\.private_segment_fixed_size:\s+(\d+)\s+\.sgpr_count:\s+(\d+)\s+\.sgpr_spill_count:\s+(\d+)\s+"
                                 r"\.symbol:\s+(\S+?)\.kd\s.*?\.vgpr_count:\s+(\d+)\s+\.vgpr_spill_count:\s+(\d+)", notes, re.S):
                found[m.group(4)] = {"scratch_bytes": int(m.group(1)), "sgprs": int(m.group(2)), "sgpr_spills": int(m.group(3)),
                                     "vgprs": int(m.group(5)), "vgpr_spills": int(m.group(6))}
        at = fat.find(magic, at + 24)
    return found


_vrt = None


def vrt() -> C.CDLL:
    """The HIP backend. Loading needs libamdhip64 but no GPU; every compute call needs a GPU."""
    global _vrt
    if _vrt is None:
        # One HIP runtime per process: the torch wheel bundles its own libamdhip64.so.7 / libhsa-runtime64 and
        # fails to initialise ("No HIP GPUs are available") if the system copy was mapped first.  Loading torch's
        # first lets libvrt.so bind to it by SONAME, so torch streams / tensors and vrt_* calls share one runtime
        # (needed by vrt_set_stream and vrt_bind_output).  A C or Rust host simply links /opt/rocm's.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        _vrt = _load("libvrt.so", VRT_SYMBOLS)
    return _vrt


_U32P = C.POINTER(C.c_uint32)
_I32P = C.POINTER(C.c_int32)
VRTH_SYMBOLS = {
    "vrth_world_new": (_P, [_I32P, C.c_uint32, C.c_uint32]),
    "vrth_world_free": (None, [_P]),
    "vrth_world_create_chunk": (C.c_int, [_P, _I32P, _P, C.c_uint32, _U32P]),
    "vrth_world_set_voxel": (C.c_int, [_P, _I32P, C.c_uint16, _U32P, _U32P]),
    "vrth_world_get_voxel": (C.c_int, [_P, _I32P, C.POINTER(C.c_uint16)]),
    "vrth_world_center_chunks": (C.c_uint32, [_P, _I32P]),
    "vrth_world_resize": (None, [_P, C.c_uint32]),
    "vrth_world_nodes": (_P, [_P]),
    "vrth_world_max_nodes": (C.c_uint32, [_P]),
    "vrth_world_chunk_roots": (C.c_uint32, [_P, _P, C.c_uint32]),
    "vrth_world_chunk_roots_ptr": (C.c_void_p, [_P]),
    "vrth_world_chunk_roots_generation": (C.c_uint64, [_P]),
    "vrth_world_info": (None, [_P, _I32P, _U32P, _U32P, _U32P]),
    "vrth_world_alloc_status": (None, [_P, _U32P, _U32P]),
    "vrth_world_chunk_state": (C.c_int, [_P, _I32P, _U32P, _U32P, _U32P, _P, C.c_uint32]),
    "vrth_world_highest_vox_at": (C.c_int, [_P, C.c_int32, C.c_int32, _I32P]),
    "vrth_world_data_from": (None, [_P, C.POINTER(WorldData)]),
    "vrth_cam_data_create": (None, [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.POINTER(C.c_float), C.POINTER(CamData)]),
    "vrth_axis_rot_to_ray": (None, [C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "vrth_std_materials": (None, [_P]),
    "vrth_std_voxel_name": (C.c_char_p, [C.c_uint32]),
    "vrth_svo_build_by_set_node": (C.c_uint32, [_P, _P, C.c_uint32]),
    "vrth_svo_build_bottom_up": (C.c_uint32, [_P, _P, C.c_uint32]),
    "vrth_svo_to_dense": (None, [_P, _P]),
    "vrth_gen_height": (C.c_int32, [C.c_uint32, C.c_int32, C.c_int32]),
    "vrth_gen_dense": (C.c_int, [C.c_uint32, _I32P, _P]),
    "vrth_gen_dense_superflat": (None, [_I32P, _P]),
    "vrth_world_generate": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.c_int]),
    "vrth_world_generate_missing": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.c_int, _P, C.c_uint32, _U32P]),
    "vrth_region_load_into_world": (C.c_int, [_P, _P, C.c_uint64, _I32P, _U32P]),
    "vrth_region_save_from_world": (C.c_uint64, [_P, _I32P, _P, C.c_uint64]),
    "vrth_chunk_msg_ingest": (C.c_int, [_P, _P, C.c_uint64, C.POINTER(C.c_uint64), _I32P, _U32P, _U32P]),
    "vrth_chunk_msg_encode": (C.c_uint64, [_P, _I32P, _P, C.c_uint64]),
    "vrth_region_of_chunk": (None, [_I32P, _I32P, _U32P]),
    "vrth_region_file_name": (C.c_uint32, [_I32P, C.c_char_p, C.c_uint32]),
}

_host = None


def host() -> C.CDLL:
    """The C++ host mirror (ClientWorld, CamData::create, world generator). CPU only."""
    global _host
    if _host is None:
        _host = _load("libvrt_host.so", VRTH_SYMBOLS)
    return _host
