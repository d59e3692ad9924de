"""GPU boundary — Python handles on libvrt.so, named after the reference's seam types.

    Gpu / GpuResources / Buffers / NodeBuffer / PixelShader  <- clientdesktop/src/graphics/{mod.rs,shader.rs}
    CamData.create, Settings, Material, WorldData            <- clientdesktop/src/graphics/mod.rs:20-143

The frame loop of main.rs:426-453 maps call for call.  There is no CPU path: every
method ends in a vrt_* call of the HIP backend and raises ``VrtError`` if that fails.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _ffi
from ._ffi import (CamData, Material, Settings, WorldData, Stats, RenderOpts,  # noqa: F401  (re-exported)
                   MODE_PRIMARY, MODE_PRIMARY_SHADOW, MODE_PATH)


class VrtError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"vrt error {code}: {msg}")
        self.code = code


def _f(vals, n):
    return (C.c_float * n)(*[float(v) for v in vals])


_scratch3a, _scratch3b, _scratch2 = (C.c_float * 3)(), (C.c_float * 3)(), (C.c_float * 2)()


def cam_data_create(rot_deg, eye, fov_deg: float, proj_size) -> CamData:
    """CamData::create(cam, eye, fov, proj_size) — mod.rs:92-111 (rot and fov in degrees)."""
    out = CamData()
    _scratch3a[0], _scratch3a[1], _scratch3a[2] = rot_deg      # (per-frame call of the frame loop: no allocations here)
    _scratch3b[0], _scratch3b[1], _scratch3b[2] = eye
    _scratch2[0], _scratch2[1] = proj_size
    _ffi.host().vrth_cam_data_create(_scratch3a, _scratch3b, fov_deg, _scratch2, C.byref(out))
    return out


def axis_rot_to_ray(rot_rad):
    """common/src/math.rs:131-146"""
    out = (C.c_float * 3)()
    _ffi.host().vrth_axis_rot_to_ray(_f(rot_rad, 3), out)
    return tuple(out)


def make_settings(max_ray_bounces=3, sun_intensity=4.0, show_step_count=0, sky_color=(0.81, 0.93, 1.0),
                  sun_pos=(0.0, 0.0, 0.0)) -> Settings:
    """Defaults are AppState::new's (clientdesktop/src/main.rs:152-156; sun_pos left at the origin there)."""
    s = Settings()
    s.max_ray_bounces, s.sun_intensity, s.show_step_count = max_ray_bounces, sun_intensity, show_step_count
    s.sky_color[:] = sky_color
    s.sun_pos[:] = sun_pos
    return s


def std_materials():
    """Material::construct_arr over the standard data pack (mod.rs:38-60): (Material*256) array."""
    arr = (Material * 256)()
    _ffi.host().vrth_std_materials(arr)
    return arr


class Gpu:
    """Owns one backend context; stands where the reference passes `&Gpu` (mod.rs:227-304).

    Arguments are GpuResources::new's (mod.rs:155-161) plus the tile shard of this process."""

    def __init__(self, max_nodes: int, world_size: int, result_size, device: int = -1,
                 shard_rank: int = 0, shard_count: int = 1, tile_major: bool = False, root_weight: int = 1,
                 row_major: bool = False, compact: bool = False, devices=None, texel_messages: bool = False,
                 staged_messages: bool = False, poison_messages: bool = False):
        """devices: a list of HIP device ordinals makes this ONE context over several devices (vrt_config.device_ids)."""
        self._lib = _ffi.vrt()
        cfg = _ffi.Config(max_nodes, world_size, result_size[0], result_size[1], device, shard_rank, shard_count,
                          (1 if tile_major else 0) | (2 if row_major else 0) | (4 if compact else 0) | (8 if texel_messages else 0) |
                          (16 if staged_messages else 0) | (32 if poison_messages else 0),
                          root_weight)
        if devices:
            cfg.n_devices = len(devices)
            for i, d in enumerate(devices):
                cfg.device_ids[i] = d
        h = C.c_void_p()
        rc = self._lib.vrt_create(C.byref(cfg), C.byref(h))
        if rc:
            raise VrtError(rc, self._lib.vrt_last_error(None).decode())
        self._h = h
        self.result_size = (result_size[0], result_size[1])
        self.shard_rank, self.shard_count = shard_rank, shard_count

    def close(self):
        if getattr(self, "_h", None):
            self._lib.vrt_destroy(self._h)
            self._h = None

    __del__ = close

    def _ck(self, rc: int):
        if rc:
            raise VrtError(rc, self._lib.vrt_last_error(self._h).decode())

    # --- Buffers (shader.rs:43-143) ---
    def write_nodes(self, pool, start: int, end: int):
        """NodeBuffer::write(gpu, src_nodes, start..end) — shader.rs:22-40. pool: u16 array or address of node 0."""
        ptr = pool if isinstance(pool, int) else pool.ctypes.data
        self._ck(self._lib.vrt_write_nodes(self._h, C.c_void_p(ptr), start, end))

    def write_chunk_roots(self, roots: np.ndarray, offset: int = 0, tag: int = 0):
        """ArrayBuffer::write (shader.rs:133-142).  tag: vrt_write_chunk_roots_tagged — e.g. ClientWorld.roots_generation()."""
        roots = np.ascontiguousarray(roots, dtype=np.uint32)
        self._ck(self._lib.vrt_write_chunk_roots_tagged(self._h, offset, roots.ctypes.data, roots.size, tag))

    def resize_chunk_buffer(self, world_size: int):
        self._ck(self._lib.vrt_resize_world(self._h, world_size))

    def write_materials(self, mats, first: int = 0, n: Optional[int] = None):
        n = len(mats) if n is None else n
        self._ck(self._lib.vrt_write_materials(self._h, first, C.cast(mats, C.c_void_p), n))

    def write_cam_data(self, cam: CamData):
        self._ck(self._lib.vrt_set_camera(self._h, C.byref(cam)))

    def write_settings(self, s: Settings):
        self._ck(self._lib.vrt_set_settings(self._h, C.byref(s)))

    def write_world_data(self, w: WorldData):
        self._ck(self._lib.vrt_set_world(self._h, C.byref(w)))

    def resize_result_texture(self, new_size):
        self._ck(self._lib.vrt_resize_output(self._h, new_size[0], new_size[1]))
        self.result_size = (new_size[0], new_size[1])

    # --- PixelShader (shader.rs:295-380) ---
    def encode_pass(self, mode: int = MODE_PRIMARY, variant: int = 0, stats: bool = False, spp: int = 1, seed: int = 0,
                    own_streams: bool = False, timed: bool = False):
        """PixelShader::encode_pass + queue.submit (shader.rs:371-379, main.rs:453,565). Asynchronous.
        own_streams: VRT_RENDER_OWN_STREAMS (include/vrt.h)."""
        o = RenderOpts(mode, variant, int(stats), spp, seed, (1 if own_streams else 0) | (2 if timed else 0))   # stats: False/True, or 2 = clock probe; timed: VRT_RENDER_TIMED
        self._ck(self._lib.vrt_render(self._h, C.byref(o)))

    render = encode_pass

    def set_frames_in_flight(self, n: int):
        """1..4 frames in flight (default 2); see vrt_set_frames_in_flight in include/vrt.h."""
        self._ck(self._lib.vrt_set_frames_in_flight(self._h, n))

    def synchronize(self):
        self._ck(self._lib.vrt_synchronize(self._h))

    # --- read-back (no reference counterpart) ---
    def read_output(self, rgb=True, ids=True, rgba8=False):
        w, h = self.result_size
        a_rgb = np.empty((h, w, 3), dtype=np.float32) if rgb else None
        a_ids = np.empty((h, w), dtype=np.uint32) if ids else None
        a_q = np.empty((h, w, 4), dtype=np.uint8) if rgba8 else None
        p = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None  # noqa: E731
        self._ck(self._lib.vrt_read_output(self._h, p(a_rgb), p(a_ids), p(a_q)))
        return a_rgb, a_ids, a_q

    def present(self, screen_size=None, color=(1.0, 1.0, 1.0, 0.33), style: int = 2, size: float = 5.0) -> np.ndarray:
        """ScreenShader::encode_pass into host memory: rgba8 [screen_h, screen_w, 4] of the last frame, sampled through the
        reference's (bilinear) sampler at any window size, under the crosshair (defaults = Crosshair::default(), mod.rs:71-80)."""
        sw, sh = screen_size or self.result_size
        ch = _ffi.Crosshair((C.c_float * 4)(*color), style, size)
        out = np.empty((sh, sw, 4), dtype=np.uint8)
        self._ck(self._lib.vrt_present(self._h, C.byref(ch), sw, sh, out.ctypes.data_as(C.c_void_p)))
        return out

    def present_device(self, screen_size=None, color=(1.0, 1.0, 1.0, 0.33), style: int = 2, size: float = 5.0):
        """vrt_present_device: (device pointer, bytes) of the presented rgba8 image, left on the GPU."""
        sw, sh = screen_size or self.result_size
        ch = _ffi.Crosshair((C.c_float * 4)(*color), style, size)
        ptr, nb = C.c_void_p(), C.c_uint64()
        self._ck(self._lib.vrt_present_device(self._h, C.byref(ch), sw, sh, C.byref(ptr), C.byref(nb)))
        return ptr.value, nb.value

    def set_presentation(self, screen_size=None, color=(1.0, 1.0, 1.0, 0.33), style: int = 2, size: float = 5.0, skip_texels: bool = False, off: bool = False):
        """vrt_set_presentation: the frames that follow are presented to a window of screen_size under this crosshair — when the window
        samples the texture texel for texel their march kernel stores the window's image itself and present / present_device with the
        same crosshair and size launch nothing (main.rs:452-454 in one launch).  skip_texels: such frames store the image only."""
        sw, sh = screen_size or self.result_size
        ch = _ffi.Crosshair((C.c_float * 4)(*color), style, size)
        self._ck(self._lib.vrt_set_presentation(self._h, None if off else C.byref(ch), sw, sh, 1 if skip_texels else 0))

    def read_steps(self) -> np.ndarray:
        w, h = self.result_size
        a = np.empty((h, w), dtype=np.uint32)
        self._ck(self._lib.vrt_read_steps(self._h, a.ctypes.data_as(C.c_void_p)))
        return a

    def stats(self) -> Stats:
        s = Stats()
        self._ck(self._lib.vrt_get_stats(self._h, C.byref(s)))
        return s

    def issue_profile(self) -> dict:
        """vrt_get_issue_profile: what issuing a frame cost the host (us, averaged over the render calls since the last call)."""
        p = _ffi.IssueProfile()
        self._ck(self._lib.vrt_get_issue_profile(self._h, C.byref(p)))
        return {k: getattr(p, k) for k, _ in p._fields_ if not k.startswith("_")}

    def accel_info(self) -> "_ffi.AccelInfo":
        """The derived lookup tables of the default march (cell grid + brick pool); rebuilt lazily by render()."""
        a = _ffi.AccelInfo()
        self._ck(self._lib.vrt_get_accel_info(self._h, C.byref(a)))
        return a

    def read_accel(self):
        """(grid[G,G,G] indexed [z,y,x], bricks[n,64]) as built on the device — see vrt_read_accel in include/vrt.h."""
        a = self.accel_info()
        g = a.world_size_chunks * 8
        grid = np.empty((g, g, g), dtype=np.uint32)
        bricks = np.empty((int(a.bricks), 64), dtype=np.uint16)
        self._ck(self._lib.vrt_read_accel(self._h, grid.ctypes.data_as(C.c_void_p), bricks.ctypes.data_as(C.c_void_p)))
        return grid, bricks

    def read_march_cells(self):
        """(cells[G,G,G,4] indexed [z,y,x], direct) — see vrt_read_march_cells in include/vrt.h."""
        g = self.accel_info().world_size_chunks * 8
        cells = np.empty((g, g, g, 4), dtype=np.uint32)
        direct = C.c_uint32()
        self._ck(self._lib.vrt_read_march_cells(self._h, cells.ctypes.data_as(C.c_void_p), C.byref(direct)))
        return cells, bool(direct.value)

    # --- device plumbing for torch / RCCL ---
    def set_stream(self, hip_stream: int):
        self._ck(self._lib.vrt_set_stream(self._h, C.c_void_p(hip_stream)))

    def bind_output(self, texels_ptr: int):
        """Render into caller-owned device memory (a torch tensor); 0 restores the context's own buffer."""
        self._ck(self._lib.vrt_bind_output(self._h, C.c_void_p(texels_ptr or None)))

    def device_output(self):
        """(device pointer, bytes) of the 16-byte-texel buffer frames are written to."""
        t, nb = C.c_void_p(), C.c_uint64()
        self._ck(self._lib.vrt_device_output(self._h, C.byref(t), C.byref(nb)))
        return t.value, nb.value

    def shard_info(self):
        a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
        self._ck(self._lib.vrt_shard_info(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def assemble(self, gathered_ptr: int, dst_ptr: int, rank_stride_bytes: int = 0, compact: bool = False):
        fn = self._lib.vrt_assemble_compact if compact else self._lib.vrt_assemble
        self._ck(fn(self._h, C.c_void_p(gathered_ptr), rank_stride_bytes, C.c_void_p(dst_ptr)))

    # --- convenience: what join_game does (main.rs:211-223) ---
    def upload_world(self, world, materials=None):
        """Upload the whole pool, chunk_roots, WorldData and materials of a ClientWorld."""
        self.write_nodes(world.nodes_ptr(), 0, world.max_nodes() & ~1)
        self.write_chunk_roots(world.chunk_roots())
        self.write_world_data(world.world_data())
        self.write_materials(materials if materials is not None else std_materials())
