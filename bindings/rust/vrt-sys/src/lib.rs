//! Raw bindings to `include/vrt.h` — the C ABI of the MI355X SVO ray-march backend.
//!
//! The four uniform structs are byte-identical to the reference's `#[repr(C)]` structs
//! (`clientdesktop/src/graphics/mod.rs:20-28, 63-70, 82-91, 113-120, 132-143`), so the client can either use these
//! or pass pointers to its own `Material` / `Crosshair` / `CamData` / `WorldData` / `Settings` values unchanged.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
pub struct vrt_ctx {
    _private: [u8; 0],
}

pub const VRT_OK: c_int = 0;
pub const VRT_ERR_INVALID_ARG: c_int = -1;
pub const VRT_ERR_OUT_OF_RANGE: c_int = -2;
pub const VRT_ERR_DEVICE: c_int = -3;
pub const VRT_ERR_OOM: c_int = -4;
pub const VRT_ERR_STATE: c_int = -5;

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct vrt_material {
    pub color: [f32; 3],
    pub is_empty: u32,
    pub is_liquid: u32,
    pub scatter: f32,
    pub _padding: [u32; 2],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct vrt_cam_data {
    pub pos: [f32; 3],
    pub _padding0: u32,
    pub inv_view_mat: [f32; 16],
    pub inv_proj_mat: [f32; 16],
    pub proj_size: [f32; 2],
    pub _padding1: [u32; 2],
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct vrt_world_data {
    pub min: [i32; 3],
    pub size: u32,
    pub size_in_chunks: u32,
    pub _padding: [u32; 3],
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct vrt_settings {
    pub max_ray_bounces: u32,
    pub sun_intensity: f32,
    pub show_step_count: u32,
    pub _padding0: u32,
    pub sky_color: [f32; 3],
    pub _padding1: u32,
    pub sun_pos: [f32; 3],
    pub _padding2: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct vrt_crosshair {
    pub color: [f32; 4],
    pub style: u32,
    pub size: f32,
    pub _padding: [u32; 2],
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct vrt_config {
    pub max_nodes: u32,
    pub world_size_chunks: u32,
    pub width: u32,
    pub height: u32,
    pub device: i32,
    pub shard_rank: u32,
    pub shard_count: u32,
    pub flags: u32,
    pub shard_root_weight: u32,
    pub n_devices: u32,
    pub device_ids: [i32; VRT_MAX_DEVICES],
}

pub const VRT_MAX_DEVICES: usize = 16;
pub const VRT_FLAG_TILE_MAJOR: u32 = 1;
pub const VRT_FLAG_ROW_MAJOR: u32 = 2;
pub const VRT_FLAG_COMPACT: u32 = 4;
pub const VRT_PRESENT_SKIP_TEXELS: u32 = 1;
pub const VRT_FLAG_TEXEL_MESSAGES: u32 = 8;
pub const VRT_FLAG_STAGED_MESSAGES: u32 = 16;
pub const VRT_FLAG_POISON_MESSAGES: u32 = 32;

pub const VRT_MODE_PRIMARY: u32 = 0;
pub const VRT_MODE_PRIMARY_SHADOW: u32 = 1;
pub const VRT_MODE_PATH: u32 = 2;

pub const VRT_RENDER_OWN_STREAMS: u32 = 1;
pub const VRT_RENDER_TIMED: u32 = 2;

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct vrt_render_opts {
    pub mode: u32,
    pub variant: u32,
    pub stats: u32,
    pub spp: u32,
    pub seed: u32,
    pub flags: u32,
    pub _reserved: [u32; 2],
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct vrt_stats {
    pub primary_rays: u64,
    pub secondary_rays: u64,
    pub hits: u64,
    pub steps: u64,
    pub node_visits: u64,
    pub primary_steps: u64,
    pub primary_node_visits: u64,
    pub ms_total: f32,
    pub ms_primary: f32,
    pub ms_secondary: f32,
    pub frames: u32,
    pub sum_ms_primary: f64,
    pub sum_ms_secondary: f64,
    pub sum_ms_total: f64,
    pub clock_shader_ticks: u64,
    pub clock_ref_ticks: u64,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct vrt_accel_info {
    pub available: u32,
    pub world_size_chunks: u32,
    pub cells: u64,
    pub bricks: u64,
    pub bytes: u64,
    pub builds: u32,
    pub last_build_ms: f32,
    pub chunk_builds: u32,
    pub ordered_frames: u32,
}

/// What issuing a frame costs the host (vrt_get_issue_profile), microseconds per vrt_render call.
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct vrt_issue_profile {
    pub frames: u32,
    pub devices: u32,
    pub issuing_threads: u32,
    pub _reserved: u32,
    pub render_us: f64,
    pub root_issue_us: f64,
    pub shard_issue_us_mean: f64,
    pub shard_issue_us_max: f64,
    pub join_wait_us: f64,
    pub tail_us: f64,
    pub message_waits_us: f64,
}

pub const VRT_ID_VOXEL_MASK: u32 = 0x7FFF;
pub const VRT_ID_HIT: u32 = 1 << 16;
pub const VRT_ID_NX: u32 = 1 << 17;
pub const VRT_ID_NY: u32 = 1 << 18;
pub const VRT_ID_NZ: u32 = 1 << 19;
pub const VRT_ID_WATER: u32 = 1 << 20;
pub const VRT_ID_SHADOW_RAY: u32 = 1 << 21;
pub const VRT_ID_SHADOWED: u32 = 1 << 22;

extern "C" {
    pub fn vrt_create(cfg: *const vrt_config, out: *mut *mut vrt_ctx) -> c_int;
    pub fn vrt_destroy(ctx: *mut vrt_ctx);
    pub fn vrt_last_error(ctx: *const vrt_ctx) -> *const c_char;
    pub fn vrt_write_nodes(ctx: *mut vrt_ctx, pool: *const u16, start: u32, end: u32) -> c_int;
    pub fn vrt_write_chunk_roots(ctx: *mut vrt_ctx, offset: u32, roots: *const u32, n: u32) -> c_int;
    pub fn vrt_write_chunk_roots_tagged(ctx: *mut vrt_ctx, offset: u32, roots: *const u32, n: u32, tag: u64) -> c_int;
    pub fn vrt_resize_world(ctx: *mut vrt_ctx, world_size_chunks: u32) -> c_int;
    pub fn vrt_write_materials(ctx: *mut vrt_ctx, first: u32, mats: *const vrt_material, n: u32) -> c_int;
    pub fn vrt_set_camera(ctx: *mut vrt_ctx, cam: *const vrt_cam_data) -> c_int;
    pub fn vrt_set_settings(ctx: *mut vrt_ctx, settings: *const vrt_settings) -> c_int;
    pub fn vrt_set_world(ctx: *mut vrt_ctx, world: *const vrt_world_data) -> c_int;
    pub fn vrt_resize_output(ctx: *mut vrt_ctx, width: u32, height: u32) -> c_int;
    pub fn vrt_render(ctx: *mut vrt_ctx, opts: *const vrt_render_opts) -> c_int;
    pub fn vrt_set_frames_in_flight(ctx: *mut vrt_ctx, n: u32) -> c_int;
    pub fn vrt_synchronize(ctx: *mut vrt_ctx) -> c_int;
    pub fn vrt_read_output(ctx: *mut vrt_ctx, rgb: *mut f32, ids: *mut u32, rgba8: *mut u8) -> c_int;
    pub fn vrt_present(ctx: *mut vrt_ctx, crosshair: *const vrt_crosshair, screen_w: u32, screen_h: u32, rgba8: *mut u8) -> c_int;
    pub fn vrt_selftest_exact_math(device: i32, n: u32, seed: u32, mismatches: *mut u64) -> c_int;
    pub fn vrt_present_device(ctx: *mut vrt_ctx, crosshair: *const vrt_crosshair, screen_w: u32, screen_h: u32, rgba8_device: *mut *mut c_void, bytes: *mut u64) -> c_int;
    pub fn vrt_set_presentation(ctx: *mut vrt_ctx, crosshair: *const vrt_crosshair, screen_w: u32, screen_h: u32, flags: u32) -> c_int;
    pub fn vrt_get_stats(ctx: *mut vrt_ctx, out: *mut vrt_stats) -> c_int;
    pub fn vrt_get_issue_profile(ctx: *mut vrt_ctx, out: *mut vrt_issue_profile) -> c_int;
    pub fn vrt_get_accel_info(ctx: *mut vrt_ctx, out: *mut vrt_accel_info) -> c_int;
    pub fn vrt_read_accel(ctx: *mut vrt_ctx, grid: *mut u32, bricks: *mut u16) -> c_int;
    pub fn vrt_read_march_cells(ctx: *mut vrt_ctx, cells: *mut u32, direct: *mut u32) -> c_int;
    pub fn vrt_read_steps(ctx: *mut vrt_ctx, steps: *mut u32) -> c_int;
    pub fn vrt_set_stream(ctx: *mut vrt_ctx, hip_stream: *mut c_void) -> c_int;
    pub fn vrt_bind_output(ctx: *mut vrt_ctx, texels: *mut c_void) -> c_int;
    pub fn vrt_device_output(ctx: *mut vrt_ctx, texels: *mut *mut c_void, bytes: *mut u64) -> c_int;
    pub fn vrt_shard_info(ctx: *mut vrt_ctx, tiles_local: *mut u32, tiles_padded: *mut u32, tiles_total: *mut u32) -> c_int;
    pub fn vrt_assemble(ctx: *mut vrt_ctx, gathered: *const c_void, rank_stride_bytes: u64, dst: *mut c_void) -> c_int;
    pub fn vrt_assemble_compact(ctx: *mut vrt_ctx, gathered: *const c_void, rank_stride_bytes: u64, dst: *mut c_void) -> c_int;
}

#[cfg(test)]
mod layout {
    use super::*;
    use std::mem::size_of;
    #[test]
    fn struct_sizes_match_the_header() {
        assert_eq!(size_of::<vrt_material>(), 32);
        assert_eq!(size_of::<vrt_cam_data>(), 160);
        assert_eq!(size_of::<vrt_world_data>(), 32);
        assert_eq!(size_of::<vrt_settings>(), 48);
        assert_eq!(size_of::<vrt_crosshair>(), 32);
        assert_eq!(size_of::<vrt_config>(), 104);
        assert_eq!(size_of::<vrt_render_opts>(), 32);
        assert_eq!(size_of::<vrt_stats>(), 112);
        assert_eq!(size_of::<vrt_accel_info>(), 48);
        assert_eq!(size_of::<vrt_issue_profile>(), 72);
    }
}
