/*
 * vrt_oracle.h — CPU ORACLE for the per-pixel SVO ray-march.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference's algorithm for the hot path
 * (MasonFeurer/VoxelRayTracing, clientdesktop/src/graphics/ray_tracer.wgsl and the host-side
 * data model that feeds it).  It exists to CHECK the HIP backend; nothing under
 * voxelraytracing_amd/ may include, link or call it.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it.
 *
 * PARITY STATUS: the reference has no tests, golden vectors or fixtures for this path and cannot be
 * built here (Rust + WGSL via wgpu; no cargo/rustc/naga in the image) — but its shader text can be executed:
 * tests/wgsl_interp.py, a generic WGSL interpreter, runs clientdesktop/src/graphics/ray_tracer.wgsl as it stands
 * (tests/golden/make_wgsl_fixtures.py, read from /root/reference at generation time) and this oracle reproduces what the
 * shader computes on thirteen scenes (config C1 at its full size among them) — voxel, hit, normal, water distance, hit position, iteration count bit for bit for every
 * pixel, colour to 1e-6 — as well as rng_next / rng_next_dir of path_tracer.wgsl and fs_main of screen_shader.wgsl
 * (tests/test_oracle_vs_reference_wgsl.py).  That pins it to the reference's SOURCE, not to a run of the reference's binary:
 * what WGSL leaves to the implementation (min with a NaN, i32(NaN), summation orders, pow's last ulps, reads past the end
 * of an array) is decided once, identically, here and in the interpreter.  Also: hand-derived known-answer tests
 * (tests/test_oracle_kat.py), golden vectors it generated itself (tests/golden/c*.npz) and a second restatement of the live
 * shader in numpy (tests/wgsl_numpy.py) that it agrees with bit for bit on ids and iteration counts.
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference).
 */
#ifndef VRT_ORACLE_H
#define VRT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- uniform structs, byte-identical to clientdesktop/src/graphics/mod.rs ---- */

/* Material, mod.rs:20-28 (32 B). */
typedef struct {
    float color[3];
    uint32_t is_empty;
    uint32_t is_liquid;
    float scatter;
    uint32_t _padding[2];
} orc_material;

/* CamData, mod.rs:82-91 (160 B). Matrices are column-major (glam Mat4): m[col*4+row]. */
typedef struct {
    float pos[3];
    uint32_t _padding0;
    float inv_view_mat[16];
    float inv_proj_mat[16];
    float proj_size[2];
    uint32_t _padding1[2];
} orc_cam_data;

/* WorldData, mod.rs:113-120 (32 B). */
typedef struct {
    int32_t min[3];
    uint32_t size;           /* world width in voxels */
    uint32_t size_in_chunks;
    uint32_t _padding[3];
} orc_world_data;

/* Settings, mod.rs:132-143 (48 B). */
typedef struct {
    uint32_t max_ray_bounces;
    float sun_intensity;
    uint32_t show_step_count;
    uint32_t _padding0;
    float sky_color[3];
    uint32_t _padding1;
    float sun_pos[3];
    uint32_t _padding2;
} orc_settings;

typedef struct {
    const uint16_t *nodes;        /* the whole flat node pool (client/src/world.rs:261) */
    uint32_t n_nodes;
    const uint32_t *chunk_roots;  /* S^3 table (client/src/world.rs:154-159) */
    uint32_t n_chunk_roots;
    const orc_material *materials; /* 256 entries (shader.rs:48) */
    orc_cam_data cam;
    orc_settings settings;
    orc_world_data world;
} orc_scene;

enum {
    ORC_MODE_PRIMARY = 0,        /* the reference's live shader: 1 primary ray / pixel */
    ORC_MODE_PRIMARY_SHADOW = 1, /* build-defined extension: + 1 shadow ray from every solid hit */
    ORC_MODE_PATH = 2            /* build-defined extension after path_tracer.wgsl (stale in the reference) */
};

/* Per-pixel id word (build-defined; the reference only writes rgba8):
 *   bits 0..14  voxel id the march stopped on (0 on a miss)
 *   bit 16      hit            (ray_tracer.wgsl:293, also set on 500-step exhaustion)
 *   bit 17..19  norm.x / norm.y / norm.z non-zero (ray_tracer.wgsl:272)
 *   bit 20      water_dist != 0 (overlay applied, ray_tracer.wgsl:137)
 *   bit 21      a shadow ray was launched from this pixel
 *   bit 22      that shadow ray was occluded
 */
#define ORC_ID_VOXEL_MASK 0x7FFFu
#define ORC_ID_HIT (1u << 16)
#define ORC_ID_NX (1u << 17)
#define ORC_ID_NY (1u << 18)
#define ORC_ID_NZ (1u << 19)
#define ORC_ID_WATER (1u << 20)
#define ORC_ID_SHADOW_RAY (1u << 21)
#define ORC_ID_SHADOWED (1u << 22)

/* Build-defined shadow-ray constants (no reference counterpart; see DESIGN.md §Shadow rays). */
#define ORC_SHADOW_BIAS 0.002f
#define ORC_SHADOW_FACTOR 0.35f

typedef struct {
    uint64_t primary_rays;
    uint64_t secondary_rays;    /* shadow rays / bounce rays actually launched */
    uint64_t hits;              /* primary rays with hit==true */
    uint64_t steps;             /* march-loop iterations over all rays (ray_tracer.wgsl:220) */
    uint64_t node_visits;       /* sum over steps of nodes visited root->leaf (L in SURVEY §8d) */
    uint64_t primary_steps;     /* the primary-ray share of the two totals above */
    uint64_t primary_node_visits;
} orc_stats;

/* Render the rectangle [x0,x1) x [y0,y1) of a w x h frame.
 * rgb: w*h*3 floats (row-major, full frame; only the rectangle is written); may be NULL.
 * ids: w*h id words; may be NULL.  steps: w*h per-pixel step counts (primary | shadow<<16); may be NULL.
 * threads <= 0: all OpenMP threads.  spp/seed are used by ORC_MODE_PATH only. */
void orc_render(const orc_scene *scene, int mode, uint32_t w, uint32_t h,
                uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1,
                float *rgb, uint32_t *ids, uint32_t *steps, orc_stats *stats,
                int threads, uint32_t spp, uint32_t seed);

/* One primary ray for known-answer tests: returns the id word; fills rgb[3], dir[3], and
 * out[8] = {hit.pos xyz, norm xyz, water_dist, iter_count}. */
uint32_t orc_trace_pixel(const orc_scene *scene, int mode, uint32_t px, uint32_t py,
                         float *rgb, float *dir, float *out);

/* March one arbitrary ray (ray_world, ray_tracer.wgsl:182-316). origin is world-local.
 * out[8] as above. Returns the id word (without shadow bits). */
uint32_t orc_ray_world(const orc_scene *scene, const float origin[3], const float dir[3],
                       float *color, float *out);

/* get_node, ray_tracer.wgsl:38-42, on the u32-pair view of the pool. */
uint32_t orc_get_node(const uint32_t *pairs, uint32_t idx);

/* find_node / find_chunk_node, ray_tracer.wgsl:76-125. out = {idx, root, depth, min xyz, max xyz, size}
 * with the floats bit-cast into uint32_t slots 3..9. */
void orc_find_node(const orc_scene *scene, const float pos[3], uint32_t max_depth, uint32_t *out10);

/* ray_sky, ray_tracer.wgsl:144-157. */
void orc_ray_sky(const orc_scene *scene, const float origin[3], const float dir[3], float *rgb);

/* Crosshair — clientdesktop/src/graphics/mod.rs:63-70 (32 B). style: 0 off, 1 dot, 2 cross. */
typedef struct {
    float color[4];
    uint32_t style;
    float size;
    uint32_t _padding[2];
} orc_crosshair;

/* Presentation of a traced frame: `textureStore` of the f32 colour into the rgba8unorm result texture
 * (ray_tracer.wgsl:179: clamp to [0,1], x255, round to nearest even) followed by fs_main of screen_shader.wgsl:43-65 for
 * every pixel of a screen_w x screen_h target: the result texture sampled at the pixel centre (Nearest: the
 * magnification filter, texture.rs:36 — screen >= texture on both axes) blended with the crosshair.  The fragment's
 * vec4 is written as unorm8 RGBA (what a non-sRGB Rgba8Unorm target stores). */
void orc_present(const float *rgb, uint32_t w, uint32_t h, uint32_t screen_w, uint32_t screen_h,
                 const orc_crosshair *crosshair, uint8_t *rgba8);

/* CamData::create, mod.rs:92-111, on a restatement of glam 0.31.0's Mat4 routines. */
void orc_cam_data_create(const float rot_deg[3], const float eye[3], float fov_deg,
                         const float proj_size[2], orc_cam_data *out);

/* axis_rot_to_ray, common/src/math.rs:131-146 (radians). */
void orc_axis_rot_to_ray(const float rot[3], float *out3);

/* PCG step of path_tracer.wgsl:56-61: advances *state, returns the f32 in [0,1]. */
float orc_rng_next(uint32_t *state);

/* ---- SVO data model (common/src/world/mod.rs) ---- */

/* NodeAlloc, common/src/world/mod.rs:213-313. */
typedef struct {
    uint32_t range_start, range_end;
    uint32_t *free_start;  /* spans of free memory */
    uint32_t *free_end;
    uint32_t n_free, cap_free;
    uint32_t last_used_addr;
} orc_node_alloc;

void orc_node_alloc_init(orc_node_alloc *a, uint32_t used_start, uint32_t used_end,
                         uint32_t free_start, uint32_t free_end);
void orc_node_alloc_destroy(orc_node_alloc *a);
int orc_node_alloc_next(orc_node_alloc *a, uint32_t *addr);   /* 1 = ok, 0 = none */
int orc_node_alloc_peek(const orc_node_alloc *a, uint32_t *addr);
void orc_node_alloc_free(orc_node_alloc *a, uint32_t addr);
void orc_node_alloc_move_end(orc_node_alloc *a, uint32_t new_end);

/* Svo::find_node, common/src/world/mod.rs:366-395. out = {idx, depth, size}; center -> c[3]. */
void orc_svo_find_node(const uint16_t *nodes, uint32_t root, uint32_t svo_size,
                       const uint32_t pos[3], uint32_t max_depth, uint32_t *out3, float *c);

/* Svo::set_node, common/src/world/mod.rs:397-459.
 * Returns 0 ok, 1 out of memory (SetVoxelErr::OutOfMemory), 2 where the assert at :416 would fire. */
int orc_svo_set_node(uint16_t *nodes, uint32_t root, uint32_t svo_size, const uint32_t pos[3],
                     uint16_t voxel, uint32_t target_depth, orc_node_alloc *alloc);

/* Convenience: build one chunk the way server/src/world/gen.rs:171-286 does for a flat rule —
 * NodeAlloc::new(0..1, 1..cap), then set_node for x in 0..32, z in 0..32, y ascending.
 * column(x,z,y) comes from a dense 32*32*32 u16 voxel array indexed [x + 32*(y + 32*z)];
 * voxel 0 entries are skipped (never set), as gen.rs only writes the terrain and water columns.
 * Returns last_used_addr + 1 (number of nodes in use incl. holes), or 0 on OOM. */
uint32_t orc_build_chunk_by_set_node(const uint16_t *dense, uint16_t *nodes, uint32_t cap);

#ifdef __cplusplus
}
#endif
#endif
