/*
 * vrt.h — C ABI of the MI355X (gfx950) SVO ray-march backend (libvrt.so).
 *
 * This is the drop-in boundary for the reference's GPU seam: everything the winit frame loop does to
 * `GpuResources` / `Buffers` / `PixelShader` (MasonFeurer/VoxelRayTracing,
 * clientdesktop/src/graphics/{mod.rs,shader.rs}, callers in clientdesktop/src/main.rs) maps to one
 * call below.  Plain pointers and sizes only; the four uniform structs are byte-identical to the
 * reference's #[repr(C)] structs, so a Rust host passes its own values unchanged (INTEGRATION.md shows
 * the `extern "C"` block a maintainer would add).
 *
 * Semantics shared by every write: the host owns the source memory, the call copies at call time
 * (wgpu `queue.write_buffer`, shader.rs:105,113,141) and the data is visible to the next vrt_render.
 * One thread per context (the reference drives this seam from the winit thread only,
 * main.rs:681-722); calls are ordered on one HIP stream.
 *
 * Every function returns VRT_OK (0) or a negative vrt_status; vrt_last_error() gives the text.
 * Nothing aborts: the reference's unwrap()/panic sites (mod.rs:223-274, client/src/world.rs:251)
 * become status codes.
 */
#ifndef VRT_H
#define VRT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vrt_ctx vrt_ctx;

typedef enum {
    VRT_OK = 0,
    VRT_ERR_INVALID_ARG = -1,
    VRT_ERR_OUT_OF_RANGE = -2,
    VRT_ERR_DEVICE = -3,
    VRT_ERR_OOM = -4,
    VRT_ERR_STATE = -5
} vrt_status;

/* Material — clientdesktop/src/graphics/mod.rs:20-28 (32 B). */
typedef struct {
    float color[3];
    uint32_t is_empty;
    uint32_t is_liquid;
    float scatter;
    uint32_t _padding[2];
} vrt_material;

/* CamData — mod.rs:82-91 (160 B). Column-major 4x4 (glam Mat4). */
typedef struct {
    float pos[3];
    uint32_t _padding0;
    float inv_view_mat[16];
    float inv_proj_mat[16];
    float proj_size[2];
    uint32_t _padding1[2];
} vrt_cam_data;

/* WorldData — mod.rs:113-120 (32 B). */
typedef struct {
    int32_t min[3];
    uint32_t size;
    uint32_t size_in_chunks;
    uint32_t _padding[3];
} vrt_world_data;

/* Settings — mod.rs:132-143 (48 B). */
typedef struct {
    uint32_t max_ray_bounces;
    float sun_intensity;
    uint32_t show_step_count;
    uint32_t _padding0;
    float sky_color[3];
    uint32_t _padding1;
    float sun_pos[3];
    uint32_t _padding2;
} vrt_settings;

/* Crosshair — mod.rs:63-70 (32 B). style: 0 off, 1 dot, 2 cross (screen_shader.wgsl:9-13). */
typedef struct {
    float color[4];
    uint32_t style;
    float size;
    uint32_t _padding[2];
} vrt_crosshair;

/* Replaces the arguments of GpuResources::new(gpu, fmt, result_size, max_nodes, world_size)
 * (mod.rs:155-195).  shard_rank/shard_count: this context traces only its share of the 8x8 screen tiles
 * (tile-interleaved multi-GPU sharding; 0/1 = whole frame).  Tiles are dealt out in periods of
 * P = shard_root_weight + shard_count - 1: the first shard_root_weight tiles of every period belong to rank 0, the
 * next ones to ranks 1, 2, ... one each.  shard_root_weight 0 or 1 = equal shares (tile t belongs to rank t % N).
 * A weight > 1 lets the gather root, whose own tiles never cross a link, take more of the frame than the ranks
 * whose tiles all arrive over one xGMI link each. */
#define VRT_MAX_DEVICES 16
typedef struct {
    uint32_t max_nodes;          /* NodeBuffer capacity in nodes (shader.rs:9-16; forced even); 2 .. 2^31 - 2^17 */
    uint32_t world_size_chunks;  /* S: chunk_roots holds S^3 entries (shader.rs:59,67) */
    uint32_t width, height;      /* result texture size, any non-zero size (main.rs:257-262: 1080 rows at the window's
                                  * aspect).  As in the reference, width/8 x height/8 tiles of 8x8 pixels are traced
                                  * (main.rs:452, integer division); the pixels beyond them stay zero, alpha included */
    int32_t device;              /* HIP device ordinal; -1 = current */
    uint32_t shard_rank, shard_count;
    uint32_t flags;              /* VRT_FLAG_* */
    uint32_t shard_root_weight;  /* tiles per period dealt to rank 0; 0 = 1 (a multi-device context: 0 = a default for n_devices) */
    /* One context, one thread, N devices (the reference drives one GpuResources from the winit thread, main.rs:398-455):
     * n_devices > 1 makes this a multi-device context over device_ids[0 .. n_devices) — the same ordinal may appear more
     * than once (a one-GPU rehearsal).  Every write is replicated to all of them (the scene is read-only during a frame),
     * vrt_render traces the frame's 8x8 tiles interleaved over the devices — device_ids[0] its share straight into the
     * row-major frame, the others as 8-byte records stored over xGMI directly into device_ids[0]'s memory
     * (hipDeviceEnablePeerAccess; no collective library, no second process) — and device_ids[0] shades those records into the
     * frame.  The frame, read-backs, vrt_present and vrt_device_output are device_ids[0]'s.  shard_rank / shard_count must be
     * 0 / 0-1; of the flags VRT_FLAG_TEXEL_MESSAGES, VRT_FLAG_STAGED_MESSAGES and VRT_FLAG_POISON_MESSAGES apply.  n_devices
     * 0 or 1: `device` alone, as before. */
    uint32_t n_devices;
    int32_t device_ids[VRT_MAX_DEVICES];
} vrt_config;

/* Write the output in the tile-major shard layout even with shard_count = 1 (a one-rank gather pipeline). */
#define VRT_FLAG_TILE_MAJOR 1u
/* A sharded context that writes its tiles at their final positions of a row-major full-frame buffer instead of a
 * compact tile-major one: the gather root rendering straight into the frame (vrt_assemble then skips its tiles). */
#define VRT_FLAG_ROW_MAJOR 2u
/* A tile-major shard context whose buffer travels over a link: store 8 bytes per pixel slot instead of the 16-byte
 * texel — {id word | bit 23 (norm.y < 0), water_dist as f32} — which is all the gather root needs, beside the frame's
 * uniforms it holds anyway, to shade the pixel itself bit for bit (vrt_assemble_compact).  Plain primary(+shadow)
 * frames with the default march only; vrt_read_output is not available on such a context. */
#define VRT_FLAG_COMPACT 4u

/* A multi-device context (vrt_config.n_devices > 1) whose other devices send whole 16-byte texels instead of 8-byte
 * records: twice the bytes over every link, but every kind of frame — also the path trace and the non-default marches,
 * whose pixels the root cannot re-shade from an id word. */
#define VRT_FLAG_TEXEL_MESSAGES 8u
/* A multi-device context whose other devices render their messages into a buffer of their own and copy it to device_ids[0]
 * afterwards (hipMemcpyPeerAsync on the sender's stream) instead of storing into device_ids[0]'s memory directly.  Taken
 * by itself for every device whose peer access to device_ids[0] is refused (hipDeviceEnablePeerAccess); the flag forces
 * it for all of them (tests, and hosts on which peer stores are slower than a copy engine). */
#define VRT_FLAG_STAGED_MESSAGES 16u
/* Testing: device_ids[0] overwrites a message slot with 0xFF bytes as soon as it has assembled the slot's frame, so a frame
 * assembled from a slot that its senders have not written again yet — a missing wait, a stale read — is visibly wrong. */
#define VRT_FLAG_POISON_MESSAGES 32u

typedef enum {
    VRT_MODE_PRIMARY = 0,        /* the reference's live shader (ray_tracer.wgsl) */
    VRT_MODE_PRIMARY_SHADOW = 1, /* + 1 shadow ray per solid hit, from a compacted hit buffer */
    VRT_MODE_PATH = 2            /* multi-bounce path trace after path_tracer.wgsl (build-defined) */
} vrt_mode;

typedef struct {
    uint32_t mode;     /* vrt_mode */
    uint32_t variant;  /* kernel variant: 0 = default (grid march over the derived cell grid / brick pool; primary +
                        * shadow fused into one launch), 1 = literal octree walk (the shader's text), 2 = ancestor-cache
                        * octree walk, 3 = grid march with the shadow rays as a second launch; DESIGN.md §Kernels */
    uint32_t stats;    /* 1: also count steps / node visits this frame (slower; not for timing); 2: clock probe — the default
                        * primary + shadow kernel with s_memtime / s_memrealtime stamps around one wave in sixteen
                        * (vrt_stats.clock_*; the frame itself is the normal one) */
    uint32_t spp;      /* VRT_MODE_PATH only */
    uint32_t seed;     /* VRT_MODE_PATH only */
    uint32_t flags;    /* VRT_RENDER_* */
    uint32_t _reserved[2];
} vrt_render_opts;

/* Run this frame on the context's own in-flight streams although the caller has set a stream (vrt_set_stream) and / or
 * bound an output (vrt_bind_output).  The caller promises that nothing it enqueues on its stream reads this frame's
 * output before vrt_synchronize (or a device-wide synchronise), and that frames in flight are bound to different
 * buffers.  For a gather root that renders its own tiles in place: they never feed the collective. */
#define VRT_RENDER_OWN_STREAMS 1u
/* This frame's launches carry timing events (vrt_stats.frames / sum_ms_*) whatever the context's sampling of plain frames
 * is (every 8th: a launch with events costs the host three times one without). */
#define VRT_RENDER_TIMED 2u

/* New relative to the reference (it presents to a swapchain and never reads back). */
typedef struct {
    uint64_t primary_rays;
    uint64_t secondary_rays;      /* shadow / bounce rays actually launched */
    uint64_t hits;
    uint64_t steps;               /* valid when the frame was rendered with opts.stats = 1 */
    uint64_t node_visits;
    uint64_t primary_steps;
    uint64_t primary_node_visits;
    float ms_total;               /* hipEvent time of the last frame's kernels, same stream */
    float ms_primary;             /* the march kernel over primary rays */
    float ms_secondary;           /* shadow / bounce kernels */
    uint32_t frames;              /* frames *timed* since the previous vrt_get_stats — every stats frame, and every 8th plain */
    double sum_ms_primary;        /* frame (the first included; a launch that carries timing events costs the host three times */
    double sum_ms_secondary;      /* one that does not) — and their summed kernel times (events stamped by the dispatches) */
    double sum_ms_total;
    uint64_t clock_shader_ticks;  /* clock-probe frames (opts.stats = 2) since the previous vrt_get_stats: summed s_memtime */
    uint64_t clock_ref_ticks;     /* ... and s_memrealtime (100 MHz) differences: shader clock = 100 MHz x their ratio */
} vrt_stats;

/* Per-pixel id word written next to the f32 radiance (build-defined; bit-exact parity target):
 * bits 0..14 voxel id; 16 hit; 17..19 norm.x/y/z != 0; 20 water overlay; 21 shadow ray launched;
 * 22 shadow ray occluded. */
#define VRT_ID_VOXEL_MASK 0x7FFFu
#define VRT_ID_HIT (1u << 16)
#define VRT_ID_NX (1u << 17)
#define VRT_ID_NY (1u << 18)
#define VRT_ID_NZ (1u << 19)
#define VRT_ID_WATER (1u << 20)
#define VRT_ID_SHADOW_RAY (1u << 21)
#define VRT_ID_SHADOWED (1u << 22)

/* GpuResources::new + Buffers::new + PixelShader::new (mod.rs:155-195, shader.rs:55-72,301-344). */
int vrt_create(const vrt_config *cfg, vrt_ctx **out);
void vrt_destroy(vrt_ctx *ctx);

/* Text of the last error on this context (ctx may be NULL: last creation error). */
const char *vrt_last_error(const vrt_ctx *ctx);

/* NodeBuffer::write(gpu, src_nodes (the whole pool), start..end) — shader.rs:22-40.
 * `pool` points at node 0 of the host pool; [start,end) is widened to even bounds exactly as the
 * reference does, so pool must be readable on [start&~1, end+(end&1)). */
int vrt_write_nodes(vrt_ctx *ctx, const uint16_t *pool, uint32_t start, uint32_t end);

/* ArrayBuffer<NodeAddr>::write(gpu, offset, items) — shader.rs:133-142; silently truncates to the
 * buffer's capacity like the reference. */
int vrt_write_chunk_roots(vrt_ctx *ctx, uint32_t offset, const uint32_t *roots, uint32_t n);

/* The same with the caller's word for "which table this is": the reference rewrites the table every frame (main.rs:446),
 * nearly always unchanged, and at 32^3 chunks the backend's compare of 128 KB is half a frame's host time.  tag != 0 that
 * equals the tag of the previous call (same offset and n) returns at once — the caller vouches that the contents are
 * the same — any other value is an ordinary write that remembers the tag (a host world's generation counter, bumped by
 * create_chunk / center_chunks / resize; include/vrt_host.h: vrth_world_chunk_roots_generation).  tag 0 = no tag. */
int vrt_write_chunk_roots_tagged(vrt_ctx *ctx, uint32_t offset, const uint32_t *roots, uint32_t n, uint64_t tag);

/* Buffers::resize_chunk_buffer(world_size) + recreate_bind_group — shader.rs:74-80, main.rs:441-445.
 * Contents are undefined afterwards (a fresh buffer in the reference). */
int vrt_resize_world(vrt_ctx *ctx, uint32_t world_size_chunks);

/* SimpleBuffer<[Material;256]>::write_slice(first, mats) — shader.rs:108-115, main.rs:219-223. */
int vrt_write_materials(vrt_ctx *ctx, uint32_t first, const vrt_material *mats, uint32_t n);

/* SimpleBuffer<T>::write — shader.rs:101-106; callers main.rs:428,439,447-449. */
int vrt_set_camera(vrt_ctx *ctx, const vrt_cam_data *cam);
int vrt_set_settings(vrt_ctx *ctx, const vrt_settings *settings);
int vrt_set_world(vrt_ctx *ctx, const vrt_world_data *world);

/* GpuResources::resize_result_texture — mod.rs:201-211. */
int vrt_resize_output(vrt_ctx *ctx, uint32_t width, uint32_t height);

/* PixelShader::encode_pass(encoder, tex_size/8) + queue.submit — shader.rs:371-379, main.rs:452-453,565.
 * Asynchronous: enqueues the frame's kernels on the context's stream. opts NULL = primary, default. */
int vrt_render(vrt_ctx *ctx, const vrt_render_opts *opts);

/* How many frames the context keeps in flight (1..4, default 2 — a swapchain gives the reference's wgpu path the same):
 * consecutive vrt_render calls of plain frames (default march, no stats, the context's own stream and output buffers;
 * also path-trace frames) alternate between that many internal streams, each with its own output (and path) buffers, so
 * one frame's tail overlaps the next one's ramp-up.  Every other call waits for all of them first; vrt_read_output / vrt_present /
 * vrt_device_output refer to the most recent frame.  1 = strictly one frame at a time; while the view is at rest (camera,
 * settings, world and materials unchanged since the previous frame) the launch's tail is then shortened from within:
 * primary + shadow frames launch their 8x8 tiles longest first, in the order of the march-loop trips the view's second frame
 * noted.  While the camera moves (and nothing else changes) a frame of up to 40 000 tiles notes its trips once in a while, one
 * small launch behind it sorts blocks of 4 x 4 tiles by their trips dilated over 5 blocks, and that order is kept for the frames
 * that follow until the camera leaves what the dilation covers (1080p at 70 degrees: ~ 9 degrees, 6.5 voxels); a view that leaps
 * stops asking.  Any other change of the view returns to screen order.  The frame is the same whatever the order;
 * VRT_TILE_ORDER=0 keeps screen order always, VRT_TILE_ORDER_MOVING=0 while the camera moves (profiles/r05_tile_order_moving.txt). */
int vrt_set_frames_in_flight(vrt_ctx *ctx, uint32_t n);

/* Block until everything enqueued on the context's stream has finished. */
int vrt_synchronize(vrt_ctx *ctx);

/* Copy the last frame to host memory (synchronises). Any pointer may be NULL.
 * rgb: width*height*3 f32 row-major; ids: width*height id words; rgba8: what textureStore would have
 * put in the reference's rgba8unorm texture (ray_tracer.wgsl:179).
 * With shard_count > 1 only this context's tiles are defined; the rest reads as zero. */
int vrt_read_output(vrt_ctx *ctx, float *rgb, uint32_t *ids, uint8_t *rgba8);

/* ScreenShader::encode_pass (shader.rs:273-293, main.rs:454) into host memory: for every pixel of a screen_w x
 * screen_h target — any size: the reference renders its 1080 rows into whatever the window is (main.rs:175,206-210) —
 * fs_main of screen_shader.wgsl:43-65: the rgba8unorm result texture sampled at the pixel centre through the reference's
 * sampler (texture.rs:31-44: ClampToEdge, mag Nearest / min Linear, lod clamped to [1, 1] — a level of detail of 1
 * selects the minification filter, so the sample is bilinear at every window size and the texel itself at 1:1) blended
 * with the crosshair, written as unorm8 RGBA, screen_w*screen_h*4 bytes (synchronises).  Whole-frame contexts only. */
int vrt_present(vrt_ctx *ctx, const vrt_crosshair *crosshair, uint32_t screen_w, uint32_t screen_h, uint8_t *rgba8);

/* The same, left on the device, and asynchronous: the blit is enqueued behind the frame it presents, on that frame's stream
 * (the caller's after vrt_set_stream), so a host that draws and presents frame after frame (main.rs:452-454) keeps its
 * frames in flight — nothing is copied to the host and nothing waits.  *rgba8_device points at screen_w*screen_h*4 bytes of
 * the screen buffer of the frame's set: valid until the present of the frame vrt_set_frames_in_flight frames later (the
 * next one with one frame in flight) — with a declared presentation (vrt_set_presentation) until that frame is RENDERED: its own launch
 * stores into the buffer — or vrt_destroy.  For a host that hands the image to its window system through GPU
 * interop (a 1080p frame is 8 MB of PCIe traffic otherwise).  vrt_synchronize before reading it from another stream. */
int vrt_present_device(vrt_ctx *ctx, const vrt_crosshair *crosshair, uint32_t screen_w, uint32_t screen_h, void **rgba8_device,
                       uint64_t *bytes);

/* The blit folded into the frame's own launch.  The reference's compute pass stores rgba8unorm (ray_tracer.wgsl:179) and its
 * blit follows in the same submission, every frame (main.rs:452-454); here the frame is a 16-byte parity texel per pixel and
 * vrt_present* a second launch that reads it again.  vrt_set_presentation declares what the frames that follow will be presented
 * with: while the window has the texture's size, that size is whole 8x8 tiles, every pixel samples its own texel's centre (to
 * 1e-4 of a texel: 1920x1080, 2560x1440, 3840x2160, ... — found once per size) and every tap of the pixels the crosshair can reach
 * lies in the pixel's own tile, a plain primary / primary + shadow frame of a whole-frame context ALSO stores the window's
 * pixel — fs_main's byte: the texel quantised, the crosshair blended in — into the screen buffer of its frame set, and
 * vrt_present_device / vrt_present with the declared crosshair and size after such a frame launch nothing.  Every other frame,
 * window size or crosshair takes the blit's own launch as before: the declaration never changes a result, only who stores it.
 * flags: VRT_PRESENT_SKIP_TEXELS — such a frame stores the window's pixel ONLY (4 bytes per pixel written instead of 20): a
 * host that presents and never reads back.  vrt_read_output, vrt_present* with another crosshair or size and vrt_assemble
 * sources then find no texels for that frame and return VRT_ERR_STATE.  crosshair NULL: off (the default). */
#define VRT_PRESENT_SKIP_TEXELS 1u
int vrt_set_presentation(vrt_ctx *ctx, const vrt_crosshair *crosshair, uint32_t screen_w, uint32_t screen_h, uint32_t flags);

int vrt_get_stats(vrt_ctx *ctx, vrt_stats *out);

/* New relative to the reference: what ISSUING a frame costs the host, averaged over the vrt_render calls since the previous call
 * of this function (wall-clock microseconds of the calling thread; nothing here waits for a device).  A multi-GPU frame period
 * cannot fall below it, and it cannot be read off a frame loop's own clock once the loop is ahead of its GPUs.  For a multi-device
 * context: the calling thread issues for device_ids[0] while one thread per other device issues for that device (all on the
 * calling thread when the ordinals repeat — a one-GPU rehearsal — or with VRT_GROUP_THREADS=0), then waits for them, then
 * enqueues the waits for the messages, the assembly and an event. */
typedef struct {
    uint32_t frames;            /* vrt_render calls averaged (0: none since the previous call; every figure below is then 0) */
    uint32_t devices;           /* 1, or the devices of a multi-device context */
    uint32_t issuing_threads;   /* threads issuing beside the caller (0: the caller issues for every device in turn) */
    uint32_t _reserved;
    double render_us;           /* the whole vrt_render call */
    double root_issue_us;       /* multi-device: the caller's own issue for device_ids[0] */
    double shard_issue_us_mean; /* ... one other device's issue (its thread's job, or its turn on the caller), mean over devices */
    double shard_issue_us_max;  /* ... the slowest of them, per frame */
    double join_wait_us;        /* ... the caller waiting for the issuing threads after its own issue (0 without threads) */
    double tail_us;             /* ... behind the join: [the stream waits for the messages,] the assembly's launch, the event record */
    double message_waits_us;    /* ... of tail_us, the N - 1 stream waits for the messages when the caller makes them (no issuing threads:
                                 * with threads every thread makes its own, inside its job) */
} vrt_issue_profile;
int vrt_get_issue_profile(vrt_ctx *ctx, vrt_issue_profile *out);

/* Self-test of the kernels' exact-arithmetic shortcuts (csrc/vrt_march.h: division and square root without the general
 * case's scaling and special-value handling, taken when every operand's magnitude is in [2^-30, 2^30]): n pseudo-random
 * operand sets (seed) on `device`, each computed both ways; *mismatches = results whose bits differ (0 on a correct
 * build).  Diagnostic only; no context needed. */
int vrt_selftest_exact_math(int32_t device, uint32_t n, uint32_t seed, uint64_t *mismatches);

/* New relative to the reference: the lookup tables the default march derives on the device from the node pool
 * and chunk_roots (brought up to date before the next frame: only the chunks a vrt_write_nodes range or a changed
 * vrt_write_chunk_roots slot touched, the whole world after a resize or a write that touches many chunks; DESIGN.md §HBM
 * layout).  `available` = 0 while a rebuild is pending or when the world is too large for them (variant 0 then runs
 * as variant 2).  While edits arrive every frame set in flight keeps its own copy of the tables, brought up to date on its
 * own stream with the chunks dirtied since its last frame (an edit then does not wait for the frame in flight); the
 * figures below are those of the copy the last frame used, vrt_read_accel returns the first copy, brought up to date. */
typedef struct {
    uint32_t available;
    uint32_t world_size_chunks;
    uint64_t cells;        /* depth-3 cells in the grid: (8S)^3, 4 B each (+ a zero border on the device) */
    uint64_t bricks;       /* bricks of the pool in use (64 x 2 B each): every chunk's region — its split cells plus slack
                            * for edits — and the regions of chunks that outgrew theirs */
    uint64_t bytes;
    uint32_t builds;       /* whole-world builds since vrt_create (first frame, resized / recentred grid) */
    float last_build_ms;   /* hipEvent time of the last whole-world build */
    uint32_t chunk_builds; /* chunks rebuilt alone since vrt_create: a vrt_write_nodes range or a changed chunk_roots slot
                            * rebuilds only the chunks it touches, stream-ordered, with no host round trip (every copy of the
                            * tables rebuilds every dirtied chunk once: the count of the copy that has rebuilt most) */
    uint32_t ordered_frames; /* (not about the tables) frames since vrt_create whose tiles were launched longest first: one-frame-at-a-time
                            * contexts, a view at rest or a view a camera step away from the frame before */
} vrt_accel_info;
int vrt_get_accel_info(vrt_ctx *ctx, vrt_accel_info *out);

/* Copy the tables to host memory for inspection (synchronises): grid[cells] entries, x-major over the whole world
 * (lo = leaf size - 1; air leaf: 0xFF800000 | lo; other leaf: voxel << 16 | lo; split depth-3 cell: 0x80000000 | brick * 64 < 0xFF800000),
 * bricks[bricks * 64] entries ((x&3) | (y&3) << 2 | (z&3) << 4 inside the cell; voxel << 1 | lo, lo = 1 for a size-2
 * leaf).  Either pointer may be NULL. */
int vrt_read_accel(vrt_ctx *ctx, uint32_t *grid, uint16_t *bricks);

/* The march cells the path trace's bounce launches read (one 16-byte entry per depth-3 cell: the cell grid's entry, the
 * size-2 mask of a split cell — bit (u >> 1) & 31, u = (x&3) | (y&3) << 2 | (z&3) << 4 —, and 64 bits "a ray passes voxel
 * u": air, or a liquid of the material table they were built with), gathered through their chunk directory and blocks
 * into cells[(8S)^3][4], x-major over the whole world, for inspection (synchronises).  *direct = 1 when the world is small
 * enough to be kept without a directory.  VRT_ERR_STATE when the tables are not up to date or this world keeps none. */
int vrt_read_march_cells(vrt_ctx *ctx, uint32_t *cells, uint32_t *direct);

/* Per-pixel march-loop iteration counts of the last frame (primary | shadow << 16); the frame must
 * have been rendered with opts.stats = 1.  Numeric twin of the reference's F2 step-count heat-map
 * (main.rs:368-370, ray_tracer.wgsl:311-314). */
int vrt_read_steps(vrt_ctx *ctx, uint32_t *steps);

/* ---- device-side plumbing for a host that owns streams / device memory (torch, RCCL) ---- */

/* Use the caller's hipStream_t for all subsequent work (NULL = the context's own stream). */
int vrt_set_stream(vrt_ctx *ctx, void *hip_stream);

/* The device output is one 16-byte texel per pixel slot: {r, g, b as f32, id word as u32}.
 * Unsharded: texel[height][width] row-major.  Sharded: the compact tile-major shard buffer
 * texel[tiles_padded][64] (tile t_local <-> the context's t_local-th screen tile in increasing order; pixel p of a
 * tile = (p&7, p>>3)); tiles_padded = ceil(total_tiles / P) is the largest count of any rank >= 1, so every gathered
 * message has the same size (with equal shares also rank 0's). */

/* Render into caller-owned device memory (e.g. a torch tensor handed to an RCCL gather) instead of the
 * context's own buffer: `texels` must hold the byte count vrt_device_output reports and be 16-byte
 * aligned.  NULL restores the context's buffer; vrt_resize_output drops the binding.  With a result size that is
 * not whole 8x8 tiles the frame's last columns / rows are never written (vrt_config.width): a row-major caller buffer
 * should start out zero, as the context's own buffers do, and the same goes for the dst of vrt_assemble*. */
int vrt_bind_output(vrt_ctx *ctx, void *texels);

/* Device pointer and size in bytes of the buffer frames are currently written to. */
int vrt_device_output(vrt_ctx *ctx, void **texels, uint64_t *bytes);

/* Number of 8x8 tiles this context traces, the padded per-rank count used for equal-sized gathers
 * (ceil(total_tiles / P)) and the total. */
int vrt_shard_info(vrt_ctx *ctx, uint32_t *tiles_local, uint32_t *tiles_padded, uint32_t *tiles_total);

/* On the gather root: scatter the shard_count gathered tile-major buffers (rank r's texels start at
 * gathered + r*rank_stride_bytes; 0 = densely packed, tiles_padded*64*16 bytes apart) into the
 * row-major device frame dst = texel[height][width].  A root created with VRT_FLAG_ROW_MAJOR has already written
 * its own tiles there: slot 0 of `gathered` is then ignored.  Asynchronous on the context's stream. */
int vrt_assemble(vrt_ctx *ctx, const void *gathered, uint64_t rank_stride_bytes, void *dst);

/* The same for VRT_FLAG_COMPACT messages (8 bytes per pixel slot; stride 0 = tiles_padded*64*8 bytes apart): the root —
 * a VRT_FLAG_ROW_MAJOR shard context holding the same camera / settings / world / materials as the senders — shades
 * every gathered pixel with the code the sender would have run, under the uniforms current at this call, and writes the
 * texel into dst. */
int vrt_assemble_compact(vrt_ctx *ctx, const void *gathered, uint64_t rank_stride_bytes, void *dst);

#ifdef __cplusplus
}
#endif
#endif
