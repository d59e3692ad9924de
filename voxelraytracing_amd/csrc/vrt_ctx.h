#pragma once
// vrt_ctx.h — what the translation units of the backend (the C ABI of include/vrt.h over the gfx950 kernels) share:
// the context, the kernels' launchers, and the helpers that cross a file boundary.
//   vrt_frames.hip   contexts, uniforms, frame sets, vrt_render, read-backs, stats
//   vrt_uploads.hip  uploads without draining, which chunks a write touched, the derived tables
//   vrt_present.hip  the presentation blit and the gather root's assembly
//   vrt_group.hip    one context over several devices
//
// Replaces the reference's wgpu seam: GpuResources / Buffers / NodeBuffer / SimpleBuffer /
// ArrayBuffer / PixelShader (clientdesktop/src/graphics/{mod.rs,shader.rs}).  Device memory layout
// (DESIGN.md §HBM layout): the node pool is kept byte-identical to the host pool (little-endian u16 =
// the reference's packed u32 pairs), chunk_roots is a dense u32[S^3], materials 256 x 32 B, output one
// 16-byte texel {r,g,b f32, id u32} per pixel slot, hit buffer 16 B per local pixel.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "vrt_device.h"
#include "vrt_exp.h"

namespace vrt {
bool variant_supported(uint32_t variant);
void launch_primary(const FrameParams &P, uint32_t variant, bool stats, bool shadow, hipStream_t st, hipEvent_t e0, hipEvent_t e1);
void launch_shadow(const FrameParams &P, uint32_t variant, bool stats, hipStream_t st, hipEvent_t e0, hipEvent_t e1);
void launch_primary_shadow_fused(const FrameParams &P, uint32_t march, bool stats, hipStream_t st, hipEvent_t e0, hipEvent_t e1);
void launch_path_primary(const FrameParams &P, bool stats, bool literal, hipStream_t st);
void launch_path_bounce(const FrameParams &P, bool stats, bool literal, hipStream_t st);
void launch_path_bounce_cells(const FrameParams &P, uint32_t refill_at, uint32_t segments, uint32_t pool_batches, hipStream_t st);
void launch_path_finish(Texel *out, uint32_t n, uint32_t spp, hipStream_t st);
void launch_tile_order(const uint32_t *cost, uint32_t n, uint32_t shift, uint32_t *scratch, uint32_t *order, hipStream_t st);
bool launch_tile_order_blocks(const uint32_t *cost, uint32_t tiles_x, uint32_t tiles_y, uint32_t shift, uint32_t radius, uint32_t *order, hipStream_t st, uint32_t threads);
void launch_path_chain_finish(Texel *out, const Texel *acc, uint32_t n, uint32_t chain, bool first, bool last, uint32_t spp, hipStream_t st);
void launch_quantize(const Texel *out, uint8_t *rgba8, uint32_t w, uint32_t h, hipStream_t st);
void launch_assemble(const Texel *gathered, Texel *dst, uint32_t width, uint32_t tiles_x, uint32_t tiles_total,
                     uint32_t root_weight, uint32_t period, bool skip_root, uint64_t rank_stride, hipStream_t st);
void launch_present(const Texel *out, uint32_t w, uint32_t h, uint32_t screen_w, uint32_t screen_h, const vrt_crosshair &ch,
                    uint8_t *rgba8, bool one_to_one, const uint32_t box[4], hipStream_t st);
void launch_assemble_shade(const FrameParams &P, const void *gathered, Texel *dst, uint32_t root_weight, uint32_t period,
                           uint64_t rank_stride, hipStream_t st);
void launch_accel_cells(const uint16_t *nodes, uint32_t n_nodes, const uint32_t *roots, uint32_t S, uint32_t *grid,
                        uint32_t *chunk_bricks, uint32_t *chunk_bases, uint32_t *chunk_caps, uint32_t *total, uint32_t *tail,
                        uint32_t *chunk_needs, uint32_t *dir, uint32_t *block_tail, uint32_t *total_blocks, hipStream_t st);
void launch_accel_bricks(const uint16_t *nodes, uint32_t n_nodes, const uint32_t *roots, uint32_t S, uint32_t *grid,
                         const uint32_t *chunk_bases, uint16_t *bricks, uint32_t brick_cap, uint32_t *dir, uint4 *blocks, uint32_t *block_tail,
                         uint32_t block_cap, const uint32_t liquid[8], hipStream_t st);
void launch_upload_words(void *dst, const void *pinned_src, uint32_t n_words, hipStream_t st);
void launch_upload_batch(void *dst0, void *dst1, const void *pinned_ring, const UploadBatch &batch, uint32_t n_pieces, hipStream_t st);
void launch_accel_chunks(const uint16_t *nodes, uint32_t n_nodes, const uint32_t *roots, uint32_t S, uint32_t *grid,
                         uint32_t *chunk_bricks, uint32_t *chunk_bases, uint32_t *chunk_caps, uint32_t *tail, uint16_t *bricks,
                         uint32_t brick_cap, uint32_t *dir, uint4 *blocks, uint32_t *block_tail, uint32_t block_cap, const uint32_t liquid[8],
                         const uint32_t *chunks, const uint32_t *extents, const uint32_t *chunk_roots_host, uint32_t n, hipStream_t st, hipEvent_t done);
}  // namespace vrt

// Host-time profile of the frame path (VRT_HOST_PROF=1: printed at process exit): where the calling thread's microseconds per
// frame go, section by section (profiles/r04_group_host_profile.txt).  Off, a section costs one test of a flag.
#include <chrono>
struct vrt_host_prof {
    bool on = getenv("VRT_HOST_PROF") != nullptr;
    double us[24] = {};
    unsigned long long n[24] = {};
    const char *name[24] = {};
    ~vrt_host_prof() {
        if (!on) return;
        for (int i = 0; i < 24; i++)
            if (n[i]) fprintf(stderr, "host-prof %-34s %9llu calls %8.2f us each\n", name[i] ? name[i] : "?", n[i], us[i] / (double)n[i]);
    }
};
extern __attribute__((visibility("hidden"))) vrt_host_prof g_host_prof;
struct vrt_prof_scope {
    int i;
    std::chrono::steady_clock::time_point t0;
    vrt_prof_scope(int i_, const char *name) : i(i_) {
        if (g_host_prof.on) { g_host_prof.name[i] = name; t0 = std::chrono::steady_clock::now(); }
    }
    ~vrt_prof_scope() {
        if (!g_host_prof.on) return;
        g_host_prof.us[i] += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        g_host_prof.n[i] += 1;
    }
};
#define VRT_PROF(i, name) vrt_prof_scope prof_scope_##i(i, name)

static_assert(sizeof(vrt_material) == 32, "Material layout (mod.rs:20-28)");
static_assert(sizeof(vrt_cam_data) == 160, "CamData layout (mod.rs:82-91)");
static_assert(sizeof(vrt_world_data) == 32, "WorldData layout (mod.rs:113-120)");
static_assert(sizeof(vrt_settings) == 48, "Settings layout (mod.rs:132-143)");
static_assert(sizeof(vrt_crosshair) == 32, "Crosshair layout (mod.rs:63-70)");
static_assert(sizeof(vrt::Texel) == 16, "texel");

struct vrt_group;
struct vrt_group;

struct vrt_ctx {
    vrt_group *grp = nullptr;   // a multi-device context (vrt_config.n_devices > 1): everything else below is unused, see vrt_group
    hipStream_t last_stream = nullptr;  // the stream the most recent frame was enqueued on
    hipEvent_t wait_before_frame = nullptr;  // set by a multi-device context: the next frame's stream waits for it first (its message slot is free)
    int device = 0;
    hipStream_t own_stream = nullptr;
    // Two frames in flight (what a swapchain gives the reference): consecutive plain frames alternate between the
    // context's two streams, each with its own output buffer, so one frame's tail overlaps the next one's ramp-up
    // instead of the in-order queue's ~5 us hand-over.  Everything else on the context waits for both (quiesce()).
    static constexpr uint32_t kMaxInFlight = 4;
    hipStream_t extra_stream[kMaxInFlight - 1] = {nullptr, nullptr, nullptr};
    vrt::Texel *extra_out[kMaxInFlight - 1] = {nullptr, nullptr, nullptr};
    uint32_t *extra_blk[kMaxInFlight - 1] = {nullptr, nullptr, nullptr};
    uint4 *extra_path[kMaxInFlight - 1] = {nullptr, nullptr, nullptr};                   // path mode: its own path buffers
    unsigned long long *extra_counters[kMaxInFlight - 1] = {nullptr, nullptr, nullptr};  // ... and segment cursors
    uint32_t in_flight = 2;        // vrt_set_frames_in_flight
    bool alt_pending = false;      // frames may still be running on the extra streams
    bool own_pending = false;      // ... or on own_stream while the caller's stream is the context's stream (VRT_RENDER_OWN_STREAMS)
    uint32_t flip = 0;             // which (stream, output, counts) set the next pipelined frame takes
    hipStream_t stream = nullptr;
    // four hipEvents per frame rendered since the last vrt_get_stats.  Primary(+shadow) frames: {begin, end} of the first
    // kernel and {begin, end} of the second, stamped by the dispatches themselves (hipExtLaunchKernel), so the stream
    // carries no marker packets between frames.  Path frames: [0], [1], [3] recorded around the launches.
    std::vector<std::array<hipEvent_t, 4>> ev_pool;
    std::vector<uint8_t> ev_kind;  // EvKind
    size_t ev_used = 0;
    double acc_ms[3] = {0, 0, 0};
    uint32_t acc_frames = 0;

    uint32_t max_nodes = 0;  // even
    uint32_t world_size = 0;
    uint32_t n_roots = 0;
    uint32_t width = 0, height = 0;
    uint32_t shard_rank = 0, shard_count = 1, shard_w0 = 1;
    uint32_t shard_first = 0, shard_run = 1, shard_period = 1;  // vrt_device.h shard_tile()
    bool whole_frame_owner = false;  // device_ids[0] of a multi-device context: its row-major buffer holds the assembled frame
    bool compact = false;     // VRT_FLAG_COMPACT: 8-byte records instead of texels (a sharded, tile-major context whose tiles cross a link)
    bool tile_major = false;  // output layout [t_local][64]: always when sharded, on request (VRT_FLAG_TILE_MAJOR) otherwise
    uint32_t tiles_x = 0, tiles_total = 0, tiles_local = 0, tiles_padded = 0;
    uint32_t slots = 0;  // pixel slots in the output buffer

    uint16_t *d_nodes = nullptr;
    uint32_t *d_roots = nullptr;
    vrt_material *d_mats = nullptr;
    vrt::Texel *d_out = nullptr;    // where frames are written: own_out or caller-bound memory
    vrt::Texel *own_out = nullptr;
    vrt::Texel *last_out = nullptr;  // the buffer holding the most recent frame
    uint32_t *last_blk = nullptr;
    uint4 *d_hits = nullptr;
    uint32_t *d_blk_counts = nullptr;  // hit records per primary workgroup
    uint32_t n_blocks = 0;
    uint32_t n_counts = 0;          // entries of blk_counts the last primary + shadow frame wrote
    uint4 *d_path = nullptr;  // path mode: 2 buffers x 3 planes x (kHitSegments * hit_seg_cap) records, lazily allocated
    unsigned long long *d_counters = nullptr;  // [kCtrCount] stats, then the hit-segment counters
    uint32_t hit_seg_cap = 0;
    uint32_t *d_steps = nullptr;
    unsigned long long *d_clock = nullptr;  // clock-probe frames: {s_memtime ticks, s_memrealtime ticks}, summed until vrt_get_stats
    uint8_t *d_rgba8 = nullptr;
    // vrt_present's targets: one per frame set, written on the stream of the frame that is presented (vrt_present.hip)
    uint8_t *d_screen[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t screen_cap[4] = {0, 0, 0, 0};
    hipStream_t screen_stream[4] = {nullptr, nullptr, nullptr, nullptr};   // the stream of the buffer's last blit

    // derived lookup tables of the grid march (vrt_accel.hip), brought up to date lazily when their inputs changed: the whole
    // world (accel_dirty) or only the chunks a write touched.  One set per frame set in use (tabs[0] always; tabs[k] once
    // frame set k has rendered): a frame in flight reads its own set, so bringing the next frame's set up to date does not
    // have to wait for it — each set keeps its own list of the chunks dirtied since *it* was last brought up to date.
    struct Tables {
        uint32_t *d_grid = nullptr;
        size_t grid_cap = 0;          // entries allocated ([8S][8S+1][8S+1] with the zero border)
        uint16_t *d_bricks = nullptr;
        uint32_t brick_cap = 0;
        // the march cells (vrt_accel.hip): a chunk directory [S][S+1][S+1] and 8-KiB blocks of 512 cells (0: outside the world,
        // 1: shared by the chunks that are one air leaf, the rest: one chunk each; the tail takes chunks that stop being air)
        uint32_t *d_cdir = nullptr;
        size_t cdir_cap = 0;
        uint4 *d_mblk = nullptr;
        uint32_t mblk_cap = 0;        // blocks
        uint32_t *d_mblk_tail = nullptr;
        uint32_t *d_chunk_bricks = nullptr, *d_chunk_bases = nullptr, *d_chunk_caps = nullptr, *d_brick_tail = nullptr;
        uint32_t chunk_cap = 0;
        bool live = false;            // a copy of tabs[0] as of the last whole-world build, plus its own chunk updates since
        std::vector<uint32_t> dirty_chunks;     // chunk slots whose nodes or root changed since this set was last brought up to date
        std::vector<uint8_t> chunk_is_dirty;    // ... as flags, [n_roots]
        std::vector<uint8_t> chunk_may_have_moved;  // rebuilt alone since the last whole-world build: may sit in the pool's tail
        uint32_t chunks_moved = 0;
        uint32_t chunk_builds = 0;              // chunks this set has rebuilt alone (vrt_accel_info reports the most advanced set's)
        hipEvent_t ev_updated = nullptr;        // behind this set's last chunk update (a reader of the node pool and chunk_roots)
        bool update_pending = false;            // ... recorded and not yet known to be over
    };
    Tables tabs[kMaxInFlight];
    // The sets are split — every frame set its own — only while edits arrive: two copies of the tables are twice the lines
    // in every XCD's 4 MB L2 (measured: + 1.0 % on the C2 frame period).  After kQuietFrames frames without a dirtied chunk
    // every frame set reads tabs[0] again; the next edit splits them (one wait for the frames in flight, then copies).
    bool tables_split = false;
    uint32_t quiet_frames = 0;
    bool shared_readers_in_flight = false;   // a frame on another frame set is reading tabs[0] (cleared with the frames in flight)
    uint32_t *d_brick_total = nullptr;
    uint32_t *d_chunk_needs = nullptr;   // whole-world build scratch: which chunks need a block of march cells
    bool march_direct = false;           // the march cells of the whole world, no chunk directory (worlds up to march_direct_max_s)
    uint32_t march_direct_max_s = 0;     // kMarchDirectMaxS, or VRT_MARCH_DIRECT_MAX_S (tests: the directory on a small world)
    uint32_t chunk_needs_cap = 0;
    uint32_t n_bricks = 0;        // bricks inside the chunks' regions after the last whole-world build
    uint32_t accel_S = 0;         // world size the tables were built for
    bool accel_dirty = true;
    bool accel_ok = false;        // false: world too large for the tables, variant 0 runs as variant 2
    uint32_t accel_max_s = 0;     // kAccelMaxS, or less through VRT_ACCEL_MAX_S (tests of the fallback)
    // Every 8th plain frame carries the dispatch-stamped timing events (VRT_TIMING_EVERY=N changes it): a launch with events
    // costs the host 13 us, one without 4 us (tools/host_cost.py) — nothing on one device, the frame period of a
    // multi-device context that issues to eight from one thread.
    uint32_t timing_every = 8, frame_no = 0;
    bool path_pool = true;         // VRT_PATH_POOL=0 / VRT_PATH_CELLS=0: bounce launches with lane = path (the round-1 structure) instead of the
    bool path_cells = true;        // pool kernel over the march cells (tests: the two structures hold each other's frames)
    // VRT_PATH_WINDOW=1 (experiments build; built and measured in round 5, not chosen: profiles/r05_window_*): the bounce launch over
    // LDS-staged windows of march cells (experiments/vrt_path_window.hip).  Per frame set: the regions' counts (two 16-byte planes
    // of per-ray state lie behind the path buffer's six)
    bool path_window = false;
    uint32_t path_window_shape = 2;   // VRT_PATH_WINDOW_SHAPE: 0 = 32^3 voxels, 1 = 48^3, 2 = 64 x 32 x 64, 3 = 64^3, 4 = no window (the rays' state in global memory)
    int32_t path_window_lift = 8;     // VRT_PATH_WINDOW_LIFT: the window's centre above the mean origin, voxels
    uint32_t *path_grp_counts[kMaxInFlight] = {nullptr, nullptr, nullptr, nullptr};
    size_t path_grp_regions[kMaxInFlight] = {0, 0, 0, 0};
    uint32_t path_samples = 8;     // VRT_PATH_SAMPLES_PER_CHAIN: samples a launch chain traces at once when spp > 1 (1: one, as round 1 did)
    vrt::Texel *path_acc[kMaxInFlight] = {nullptr, nullptr, nullptr, nullptr};   // ... their accumulation planes, per frame set
    size_t path_acc_texels[kMaxInFlight] = {0, 0, 0, 0}, path_buf_records[kMaxInFlight] = {0, 0, 0, 0};
    uint32_t path_pool_batches = 0;   // VRT_PATH_POOL_K = 4 | 5: the bounce waves' pools; 0: 5 for small worlds with two frames in flight, else 4
    uint32_t path_refill = 0;      // VRT_PATH_POOL_REFILL: idle lanes that send a bounce wave back to its pool (0: the default, 16)
    uint32_t accel_builds = 0;
    // longest tiles first (vrt_kernels.hip: tile_order_*), for a context that renders one frame at a time
    // (vrt_set_frames_in_flight(1)) and whose view is at rest: the second plain frame of an unchanged view (camera, settings,
    // world, materials) notes its tiles' march-loop trips, right behind it on the stream four small launches turn them into the
    // order the following frames of that view launch their tiles in.  The order is only worth anything for the very view it
    // was made from (a launch's tail is a handful of tiles with grazing rays, and which tiles those are changes with a hundredth
    // of a voxel of camera travel; measured): any change of the view goes back to screen order.  With two
    // frames in flight the other frame already fills a launch's tail and the order buys nothing.
    bool tile_lpt = true;               // VRT_TILE_ORDER=0: screen order always
    uint32_t *d_tile_cost = nullptr, *d_tile_order = nullptr, *d_tile_scratch = nullptr;
    uint32_t tile_buf_tiles = 0;        // what the buffers are sized for
    bool tile_order_valid = false;
    uint32_t view_gen = 0;              // counts the changes of anything a tile's trips depend on
    uint32_t frame_view_gen = ~0u;      // ... as of the last frame rendered
    uint32_t order_view_gen = ~0u;      // ... as of the frame the order was made from
    // ... and while the view MOVES (1, the default): a frame notes its trips, ONE launch behind it sorts blocks of 4 x 4 tiles by
    // their trips dilated over `mov_radius` blocks (vrt_kernels.hip: launch_tile_order_blocks, ~ 17 us), and the order is KEPT for
    // the frames that follow while their camera stays within what the dilation covers (vrt_order.hip, hold_limits: ~ 8 of the bench's orbit steps)
    // and nothing but the camera has changed; the frame that comes near the edge of that notes its trips for the next order.  A
    // view that moves too fast for its orders to be used stops asking for them (mov_backoff).  One frame at a time, orbit:
    // 112.8 -> 109.1 us per frame (profiles/r05_tile_order_moving.txt).  0 (VRT_TILE_ORDER_MOVING=0): screen order while the
    // view moves.
    uint32_t tile_lpt_moving = 1;
    uint32_t order_uses = 0;            // frames that used the dilated order in d_tile_order
    uint32_t mov_backoff = 0, mov_skip = 0;   // moving frames that go without asking for an order (doubles while orders go unused)
    bool mov_any_size = false;          // VRT_TILE_ORDER_MOVING set by name: also frames of more than kMovingTilesMax tiles
    uint32_t mov_radius = 5;            // VRT_TILE_ORDER_RADIUS: blocks the order is dilated over
    // vrt_present*: whether a window of (one_w x one_h) over a texture of the same size samples every texel at its centre
    uint32_t one_w = 0, one_h = 0;
    bool one_to_one = false;
    // vrt_set_presentation: the declared crosshair / window / flags; whether frames of (pres_for_w x pres_for_h) can store their own
    // window pixels (found once per size and declaration) and the crosshair's box; which frame sets' screen buffers hold the
    // image their last frame stored itself; whether the last frame stored one, and whether it stored texels at all
    bool pres_on = false;
    vrt_crosshair pres_ch{};
    uint32_t pres_w = 0, pres_h = 0, pres_flags = 0;
    uint32_t pres_for_w = 0, pres_for_h = 0, pres_box[4] = {0, 0, 0, 0};
    bool pres_fusable = false;
    bool last_fused = false, last_has_texels = true;
    uint32_t ordered_frames = 0;        // frames launched in an order (vrt_accel_info.ordered_frames)
    bool order_dilated = false;         // the order in d_tile_order is a dilated one
    uint32_t cam_gen = 0, order_cam_gen = 0;   // counts the changes of the camera (each is a change of the view too)
    vrt_cam_data order_cam{};           // the camera of the frame the order was made from
    bool tile_order_stale = false;      // a chunk was edited since the order was made: still used, made again by the next frame without an edit in front of it
    uint32_t frame_mode = ~0u;          // vrt_mode of the last frame rendered (a change of mode is a change of view)
    uint32_t last_slot = 0, last_tab = 0;   // the frame set and the table set of the last frame
    float accel_last_ms = 0.f;
    uint64_t roots_tag = 0;         // vrt_write_chunk_roots_tagged: the caller's tag of the table as last written (0: none)
    uint32_t roots_tag_offset = 0, roots_tag_n = 0;
    std::vector<uint32_t> h_roots;  // what chunk_roots holds, to recognise the reference's per-frame rewrite of the same table
    std::vector<std::pair<uint32_t, uint32_t>> roots_index;  // (root, chunk slot) sorted by root, roots != 0: which chunk owns a node
    bool roots_index_stale = true;

    // uploads are staged through pinned memory (copy-at-call semantics without waiting for the device) and ordered with
    // the frames in flight by events, not by draining them
    uint8_t *h_ring = nullptr, *d_ring = nullptr;   // the pinned ring, and where the device sees it
    static constexpr size_t kRingSegBytes = 1u << 20, kRingSegs = 8;
    hipEvent_t ring_ev[kRingSegs][2] = {};   // behind a segment's last copy on c->stream [0] / the upload stream [1]
    bool ring_ev_used[kRingSegs][2] = {};
    uint32_t ring_seg = 0;
    size_t ring_off = 0;
    // node-pool / chunk_roots uploads staged since the last flush: copied into the ring at call time, launched together — one
    // copy kernel for all of them — before the next thing that reads those buffers (a frame's table update, a whole-world
    // build, a synchronise).  The reference drains every pending GiveChunkData per frame (main.rs:289-295): a launch per
    // range made the frame loop host-bound beyond two uploads per frame (27 us each).
    struct Staged { uint32_t buf, dst_word, n_words; size_t ring_at; };
    std::vector<Staged> staged;
    size_t staged_bytes = 0;
    bool flushed_at_call = false;     // a staged range went out at its vrt_write_* call since the last frame (the device was idle): the next ones wait for the batch
    bool staged_seg[kRingSegs] = {};  // ring segments the staged ranges lie in (their events are recorded at the flush)
    hipEvent_t ev_frames = nullptr;   // scratch: "everything enqueued on that frame stream so far"
    hipEvent_t ev_upload = nullptr;   // the last upload / table rebuild on c->stream
    // Node-pool and chunk_roots uploads have a stream of their own: frame set 0 runs on c->stream, and an upload queued
    // behind a frame there would wait for it.  What reads those two buffers — the table updates, and frames that walk the
    // octree itself (variants 1 / 2, worlds beyond the tables) — is what an upload waits for, nothing else.
    hipStream_t up_stream = nullptr;
    hipEvent_t ev_pool_upload = nullptr;      // the last upload on up_stream
    uint64_t pool_gen = 0;                    // bumped by every upload on up_stream
    uint64_t seen_pool_gen[kMaxInFlight + 1] = {0, 0, 0, 0, 0};  // [slot] of the frame streams as seen_gen, [kMaxInFlight] c->stream
    bool walkers_in_flight = false;           // a frame that reads the node pool has been enqueued since the last full synchronise
    hipEvent_t ev_walkers = nullptr;
    uint64_t upload_gen = 0;          // bumped by every upload; a frame stream waits for ev_upload when it has not seen it
    uint64_t seen_gen[kMaxInFlight] = {0, 0, 0, 0};  // [0] own_stream, [k] extra_stream[k - 1]

    float *d_ndc = nullptr;       // ndc_x[width] then ndc_y[height] (FrameParams), rebuilt when proj_size or the output size change
    uint32_t ndc_w = 0, ndc_h = 0;
    float ndc_proj[2] = {0.f, 0.f};

    vrt_material h_mats[256];
    uint32_t liquid_mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // bit v <=> h_mats[v].is_liquid == 1 (kept by vrt_write_materials)
    bool liquid_is_range = true;                          // the liquid ids are one range below 255, or none
    uint32_t liquid_lo = 0x80000000u, liquid_span = 0u;   // (none: no 15-bit voxel id is 0x80000000)
    vrt_cam_data cam;
    vrt_settings settings;
    vrt_world_data world;

    // vrt_get_issue_profile: the calling thread's time inside vrt_render since the last call of it
    double prof_render_us = 0.0;
    uint32_t prof_frames = 0;

    uint32_t last_spp = 1;
    bool rendered = false;
    bool last_stats = false;
    uint32_t last_mode = 0;
    bool timing_pending = false;
    vrt_stats stats;

    std::string err;
};

static constexpr size_t kSegBytes = (size_t)vrt::kHitSegments * vrt::kSegStride * sizeof(uint32_t);
// three sets of segment cursors: launch g of a path frame appends to set g % 3, reads set (g - 1) % 3 and clears set
// (g + 1) % 3 for its successor, so no memset sits between two launches
static constexpr size_t kCounterBytes = vrt::kCtrCount * sizeof(unsigned long long) + 3 * kSegBytes;   // + the path trace's 3 cursor sets

enum EvKind : uint8_t { kEvNone = 0, kEvOneKernel = 1, kEvTwoKernels = 2, kEvRecorded = 3 };

// Largest world the grid march's tables cover: the cell grid is addressed by a 32-bit byte offset built with signed
// 24-bit multiplies (8S (8S+1)^2 * 4 B < 2^31, (8S+1)^2 * 4 < 2^23), bricks by brick * 128 B < 2^32.
static constexpr uint32_t kAccelMaxS = 100;
// ... the march cells (16 bytes per cell) are addressed the same way: 8S (8S+1)^2 * 16 B < 2^31
static constexpr uint32_t kMarchBlocksMax = 1u << 18;   // 2 GiB of march-cell blocks (byte offsets stay below 2^31)
// Small worlds skip the directory: the cells of the whole world, [4S][4S+1][4S+1] lines of 2 x 2 x 2 cells (one dependent
// load and a divergent branch less per change of chunk: 17.5 against 16.1 Grays/s on C4).  16^3 chunks: 35 MB; larger
// worlds go through the directory (32^3: 23 MB instead of 273).
static constexpr uint32_t kMarchDirectMaxS = 16;
static constexpr uint32_t kAccelMaxBricks = 0x1FE0000u - 1u;   // 0x80000000 | brick * 64 stays below vrt::kAirLeaf
// Chunks that can be rebuilt alone between two whole-world builds: each may move, once, into a 512-brick region
// (64 KiB) at the tail of the brick pool.
static constexpr uint32_t kTailChunks = 128;
// A write that touches more chunks than this is cheaper as a whole-world build.
static constexpr uint32_t kMaxDirtyChunks = 256;

// ---- helpers that cross a translation unit (hidden: the library exports include/vrt.h and nothing else) ----
#define VRT_HIDDEN __attribute__((visibility("hidden")))
extern VRT_HIDDEN thread_local std::string g_create_err;
VRT_HIDDEN int fail(vrt_ctx *ctx, int code, const char *fmt, ...);
#define HIP_TRY(ctx, expr)                                                                          \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return fail(ctx, e_ == hipErrorOutOfMemory ? VRT_ERR_OOM : VRT_ERR_DEVICE, "%s: %s", #expr, \
                        hipGetErrorString(e_));                                                     \
    } while (0)

// Multi-device entry points switch the calling thread's current HIP device; the caller gets its own back.
struct DeviceRestore {
    int dev = -1;
    DeviceRestore() { if (hipGetDevice(&dev) != hipSuccess) dev = -1; }
    ~DeviceRestore() { if (dev >= 0) (void)hipSetDevice(dev); }
};

VRT_HIDDEN int quiesce(vrt_ctx *c);   // wait for the frames that may still be running on the context's other streams
#define QUIESCE(c)                     \
    do {                               \
        const int q_ = quiesce(c);     \
        if (q_) return q_;             \
    } while (0)

// vrt_frames.hip
VRT_HIDDEN bool ragged_output(const vrt_ctx *c);
VRT_HIDDEN hipError_t zero_now(vrt_ctx *c, void *p, size_t bytes);
VRT_HIDDEN int validate_frame(vrt_ctx *c);
VRT_HIDDEN int ensure_ndc(vrt_ctx *c);
VRT_HIDDEN void fill_uniforms(const vrt_ctx *c, vrt::FrameParams &P);
// vrt_present.hip: whether this frame can store its own window pixels (vrt_set_presentation) and, then, its screen buffer
VRT_HIDDEN bool presentation_fusable(vrt_ctx *c);
VRT_HIDDEN int screen_buffer_for_frame(vrt_ctx *c, uint32_t slot, hipStream_t st, uint32_t screen_w, uint32_t screen_h);
// vrt_order.hip: the order a one-frame-at-a-time context launches its tiles in, around the frame's launch in vrt_render
struct TileOrderPlan { bool sort = false, dilate = false; };
VRT_HIDDEN int tile_order_before_frame(vrt_ctx *c, vrt::FrameParams &P, hipStream_t st, const vrt_render_opts &o, uint32_t variant, bool kstats,
                                       bool edit_in_front, TileOrderPlan &plan);
VRT_HIDDEN int tile_order_after_frame(vrt_ctx *c, const vrt::FrameParams &P, hipStream_t st, const TileOrderPlan &plan);
// vrt_uploads.hip
VRT_HIDDEN int alloc_roots(vrt_ctx *c, uint32_t world_size);
VRT_HIDDEN int flush_staged(vrt_ctx *c);
VRT_HIDDEN int order_after_frames(vrt_ctx *c, hipStream_t target);
VRT_HIDDEN int order_after_frames(vrt_ctx *c);
VRT_HIDDEN int publish_upload(vrt_ctx *c);
VRT_HIDDEN int wait_for_pool_uploads(vrt_ctx *c, hipStream_t st, uint32_t slot);
VRT_HIDDEN int frame_waits_for_uploads(vrt_ctx *c, hipStream_t st, uint32_t slot);
VRT_HIDDEN int stage_upload(vrt_ctx *c, void *dst, const void *src, size_t bytes, bool pool = false);
VRT_HIDDEN void mark_all_dirty(vrt_ctx *c);
VRT_HIDDEN int free_tables(vrt_ctx *c, vrt_ctx::Tables &T);
VRT_HIDDEN int ensure_accel_world(vrt_ctx *c);
VRT_HIDDEN int update_tables(vrt_ctx *c, uint32_t k, hipStream_t st);
VRT_HIDDEN size_t chunk_dir_entries(uint32_t S);
VRT_HIDDEN size_t direct_cell_entries(uint32_t S);

// ---- one context over several devices (vrt_config.n_devices > 1): vrt_group.hip ----
// ---------------------------------------------------------------------------------------------------------------------
// One context over several devices (vrt_config.n_devices > 1).
//
// The reference is one process on one thread driving one GpuResources (main.rs:398-455); this keeps that shape for a
// node with several GPUs.  A group is N ordinary contexts, one per entry of device_ids: device 0's is a row-major shard
// root, the others are tile-major shard contexts that write 8-byte records (or texels, VRT_FLAG_TEXEL_MESSAGES).  Every
// upload is replicated (each context stages its own copy); a frame is
//     for r >= 1:  [device r] wait until device 0 has consumed message slot k -> render into device 0's memory
//                             (peer stores over xGMI, hipDeviceEnablePeerAccess) -> record done[r][k]
//     device 0:    render its own tiles straight into frame buffer k (its in-flight stream X) -> on X: wait for every
//                  done[r][k] -> shade / scatter the messages into the frame -> record consumed[k]
// all enqueued from the calling thread without waiting for anything: two message slots and device 0's two frame buffers
// give the same two frames in flight a single device has.  No collective library, no second process.
// ---------------------------------------------------------------------------------------------------------------------
// One issuing thread per device other than the root: the caller stays one thread (the reference's shape), but issuing a
// frame to N devices from it alone costs N x (launch + event record + waits) and makes eight devices host-bound
// (DESIGN.md §7).  A worker sleeps on a condition variable between frames' bursts and spins briefly first, so a frame
// loop finds it awake; every API call joins the workers before it returns — nothing of a context is ever touched by
// two threads at once.
struct GrpWorker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::atomic<uint64_t> posted{0}, finished{0};
    std::function<int()> job;
    int rc = 0;
    bool quit = false;
    double last_job_us = 0.0;   // how long the last job took this thread (read by the caller after join())

    void run() {
        uint64_t seen = 0;
        for (;;) {
            for (int spin = 0; spin < 20000 && posted.load(std::memory_order_acquire) == seen; spin++) __builtin_ia32_pause();
            if (posted.load(std::memory_order_acquire) == seen) {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return quit || posted.load(std::memory_order_acquire) != seen; });
                if (quit) return;
            }
            seen = posted.load(std::memory_order_acquire);
            const auto t0 = std::chrono::steady_clock::now();
            rc = job();
            last_job_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            finished.store(seen, std::memory_order_release);
        }
    }
    void post(std::function<int()> f) {
        job = std::move(f);
        {
            std::lock_guard<std::mutex> lk(m);
            posted.fetch_add(1, std::memory_order_release);
        }
        cv.notify_one();
    }
    int join() {
        const uint64_t want = posted.load(std::memory_order_acquire);
        while (finished.load(std::memory_order_acquire) != want) __builtin_ia32_pause();
        return rc;
    }
    void stop() {
        {
            std::lock_guard<std::mutex> lk(m);
            quit = true;
        }
        cv.notify_one();
        if (th.joinable()) th.join();
    }
};

struct vrt_group {
    std::vector<vrt_ctx *> dev;      // dev[0] = the root
    std::vector<std::unique_ptr<GrpWorker>> workers;   // [r - 1] issues for dev[r]; empty: the calling thread issues for all (VRT_GROUP_THREADS=0)
    bool texels = false;             // VRT_FLAG_TEXEL_MESSAGES
    bool poison = false;             // VRT_FLAG_POISON_MESSAGES
    // [r] device r cannot store into device 0's memory (peer access refused), or VRT_FLAG_STAGED_MESSAGES: it renders into
    // stage[r][slot], a buffer of its own, and copies the message over afterwards (hipMemcpyPeerAsync, its own stream)
    std::vector<uint8_t> staged;
    std::vector<std::array<void *, 2>> stage;
    static constexpr uint32_t kSlots = 2;
    void *recv[kSlots] = {nullptr, nullptr};               // on device 0: [n_devices][tiles_padded * 64] records or texels
    size_t rank_stride = 0;                                // bytes between two devices' messages
    hipEvent_t consumed[kSlots] = {nullptr, nullptr};      // device 0 has assembled the frame of this slot
    bool consumed_used[kSlots] = {false, false};
    std::vector<std::array<hipEvent_t, kSlots>> done;      // [r][slot]: device r's message is complete
    uint32_t slot = 0, in_flight = 2;
    bool last_was_stats = false;
    // The stream device 0's share of the current frame was enqueued on, published by the calling thread once its own vrt_render has
    // returned: every issuing thread then makes that stream wait for ITS message itself (one hipStreamWaitEvent each, in parallel)
    // instead of the caller making N - 1 of them in turn behind the join — they were ~ 6 us each, the part of the caller's frame that
    // grew with N (profiles/r06_bench_modes_rehearsal.txt: 10 / 20 / 48 us at N = 2 / 4 / 8).  nullptr: not yet; kNoFrameStream: the
    // root's render failed, nobody waits for anything.
    std::atomic<hipStream_t> frame_stream{nullptr};
    static inline hipStream_t no_frame_stream() { return reinterpret_cast<hipStream_t>(static_cast<uintptr_t>(1)); }
    // vrt_get_issue_profile: sums over the frames since its last call
    struct { uint32_t frames = 0; double render = 0, root = 0, shard_sum = 0, shard_max = 0, join = 0, tail = 0, waits = 0; } prof;
};

VRT_HIDDEN int grp_create(const vrt_config *cfg, vrt_ctx **out);
VRT_HIDDEN void grp_destroy(vrt_ctx *c);
VRT_HIDDEN int grp_render(vrt_ctx *c, const vrt_render_opts *opts);
VRT_HIDDEN int grp_synchronize(vrt_ctx *c);
VRT_HIDDEN int grp_get_stats(vrt_ctx *c, vrt_stats *out);
VRT_HIDDEN int grp_get_issue_profile(vrt_ctx *c, vrt_issue_profile *out);
VRT_HIDDEN int grp_resize_output(vrt_ctx *c, uint32_t w, uint32_t h);
VRT_HIDDEN int grp_set_frames_in_flight(vrt_ctx *c, uint32_t n);
inline vrt_ctx *grp_root(vrt_ctx *c) { return c->grp->dev[0]; }
template <typename F>
inline int grp_each(vrt_ctx *c, F f) {
    DeviceRestore restore;
    for (vrt_ctx *d : c->grp->dev) {
        const int rc = f(d);
        if (rc) { c->err = d->err; return rc; }
    }
    return VRT_OK;
}
#define GRP_EACH(c, call)                                                   \
    do {                                                                    \
        if ((c) && (c)->grp) return grp_each((c), [&](vrt_ctx *d) { return call; }); \
    } while (0)
#define GRP_ROOT(c, call)                                                   \
    do {                                                                    \
        if ((c) && (c)->grp) { DeviceRestore restore_; vrt_ctx *d = grp_root(c); const int rc_ = call; if (rc_) (c)->err = d->err; return rc_; } \
    } while (0)
#define GRP_REFUSE(c, what)                                                 \
    do {                                                                    \
        if ((c) && (c)->grp) return fail((c), VRT_ERR_STATE, what ": not on a multi-device context (it owns its streams and message buffers)"); \
    } while (0)
