// vrt_backend.hip — the C ABI of include/vrt.h over the gfx950 kernels in vrt_kernels.hip.
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

namespace vrt {
bool variant_supported(uint32_t variant);
void launch_primary(const FrameParams &P, uint32_t variant, bool stats, bool shadow, hipStream_t st, hipEvent_t e0, hipEvent_t e1);
void launch_shadow(const FrameParams &P, uint32_t variant, bool stats, hipStream_t st, hipEvent_t e0, hipEvent_t e1);
void launch_primary_shadow_fused(const FrameParams &P, uint32_t march, bool stats, hipStream_t st, hipEvent_t e0, hipEvent_t e1);
#ifdef VRT_EXPERIMENTS   // measured and rejected structures, kept for tools/ab (make experiments): DESIGN.md section 5
void launch_primary_shadow_persistent(const FrameParams &P, uint32_t *heads, uint32_t n_cus, hipStream_t st, hipEvent_t e0, hipEvent_t e1);
void launch_path_bounce_pool(const FrameParams &P, bool continuations, uint32_t refill_at, uint32_t eject_at, hipStream_t st);
void launch_path_persistent(const FrameParams &P, uint32_t *heads, uint32_t n_cus, hipStream_t st);
#endif
void launch_path_primary(const FrameParams &P, bool stats, bool literal, hipStream_t st);
void launch_path_bounce(const FrameParams &P, bool stats, bool literal, hipStream_t st);
void launch_path_bounce_cells(const FrameParams &P, uint32_t refill_at, hipStream_t st);
void launch_path_finish(Texel *out, uint32_t n, uint32_t spp, hipStream_t st);
void launch_tile_order(const uint32_t *cost, uint32_t n, uint32_t shift, uint32_t *scratch, uint32_t *order, hipStream_t st);
void launch_path_chain_finish(Texel *out, const Texel *acc, uint32_t n, uint32_t chain, bool first, bool last, uint32_t spp, hipStream_t st);
void launch_quantize(const Texel *out, uint8_t *rgba8, uint32_t w, uint32_t h, hipStream_t st);
void launch_assemble(const Texel *gathered, Texel *dst, uint32_t width, uint32_t tiles_x, uint32_t tiles_total,
                     uint32_t root_weight, uint32_t period, bool skip_root, uint64_t rank_stride, hipStream_t st);
void launch_present(const Texel *out, uint32_t w, uint32_t h, uint32_t screen_w, uint32_t screen_h, const vrt_crosshair &ch,
                    uint8_t *rgba8, hipStream_t st);
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
                         const uint32_t *chunks, const uint32_t *extents, uint32_t n, hipStream_t st);
}  // namespace vrt

static_assert(sizeof(vrt_material) == 32, "Material layout (mod.rs:20-28)");
static_assert(sizeof(vrt_cam_data) == 160, "CamData layout (mod.rs:82-91)");
static_assert(sizeof(vrt_world_data) == 32, "WorldData layout (mod.rs:113-120)");
static_assert(sizeof(vrt_settings) == 48, "Settings layout (mod.rs:132-143)");
static_assert(sizeof(vrt_crosshair) == 32, "Crosshair layout (mod.rs:63-70)");
static_assert(sizeof(vrt::Texel) == 16, "texel");

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
    // path mode, straggler chain (launch_path_frame): per frame set [0] = the context's own, [k] = extra set k - 1
    uint4 *path_cont[kMaxInFlight] = {nullptr, nullptr, nullptr, nullptr};        // kContSets x 4 planes
    hipStream_t side_stream[kMaxInFlight] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t side_ev[kMaxInFlight][6] = {};                                     // [0..3] bounce launch done, [4] chain done, [5] frame start
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
    uint32_t *d_heads = nullptr;   // variant 4: the per-XCD queue heads (8 x 64 B)
    uint32_t n_cus = 0;
    uint32_t n_blocks = 0;
    uint32_t n_counts = 0;          // entries of blk_counts the last primary + shadow frame wrote
    uint4 *d_path = nullptr;  // path mode: 2 buffers x 3 planes x (kHitSegments * hit_seg_cap) records, lazily allocated
    unsigned long long *d_counters = nullptr;  // [kCtrCount] stats, then the hit-segment counters
    uint32_t hit_seg_cap = 0;
    uint32_t *d_steps = nullptr;
    unsigned long long *d_clock = nullptr;  // clock-probe frames: {s_memtime ticks, s_memrealtime ticks}, summed until vrt_get_stats
    uint8_t *d_rgba8 = nullptr;
    uint8_t *d_screen = nullptr;   // vrt_present's target
    size_t screen_cap = 0;

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
    bool path_persistent = false;  // VRT_PATH_PERSISTENT=1: plain path frames as one persistent launch instead of one launch per bounce
    bool path_pool = true;         // VRT_PATH_POOL=0: bounce launches with lane = path (the round-1 structure) instead of the pool kernel
    bool path_chain = false;       // VRT_PATH_POOL_CHAIN=1: the pool kernel's stragglers go to a chain of launches on a side stream
    bool path_cells = true;        // VRT_PATH_CELLS=0: the pool kernel over cell grid + bricks instead of the one over the march cells
    uint32_t path_samples = 8;     // VRT_PATH_SAMPLES_PER_CHAIN: samples a launch chain traces at once when spp > 1 (1: one, as round 1 did)
    vrt::Texel *path_acc[kMaxInFlight] = {nullptr, nullptr, nullptr, nullptr};   // ... their accumulation planes, per frame set
    size_t path_acc_texels[kMaxInFlight] = {0, 0, 0, 0}, path_buf_records[kMaxInFlight] = {0, 0, 0, 0}, path_cont_records[kMaxInFlight] = {0, 0, 0, 0};
    uint32_t path_refill = 0, path_eject = ~0u;   // VRT_PATH_POOL_REFILL / _EJECT: the pool kernel's thresholds (experiments; 0 / ~0: defaults)
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
static constexpr uint32_t kContSets = 4;   // straggler-chain record sets of a path frame (one per bounce launch; more bounces: no chain)
static constexpr size_t kCounterBytes = vrt::kCtrCount * sizeof(unsigned long long) + (3 + kContSets) * kSegBytes;   // 3 path cursor sets + the chain's

enum EvKind : uint8_t { kEvNone = 0, kEvOneKernel = 1, kEvTwoKernels = 2, kEvRecorded = 3 };

static thread_local std::string g_create_err;

static int fail(vrt_ctx *ctx, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    else g_create_err = buf;
    return code;
}

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

// Wait for the frame that may still be running on the second stream.
static int quiesce(vrt_ctx *c) {
    if (c->alt_pending) {
        for (hipStream_t st : c->extra_stream)
            if (st) HIP_TRY(c, hipStreamSynchronize(st));
        c->alt_pending = false;
    }
    if (c->own_pending) {
        HIP_TRY(c, hipStreamSynchronize(c->own_stream));
        c->own_pending = false;
    }
    c->shared_readers_in_flight = false;
    return VRT_OK;
}
#define QUIESCE(c)                     \
    do {                               \
        const int q_ = quiesce(c);     \
        if (q_) return q_;             \
    } while (0)

// The reference dispatches tex_size / 8 workgroups per axis (main.rs:452) over a result texture of any size
// (main.rs:257-262: 1080 rows, the window's aspect): the columns and rows beyond the last whole 8x8 tile are never stored to
// and keep the fresh texture's zeros.  Buffers that are read as whole frames start out zero for such a size.
static bool ragged_output(const vrt_ctx *c) { return ((c->width | c->height) & 7u) != 0u; }
static hipError_t zero_now(vrt_ctx *c, void *p, size_t bytes) {
    const hipError_t e = hipMemsetAsync(p, 0, bytes, c->stream);
    return e != hipSuccess ? e : hipStreamSynchronize(c->stream);   // (done before a launch on any other stream can follow)
}

static void layout_tiles(vrt_ctx *c) {
    c->tiles_x = c->width / 8u;
    c->tiles_total = c->tiles_x * (c->height / 8u);
    // tiles are dealt out in periods of P = w0 + N - 1: w0 to rank 0, then one to each of ranks 1..N-1
    const uint32_t P = c->shard_w0 + c->shard_count - 1u;
    c->shard_period = P;
    c->shard_run = c->shard_rank == 0 ? c->shard_w0 : 1u;
    c->shard_first = c->shard_rank == 0 ? 0u : c->shard_w0 + c->shard_rank - 1u;
    const uint32_t full = c->tiles_total / P, rem = c->tiles_total % P;
    c->tiles_padded = (c->tiles_total + P - 1u) / P;
    c->tiles_local = full * c->shard_run + (rem > c->shard_first ? (rem - c->shard_first < c->shard_run ? rem - c->shard_first : c->shard_run) : 0u);
    const uint32_t tm_tiles = c->tiles_local > c->tiles_padded ? c->tiles_local : c->tiles_padded;
    c->slots = c->tile_major ? tm_tiles * 64u : c->width * c->height;
}

static int alloc_output(vrt_ctx *c) {
    (void)hipFree(c->own_out); (void)hipFree(c->d_hits); (void)hipFree(c->d_steps); (void)hipFree(c->d_rgba8); (void)hipFree(c->d_path);
    (void)hipFree(c->d_blk_counts);
    c->own_out = nullptr; c->d_hits = nullptr; c->d_steps = nullptr; c->d_rgba8 = nullptr; c->d_path = nullptr; c->d_blk_counts = nullptr;
    for (auto &p : c->extra_out) { (void)hipFree(p); p = nullptr; }
    for (auto &p : c->extra_blk) { (void)hipFree(p); p = nullptr; }
    for (auto &p : c->extra_path) { (void)hipFree(p); p = nullptr; }
    for (auto &p : c->path_cont) { (void)hipFree(p); p = nullptr; }
    for (auto &p : c->path_acc) { (void)hipFree(p); p = nullptr; }
    (void)hipFree(c->d_tile_cost); c->d_tile_cost = nullptr;
    (void)hipFree(c->d_tile_order); c->d_tile_order = nullptr;
    (void)hipFree(c->d_tile_scratch); c->d_tile_scratch = nullptr;
    c->tile_buf_tiles = 0;
    c->tile_order_valid = false;
    for (auto &n : c->path_acc_texels) n = 0;
    for (auto &n : c->path_buf_records) n = 0;
    for (auto &n : c->path_cont_records) n = 0;
    layout_tiles(c);
    const size_t n = c->slots ? c->slots : 1;
    HIP_TRY(c, hipMalloc(&c->own_out, n * sizeof(vrt::Texel)));
    // hit buffer: kHitSegments segments, each able to hold every record its workgroups can produce
    const uint32_t nblocks = (c->tiles_local + 3u) / 4u;
    c->hit_seg_cap = ((nblocks + vrt::kHitSegments - 1u) / vrt::kHitSegments) * 256u;
    if (c->hit_seg_cap == 0) c->hit_seg_cap = 256u;
    HIP_TRY(c, hipMalloc(&c->d_hits, (size_t)vrt::kHitSegments * c->hit_seg_cap * sizeof(uint4)));  // >= nblocks * 256
    c->n_blocks = nblocks;
    // launched-ray counts: one per primary workgroup (two-launch variants) or one per tile (the one-launch kernel)
    const size_t ncnt = c->tiles_local ? c->tiles_local : 1;
    HIP_TRY(c, hipMalloc(&c->d_blk_counts, ncnt * sizeof(uint32_t)));
    HIP_TRY(c, hipMemsetAsync(c->d_blk_counts, 0, ncnt * sizeof(uint32_t), c->stream));
    HIP_TRY(c, hipMemsetAsync(c->own_out, 0, n * sizeof(vrt::Texel), c->stream));
    // (c->stream may be the caller's: a VRT_RENDER_OWN_STREAMS frame on own_stream is not ordered behind these memsets, and
    // result sizes that are not whole tiles rely on the zeros)
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    // the extra (stream, output, counts) sets of frames in flight are created when first used (vrt_render)
    c->d_out = c->own_out;  // a resize drops any caller-bound output (its size no longer matches)
    c->last_out = c->own_out;
    c->last_blk = c->d_blk_counts;
    c->rendered = false;
    return VRT_OK;
}

static int flush_staged(vrt_ctx *c);
static int alloc_roots(vrt_ctx *c, uint32_t world_size) {
    const uint64_t n = (uint64_t)world_size * world_size * world_size;
    if (world_size == 0 || n > (1ull << 28)) return fail(c, VRT_ERR_INVALID_ARG, "world_size_chunks %u out of range", world_size);
    {   // uploads into the table that goes away: launched, then waited for
        const int rc = flush_staged(c);
        if (rc) return rc;
    }
    if (c->up_stream) HIP_TRY(c, hipStreamSynchronize(c->up_stream));
    c->roots_tag = 0;
    (void)hipFree(c->d_roots);
    c->d_roots = nullptr;
    HIP_TRY(c, hipMalloc(&c->d_roots, n * sizeof(uint32_t)));
    // A fresh wgpu buffer is zero-initialised: every chunk resolves to pool[0], the air leaf.
    HIP_TRY(c, hipMemsetAsync(c->d_roots, 0, n * sizeof(uint32_t), c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));   // (later uploads run on their own stream)
    c->world_size = world_size;
    c->n_roots = (uint32_t)n;
    c->h_roots.assign((size_t)n, 0u);
    for (auto &T : c->tabs) {
        T.dirty_chunks.clear();
        T.chunk_is_dirty.assign((size_t)n, 0);
        T.chunk_may_have_moved.assign((size_t)n, 0);
        T.chunks_moved = 0;
    }
    c->roots_index_stale = true;
    c->accel_dirty = true;
    return VRT_OK;
}

// Largest world the grid march's tables cover: the cell grid is addressed by a 32-bit byte offset built with signed
// 24-bit multiplies (8S (8S+1)^2 * 4 B < 2^31, (8S+1)^2 * 4 < 2^23), bricks by brick * 128 B < 2^32.
static constexpr uint32_t kAccelMaxS = 100;
// ... the march cells (16 bytes per cell) are addressed the same way: 8S (8S+1)^2 * 16 B < 2^31
static constexpr uint32_t kMarchBlocksMax = 1u << 18;   // 2 GiB of march-cell blocks (byte offsets stay below 2^31)
static size_t chunk_dir_entries(uint32_t S) { return (size_t)S * (S + 1u) * (S + 1u); }
// Small worlds skip the directory: the cells of the whole world, [4S][4S+1][4S+1] lines of 2 x 2 x 2 cells (one dependent
// load and a divergent branch less per change of chunk: 17.5 against 16.1 Grays/s on C4).  16^3 chunks: 35 MB; larger
// worlds go through the directory (32^3: 23 MB instead of 273).
static constexpr uint32_t kMarchDirectMaxS = 16;
static size_t direct_cell_entries(uint32_t S) { const size_t B = (size_t)S * 4u; return B * (B + 1u) * (B + 1u) * 8u; }
static constexpr uint32_t kAccelMaxBricks = (1u << 25) - 1u;
// Chunks that can be rebuilt alone between two whole-world builds: each may move, once, into a 512-brick region
// (64 KiB) at the tail of the brick pool.
static constexpr uint32_t kTailChunks = 128;
// A write that touches more chunks than this is cheaper as a whole-world build.
static constexpr uint32_t kMaxDirtyChunks = 256;

// ---- ordering without draining -------------------------------------------------------------------------------------
// Frames in flight run on the context's own streams; uploads and table rebuilds run on c->stream.  An upload must come
// after every frame enqueued before it (they read what it overwrites) and before every frame enqueued after it: both are
// stream waits on events, the host never blocks on the device here.

// `target` waits for everything enqueued so far on the frame streams other than itself.
static int order_after_frames(vrt_ctx *c, hipStream_t target) {
    auto wait_for = [&](hipStream_t st) -> int {
        if (!st || st == target) return VRT_OK;
        if (!c->ev_frames) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_frames, hipEventDisableTiming));
        HIP_TRY(c, hipEventRecord(c->ev_frames, st));
        HIP_TRY(c, hipStreamWaitEvent(target, c->ev_frames, 0));
        return VRT_OK;
    };
    if (c->alt_pending)
        for (hipStream_t st : c->extra_stream) { const int rc = wait_for(st); if (rc) return rc; }
    if (c->own_pending) { const int rc = wait_for(c->own_stream); if (rc) return rc; }
    if (target != c->stream) { const int rc = wait_for(c->stream); if (rc) return rc; }
    return VRT_OK;
}
static int order_after_frames(vrt_ctx *c) { return order_after_frames(c, c->stream); }

// Everything enqueued on c->stream so far (an upload, a table rebuild) happens before later frames on other streams.
static int publish_upload(vrt_ctx *c) {
    if (!c->ev_upload) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_upload, hipEventDisableTiming));
    HIP_TRY(c, hipEventRecord(c->ev_upload, c->stream));
    c->upload_gen += 1;
    return VRT_OK;
}

// `st` (a frame stream's slot, or kMaxInFlight for c->stream) waits for the node-pool / chunk_roots uploads so far.
static int flush_staged(vrt_ctx *c);
static int wait_for_pool_uploads(vrt_ctx *c, hipStream_t st, uint32_t slot) {
    {
        const int rc = flush_staged(c);   // (what vrt_write_nodes / vrt_write_chunk_roots staged since the last flush: one launch)
        if (rc) return rc;
    }
    if (!c->ev_pool_upload || c->seen_pool_gen[slot] == c->pool_gen) return VRT_OK;
    HIP_TRY(c, hipStreamWaitEvent(st, c->ev_pool_upload, 0));
    c->seen_pool_gen[slot] = c->pool_gen;
    return VRT_OK;
}

// Called before a frame is enqueued on frame stream `st` (slot 0 = own_stream, k = extra_stream[k - 1]).
static int frame_waits_for_uploads(vrt_ctx *c, hipStream_t st, uint32_t slot) {
    const int rc = wait_for_pool_uploads(c, st, st == c->stream ? vrt_ctx::kMaxInFlight : slot);
    if (rc) return rc;
    if (st == c->stream || c->seen_gen[slot] == c->upload_gen) return VRT_OK;
    HIP_TRY(c, hipStreamWaitEvent(st, c->ev_upload, 0));
    c->seen_gen[slot] = c->upload_gen;
    return VRT_OK;
}

// Copy `bytes` of host memory to the device with wgpu's write_buffer semantics — the caller may reuse `src` as soon as
// this returns, the data is visible to the next frame — without waiting for the device: the bytes are copied into a
// pinned ring now, the ring feeds an asynchronous copy kernel.  `pool`: the destination is the node pool or chunk_roots —
// the copy runs on the upload stream behind the readers of those two buffers (the table updates; every frame only if one
// that walks the octree is in flight); otherwise on c->stream behind the frames in flight.  Transfers larger than a ring
// segment (the initial pool upload) take the synchronous route.
// `bytes` of the pinned ring (64-byte aligned), valid until the segment comes round again (eight segments on)
static int ring_place(vrt_ctx *c, size_t bytes, size_t *at) {
    if (!c->h_ring) {
        HIP_TRY(c, hipHostMalloc((void **)&c->h_ring, vrt_ctx::kRingSegBytes * vrt_ctx::kRingSegs, hipHostMallocMapped));
        HIP_TRY(c, hipHostGetDevicePointer((void **)&c->d_ring, c->h_ring, 0));
    }
    const size_t need = (bytes + 63u) & ~(size_t)63u;
    if (c->ring_off + need > vrt_ctx::kRingSegBytes) {
        c->ring_seg = (c->ring_seg + 1u) % vrt_ctx::kRingSegs;
        c->ring_off = 0;
        // the segment's previous copies must have left it (seven segments ago: practically always long done)
        for (int k = 0; k < 2; k++)
            if (c->ring_ev_used[c->ring_seg][k]) HIP_TRY(c, hipEventSynchronize(c->ring_ev[c->ring_seg][k]));
    }
    *at = (size_t)c->ring_seg * vrt_ctx::kRingSegBytes + c->ring_off;
    c->ring_off += need;
    return VRT_OK;
}

// The node-pool / chunk_roots uploads staged so far, as one launch on the upload stream: behind the readers of those two
// buffers (whole-world builds and the other uploads so far, every table set's last update — those that are not over yet: a
// wait is a barrier packet on the stream, a query is a load; every frame only if one that walks the octree is in flight).
static int order_after_frames(vrt_ctx *c, hipStream_t target);
static int flush_staged(vrt_ctx *c) {
    if (c->staged.empty()) return VRT_OK;
    hipStream_t st = c->up_stream;
    if (c->ev_upload && hipEventQuery(c->ev_upload) != hipSuccess) HIP_TRY(c, hipStreamWaitEvent(st, c->ev_upload, 0));
    for (auto &T : c->tabs)
        if (T.update_pending) {
            if (hipEventQuery(T.ev_updated) == hipSuccess) T.update_pending = false;
            else HIP_TRY(c, hipStreamWaitEvent(st, T.ev_updated, 0));
        }
    (void)hipGetLastError();   // (hipErrorNotReady from the queries is not an error)
    if (c->walkers_in_flight) {
        const int rc = order_after_frames(c, st);
        if (rc) return rc;
    }
    vrt::UploadBatch batch;
    uint32_t n = 0;
    auto launch = [&]() -> int {
        vrt::launch_upload_batch(c->d_nodes, c->d_roots, c->d_ring, batch, n, st);
        HIP_TRY(c, hipGetLastError());
        n = 0;
        return VRT_OK;
    };
    for (const auto &s : c->staged)
        for (uint32_t done = 0; done < s.n_words; done += vrt::kUploadPieceWords) {
            const uint32_t words = s.n_words - done < vrt::kUploadPieceWords ? s.n_words - done : vrt::kUploadPieceWords;
            batch.piece[n++] = vrt::UploadPiece{s.dst_word + done, (uint32_t)(s.ring_at / 4u) + done, words | (s.buf ? 0x80000000u : 0u)};
            if (n == vrt::kUploadBatchPieces) { const int rc = launch(); if (rc) return rc; }
        }
    if (n) { const int rc = launch(); if (rc) return rc; }
    for (uint32_t k = 0; k < vrt_ctx::kRingSegs; k++)
        if (c->staged_seg[k]) {
            hipEvent_t &rev = c->ring_ev[k][1];
            if (!rev) HIP_TRY(c, hipEventCreateWithFlags(&rev, hipEventDisableTiming));
            HIP_TRY(c, hipEventRecord(rev, st));
            c->ring_ev_used[k][1] = true;
            c->staged_seg[k] = false;
        }
    c->staged.clear();
    c->staged_bytes = 0;
    HIP_TRY(c, hipEventRecord(c->ev_pool_upload, st));
    c->pool_gen += 1;
    return VRT_OK;
}

// Stage `bytes` for words [dst_word, ...) of the node pool (buf 0) or chunk_roots (buf 1): copied now, launched at the flush.
static int stage_pool_upload(vrt_ctx *c, uint32_t buf, uint32_t dst_word, const void *src, size_t bytes) {
    // a range that overlaps one staged earlier must land after it: the batch's pieces run side by side
    for (const auto &s : c->staged)
        if (s.buf == buf && dst_word < s.dst_word + s.n_words && s.dst_word < dst_word + (uint32_t)(bytes / 4u)) {
            const int rc = flush_staged(c);
            if (rc) return rc;
            break;
        }
    if (c->staged_bytes + bytes > 4u * vrt_ctx::kRingSegBytes) {   // (half the ring: staged data is never overwritten by what follows)
        const int rc = flush_staged(c);
        if (rc) return rc;
    }
    size_t at = 0;
    const int rc = ring_place(c, bytes, &at);
    if (rc) return rc;
    memcpy(c->h_ring + at, src, bytes);
    c->staged.push_back({buf, dst_word, (uint32_t)(bytes / 4u), at});
    c->staged_bytes += bytes;
    c->staged_seg[at / vrt_ctx::kRingSegBytes] = true;
    return VRT_OK;
}

static int stage_upload(vrt_ctx *c, void *dst, const void *src, size_t bytes, bool pool = false) {
    if (bytes == 0) return VRT_OK;
    hipStream_t st = c->stream;
    if (pool) {
        if (!c->up_stream) HIP_TRY(c, hipStreamCreateWithFlags(&c->up_stream, hipStreamNonBlocking));
        if (!c->ev_pool_upload) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_pool_upload, hipEventDisableTiming));
        st = c->up_stream;
        const bool is_roots = dst >= (void *)c->d_roots && dst < (void *)(c->d_roots + c->n_roots);
        if (bytes <= vrt_ctx::kRingSegBytes && !(bytes & 3u) && !((uintptr_t)dst & 3u))
            return stage_pool_upload(c, is_roots ? 1u : 0u,
                                     (uint32_t)(((uintptr_t)dst - (uintptr_t)(is_roots ? (void *)c->d_roots : (void *)c->d_nodes)) / 4u), src, bytes);
        {   // (the whole pool at join time: the synchronous route below, behind what is staged)
            const int rc = flush_staged(c);
            if (rc) return rc;
        }
        // whole-world builds and the other uploads so far, every set's last update — those that are not over yet (a wait is
        // a barrier packet on the stream, a query is a load)
        if (c->ev_upload && hipEventQuery(c->ev_upload) != hipSuccess) HIP_TRY(c, hipStreamWaitEvent(st, c->ev_upload, 0));
        for (auto &T : c->tabs)
            if (T.update_pending) {
                if (hipEventQuery(T.ev_updated) == hipSuccess) T.update_pending = false;
                else HIP_TRY(c, hipStreamWaitEvent(st, T.ev_updated, 0));
            }
        (void)hipGetLastError();   // (hipErrorNotReady from the queries is not an error)
        if (c->walkers_in_flight) {
            const int rc = order_after_frames(c, st);
            if (rc) return rc;
        }
    } else {
        const int rc = order_after_frames(c);
        if (rc) return rc;
    }
    auto publish = [&]() -> int {
        if (!pool) return publish_upload(c);
        HIP_TRY(c, hipEventRecord(c->ev_pool_upload, st));
        c->pool_gen += 1;
        return VRT_OK;
    };
    if (bytes > vrt_ctx::kRingSegBytes || (bytes & 3u) || ((uintptr_t)dst & 3u)) {
        HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st));
        HIP_TRY(c, hipStreamSynchronize(st));
        return publish();
    }
    size_t at = 0;
    {
        const int rc = ring_place(c, bytes, &at);
        if (rc) return rc;
    }
    memcpy(c->h_ring + at, src, bytes);
    vrt::launch_upload_words(dst, c->d_ring + at, (uint32_t)(bytes / 4u), st);
    HIP_TRY(c, hipGetLastError());
    const size_t seg = at / vrt_ctx::kRingSegBytes;
    hipEvent_t &rev = c->ring_ev[seg][pool ? 1 : 0];
    if (!rev) HIP_TRY(c, hipEventCreateWithFlags(&rev, hipEventDisableTiming));
    HIP_TRY(c, hipEventRecord(rev, st));
    c->ring_ev_used[seg][pool ? 1 : 0] = true;
    return publish();
}

// ---- which chunks a write touched ----------------------------------------------------------------------------------
static void mark_all_dirty(vrt_ctx *c) {
    c->view_gen++;
    c->accel_dirty = true;
    for (auto &T : c->tabs) {
        for (uint32_t ch : T.dirty_chunks) T.chunk_is_dirty[ch] = 0;
        T.dirty_chunks.clear();
    }
}

// every table set in use hears of it (a set's list: what changed since *that set* was last brought up to date)
static void mark_chunk_dirty(vrt_ctx *c, uint32_t chunk) {
    c->view_gen++;
    if (c->accel_dirty) return;
    c->tables_split = true;   // (from the next frame on; see vrt_render)
    c->quiet_frames = 0;
    for (uint32_t k = 0; k < vrt_ctx::kMaxInFlight; k++) {
        auto &T = c->tabs[k];
        if ((k && !T.live) || T.chunk_is_dirty[chunk]) continue;
        if (T.dirty_chunks.size() >= kMaxDirtyChunks) { mark_all_dirty(c); return; }
        T.chunk_is_dirty[chunk] = 1;
        T.dirty_chunks.push_back(chunk);
    }
}

static void refresh_roots_index(vrt_ctx *c) {   // (root, chunk) of every present chunk, sorted
    if (!c->roots_index_stale) return;
    c->roots_index.clear();
    for (uint32_t i = 0; i < c->n_roots; i++)
        if (c->h_roots[i]) c->roots_index.emplace_back(c->h_roots[i], i);
    std::sort(c->roots_index.begin(), c->roots_index.end());
    c->roots_index_stale = false;
}

// Nodes [start, end) were overwritten: the chunks whose octrees may have changed are the ones whose root lies in the range
// and the one whose root precedes it (a chunk's nodes follow its root up to the next chunk's root: ChunkAlloc hands out
// disjoint ranges, client/src/world.rs:239-256).  Node 0 is the root of every missing chunk: a write to it is everything.
static void mark_node_range_dirty(vrt_ctx *c, uint32_t start, uint32_t end) {
    if (c->accel_dirty) return;
    if (start == 0u) { mark_all_dirty(c); return; }
    refresh_roots_index(c);
    auto it = std::upper_bound(c->roots_index.begin(), c->roots_index.end(), std::make_pair(start, 0xFFFFFFFFu));
    // chunks sharing the root that precedes the range (normally one), then every chunk rooted inside it
    if (it != c->roots_index.begin()) {
        const uint32_t r = std::prev(it)->first;
        for (auto k = std::prev(it);; --k) {
            if (k->first != r) break;
            mark_chunk_dirty(c, k->second);
            if (c->accel_dirty || k == c->roots_index.begin()) break;
        }
    }
    for (; it != c->roots_index.end() && it->first < end && !c->accel_dirty; ++it) mark_chunk_dirty(c, it->second);
}

// The cell grid and brick pool (vrt_accel.hip) follow the node pool and chunk_roots in two steps.
//   ensure_accel_world  before a frame picks its frame set: the whole-world build when it is due (first frame, resized or
//                       recentred grid, too many single-chunk updates since the last one) — on c->stream with the frames
//                       in flight waited for, into tabs[0], copied to the other sets in use;
//   update_tables       once the frame has its set and stream: the chunks dirtied since *that set* was last brought up to
//                       date, rebuilt alone on the frame's own stream — nothing waits for the frames in flight, which read
//                       other sets (or are earlier on this very stream).
static int free_tables(vrt_ctx *c, vrt_ctx::Tables &T) {
    (void)c;
    (void)hipFree(T.d_grid); (void)hipFree(T.d_bricks); (void)hipFree(T.d_chunk_bricks); (void)hipFree(T.d_chunk_bases);
    (void)hipFree(T.d_chunk_caps); (void)hipFree(T.d_brick_tail); (void)hipFree(T.d_cdir); (void)hipFree(T.d_mblk); (void)hipFree(T.d_mblk_tail);
    T.d_grid = nullptr; T.d_bricks = nullptr; T.d_chunk_bricks = T.d_chunk_bases = T.d_chunk_caps = T.d_brick_tail = nullptr;
    T.d_cdir = nullptr; T.d_mblk = nullptr; T.d_mblk_tail = nullptr;
    T.grid_cap = 0; T.brick_cap = 0; T.chunk_cap = 0; T.cdir_cap = 0; T.mblk_cap = 0;
    T.live = false;
    return VRT_OK;
}

// tabs[k] becomes a copy of tabs[0] (device tables and host bookkeeping); everything on c->stream, the caller has waited
// for the frames in flight.
static int alloc_tables_like_first(vrt_ctx *c, uint32_t k) {
    auto &A = c->tabs[0];
    auto &T = c->tabs[k];
    const uint32_t S = c->accel_S, n_chunks = S * S * S;
    const size_t G = (size_t)S * 8u, entries = G * (G + 1u) * (G + 1u);
    if (T.grid_cap < entries) {
        (void)hipFree(T.d_grid); T.d_grid = nullptr; T.grid_cap = 0;
        HIP_TRY(c, hipMalloc(&T.d_grid, entries * sizeof(uint32_t)));
        T.grid_cap = entries;
    }
    if (A.d_mblk) {
        if (T.cdir_cap < chunk_dir_entries(S)) {
            (void)hipFree(T.d_cdir); T.d_cdir = nullptr; T.cdir_cap = 0;
            HIP_TRY(c, hipMalloc(&T.d_cdir, chunk_dir_entries(S) * sizeof(uint32_t)));
            T.cdir_cap = chunk_dir_entries(S);
        }
        if (T.mblk_cap != A.mblk_cap || !T.d_mblk) {
            (void)hipFree(T.d_mblk); T.d_mblk = nullptr; T.mblk_cap = 0;
            HIP_TRY(c, hipMalloc(&T.d_mblk, (size_t)A.mblk_cap * 512u * sizeof(uint4)));
            T.mblk_cap = A.mblk_cap;
        }
        if (!T.d_mblk_tail) HIP_TRY(c, hipMalloc(&T.d_mblk_tail, sizeof(uint32_t)));
    } else if (T.d_mblk) {
        (void)hipFree(T.d_mblk); T.d_mblk = nullptr; T.mblk_cap = 0;
    }
    if (T.chunk_cap < n_chunks) {
        (void)hipFree(T.d_chunk_bricks); (void)hipFree(T.d_chunk_bases); (void)hipFree(T.d_chunk_caps);
        T.d_chunk_bricks = T.d_chunk_bases = T.d_chunk_caps = nullptr; T.chunk_cap = 0;
        HIP_TRY(c, hipMalloc(&T.d_chunk_bricks, (size_t)n_chunks * sizeof(uint32_t)));
        HIP_TRY(c, hipMalloc(&T.d_chunk_bases, (size_t)n_chunks * sizeof(uint32_t)));
        HIP_TRY(c, hipMalloc(&T.d_chunk_caps, (size_t)n_chunks * sizeof(uint32_t)));
        T.chunk_cap = n_chunks;
    }
    if (T.brick_cap != A.brick_cap || !T.d_bricks) {
        (void)hipFree(T.d_bricks); T.d_bricks = nullptr; T.brick_cap = 0;
        HIP_TRY(c, hipMalloc(&T.d_bricks, (size_t)A.brick_cap * 64u * sizeof(uint16_t)));
        T.brick_cap = A.brick_cap;
    }
    if (!T.d_brick_tail) HIP_TRY(c, hipMalloc(&T.d_brick_tail, sizeof(uint32_t)));
    return VRT_OK;
}

static int copy_tables_from_first(vrt_ctx *c, uint32_t k) {
    auto &A = c->tabs[0];
    auto &T = c->tabs[k];
    const uint32_t S = c->accel_S, n_chunks = S * S * S;
    const size_t G = (size_t)S * 8u, entries = G * (G + 1u) * (G + 1u);
    {
        const int rc = alloc_tables_like_first(c, k);
        if (rc) return rc;
    }
    HIP_TRY(c, hipMemcpyAsync(T.d_grid, A.d_grid, entries * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
    if (A.d_mblk) {
        HIP_TRY(c, hipMemcpyAsync(T.d_cdir, A.d_cdir, chunk_dir_entries(S) * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(T.d_mblk, A.d_mblk, (size_t)A.mblk_cap * 512u * sizeof(uint4), hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(T.d_mblk_tail, A.d_mblk_tail, sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
    }
    HIP_TRY(c, hipMemcpyAsync(T.d_chunk_bricks, A.d_chunk_bricks, (size_t)n_chunks * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(T.d_chunk_bases, A.d_chunk_bases, (size_t)n_chunks * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(T.d_chunk_caps, A.d_chunk_caps, (size_t)n_chunks * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(T.d_brick_tail, A.d_brick_tail, sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(T.d_bricks, A.d_bricks, (size_t)A.brick_cap * 64u * sizeof(uint16_t), hipMemcpyDeviceToDevice, c->stream));
    T.chunk_may_have_moved = A.chunk_may_have_moved;
    T.chunks_moved = A.chunks_moved;
    T.chunk_builds = A.chunk_builds;
    T.dirty_chunks = A.dirty_chunks;   // what tabs[0] has not caught up with yet, this copy has not either
    T.chunk_is_dirty = A.chunk_is_dirty;
    T.update_pending = false;
    T.live = true;
    return VRT_OK;
}

static int ensure_accel_world(vrt_ctx *c) {
    const uint32_t S = c->world.size_in_chunks;
    if (S > c->accel_max_s) {  // too large for the tables: nothing to keep up to date, the octree walk reads the pool itself
        if (c->accel_dirty || c->accel_S != S) {
            mark_all_dirty(c);
            c->accel_ok = false;
            c->accel_S = S;
            c->accel_dirty = false;
        }
        return VRT_OK;
    }
    if (!c->accel_dirty && c->accel_S == S) {
        if (!c->accel_ok) return VRT_OK;   // (the brick pool would be too large: stays off until something changes)
        // single chunks; a chunk that outgrew its region moves to the tail once — every set's tail must have room for all of them
        bool room = true;
        for (uint32_t k = 0; k < vrt_ctx::kMaxInFlight; k++) {
            const auto &T = c->tabs[k];
            if (k && !T.live) continue;
            uint32_t fresh = 0;
            for (uint32_t ch : T.dirty_chunks) fresh += T.chunk_may_have_moved[ch] ? 0u : 1u;
            room = room && T.chunks_moved + fresh <= kTailChunks;
        }
        if (room) return VRT_OK;
    }
    mark_all_dirty(c);   // (clears the chunk lists: a whole-world build covers them)
    QUIESCE(c);  // frames on the other streams may still be reading the old tables; this path reads a count back anyway
    {   // ... and the node pool and chunk_roots as uploaded so far
        const int rc = wait_for_pool_uploads(c, c->stream, vrt_ctx::kMaxInFlight);
        if (rc) return rc;
    }
    auto &A = c->tabs[0];
    c->accel_ok = false;
    c->accel_S = S;
    c->accel_dirty = false;
    const uint32_t n_chunks = S * S * S;
    const size_t G = (size_t)S * 8u;
    const size_t entries = G * (G + 1u) * (G + 1u);
    if (entries > A.grid_cap) {
        (void)hipFree(A.d_grid);
        A.d_grid = nullptr; A.grid_cap = 0;
        HIP_TRY(c, hipMalloc(&A.d_grid, entries * sizeof(uint32_t)));
        A.grid_cap = entries;
    }
    // the border rows / entries are never written by the kernels: zero = "outside the world"
    HIP_TRY(c, hipMemsetAsync(A.d_grid, 0, entries * sizeof(uint32_t), c->stream));
    if (chunk_dir_entries(S) > A.cdir_cap) {
        (void)hipFree(A.d_cdir);
        A.d_cdir = nullptr; A.cdir_cap = 0;
        HIP_TRY(c, hipMalloc(&A.d_cdir, chunk_dir_entries(S) * sizeof(uint32_t)));
        A.cdir_cap = chunk_dir_entries(S);
    }
    HIP_TRY(c, hipMemsetAsync(A.d_cdir, 0, chunk_dir_entries(S) * sizeof(uint32_t), c->stream));   // the border: outside the world
    if (!A.d_mblk_tail) HIP_TRY(c, hipMalloc(&A.d_mblk_tail, sizeof(uint32_t)));
    if (n_chunks > c->chunk_needs_cap) {
        (void)hipFree(c->d_chunk_needs);
        c->d_chunk_needs = nullptr; c->chunk_needs_cap = 0;
        HIP_TRY(c, hipMalloc(&c->d_chunk_needs, (size_t)n_chunks * sizeof(uint32_t)));
        c->chunk_needs_cap = n_chunks;
    }
    if (n_chunks > A.chunk_cap) {
        (void)hipFree(A.d_chunk_bricks); (void)hipFree(A.d_chunk_bases); (void)hipFree(A.d_chunk_caps);
        A.d_chunk_bricks = A.d_chunk_bases = A.d_chunk_caps = nullptr; A.chunk_cap = 0;
        HIP_TRY(c, hipMalloc(&A.d_chunk_bricks, (size_t)n_chunks * sizeof(uint32_t)));
        HIP_TRY(c, hipMalloc(&A.d_chunk_bases, (size_t)n_chunks * sizeof(uint32_t)));
        HIP_TRY(c, hipMalloc(&A.d_chunk_caps, (size_t)n_chunks * sizeof(uint32_t)));
        A.chunk_cap = n_chunks;
    }
    if (!c->d_brick_total) HIP_TRY(c, hipMalloc(&c->d_brick_total, 2 * sizeof(uint32_t)));   // [0] bricks, [1] march-cell blocks
    if (!A.d_brick_tail) HIP_TRY(c, hipMalloc(&A.d_brick_tail, sizeof(uint32_t)));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIP_TRY(c, hipEventCreate(&e0));
    HIP_TRY(c, hipEventCreate(&e1));
    const bool direct = S <= c->march_direct_max_s;
    c->march_direct = direct;
    auto body = [&]() -> int {
        HIP_TRY(c, hipEventRecord(e0, c->stream));
        vrt::launch_accel_cells(c->d_nodes, c->max_nodes, c->d_roots, S, A.d_grid, A.d_chunk_bricks, A.d_chunk_bases, A.d_chunk_caps,
                                c->d_brick_total, A.d_brick_tail, direct ? nullptr : c->d_chunk_needs, A.d_cdir, A.d_mblk_tail, c->d_brick_total + 1, c->stream);
        HIP_TRY(c, hipGetLastError());
        uint32_t totals[2] = {0, 0};  // bricks in all chunk regions (counts + slack); chunks that need a block of march cells
        HIP_TRY(c, hipMemcpyAsync(totals, c->d_brick_total, sizeof totals, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        const uint32_t total = totals[0];
        // the march cells: blocks 0 and 1, one per chunk that needs its own, and room for the chunks that may come to need one
        // before the next whole-world build (every one of them is a chunk rebuilt alone: at most kTailChunks)
        const uint64_t want_blocks = direct ? (direct_cell_entries(S) + 511u) / 512u : 2ull + totals[1] + kTailChunks;
        if (want_blocks > kMarchBlocksMax) {
            for (auto &T : c->tabs) { (void)hipFree(T.d_mblk); T.d_mblk = nullptr; T.mblk_cap = 0; }
        } else {
            if (want_blocks > A.mblk_cap || !A.d_mblk) {
                (void)hipFree(A.d_mblk);
                A.d_mblk = nullptr; A.mblk_cap = 0;
                uint64_t cap = want_blocks + (direct ? 0u : totals[1] / 8u);
                if (cap > kMarchBlocksMax) cap = kMarchBlocksMax;
                HIP_TRY(c, hipMalloc(&A.d_mblk, (size_t)cap * 512u * sizeof(uint4)));
                A.mblk_cap = (uint32_t)cap;
            }
            // block 0: every cell stops the ray (direct: so do the blocks of the directory's border)
            HIP_TRY(c, hipMemsetAsync(A.d_mblk, 0, (direct ? (size_t)want_blocks : (size_t)1) * 512u * sizeof(uint4), c->stream));
        }
        const uint64_t want = (uint64_t)total + (uint64_t)kTailChunks * 512u;
        if (want > kAccelMaxBricks) return VRT_OK;  // accel_ok stays false
        if (want > A.brick_cap || !A.d_bricks) {
            (void)hipFree(A.d_bricks);
            A.d_bricks = nullptr; A.brick_cap = 0;
            uint64_t cap = want + total / 4u;  // room to grow before the next reallocation
            if (cap > kAccelMaxBricks) cap = kAccelMaxBricks;
            HIP_TRY(c, hipMalloc(&A.d_bricks, (size_t)cap * 64u * sizeof(uint16_t)));
            A.brick_cap = (uint32_t)cap;
        }
        vrt::launch_accel_bricks(c->d_nodes, c->max_nodes, c->d_roots, S, A.d_grid, A.d_chunk_bases, A.d_bricks, A.brick_cap, direct ? nullptr : A.d_cdir, A.d_mblk,
                                 A.d_mblk_tail, A.mblk_cap, c->liquid_mask, c->stream);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipEventRecord(e1, c->stream));
        HIP_TRY(c, hipEventSynchronize(e1));
        HIP_TRY(c, hipEventElapsedTime(&c->accel_last_ms, e0, e1));
        c->n_bricks = total;
        c->accel_builds += 1;
        c->accel_ok = true;
        std::fill(A.chunk_may_have_moved.begin(), A.chunk_may_have_moved.end(), (uint8_t)0);
        A.chunks_moved = 0;
        A.update_pending = false;
        A.live = true;
        for (uint32_t k = 1; k < vrt_ctx::kMaxInFlight; k++) {   // the other sets in use start over as copies
            if (c->tabs[k].live) {
                const int rc = copy_tables_from_first(c, k);
                if (rc) return rc;
            } else if (k < c->in_flight) {   // (memory for the sets the first edit will want: an allocation is milliseconds)
                const int rc = alloc_tables_like_first(c, k);
                if (rc) return rc;
            }
        }
        return publish_upload(c);
    };
    const int rc = body();
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc) c->accel_dirty = true;
    return rc;
}

// The frame about to be enqueued on `st` uses table set `k`: bring it up to date there.  `st` has been made to wait for the
// uploads so far (frame_waits_for_uploads).
static int update_tables(vrt_ctx *c, uint32_t k, hipStream_t st) {
    if (!c->accel_ok || c->accel_dirty) return VRT_OK;
    auto &T = c->tabs[k];
    if (!T.live) {   // this frame set's first frame since the last whole-world build of a smaller crowd: a copy of tabs[0]
        // (wait for the frames in flight; the frame about to be enqueued has already been announced on its stream)
        const bool alt = c->alt_pending, own = c->own_pending;
        QUIESCE(c);
        c->alt_pending = alt;
        c->own_pending = own;
        {
            const int rc = wait_for_pool_uploads(c, c->stream, vrt_ctx::kMaxInFlight);
            if (rc) return rc;
        }
        if (c->tabs[0].update_pending) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->tabs[0].ev_updated, 0));
        int rc = copy_tables_from_first(c, k);
        if (rc) return rc;
        rc = publish_upload(c);
        if (rc) return rc;
        if (st != c->stream) HIP_TRY(c, hipStreamWaitEvent(st, c->ev_upload, 0));
    }
    if (T.dirty_chunks.empty()) return VRT_OK;
    if (k == 0u && c->shared_readers_in_flight) {   // frames of other frame sets still read this set (it was shared until now)
        const bool alt = c->alt_pending, own = c->own_pending;
        QUIESCE(c);
        c->alt_pending = alt;
        c->own_pending = own;
    }
    // how far each chunk's nodes can reach: up to the next chunk's root (ChunkAlloc's ranges are disjoint); the kernel
    // stages that much of the pool and reads anything beyond from the pool itself
    refresh_roots_index(c);
    std::vector<uint32_t> extents(T.dirty_chunks.size());
    for (size_t i = 0; i < extents.size(); i++) {
        const uint32_t r = T.dirty_chunks[i] < c->n_roots ? c->h_roots[T.dirty_chunks[i]] : 0u;
        auto nx = std::upper_bound(c->roots_index.begin(), c->roots_index.end(), std::make_pair(r, 0xFFFFFFFFu));
        const uint32_t end = nx != c->roots_index.end() ? nx->first : c->max_nodes;
        extents[i] = r ? (end > r ? end - r : 0u) : 1u;   // (a missing chunk is node 0 alone: one air leaf)
    }
    vrt::launch_accel_chunks(c->d_nodes, c->max_nodes, c->d_roots, c->accel_S, T.d_grid, T.d_chunk_bricks, T.d_chunk_bases, T.d_chunk_caps,
                             T.d_brick_tail, T.d_bricks, T.brick_cap, c->march_direct ? nullptr : T.d_cdir, T.d_mblk, T.d_mblk_tail, T.mblk_cap, c->liquid_mask,
                             T.dirty_chunks.data(), extents.data(), (uint32_t)T.dirty_chunks.size(), st);
    HIP_TRY(c, hipGetLastError());
    // the next upload of nodes or roots waits for this reader
    if (!T.ev_updated) HIP_TRY(c, hipEventCreateWithFlags(&T.ev_updated, hipEventDisableTiming));
    HIP_TRY(c, hipEventRecord(T.ev_updated, st));
    T.update_pending = true;
    for (uint32_t ch : T.dirty_chunks) {
        if (!T.chunk_may_have_moved[ch]) { T.chunk_may_have_moved[ch] = 1; T.chunks_moved += 1; }
        T.chunk_is_dirty[ch] = 0;
    }
    T.chunk_builds += (uint32_t)T.dirty_chunks.size();
    T.dirty_chunks.clear();
    return VRT_OK;
}

// ---- one context over several devices (vrt_config.n_devices > 1); defined behind the C ABI below ----
static int grp_create(const vrt_config *cfg, vrt_ctx **out);
static void grp_destroy(vrt_ctx *c);
static int grp_render(vrt_ctx *c, const vrt_render_opts *opts);
static int grp_synchronize(vrt_ctx *c);
static int grp_get_stats(vrt_ctx *c, vrt_stats *out);
static int grp_resize_output(vrt_ctx *c, uint32_t w, uint32_t h);
static int grp_set_frames_in_flight(vrt_ctx *c, uint32_t n);
static vrt_ctx *grp_root(vrt_ctx *c);
template <typename F> static int grp_each(vrt_ctx *c, F f);
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

extern "C" {

int vrt_create(const vrt_config *cfg, vrt_ctx **out) {
    if (!cfg || !out) return fail(nullptr, VRT_ERR_INVALID_ARG, "vrt_create: null argument");
    *out = nullptr;
    if (cfg->n_devices > 1u) return grp_create(cfg, out);
    if (cfg->max_nodes < 2 || cfg->max_nodes > 0x7FFFFFFEu)
        return fail(nullptr, VRT_ERR_INVALID_ARG, "max_nodes must be in [2, 2^31 - 2] (the pool is addressed through a 32-bit byte offset)");
    if (cfg->width == 0 || cfg->height == 0)
        return fail(nullptr, VRT_ERR_INVALID_ARG, "output %ux%u: dimensions must be non-zero", cfg->width, cfg->height);
    if ((uint64_t)cfg->width * cfg->height > (1ull << 28))
        return fail(nullptr, VRT_ERR_INVALID_ARG, "output too large");
    const uint32_t sc = cfg->shard_count ? cfg->shard_count : 1u;
    if (cfg->shard_rank >= sc) return fail(nullptr, VRT_ERR_INVALID_ARG, "shard_rank %u >= shard_count %u", cfg->shard_rank, sc);
    if (cfg->shard_root_weight > 4096u) return fail(nullptr, VRT_ERR_INVALID_ARG, "shard_root_weight %u out of range", cfg->shard_root_weight);
    if ((cfg->flags & VRT_FLAG_ROW_MAJOR) && (cfg->flags & VRT_FLAG_TILE_MAJOR))
        return fail(nullptr, VRT_ERR_INVALID_ARG, "VRT_FLAG_ROW_MAJOR and VRT_FLAG_TILE_MAJOR exclude each other");
    if ((cfg->flags & VRT_FLAG_COMPACT) && ((cfg->flags & VRT_FLAG_ROW_MAJOR) || (sc == 1u && !(cfg->flags & VRT_FLAG_TILE_MAJOR))))
        return fail(nullptr, VRT_ERR_INVALID_ARG, "VRT_FLAG_COMPACT is for tile-major shard buffers");

    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
        return fail(nullptr, VRT_ERR_DEVICE, "no HIP device available (%s)", hipGetErrorString(e));
    int dev = cfg->device;
    if (dev < 0) {
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    }
    if (dev >= ndev) return fail(nullptr, VRT_ERR_INVALID_ARG, "device %d out of range (%d devices)", dev, ndev);

    vrt_ctx *c = new (std::nothrow) vrt_ctx();
    if (!c) return fail(nullptr, VRT_ERR_OOM, "host allocation failed");
    c->device = dev;
    c->shard_rank = cfg->shard_rank;
    c->shard_count = sc;
    c->shard_w0 = cfg->shard_root_weight ? cfg->shard_root_weight : 1u;
    c->tile_major = (sc > 1u && !(cfg->flags & VRT_FLAG_ROW_MAJOR)) || (cfg->flags & VRT_FLAG_TILE_MAJOR);
    c->compact = (cfg->flags & VRT_FLAG_COMPACT) != 0;
    c->width = cfg->width;
    c->height = cfg->height;
    c->max_nodes = cfg->max_nodes & ~1u;  // NodeBuffer::new forces an even size (shader.rs:10-12)
    c->accel_max_s = kAccelMaxS;
    if (const char *e = getenv("VRT_ACCEL_MAX_S")) {
        const long v = strtol(e, nullptr, 10);
        if (v >= 0 && v < (long)kAccelMaxS) c->accel_max_s = (uint32_t)v;
    }
    c->march_direct_max_s = kMarchDirectMaxS;
    if (const char *e = getenv("VRT_MARCH_DIRECT_MAX_S")) {
        const long v = strtol(e, nullptr, 10);
        if (v >= 0 && v <= (long)kMarchDirectMaxS) c->march_direct_max_s = (uint32_t)v;
    }
    if (const char *e = getenv("VRT_TILE_ORDER")) c->tile_lpt = e[0] != '0';
#ifdef VRT_EXPERIMENTS
    if (const char *e = getenv("VRT_PATH_PERSISTENT")) c->path_persistent = e[0] == '1';
    if (const char *e = getenv("VRT_PATH_POOL")) c->path_pool = e[0] != '0';
    if (const char *e = getenv("VRT_PATH_POOL_CHAIN")) c->path_chain = e[0] == '1';
    if (const char *e = getenv("VRT_PATH_CELLS")) c->path_cells = e[0] != '0';
    if (const char *e = getenv("VRT_PATH_POOL_REFILL")) c->path_refill = (uint32_t)atoi(e);
    if (const char *e = getenv("VRT_PATH_POOL_EJECT")) c->path_eject = (uint32_t)atoi(e);
#endif
    if (const char *e = getenv("VRT_PATH_SAMPLES_PER_CHAIN")) { const int v = atoi(e); if (v >= 1 && v <= 16) c->path_samples = (uint32_t)v; }
    if (const char *e = getenv("VRT_TIMING_EVERY")) { const long v = strtol(e, nullptr, 10); if (v >= 1 && v <= 1000000) c->timing_every = (uint32_t)v; }
    memset(c->h_mats, 0, sizeof c->h_mats);
    memset(&c->cam, 0, sizeof c->cam);
    memset(&c->settings, 0, sizeof c->settings);
    memset(&c->world, 0, sizeof c->world);
    memset(&c->stats, 0, sizeof c->stats);

    auto body = [&]() -> int {
        HIP_TRY(c, hipSetDevice(dev));
        HIP_TRY(c, hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
        c->stream = c->own_stream;
        HIP_TRY(c, hipMalloc(&c->d_nodes, (size_t)c->max_nodes * sizeof(uint16_t)));
        // fresh buffer = zeros = every node an air leaf (client/src/world.rs:273-274)
        HIP_TRY(c, hipMemsetAsync(c->d_nodes, 0, (size_t)c->max_nodes * sizeof(uint16_t), c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));   // (uploads run on their own stream)
        // the upload path's fixtures now, not at the first edit (a mapped pinned allocation is tens of milliseconds)
        HIP_TRY(c, hipStreamCreateWithFlags(&c->up_stream, hipStreamNonBlocking));
        HIP_TRY(c, hipEventCreateWithFlags(&c->ev_pool_upload, hipEventDisableTiming));
        HIP_TRY(c, hipHostMalloc((void **)&c->h_ring, vrt_ctx::kRingSegBytes * vrt_ctx::kRingSegs, hipHostMallocMapped));
        HIP_TRY(c, hipHostGetDevicePointer((void **)&c->d_ring, c->h_ring, 0));
        HIP_TRY(c, hipMalloc(&c->d_mats, sizeof c->h_mats));
        HIP_TRY(c, hipMemsetAsync(c->d_mats, 0, sizeof c->h_mats, c->stream));
        HIP_TRY(c, hipMalloc(&c->d_counters, kCounterBytes));
        HIP_TRY(c, hipMemsetAsync(c->d_counters, 0, kCounterBytes, c->stream));
        int r = alloc_roots(c, cfg->world_size_chunks);
        if (r) return r;
        r = alloc_output(c);
        if (r) return r;
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        return VRT_OK;
    };
    const int rc = body();
    if (rc != VRT_OK) {
        g_create_err = c->err;
        vrt_destroy(c);
        return rc;
    }
    *out = c;
    return VRT_OK;
}

void vrt_destroy(vrt_ctx *c) {
    if (!c) return;
    if (c->grp) { grp_destroy(c); return; }
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
    if (c->up_stream) (void)hipStreamSynchronize(c->up_stream);
    for (hipStream_t st : c->extra_stream)
        if (st) (void)hipStreamSynchronize(st);
    for (auto p : c->extra_out) (void)hipFree(p);
    for (auto p : c->extra_blk) (void)hipFree(p);
    for (auto p : c->extra_path) (void)hipFree(p);
    for (auto p : c->extra_counters) (void)hipFree(p);
    for (auto p : c->path_cont) (void)hipFree(p);
    for (auto p : c->path_acc) (void)hipFree(p);
    (void)hipFree(c->d_tile_cost); (void)hipFree(c->d_tile_order); (void)hipFree(c->d_tile_scratch);
    for (auto st : c->side_stream)
        if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    for (auto &evs : c->side_ev)
        for (auto ev : evs)
            if (ev) (void)hipEventDestroy(ev);
    (void)hipFree(c->d_nodes); (void)hipFree(c->d_roots); (void)hipFree(c->d_mats); (void)hipFree(c->own_out);
    (void)hipFree(c->d_hits); (void)hipFree(c->d_counters); (void)hipFree(c->d_steps); (void)hipFree(c->d_rgba8); (void)hipFree(c->d_path);
    (void)hipFree(c->d_blk_counts); (void)hipFree(c->d_clock);
    for (auto &T : c->tabs) {
        (void)free_tables(c, T);
        if (T.ev_updated) (void)hipEventDestroy(T.ev_updated);
    }
    (void)hipFree(c->d_brick_total); (void)hipFree(c->d_chunk_needs);
    if (c->up_stream) { (void)hipStreamSynchronize(c->up_stream); (void)hipStreamDestroy(c->up_stream); }
    if (c->ev_pool_upload) (void)hipEventDestroy(c->ev_pool_upload);
    if (c->ev_walkers) (void)hipEventDestroy(c->ev_walkers);
    if (c->h_ring) (void)hipHostFree(c->h_ring);
    for (auto &evs : c->ring_ev)
        for (auto ev : evs)
            if (ev) (void)hipEventDestroy(ev);
    if (c->ev_frames) (void)hipEventDestroy(c->ev_frames);
    if (c->ev_upload) (void)hipEventDestroy(c->ev_upload); (void)hipFree(c->d_ndc); (void)hipFree(c->d_screen); (void)hipFree(c->d_heads);
    for (auto &t : c->ev_pool)
        for (auto &ev : t)
            if (ev) (void)hipEventDestroy(ev);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    for (hipStream_t st : c->extra_stream)
        if (st) (void)hipStreamDestroy(st);
    delete c;
}

const char *vrt_last_error(const vrt_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int vrt_write_nodes(vrt_ctx *c, const uint16_t *pool, uint32_t start, uint32_t end) {
    GRP_EACH(c, vrt_write_nodes(d, pool, start, end));
    if (!c || !pool) return fail(c, VRT_ERR_INVALID_ARG, "vrt_write_nodes: null argument");
    if (end < start) return fail(c, VRT_ERR_INVALID_ARG, "vrt_write_nodes: end %u < start %u", end, start);
    // NodeBuffer::write, shader.rs:24-33: widen to even bounds
    uint32_t root = start, count = end - start;
    if (root % 2 == 1) { root -= 1; count += 1; }
    if (count % 2 == 1) count += 1;
    if (count == 0) return VRT_OK;
    if ((uint64_t)root + count > c->max_nodes)
        return fail(c, VRT_ERR_OUT_OF_RANGE, "vrt_write_nodes: [%u,%u) exceeds the %u-node buffer", start, end, c->max_nodes);
    HIP_TRY(c, hipSetDevice(c->device));
    // copy-at-call-time (write_buffer semantics: the caller may reuse `pool` as soon as this returns) through the pinned
    // ring, ordered after the frames in flight without waiting for them
    const int rc = stage_upload(c, c->d_nodes + root, pool + root, (size_t)count * sizeof(uint16_t), true);
    if (rc) return rc;
    mark_node_range_dirty(c, start, end);   // (the widening repeats a neighbour's node: nothing of its octree changes)
    return VRT_OK;
}

int vrt_write_chunk_roots(vrt_ctx *c, uint32_t offset, const uint32_t *roots, uint32_t n) { return vrt_write_chunk_roots_tagged(c, offset, roots, n, 0); }

int vrt_write_chunk_roots_tagged(vrt_ctx *c, uint32_t offset, const uint32_t *roots, uint32_t n, uint64_t tag) {
    GRP_EACH(c, vrt_write_chunk_roots_tagged(d, offset, roots, n, tag));
    if (!c || (!roots && n)) return fail(c, VRT_ERR_INVALID_ARG, "vrt_write_chunk_roots: null argument");
    // the reference rewrites the whole table every frame (main.rs:446).  A caller that can say "nothing changed since the
    // write I tagged like this" is believed: the 128 KB compare of a 32^3 table is half a frame's host time
    if (tag != 0 && tag == c->roots_tag && offset == c->roots_tag_offset && n == c->roots_tag_n) return VRT_OK;
    c->roots_tag = 0;
    if (offset > c->n_roots) return fail(c, VRT_ERR_OUT_OF_RANGE, "vrt_write_chunk_roots: offset %u > %u", offset, c->n_roots);
    // ArrayBuffer::write truncates to capacity (shader.rs:134-135)
    const uint32_t cut = n < c->n_roots - offset ? n : c->n_roots - offset;
    if (cut == 0) return VRT_OK;
    auto remember = [&]() { c->roots_tag = tag; c->roots_tag_offset = offset; c->roots_tag_n = n; };
    // ... an identical rewrite changes nothing
    if (memcmp(c->h_roots.data() + offset, roots, (size_t)cut * sizeof(uint32_t)) == 0) { remember(); return VRT_OK; }
    HIP_TRY(c, hipSetDevice(c->device));
    const int rc = stage_upload(c, c->d_roots + offset, roots, (size_t)cut * sizeof(uint32_t), true);
    if (rc) return rc;
    // the slots whose root changed are the chunks to rebuild (a chunk arrived or was dropped); a recentred grid changes
    // nearly all of them and becomes a whole-world build
    for (uint32_t i = 0; i < cut && !c->accel_dirty; i++)
        if (c->h_roots[offset + i] != roots[i]) mark_chunk_dirty(c, offset + i);
    memcpy(c->h_roots.data() + offset, roots, (size_t)cut * sizeof(uint32_t));
    c->roots_index_stale = true;
    remember();
    return VRT_OK;
}

int vrt_resize_world(vrt_ctx *c, uint32_t world_size_chunks) {
    GRP_EACH(c, vrt_resize_world(d, world_size_chunks));
    if (!c) return VRT_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    QUIESCE(c);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return alloc_roots(c, world_size_chunks);
}

int vrt_write_materials(vrt_ctx *c, uint32_t first, const vrt_material *mats, uint32_t n) {
    GRP_EACH(c, vrt_write_materials(d, first, mats, n));
    if (!c || (!mats && n)) return fail(c, VRT_ERR_INVALID_ARG, "vrt_write_materials: null argument");
    if ((uint64_t)first + n > 256) return fail(c, VRT_ERR_OUT_OF_RANGE, "vrt_write_materials: %u+%u > 256", first, n);
    if (n == 0) return VRT_OK;
    c->view_gen++;
    memcpy(c->h_mats + first, mats, (size_t)n * sizeof(vrt_material));
    uint32_t old_mask[8];
    memcpy(old_mask, c->liquid_mask, sizeof old_mask);
    memset(c->liquid_mask, 0, sizeof c->liquid_mask);
    int lo = -1, hi = -1, n_liquid = 0;
    for (int v = 0; v < 256; v++)
        if (c->h_mats[v].is_liquid == 1u) {
            c->liquid_mask[v >> 5] |= 1u << (v & 31);
            if (lo < 0) lo = v;
            hi = v;
            n_liquid++;
        }
    // one contiguous id range, and material 255 (which every id >= 255 clamps to) not in it
    c->liquid_is_range = n_liquid == 0 || (hi - lo + 1 == n_liquid && hi < 255);
    c->liquid_lo = n_liquid ? (uint32_t)lo : 0x80000000u;
    c->liquid_span = n_liquid ? (uint32_t)(hi - lo) : 0u;
    // the march cells say which voxels stop a ray: another set of liquids is another set of tables (a join-time event)
    if (memcmp(old_mask, c->liquid_mask, sizeof old_mask) != 0) mark_all_dirty(c);
    HIP_TRY(c, hipSetDevice(c->device));
    return stage_upload(c, c->d_mats + first, mats, (size_t)n * sizeof(vrt_material));
}

int vrt_set_camera(vrt_ctx *c, const vrt_cam_data *cam) {
    GRP_EACH(c, vrt_set_camera(d, cam));
    if (!c || !cam) return fail(c, VRT_ERR_INVALID_ARG, "vrt_set_camera: null argument");
    if (memcmp(&c->cam, cam, sizeof *cam) != 0) c->view_gen++;
    c->cam = *cam;
    return VRT_OK;
}

int vrt_set_settings(vrt_ctx *c, const vrt_settings *s) {
    GRP_EACH(c, vrt_set_settings(d, s));
    if (!c || !s) return fail(c, VRT_ERR_INVALID_ARG, "vrt_set_settings: null argument");
    if (memcmp(&c->settings, s, sizeof *s) != 0) c->view_gen++;
    c->settings = *s;
    return VRT_OK;
}

int vrt_set_world(vrt_ctx *c, const vrt_world_data *w) {
    GRP_EACH(c, vrt_set_world(d, w));
    if (!c || !w) return fail(c, VRT_ERR_INVALID_ARG, "vrt_set_world: null argument");
    if (memcmp(&c->world, w, sizeof *w) != 0) c->view_gen++;
    c->world = *w;
    return VRT_OK;
}

int vrt_resize_output(vrt_ctx *c, uint32_t width, uint32_t height) {
    if (c && c->grp) return grp_resize_output(c, width, height);
    if (!c) return VRT_ERR_INVALID_ARG;
    if (width == 0 || height == 0 || (uint64_t)width * height > (1ull << 28))
        return fail(c, VRT_ERR_INVALID_ARG, "output %ux%u: dimensions must be non-zero, at most 2^28 pixels", width, height);
    HIP_TRY(c, hipSetDevice(c->device));
    QUIESCE(c);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->width = width;
    c->height = height;
    return alloc_output(c);
}

static int validate_frame(vrt_ctx *c) {
    const uint32_t S = c->world.size_in_chunks;
    if (S == 0 || (uint64_t)S * S * S > c->n_roots)
        return fail(c, VRT_ERR_STATE, "world.size_in_chunks %u does not fit the %u-entry chunk_roots buffer "
                    "(call vrt_resize_world first, main.rs:441-445)", S, c->n_roots);
    if (c->world.size != S * 32u)
        return fail(c, VRT_ERR_STATE, "world.size %u != size_in_chunks*32 (%u)", c->world.size, S * 32u);
    return VRT_OK;
}

// Frame-uniform pieces of create_ray_from_screen (ray_tracer.wgsl:160-161): one IEEE divide per column and row
// instead of two per pixel.  Same operations in the same order as the shader text, in binary32.
static int ensure_ndc(vrt_ctx *c) {
    if (c->d_ndc && c->ndc_w == c->width && c->ndc_h == c->height &&
        memcmp(c->ndc_proj, c->cam.proj_size, sizeof c->ndc_proj) == 0)
        return VRT_OK;
    if (!c->d_ndc || c->ndc_w + c->ndc_h < c->width + c->height) {
        QUIESCE(c);
        (void)hipFree(c->d_ndc);
        c->d_ndc = nullptr;
        HIP_TRY(c, hipMalloc(&c->d_ndc, (size_t)(c->width + c->height) * sizeof(float)));
    }
    std::vector<float> t((size_t)c->width + c->height);
    volatile float px = c->cam.proj_size[0], py = c->cam.proj_size[1];
    for (uint32_t i = 0; i < c->width; i++) { volatile float q = ((float)(int)i * 2.0f) / px; t[i] = q - 1.0f; }
    for (uint32_t i = 0; i < c->height; i++) { volatile float q = ((float)(int)i * 2.0f) / py; t[c->width + i] = q - 1.0f; }
    const int rc = stage_upload(c, c->d_ndc, t.data(), t.size() * sizeof(float));
    if (rc) return rc;
    c->ndc_w = c->width;
    c->ndc_h = c->height;
    memcpy(c->ndc_proj, c->cam.proj_size, sizeof c->ndc_proj);
    return VRT_OK;
}

// ray_sky's sun_dir for a ray starting at the camera (ray_tracer.wgsl:149, origin = cam.pos - world.min :169).
static void cam_sun_dir(const vrt_ctx *c, float out[3]) {
    volatile float d[3];
    for (int k = 0; k < 3; k++) {
        volatile float wm = (float)c->world.min[k];
        volatile float origin = c->cam.pos[k] - wm;
        volatile float a = c->settings.sun_pos[k] - wm;
        d[k] = a - origin;
    }
    volatile float xx = d[0] * d[0], yy = d[1] * d[1], zz = d[2] * d[2];
    volatile float s = xx + yy;
    volatile float dot = s + zz;
    volatile float len = sqrtf(dot);
    for (int k = 0; k < 3; k++) { volatile float q = d[k] / len; out[k] = q; }
}

// Fold the event triples of all frames rendered since the last call into acc_ms (synchronises).
static int fold_events(vrt_ctx *c, float last[3]) {
    if (c->ev_used == 0) return VRT_OK;
    QUIESCE(c);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < c->ev_used; i++) {
        auto &t = c->ev_pool[i];
        float a = 0, b = 0, tot = 0;
        switch (c->ev_kind[i]) {
            case kEvOneKernel:
                HIP_TRY(c, hipEventElapsedTime(&a, t[0], t[1]));
                tot = a;
                break;
            case kEvTwoKernels:
                HIP_TRY(c, hipEventElapsedTime(&a, t[0], t[1]));
                HIP_TRY(c, hipEventElapsedTime(&b, t[2], t[3]));
                HIP_TRY(c, hipEventElapsedTime(&tot, t[0], t[3]));
                break;
            case kEvRecorded:
                HIP_TRY(c, hipEventElapsedTime(&a, t[0], t[1]));
                HIP_TRY(c, hipEventElapsedTime(&b, t[1], t[3]));
                HIP_TRY(c, hipEventElapsedTime(&tot, t[0], t[3]));
                break;
            default:
                continue;  // an empty shard: nothing was launched
        }
        c->acc_ms[0] += a; c->acc_ms[1] += b; c->acc_ms[2] += tot;
        c->acc_frames += 1;
        if (last) { last[0] = a; last[1] = b; last[2] = tot; }
    }
    c->ev_used = 0;
    return VRT_OK;
}

// The frame's uniforms and scene pointers (everything of FrameParams that does not depend on where a frame is written).
static void fill_uniforms(const vrt_ctx *c, vrt::FrameParams &P) {
    P.n_nodes = c->max_nodes;
    // only the S^3 entries the frame's WorldData describes are addressable (find_node :120-123)
    P.n_roots = c->world.size_in_chunks * c->world.size_in_chunks * c->world.size_in_chunks;
    P.width = c->width;
    P.height = c->height;
    P.tiles_x = c->tiles_x;
    P.tiles_total = c->tiles_total;
    P.shard_first = c->shard_first;
    P.shard_run = c->shard_run;
    P.shard_period = c->shard_period;
    P.tiles_local = c->tiles_local;
    P.tile_major = c->tile_major ? 1u : 0u;
    P.compact = c->compact ? 1u : 0u;
    P.cam = c->cam;
    P.settings = c->settings;
    P.world = c->world;
    const vrt_settings &s = c->settings;
    P.finite_settings = std::isfinite(s.sun_intensity) && std::isfinite(s.sky_color[0]) && std::isfinite(s.sky_color[1]) &&
                        std::isfinite(s.sky_color[2]) && std::isfinite(s.sun_pos[0]) && std::isfinite(s.sun_pos[1]) &&
                        std::isfinite(s.sun_pos[2]);
    memcpy(P.liquid, c->liquid_mask, sizeof P.liquid);
    P.liquid_is_range = c->liquid_is_range ? 1u : 0u;
    P.liquid_lo = c->liquid_lo;
    P.liquid_span = c->liquid_span;

    P.ndc_x = c->d_ndc;
    P.ndc_y = c->d_ndc + c->width;
    cam_sun_dir(c, P.cam_sun_dir);

}

// Where one frame runs and what it writes: the caller's stream and the current output, or — frames in flight — one of
// the context's own (stream, output, launched-ray counts, path buffers, cursors) sets.
struct FrameSet {
    uint32_t slot = 0;   // 0: the context's own set, k: extra set k - 1
    hipStream_t st;
    vrt::Texel *out;
    uint32_t *blk;
    uint4 **path_buf;
    unsigned long long *counters;
};

// Two (or more) frames in flight: plain frames — one launch, or the path trace's chain of launches — alternate between
// the context's sets; anything else (stats, the two-launch variants, a caller's stream or bound buffer without
// VRT_RENDER_OWN_STREAMS) waits for them and runs alone on c->stream.
static int pick_frame_set(vrt_ctx *c, const vrt_render_opts &o, uint32_t variant, bool kstats, FrameSet &f) {
    const bool chain = o.mode == VRT_MODE_PRIMARY || (o.mode == VRT_MODE_PRIMARY_SHADOW && (variant == 0u || (variant == 2u && c->compact))) ||
                       o.mode == VRT_MODE_PATH;
    // VRT_RENDER_OWN_STREAMS: the caller set a stream and / or bound an output but lets this frame run on the context's own
    // streams (nothing on the caller's stream consumes it before a synchronise; frames in flight are bound to different
    // buffers) — the gather root's own tiles in bench.py
    const bool own_streams = (o.flags & VRT_RENDER_OWN_STREAMS) != 0u;
    const bool pipelined = c->in_flight > 1u && chain && !kstats && (own_streams || (c->stream == c->own_stream && c->d_out == c->own_out));
    const bool bound = c->d_out != c->own_out;
    f = FrameSet{0u, c->stream, c->d_out, c->d_blk_counts, &c->d_path, c->d_counters};
    if (!pipelined) {
        QUIESCE(c);
        return VRT_OK;
    }
    if (c->flip) {
        const uint32_t k = c->flip - 1u;
        f.slot = k + 1u;
        if (o.mode == VRT_MODE_PATH) {
            if (!c->extra_counters[k]) HIP_TRY(c, hipMalloc(&c->extra_counters[k], kCounterBytes));
            f.counters = c->extra_counters[k];
            f.path_buf = &c->extra_path[k];
        }
        if (!c->extra_stream[k]) HIP_TRY(c, hipStreamCreateWithFlags(&c->extra_stream[k], hipStreamNonBlocking));
        if (!bound && !c->extra_out[k]) {
            const size_t bytes = (size_t)(c->slots ? c->slots : 1) * sizeof(vrt::Texel);
            HIP_TRY(c, hipMalloc(&c->extra_out[k], bytes));
            if (ragged_output(c)) HIP_TRY(c, zero_now(c, c->extra_out[k], bytes));   // texels no workgroup covers stay zero (main.rs:452)
        }
        if (!c->extra_blk[k]) HIP_TRY(c, hipMalloc(&c->extra_blk[k], (size_t)(c->tiles_local ? c->tiles_local : 1) * sizeof(uint32_t)));
        f.st = c->extra_stream[k];
        f.blk = c->extra_blk[k];
        if (!bound) f.out = c->extra_out[k];  // a bound output is the caller's buffer for this very frame
        c->alt_pending = true;
        const int rc = frame_waits_for_uploads(c, f.st, k + 1u);
        if (rc) return rc;
    } else if (c->stream != c->own_stream) {
        f.st = c->own_stream;
        c->own_pending = true;
        const int rc = frame_waits_for_uploads(c, f.st, 0u);
        if (rc) return rc;
    }
    c->flip = (c->flip + 1u) % c->in_flight;
    return VRT_OK;
}

// The next free event quadruple of the pool (folding the pool into the accumulated times when it is full).
static int next_events(vrt_ctx *c, std::array<hipEvent_t, 4> **ev, uint8_t **kind) {
    if (c->ev_used == c->ev_pool.size()) {
        // (512 frames of events: creating one costs the host a few microseconds, so a context is at full speed once it has
        // rendered that many frames between two vrt_get_stats calls; folding costs one drain per 512 frames)
        if (c->ev_pool.size() >= 512) {
            const int rc = fold_events(c, nullptr);
            if (rc) return rc;
        } else {
            std::array<hipEvent_t, 4> t{nullptr, nullptr, nullptr, nullptr};
            for (auto &e : t) HIP_TRY(c, hipEventCreate(&e));
            c->ev_pool.push_back(t);
            c->ev_kind.push_back(kEvNone);
        }
    }
    *kind = &c->ev_kind[c->ev_used];
    **kind = kEvNone;
    *ev = &c->ev_pool[c->ev_used++];
    return VRT_OK;
}

// Wavefront path trace: per sample one launch per bounce over the compacted live-path buffer.
static int launch_path_frame(vrt_ctx *c, vrt::FrameParams &P, const FrameSet &f, const vrt_render_opts &o, bool kstats, bool literal,
                             std::array<hipEvent_t, 4> &ev, uint8_t &ev_kind) {
    const uint32_t spp = o.spp ? o.spp : 1u, bounces = c->settings.max_ray_bounces;
#ifdef VRT_EXPERIMENTS
    if (bounces > 0 && !kstats && !literal && P.grid && c->path_persistent) {
        // VRT_PATH_PERSISTENT=1 (built and measured, not the default: 10.8 against 13.0 Grays/s on C4):
        // persistent waves whose lanes own pixels and are refilled in batches (vrt_path.hip); one launch per frame
        // whatever spp and the bounce count are, no path buffers.  The tile queues' eight heads live at the start of this
        // frame set's segment-counter area (zeroed with the counters just before).
        if (!c->n_cus) {
            hipDeviceProp_t prop;
            HIP_TRY(c, hipGetDeviceProperties(&prop, c->device));
            c->n_cus = (uint32_t)prop.multiProcessorCount;
        }
        P.spp = spp;
        P.seed = o.seed;
        if (ev[0]) HIP_TRY(c, hipEventRecord(ev[0], f.st));
        vrt::launch_path_persistent(P, P.seg_counts, c->n_cus, f.st);
        HIP_TRY(c, hipGetLastError());
        if (ev[0]) {
            HIP_TRY(c, hipEventRecord(ev[1], f.st));
            HIP_TRY(c, hipEventRecord(ev[3], f.st));
            ev_kind = kEvRecorded;
        }
        c->last_spp = spp;
        return VRT_OK;
    }
#endif
    // Several samples per launch chain (plain frames, spp > 1): every launch of the chain carries `samples` times the rays —
    // 2.7 rays per lane are not enough to cover a bounce launch's tail — and a frame of 16 spp is 4 x 4
    // launches instead of 16 x 4.  Each sample accumulates into its own plane; the chain's finishing pass adds the planes
    // to the frame in sample order, which is the order one sample per chain adds them in.
    const uint32_t samples = (spp > 1u && !kstats && !literal && P.grid && bounces > 0) ? (spp < c->path_samples ? spp : c->path_samples) : 1u;
    const bool planes = samples > 1u;
    const uint32_t seg_cap = c->hit_seg_cap * samples;
    const size_t cap = (size_t)vrt::kHitSegments * seg_cap;
    if (c->path_buf_records[f.slot] < cap) {   // (grows only; hipFree waits for whatever still uses the old one)
        (void)hipFree(*f.path_buf);
        *f.path_buf = nullptr; c->path_buf_records[f.slot] = 0;
        HIP_TRY(c, hipMalloc(f.path_buf, 2 * 3 * cap * sizeof(uint4)));
        c->path_buf_records[f.slot] = cap;
    }
    if (planes && c->path_acc_texels[f.slot] < (size_t)samples * c->slots) {
        (void)hipFree(c->path_acc[f.slot]);
        c->path_acc[f.slot] = nullptr; c->path_acc_texels[f.slot] = 0;
        HIP_TRY(c, hipMalloc(&c->path_acc[f.slot], (size_t)samples * c->slots * sizeof(vrt::Texel)));
        if (ragged_output(c)) HIP_TRY(c, zero_now(c, c->path_acc[f.slot], (size_t)samples * c->slots * sizeof(vrt::Texel)));
        c->path_acc_texels[f.slot] = (size_t)samples * c->slots;
    }
    vrt::Texel *const frame_out = P.out;
    P.hit_seg_cap = seg_cap;
    P.acc = planes ? c->path_acc[f.slot] : nullptr;
    P.acc_slots = c->slots;
    P.chain = 1u;
    if (planes) P.out = c->path_acc[f.slot];   // what the bounce launches accumulate into, through slots that carry the plane
    constexpr uint32_t kSegWords = vrt::kHitSegments * vrt::kSegStride;
    uint32_t *seg[3] = {P.seg_counts, P.seg_counts + kSegWords, P.seg_counts + 2 * kSegWords};
    uint4 *buf[2] = {*f.path_buf, *f.path_buf + 3 * cap};
    P.path_cap = (uint32_t)cap;
    P.in_cap = (uint32_t)cap;
    P.in_seg_cap = seg_cap;
    P.cont_out = nullptr;
    P.cont_counts = nullptr;
    P.spp = spp;
    P.seed = o.seed;
    // Bounce launches over the derived tables use the pool kernel (vrt_path.hip).  With up to kContSets of them per sample
    // they hand the rays still marching when a wave's pool runs dry to a *straggler chain* on a side stream: launch S(b)
    // marches what bounce launch b handed on plus the next segments of S(b - 1)'s own survivors, while bounce launch
    // b + 1 already runs — the few rays that graze the terrain for a hundred steps, which every bounce launch used to
    // wait for, are off the frame's critical path.  A path is in exactly one of the two chains, so nothing is shared but
    // the record sets' cursors (atomics).  The chains join at the end of every sample.
    const bool pool = !kstats && !literal && P.grid && bounces > 1 && c->path_pool;
    const bool cells = pool && P.mblk && c->path_cells;
#ifdef VRT_EXPERIMENTS
    const bool chain = pool && !cells && bounces - 1u <= kContSets && c->path_chain;
#else
    const bool chain = false;   // (the straggler chain on a side stream: the experiments build)
#endif
    uint32_t *cont_seg[kContSets];
    for (uint32_t i = 0; i < kContSets; i++) cont_seg[i] = P.seg_counts + (3 + i) * kSegWords;
    hipStream_t side = nullptr;
    hipEvent_t *sev = c->side_ev[f.slot];
    if (chain) {
        if (c->path_cont_records[f.slot] < cap) {
            (void)hipFree(c->path_cont[f.slot]);
            c->path_cont[f.slot] = nullptr; c->path_cont_records[f.slot] = 0;
            HIP_TRY(c, hipMalloc(&c->path_cont[f.slot], (size_t)kContSets * 4 * cap * sizeof(uint4)));
            c->path_cont_records[f.slot] = cap;
        }
        if (!c->side_stream[f.slot]) HIP_TRY(c, hipStreamCreateWithFlags(&c->side_stream[f.slot], hipStreamNonBlocking));
        for (int i = 0; i < 6; i++)
            if (!sev[i]) HIP_TRY(c, hipEventCreateWithFlags(&sev[i], hipEventDisableTiming));
        side = c->side_stream[f.slot];
    }
    uint4 *cont = c->path_cont[f.slot];
    (void)cont;
    const bool timed = ev[0] != nullptr;
    if (timed) HIP_TRY(c, hipEventRecord(ev[0], f.st));
    if (bounces == 0) HIP_TRY(c, hipMemsetAsync(f.out, 0, (size_t)c->slots * sizeof(vrt::Texel), f.st));
    bool first = true;
    uint32_t g = 0;   // launch number within the frame (all three cursor sets are zero when it starts: vrt_render cleared them)
    for (uint32_t smp = 0; smp < spp && bounces > 0; smp += samples) {
        P.sample = smp;
        P.chain = spp - smp < samples ? spp - smp : samples;
        // (the chain's cursors are zero at the start of a frame — vrt_render cleared the counters — and again for every
        // further sample; the chains have joined by then)
        if (chain && smp > 0) HIP_TRY(c, hipMemsetAsync(cont_seg[0], 0, kContSets * kSegBytes, f.st));
        for (uint32_t b = 0; b < bounces; b++, g++) {
            P.seg_counts = seg[g % 3u];
            P.seg_in = seg[(g + 2u) % 3u];
            P.seg_clear = seg[(g + 1u) % 3u];
            P.path_out = buf[g & 1u];
            P.path_in = buf[(g + 1u) & 1u];
            P.last_bounce = b + 1 == bounces;
            P.cont_out = nullptr;
            P.cont_counts = nullptr;
            // one sample per pixel: the lane that ends a path has the pixel's final value (x / 1 = x) — no finishing pass
            if (b == 0) {
                vrt::launch_path_primary(P, kstats, literal, f.st);
            } else if (!pool) {
                vrt::launch_path_bounce(P, kstats, literal, f.st);
            } else if (cells) {
                vrt::launch_path_bounce_cells(P, c->path_refill, f.st);
            } else {
#ifndef VRT_EXPERIMENTS
                vrt::launch_path_bounce(P, kstats, literal, f.st);   // (a world without march cells: lane = path)
#else
                if (chain) {
                    P.cont_out = cont + (size_t)(b - 1u) * 4 * cap;
                    P.cont_counts = cont_seg[b - 1u];
                }
                vrt::launch_path_bounce_pool(P, false, c->path_refill, c->path_eject, f.st);
                if (chain) {
                    HIP_TRY(c, hipGetLastError());
                    // S(b): after bounce launch b (its hand-overs) and S(b - 1) (stream order: its survivors)
                    HIP_TRY(c, hipEventRecord(sev[b - 1u], f.st));
                    HIP_TRY(c, hipStreamWaitEvent(side, sev[b - 1u], 0));
                    vrt::FrameParams Q = P;
                    Q.path_in = P.cont_out;
                    Q.seg_in = P.cont_counts;
                    Q.seg_clear = nullptr;
                    Q.path_out = nullptr;
                    Q.seg_counts = nullptr;
                    Q.cont_out = P.last_bounce ? nullptr : cont + (size_t)b * 4 * cap;
                    Q.cont_counts = P.last_bounce ? nullptr : cont_seg[b];
                    vrt::launch_path_bounce_pool(Q, true, c->path_refill, 0u, side);
                }
#endif
            }
            HIP_TRY(c, hipGetLastError());
            if (first) { if (timed) HIP_TRY(c, hipEventRecord(ev[1], f.st)); first = false; }
        }
        if (chain) {   // the sample's two chains join
            HIP_TRY(c, hipEventRecord(sev[4], side));
            HIP_TRY(c, hipStreamWaitEvent(f.st, sev[4], 0));
        }
        if (planes) {
            vrt::launch_path_chain_finish(frame_out, c->path_acc[f.slot], c->slots, P.chain, smp == 0u, smp + P.chain >= spp, spp, f.st);
            HIP_TRY(c, hipGetLastError());
        }
    }
    P.out = frame_out;
    if (first && timed) HIP_TRY(c, hipEventRecord(ev[1], f.st));
    if (bounces > 0 && spp > 1u && !planes) {
        vrt::launch_path_finish(f.out, c->slots, spp, f.st);
        HIP_TRY(c, hipGetLastError());
    }
    if (timed) {
        HIP_TRY(c, hipEventRecord(ev[3], f.st));
        ev_kind = kEvRecorded;
    }
    c->last_spp = spp;
    return VRT_OK;
}

// Primary (+ shadow) rays: one launch (variant 0, 4; primary only) or two (variants 1-3).
static int launch_march_frame(vrt_ctx *c, const vrt::FrameParams &P, const FrameSet &f, bool shadow, uint32_t variant, bool kstats,
                              std::array<hipEvent_t, 4> &ev, uint8_t &ev_kind) {
    if (!c->tiles_local) return VRT_OK;  // an empty shard
    const uint32_t march = variant == 3u ? 0u : variant;  // variant 3 = the grid march in two launches
#ifdef VRT_EXPERIMENTS
    if (variant == 4u) {
        if (!c->d_heads) {
            HIP_TRY(c, hipMalloc(&c->d_heads, 8 * 64));
            hipDeviceProp_t prop;
            HIP_TRY(c, hipGetDeviceProperties(&prop, c->device));
            c->n_cus = (uint32_t)prop.multiProcessorCount;
        }
        HIP_TRY(c, hipMemsetAsync(c->d_heads, 0, 8 * 64, f.st));
        c->n_counts = c->tiles_local;
        vrt::launch_primary_shadow_persistent(P, c->d_heads, c->n_cus, f.st, ev[0], ev[1]);
        HIP_TRY(c, hipGetLastError());
        if (ev[0]) ev_kind = kEvOneKernel;
        return VRT_OK;
    }
#endif
    // primary + shadow in one launch: the default march, and — on a context whose pixel slots are 8-byte records — the
    // octree walk it falls back to when the world is too large for the derived tables (the two-launch kernels store and
    // re-read 16-byte texels, which such a buffer has no room for)
    const bool fused = shadow && (variant == 0u || (variant == 2u && c->compact));
    c->n_counts = fused ? c->tiles_local : c->n_blocks;
    if (fused) vrt::launch_primary_shadow_fused(P, march, kstats, f.st, ev[0], ev[1]);
    else vrt::launch_primary(P, march, kstats, shadow, f.st, ev[0], ev[1]);
    HIP_TRY(c, hipGetLastError());
    if (ev[0]) ev_kind = kEvOneKernel;
    if (shadow && !fused) {
        vrt::launch_shadow(P, march, kstats, f.st, ev[2], ev[3]);
        HIP_TRY(c, hipGetLastError());
        if (ev[0]) ev_kind = kEvTwoKernels;
    }
    return VRT_OK;
}

int vrt_render(vrt_ctx *c, const vrt_render_opts *opts) {
    if (c && c->grp) return grp_render(c, opts);
    if (!c) return VRT_ERR_INVALID_ARG;
    vrt_render_opts o;
    memset(&o, 0, sizeof o);
    if (opts) o = *opts;
    if (o.mode > VRT_MODE_PATH) return fail(c, VRT_ERR_INVALID_ARG, "vrt_render: mode %u not supported", o.mode);
    if (o.mode == VRT_MODE_PATH && o.variant != 0) return fail(c, VRT_ERR_INVALID_ARG, "vrt_render: the path trace has one kernel variant");
    if (o.stats > 2u) return fail(c, VRT_ERR_INVALID_ARG, "vrt_render: stats %u (0, 1 = count steps, 2 = clock probe)", o.stats);
    if (o.stats == 2u && (o.mode != VRT_MODE_PRIMARY_SHADOW || (o.variant != 0u)))
        return fail(c, VRT_ERR_INVALID_ARG, "vrt_render: the clock probe (stats = 2) is a build of the default primary + shadow kernel");
    if (o.variant == 4u && (o.mode != VRT_MODE_PRIMARY_SHADOW || o.stats))
        return fail(c, VRT_ERR_INVALID_ARG, "vrt_render: variant 4 (persistent grid) renders plain primary + shadow frames only");
    if (!vrt::variant_supported(o.variant)) return fail(c, VRT_ERR_INVALID_ARG, "vrt_render: unknown kernel variant %u", o.variant);
    int rc = validate_frame(c);
    if (rc) return rc;
    HIP_TRY(c, hipSetDevice(c->device));

    if (c->compact && (o.mode == VRT_MODE_PATH || (o.variant != 0u && o.variant != 2u) || c->settings.show_step_count == 1u))
        return fail(c, VRT_ERR_STATE, "vrt_render: a VRT_FLAG_COMPACT context renders primary(+shadow) frames with the default march "
                    "only (no path trace, step-count view, literal or two-launch variants)");
    if (o.stats == 1u && !c->d_steps) {
        HIP_TRY(c, hipMalloc(&c->d_steps, (size_t)(c->slots ? c->slots : 1) * sizeof(uint32_t)));
        if (ragged_output(c)) HIP_TRY(c, zero_now(c, c->d_steps, (size_t)(c->slots ? c->slots : 1) * sizeof(uint32_t)));
    }
    if (o.stats == 2u && !c->d_clock) {
        HIP_TRY(c, hipMalloc(&c->d_clock, 2 * sizeof(unsigned long long)));
        HIP_TRY(c, hipMemsetAsync(c->d_clock, 0, 2 * sizeof(unsigned long long), c->stream));
        const int rc2 = publish_upload(c);
        if (rc2) return rc2;
    }
    rc = ensure_ndc(c);
    if (rc) return rc;
    uint32_t variant = o.variant;
    // The fast marches never ask whether *air* is liquid (ray_tracer.wgsl:226 asks for every voxel, voxel 0 included): a
    // material table that flags voxel 0 as liquid — nothing the reference's data packs do — is traced by the literal march.
    const bool air_liquid = c->h_mats[0].is_liquid == 1u;
    if (air_liquid) {
        if (c->compact || o.variant == 4u)
            return fail(c, VRT_ERR_STATE, "vrt_render: materials[0].is_liquid == 1 (air flagged liquid) is traced by the literal march only");
        variant = 1u;
    }
    if (variant == 0u || variant == 3u || variant == 4u || (o.mode == VRT_MODE_PATH && !air_liquid)) {
        rc = ensure_accel_world(c);
        if (rc) return rc;
        if (!c->accel_ok && (variant == 0u || variant == 3u || variant == 4u)) variant = 2u;  // world too large for the tables: walk the octree
    }
    // per-lane iteration counts exist in the STATS kernels only; the step-count debug view (F2 in the reference,
    // main.rs:368-370) needs them, so it runs those kernels too
    const bool kstats = o.stats == 1u || c->settings.show_step_count == 1u;

    FrameSet f;
    rc = pick_frame_set(c, o, variant, kstats, f);
    if (rc) return rc;
    c->last_out = f.out;
    c->last_blk = f.blk;
    c->last_stream = f.st;
    c->last_slot = f.slot;
    if (c->wait_before_frame) {
        HIP_TRY(c, hipStreamWaitEvent(f.st, c->wait_before_frame, 0));
        c->wait_before_frame = nullptr;
    }

    // this frame's table set, brought up to date on its own stream (which waits for the uploads so far first)
    const bool wants_tables = variant == 0u || variant == 3u || variant == 4u || (o.mode == VRT_MODE_PATH && !air_liquid);
    constexpr uint32_t kQuietFrames = 64;
    if (c->tables_split && ++c->quiet_frames > kQuietFrames && c->tabs[0].dirty_chunks.empty()) {
        c->tables_split = false;   // no edit for a while: everybody reads tabs[0] again; the other sets go stale
        for (uint32_t k = 1; k < vrt_ctx::kMaxInFlight; k++) c->tabs[k].live = false;
    }
    const uint32_t tab = c->tables_split ? f.slot : 0u;
    const vrt_ctx::Tables &T = c->tabs[tab];
    c->last_tab = tab;
    if (wants_tables && c->accel_ok) {
        rc = frame_waits_for_uploads(c, f.st, f.slot);
        if (rc) return rc;
        rc = update_tables(c, tab, f.st);
        if (rc) return rc;
        if (tab == 0u && f.slot != 0u) {   // the shared set from another frame set's stream: behind its last update
            if (c->tabs[0].update_pending && f.st != c->stream) HIP_TRY(c, hipStreamWaitEvent(f.st, c->tabs[0].ev_updated, 0));
            c->shared_readers_in_flight = true;
        }
    } else {
        rc = frame_waits_for_uploads(c, f.st, f.slot);   // (a frame on c->stream too: the node pool's uploads have their own stream)
        if (rc) return rc;
    }

    vrt::FrameParams P;
    memset(&P, 0, sizeof P);
    P.nodes = c->d_nodes;
    P.roots = c->d_roots;
    P.mats = c->d_mats;
    if (wants_tables && c->accel_ok && c->accel_S == c->world.size_in_chunks && !c->accel_dirty && T.live && T.dirty_chunks.empty() && !air_liquid) {
        P.grid = T.d_grid;
        P.bricks = T.d_bricks;
        P.grid_dim = c->accel_S * 8u;
        const size_t G = (size_t)c->accel_S * 8u;
        P.grid_bytes = (uint32_t)(G * (G + 1u) * (G + 1u) * sizeof(uint32_t));  // [8S][8S+1][8S+1]: the zero border
        P.brick_bytes = (uint32_t)((size_t)T.brick_cap * 64u * sizeof(uint16_t));
        if (T.d_mblk) {
            P.cdir = T.d_cdir;
            P.cdir_bytes = (uint32_t)(chunk_dir_entries(c->accel_S) * sizeof(uint32_t));
            P.mblk = T.d_mblk;
            // (a direct world: exactly its lines — a position beyond the last slab must be out of range, it reads as zeros)
            P.mblk_bytes = (uint32_t)(c->march_direct ? direct_cell_entries(c->accel_S) * sizeof(uint4) : (size_t)T.mblk_cap * 512u * sizeof(uint4));
            P.march_direct = c->march_direct ? 1u : 0u;
        }
    }
    // a march that walks the octree reads the node pool and chunk_roots: uploads then wait for the frames in flight
    if (!P.grid || variant == 1u || variant == 2u) c->walkers_in_flight = true;
    P.out = f.out;
    P.hits = c->d_hits;
    P.blk_counts = f.blk;
    P.counters = f.counters;
    P.seg_counts = reinterpret_cast<uint32_t *>(f.counters + vrt::kCtrCount);
    P.hit_seg_cap = c->hit_seg_cap;
    P.steps = o.stats == 1u ? c->d_steps : nullptr;
    P.clock = o.stats == 2u ? c->d_clock : nullptr;
    fill_uniforms(c, P);

    std::array<hipEvent_t, 4> *ev = nullptr;
    uint8_t *ev_kind = nullptr;
    static std::array<hipEvent_t, 4> no_events{nullptr, nullptr, nullptr, nullptr};
    static uint8_t no_kind = 0;
    if (c->timing_every > 1u && (c->frame_no++ % c->timing_every) != 0u && !kstats && !(o.flags & VRT_RENDER_TIMED)) {
        ev = &no_events;   // an untimed frame: the launches carry no events (vrt_stats' kernel times average the timed ones)
        ev_kind = &no_kind;
    } else {
        rc = next_events(c, &ev, &ev_kind);
        if (rc) return rc;
    }
    // longest tiles first: the one-launch primary + shadow kernel over the derived tables, plain frames, one frame at a time
    // on the context's own stream (a frame, the sort behind it and the next frame are then ordered by the stream alone)
    const bool lpt = c->tile_lpt && c->in_flight == 1u && f.st == c->stream && (o.mode == VRT_MODE_PRIMARY_SHADOW || o.mode == VRT_MODE_PRIMARY) && variant == 0u && !kstats &&
                     o.stats == 0u && P.grid && c->tiles_local >= 128u;
    bool tile_sort = false;
    // (a tile's trips depend on the mode too — a primary-only frame has no shadow march: an order made from the other
    // mode's frame is a stale order, and the frame before a sort must be of the same kind)
    if (c->frame_mode != o.mode) c->view_gen++;
    if (lpt) {
        if (c->tile_buf_tiles != c->tiles_local) {
            const uint32_t chunks = (c->tiles_local + 63u) / 64u;
            HIP_TRY(c, hipMalloc(&c->d_tile_cost, (size_t)c->tiles_local * sizeof(uint32_t)));
            HIP_TRY(c, hipMalloc(&c->d_tile_order, (size_t)c->tiles_local * sizeof(uint32_t)));
            HIP_TRY(c, hipMalloc(&c->d_tile_scratch, (size_t)64u * (chunks + 1u) * sizeof(uint32_t)));
            c->tile_buf_tiles = c->tiles_local;
            c->tile_order_valid = false;
        }
        if (c->order_view_gen != c->view_gen) c->tile_order_valid = false;   // the order of another view: worse than none
        if (c->tile_order_valid) P.tile_order = c->d_tile_order;
        tile_sort = !c->tile_order_valid && c->frame_view_gen == c->view_gen;   // the view has come to rest: this frame notes its trips
        if (tile_sort) P.tile_cost = c->d_tile_cost;
    }
    c->frame_view_gen = c->view_gen;
    c->frame_mode = o.mode;
    // the counters feed stats frames and the path trace's segment cursors; a plain primary(+shadow) frame reads none
    if (kstats || o.mode == VRT_MODE_PATH) HIP_TRY(c, hipMemsetAsync(f.counters, 0, kCounterBytes, f.st));
    if (o.mode == VRT_MODE_PATH) rc = launch_path_frame(c, P, f, o, kstats, air_liquid, *ev, *ev_kind);
    else rc = launch_march_frame(c, P, f, o.mode == VRT_MODE_PRIMARY_SHADOW, variant, kstats, *ev, *ev_kind);
    if (rc) return rc;
    if (tile_sort) {   // (the frame above read the old order and is over when this runs; the next frame starts after it)
        vrt::launch_tile_order(c->d_tile_cost, c->tiles_local, 1u, c->d_tile_scratch, c->d_tile_order, f.st);   // classes of two trips
        HIP_TRY(c, hipGetLastError());
        c->tile_order_valid = true;
        c->order_view_gen = c->view_gen;
    }
    c->rendered = true;
    c->last_stats = o.stats == 1u;
    c->last_mode = o.mode;
    c->timing_pending = true;
    return VRT_OK;
}

int vrt_synchronize(vrt_ctx *c) {
    if (c && c->grp) return grp_synchronize(c);
    if (!c) return VRT_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    QUIESCE(c);
    {
        const int rc = flush_staged(c);
        if (rc) return rc;
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->up_stream) HIP_TRY(c, hipStreamSynchronize(c->up_stream));
    c->walkers_in_flight = false;   // nothing is in flight any more
    for (auto &T : c->tabs) T.update_pending = false;
    return VRT_OK;
}

int vrt_read_output(vrt_ctx *c, float *rgb, uint32_t *ids, uint8_t *rgba8) {
    if (c && c->grp) { const int rc_ = grp_synchronize(c); if (rc_) return rc_; }
    GRP_ROOT(c, vrt_read_output(d, rgb, ids, rgba8));
    if (!c) return VRT_ERR_INVALID_ARG;
    if (!c->rendered) return fail(c, VRT_ERR_STATE, "vrt_read_output: nothing rendered yet");
    if (c->compact) return fail(c, VRT_ERR_STATE, "vrt_read_output: a VRT_FLAG_COMPACT context holds 8-byte records, not texels (vrt_assemble_compact shades them)");
    HIP_TRY(c, hipSetDevice(c->device));
    QUIESCE(c);
    const size_t npix = (size_t)c->width * c->height;
    if (rgba8) {
        if (c->tile_major) return fail(c, VRT_ERR_STATE, "vrt_read_output: rgba8 readback needs the row-major (unsharded) layout");
        if (!c->d_rgba8) HIP_TRY(c, hipMalloc(&c->d_rgba8, npix * 4));
        vrt::launch_quantize(c->last_out, c->d_rgba8, c->width, c->height, c->stream);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipMemcpyAsync(rgba8, c->d_rgba8, npix * 4, hipMemcpyDeviceToHost, c->stream));
    }
    std::vector<vrt::Texel> t;
    if (rgb || ids) {
        t.resize(c->slots);
        HIP_TRY(c, hipMemcpyAsync(t.data(), c->last_out, t.size() * sizeof(vrt::Texel), hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (!rgb && !ids) return VRT_OK;
    auto put = [&](size_t dst, const vrt::Texel &x) {
        if (rgb) { memcpy(rgb + dst * 3, &x, 12); }
        if (ids) ids[dst] = x.w;
    };
    if (!c->tile_major) {
        for (size_t i = 0; i < npix; i++) put(i, t[i]);
        return VRT_OK;
    }
    // sharded: de-interleave this context's tiles; foreign tiles read as zero
    if (rgb) memset(rgb, 0, npix * 3 * sizeof(float));
    if (ids) memset(ids, 0, npix * sizeof(uint32_t));
    for (uint32_t tl = 0; tl < c->tiles_local; tl++) {
        const uint32_t tile = vrt::shard_tile(tl, c->shard_first, c->shard_run, c->shard_period);
        const uint32_t tx = (tile % c->tiles_x) * 8u, ty = (tile / c->tiles_x) * 8u;
        for (uint32_t p = 0; p < 64; p++) put((size_t)(ty + (p >> 3)) * c->width + tx + (p & 7u), t[(size_t)tl * 64 + p]);
    }
    return VRT_OK;
}

// ScreenShader::encode_pass into the context's screen buffer on the device; asynchronous on c->stream.
static int present_on_device(vrt_ctx *c, const vrt_crosshair *crosshair, uint32_t screen_w, uint32_t screen_h, const char *who) {
    if (!c->rendered) return fail(c, VRT_ERR_STATE, "%s: nothing rendered yet", who);
    if (c->tile_major || (c->shard_count > 1u && !c->whole_frame_owner))
        return fail(c, VRT_ERR_STATE, "%s: needs the whole row-major frame", who);
    if (screen_w == 0u || screen_h == 0u || (uint64_t)screen_w * screen_h > (1ull << 28))
        return fail(c, VRT_ERR_INVALID_ARG, "%s: screen %ux%u out of range", who, screen_w, screen_h);
    HIP_TRY(c, hipSetDevice(c->device));
    QUIESCE(c);
    const size_t bytes = (size_t)screen_w * screen_h * 4u;
    if (bytes > c->screen_cap) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));   // an earlier present may still be writing the old buffer
        (void)hipFree(c->d_screen);
        c->d_screen = nullptr; c->screen_cap = 0;
        HIP_TRY(c, hipMalloc(&c->d_screen, bytes));
        c->screen_cap = bytes;
    }
    vrt::launch_present(c->last_out, c->width, c->height, screen_w, screen_h, *crosshair, c->d_screen, c->stream);
    HIP_TRY(c, hipGetLastError());
    return VRT_OK;
}

int vrt_present(vrt_ctx *c, const vrt_crosshair *crosshair, uint32_t screen_w, uint32_t screen_h, uint8_t *rgba8) {
    if (c && c->grp) { const int rc_ = grp_synchronize(c); if (rc_) return rc_; }
    GRP_ROOT(c, vrt_present(d, crosshair, screen_w, screen_h, rgba8));
    if (!c || !crosshair || !rgba8) return fail(c, VRT_ERR_INVALID_ARG, "vrt_present: null argument");
    const int rc = present_on_device(c, crosshair, screen_w, screen_h, "vrt_present");
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(rgba8, c->d_screen, (size_t)screen_w * screen_h * 4u, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return VRT_OK;
}

int vrt_present_device(vrt_ctx *c, const vrt_crosshair *crosshair, uint32_t screen_w, uint32_t screen_h, void **rgba8_device, uint64_t *bytes) {
    if (c && c->grp) { const int rc_ = grp_synchronize(c); if (rc_) return rc_; }
    GRP_ROOT(c, vrt_present_device(d, crosshair, screen_w, screen_h, rgba8_device, bytes));
    if (!c || !crosshair || !rgba8_device) return fail(c, VRT_ERR_INVALID_ARG, "vrt_present_device: null argument");
    const int rc = present_on_device(c, crosshair, screen_w, screen_h, "vrt_present_device");
    if (rc) return rc;
    *rgba8_device = c->d_screen;
    if (bytes) *bytes = (uint64_t)screen_w * screen_h * 4u;
    return VRT_OK;
}

int vrt_read_steps(vrt_ctx *c, uint32_t *steps) {
    GRP_REFUSE(c, "vrt_read_steps");
    if (!c || !steps) return fail(c, VRT_ERR_INVALID_ARG, "vrt_read_steps: null argument");
    if (!c->rendered || !c->last_stats || !c->d_steps)
        return fail(c, VRT_ERR_STATE, "vrt_read_steps: the last frame was not rendered with opts.stats = 1");
    if (c->tile_major) return fail(c, VRT_ERR_STATE, "vrt_read_steps: needs the row-major (unsharded) layout");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(steps, c->d_steps, (size_t)c->width * c->height * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return VRT_OK;
}

int vrt_get_stats(vrt_ctx *c, vrt_stats *out) {
    if (c && c->grp) return grp_get_stats(c, out);
    if (!c || !out) return fail(c, VRT_ERR_INVALID_ARG, "vrt_get_stats: null argument");
    if (!c->rendered) return fail(c, VRT_ERR_STATE, "vrt_get_stats: nothing rendered yet");
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->timing_pending) {
        float last[3] = {0, 0, 0};
        int rc = fold_events(c, last);
        if (rc) return rc;
        std::vector<unsigned long long> hbuf(kCounterBytes / sizeof(unsigned long long));
        HIP_TRY(c, hipMemcpyAsync(hbuf.data(), c->d_counters, kCounterBytes, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        unsigned long long *h = hbuf.data();
        const uint32_t *seg = reinterpret_cast<const uint32_t *>(h + vrt::kCtrCount);
        unsigned long long launched = 0;
        if (c->last_mode == VRT_MODE_PRIMARY_SHADOW && c->n_counts) {
            std::vector<uint32_t> bc(c->n_counts);
            HIP_TRY(c, hipMemcpyAsync(bc.data(), c->last_blk, bc.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            for (uint32_t v : bc) launched += v;
        }
        (void)seg;
        vrt_stats s;
        memset(&s, 0, sizeof s);
        s.primary_rays = (uint64_t)c->tiles_local * 64u * (c->last_mode == VRT_MODE_PATH ? c->last_spp : 1u);
        s.secondary_rays = c->last_mode == VRT_MODE_PRIMARY_SHADOW ? launched : 0;
        if (c->last_mode == VRT_MODE_PATH && c->last_stats) s.secondary_rays = h[vrt::kCtrSecondary];
        if (c->last_stats) {
            s.hits = h[vrt::kCtrHits];
            s.steps = h[vrt::kCtrSteps];
            s.node_visits = h[vrt::kCtrVisits];
            s.primary_steps = h[vrt::kCtrPrimarySteps];
            s.primary_node_visits = h[vrt::kCtrPrimaryVisits];
        }
        if (c->d_clock) {  // clock-probe frames since the last call
            unsigned long long clk[2] = {0, 0};
            HIP_TRY(c, hipMemcpyAsync(clk, c->d_clock, sizeof clk, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipMemsetAsync(c->d_clock, 0, sizeof clk, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            s.clock_shader_ticks = clk[0];
            s.clock_ref_ticks = clk[1];
        }
        s.ms_primary = last[0]; s.ms_secondary = last[1]; s.ms_total = last[2];
        s.frames = c->acc_frames;
        s.sum_ms_primary = c->acc_ms[0]; s.sum_ms_secondary = c->acc_ms[1]; s.sum_ms_total = c->acc_ms[2];
        c->acc_frames = 0;
        c->acc_ms[0] = c->acc_ms[1] = c->acc_ms[2] = 0;
        c->stats = s;
        c->timing_pending = false;
    }
    *out = c->stats;
    return VRT_OK;
}

// tabs[0] is what vrt_get_accel_info / vrt_read_accel report: apply what it has not caught up with (the frames since may
// have run on other frame sets).  Waits for the frames in flight.
static int first_tables_up_to_date(vrt_ctx *c) {
    if (!c->accel_ok || c->accel_dirty || !c->tabs[0].live) return VRT_OK;
    QUIESCE(c);
    int rc = wait_for_pool_uploads(c, c->stream, vrt_ctx::kMaxInFlight);
    if (rc) return rc;
    if (c->ev_upload && c->stream != c->own_stream) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_upload, 0));
    rc = update_tables(c, 0, c->stream);
    if (rc) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return VRT_OK;
}

// Bricks of the pool in use: the chunks' regions plus the tail regions of chunks that moved.
static int bricks_in_use(vrt_ctx *c, const vrt_ctx::Tables &T, uint32_t *n) {
    *n = 0;
    if (!c->accel_ok || !T.live || !T.d_brick_tail) return VRT_OK;
    QUIESCE(c);   // (the set's last update may be on another frame stream)
    HIP_TRY(c, hipMemcpyAsync(n, T.d_brick_tail, sizeof *n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (*n > T.brick_cap) *n = T.brick_cap;
    return VRT_OK;
}

int vrt_get_accel_info(vrt_ctx *c, vrt_accel_info *out) {
    GRP_ROOT(c, vrt_get_accel_info(d, out));
    if (!c || !out) return fail(c, VRT_ERR_INVALID_ARG, "vrt_get_accel_info: null argument");
    memset(out, 0, sizeof *out);
    HIP_TRY(c, hipSetDevice(c->device));
    // (as of the table set the last frame used; the other sets catch up when their frame set renders next)
    const vrt_ctx::Tables &L = c->tabs[c->last_tab];
    const bool up_to_date = c->accel_ok && !c->accel_dirty && L.live && L.dirty_chunks.empty();
    out->available = up_to_date ? 1u : 0u;
    out->world_size_chunks = c->accel_S;
    out->cells = (uint64_t)c->accel_S * c->accel_S * c->accel_S * 512u;
    uint32_t used = 0;
    const int rc = bricks_in_use(c, L, &used);
    if (rc) return rc;
    out->bricks = used;
    const uint64_t G = (uint64_t)c->accel_S * 8u;
    out->bytes = G * (G + 1u) * (G + 1u) * sizeof(uint32_t) + (uint64_t)used * 64u * sizeof(uint16_t);
    out->builds = c->accel_builds;
    out->last_build_ms = c->accel_last_ms;
    for (const auto &T : c->tabs)
        if (T.live && T.chunk_builds > out->chunk_builds) out->chunk_builds = T.chunk_builds;   // every set rebuilds every dirty chunk once
    return VRT_OK;
}

int vrt_read_accel(vrt_ctx *c, uint32_t *grid, uint16_t *bricks) {
    GRP_ROOT(c, vrt_read_accel(d, grid, bricks));
    if (!c) return VRT_ERR_INVALID_ARG;
    if (!c->accel_ok || c->accel_dirty)
        return fail(c, VRT_ERR_STATE, "vrt_read_accel: the tables are not up to date (render a frame first)");
    HIP_TRY(c, hipSetDevice(c->device));
    {
        const int rc = first_tables_up_to_date(c);   // (the last frame may have used another frame set's tables)
        if (rc) return rc;
    }
    const size_t G = (size_t)c->accel_S * 8u, G1 = G + 1u;
    if (grid) {  // the device layout carries a zero border row / entry ([G][G+1][G+1]); the caller gets the G^3 cells
        std::vector<uint32_t> t(G * G1 * G1);
        HIP_TRY(c, hipMemcpyAsync(t.data(), c->tabs[0].d_grid, t.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        for (size_t z = 0; z < G; z++)
            for (size_t y = 0; y < G; y++) memcpy(grid + (z * G + y) * G, t.data() + (z * G1 + y) * G1, G * sizeof(uint32_t));
    }
    uint32_t used = 0;
    const int rc = bricks_in_use(c, c->tabs[0], &used);
    if (rc) return rc;
    if (bricks && used)
        HIP_TRY(c, hipMemcpyAsync(bricks, c->tabs[0].d_bricks, (size_t)used * 64u * sizeof(uint16_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return VRT_OK;
}

int vrt_read_march_cells(vrt_ctx *c, uint32_t *cells, uint32_t *direct) {
    GRP_ROOT(c, vrt_read_march_cells(d, cells, direct));
    if (!c || !cells) return fail(c, VRT_ERR_INVALID_ARG, "vrt_read_march_cells: null argument");
    if (!c->accel_ok || c->accel_dirty || !c->tabs[0].d_mblk)
        return fail(c, VRT_ERR_STATE, "vrt_read_march_cells: no march cells, or not up to date (render a frame first)");
    HIP_TRY(c, hipSetDevice(c->device));
    {
        const int rc = first_tables_up_to_date(c);   // (the last frame may have used another frame set's tables)
        if (rc) return rc;
    }
    const auto &T = c->tabs[0];
    const uint32_t S = c->accel_S;
    const size_t G = (size_t)S * 8u;
    if (direct) *direct = c->march_direct ? 1u : 0u;
    std::vector<uint4> blocks(c->march_direct ? direct_cell_entries(S) : (size_t)T.mblk_cap * 512u);
    std::vector<uint32_t> dir(chunk_dir_entries(S));
    HIP_TRY(c, hipMemcpyAsync(blocks.data(), T.d_mblk, blocks.size() * sizeof(uint4), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(dir.data(), T.d_cdir, dir.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const size_t B1 = (size_t)S * 4u + 1u;
    for (size_t z = 0; z < G; z++)
        for (size_t y = 0; y < G; y++)
            for (size_t x = 0; x < G; x++) {
                const size_t sub = (x & 1u) | ((y & 1u) << 1) | ((z & 1u) << 2);
                size_t at;
                if (c->march_direct) {
                    at = (((z >> 1) * B1 + (y >> 1)) * B1 + (x >> 1)) * 8u + sub;
                } else {
                    const uint32_t blk = dir[((z >> 3) * (S + 1u) + (y >> 3)) * (S + 1u) + (x >> 3)];
                    if (blk >= T.mblk_cap) return fail(c, VRT_ERR_DEVICE, "vrt_read_march_cells: the directory names block %u of %u", blk, T.mblk_cap);
                    const size_t line = ((((z >> 1) & 3u) << 2 | ((y >> 1) & 3u)) << 2) | ((x >> 1) & 3u);
                    at = (size_t)blk * 512u + line * 8u + sub;
                }
                memcpy(cells + ((z * G + y) * G + x) * 4u, &blocks[at], sizeof(uint4));
            }
    return VRT_OK;
}

int vrt_set_frames_in_flight(vrt_ctx *c, uint32_t n) {
    if (c && c->grp) return grp_set_frames_in_flight(c, n);
    if (!c) return VRT_ERR_INVALID_ARG;
    if (n < 1u || n > vrt_ctx::kMaxInFlight) return fail(c, VRT_ERR_INVALID_ARG, "vrt_set_frames_in_flight: 1..%u", vrt_ctx::kMaxInFlight);
    HIP_TRY(c, hipSetDevice(c->device));
    QUIESCE(c);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->in_flight = n;
    c->flip = 0;
    // the table sets of frame slots that no longer render go stale: nobody brings them up to date, so their dirty lists
    // would only grow (and, full, force whole-world builds on a context whose active sets are fine)
    for (uint32_t k = n; k < vrt_ctx::kMaxInFlight; k++) {
        auto &T = c->tabs[k];
        for (uint32_t ch : T.dirty_chunks) T.chunk_is_dirty[ch] = 0;
        T.dirty_chunks.clear();
        T.live = false;
    }
    return VRT_OK;
}

int vrt_set_stream(vrt_ctx *c, void *hip_stream) {
    GRP_REFUSE(c, "vrt_set_stream");
    if (!c) return VRT_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    QUIESCE(c);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return VRT_OK;
}

int vrt_bind_output(vrt_ctx *c, void *texels) {
    GRP_REFUSE(c, "vrt_bind_output");
    if (!c) return VRT_ERR_INVALID_ARG;
    if (texels && ((uintptr_t)texels % 16u)) return fail(c, VRT_ERR_INVALID_ARG, "vrt_bind_output: texels must be 16-byte aligned");
    // stream-ordered: launches capture the pointer, so frames already enqueued keep writing where they were
    // told to and the next vrt_render uses the new buffer (lets a host ping-pong two gather messages)
    c->d_out = texels ? (vrt::Texel *)texels : c->own_out;
    c->last_out = c->d_out;
    c->rendered = false;
    return VRT_OK;
}

int vrt_device_output(vrt_ctx *c, void **texels, uint64_t *bytes) {
    GRP_ROOT(c, vrt_device_output(d, texels, bytes));
    if (!c) return VRT_ERR_INVALID_ARG;
    if (texels) *texels = c->d_out == c->own_out ? c->last_out : c->d_out;  // own buffers: the one holding the last frame
    if (bytes) *bytes = (uint64_t)c->slots * (c->compact ? 8u : sizeof(vrt::Texel));
    return VRT_OK;
}

int vrt_shard_info(vrt_ctx *c, uint32_t *tiles_local, uint32_t *tiles_padded, uint32_t *tiles_total) {
    GRP_ROOT(c, vrt_shard_info(d, tiles_local, tiles_padded, tiles_total));
    if (!c) return VRT_ERR_INVALID_ARG;
    if (tiles_local) *tiles_local = c->tiles_local;
    if (tiles_padded) *tiles_padded = c->tiles_padded;
    if (tiles_total) *tiles_total = c->tiles_total;
    return VRT_OK;
}

int vrt_assemble(vrt_ctx *c, const void *gathered, uint64_t rank_stride_bytes, void *dst) {
    GRP_REFUSE(c, "vrt_assemble");
    if (!c) return VRT_ERR_INVALID_ARG;
    if (!gathered || !dst) return fail(c, VRT_ERR_INVALID_ARG, "vrt_assemble: null argument");
    if (rank_stride_bytes % 16u) return fail(c, VRT_ERR_INVALID_ARG, "vrt_assemble: rank stride must be a multiple of 16 bytes");
    HIP_TRY(c, hipSetDevice(c->device));
    const uint64_t stride = rank_stride_bytes ? rank_stride_bytes / 16u : (uint64_t)c->tiles_padded * 64u;
    const bool in_place = c->shard_count > 1u && !c->tile_major;  // VRT_FLAG_ROW_MAJOR root: its tiles are already in dst
    vrt::launch_assemble((const vrt::Texel *)gathered, (vrt::Texel *)dst, c->width, c->tiles_x, c->tiles_total, c->shard_w0,
                         c->shard_period, in_place, stride, c->stream);
    HIP_TRY(c, hipGetLastError());
    return VRT_OK;
}

int vrt_assemble_compact(vrt_ctx *c, const void *gathered, uint64_t rank_stride_bytes, void *dst) {
    GRP_REFUSE(c, "vrt_assemble_compact");
    if (!c) return VRT_ERR_INVALID_ARG;
    if (!gathered || !dst) return fail(c, VRT_ERR_INVALID_ARG, "vrt_assemble_compact: null argument");
    if (rank_stride_bytes % 8u) return fail(c, VRT_ERR_INVALID_ARG, "vrt_assemble_compact: rank stride must be a multiple of 8 bytes");
    if (!(c->shard_count > 1u && !c->tile_major))
        return fail(c, VRT_ERR_STATE, "vrt_assemble_compact: the gather root must be a VRT_FLAG_ROW_MAJOR shard context (it shades the "
                    "other ranks' records with its own uniforms and has its own tiles in the frame already)");
    int rc = validate_frame(c);
    if (rc) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    rc = ensure_ndc(c);
    if (rc) return rc;
    vrt::FrameParams P;
    memset(&P, 0, sizeof P);
    P.mats = c->d_mats;
    fill_uniforms(c, P);
    const uint64_t stride = rank_stride_bytes ? rank_stride_bytes / 8u : (uint64_t)c->tiles_padded * 64u;
    vrt::launch_assemble_shade(P, gathered, (vrt::Texel *)dst, c->shard_w0, c->shard_period, stride, c->stream);
    HIP_TRY(c, hipGetLastError());
    return VRT_OK;
}

#ifdef VRT_EXPERIMENTS
int vrt_experiments_build(void) { return 1; }   // (not in include/vrt.h: only `make experiments` exports it)
#endif

}  // extern "C"

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
            rc = job();
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
};

static vrt_ctx *grp_root(vrt_ctx *c) { return c->grp->dev[0]; }

template <typename F>
static int grp_each(vrt_ctx *c, F f) {
    DeviceRestore restore;
    for (vrt_ctx *d : c->grp->dev) {
        const int rc = f(d);
        if (rc) { c->err = d->err; return rc; }
    }
    return VRT_OK;
}

static int grp_alloc_messages(vrt_ctx *c) {
    vrt_group *g = c->grp;
    vrt_ctx *root = g->dev[0];
    HIP_TRY(c, hipSetDevice(root->device));
    for (auto &p : g->recv) { (void)hipFree(p); p = nullptr; }
    g->rank_stride = (size_t)root->tiles_padded * 64u * (g->texels ? 16u : 8u);
    for (auto &p : g->recv) {
        HIP_TRY(c, hipMalloc(&p, g->rank_stride * g->dev.size()));
        HIP_TRY(c, hipMemset(p, 0, g->rank_stride * g->dev.size()));
    }
    for (size_t r = 1; r < g->dev.size(); r++) {
        if (!g->staged[r]) continue;
        HIP_TRY(c, hipSetDevice(g->dev[r]->device));
        for (auto &p : g->stage[r]) {
            (void)hipFree(p); p = nullptr;
            HIP_TRY(c, hipMalloc(&p, g->rank_stride));
            HIP_TRY(c, hipMemset(p, 0, g->rank_stride));
        }
    }
    HIP_TRY(c, hipSetDevice(root->device));
    g->slot = 0;
    g->consumed_used[0] = g->consumed_used[1] = false;
    return VRT_OK;
}

static int grp_create(const vrt_config *cfg, vrt_ctx **out) {
    const uint32_t n = cfg->n_devices;
    if (n > VRT_MAX_DEVICES) return fail(nullptr, VRT_ERR_INVALID_ARG, "n_devices %u > VRT_MAX_DEVICES", n);
    if (cfg->shard_rank != 0u || cfg->shard_count > 1u)
        return fail(nullptr, VRT_ERR_INVALID_ARG, "a multi-device context shards by itself: shard_rank / shard_count must be 0");
    if (cfg->flags & ~(VRT_FLAG_TEXEL_MESSAGES | VRT_FLAG_STAGED_MESSAGES | VRT_FLAG_POISON_MESSAGES))
        return fail(nullptr, VRT_ERR_INVALID_ARG, "a multi-device context takes VRT_FLAG_TEXEL_MESSAGES, _STAGED_MESSAGES and _POISON_MESSAGES only");
    vrt_ctx *c = new (std::nothrow) vrt_ctx();
    vrt_group *g = new (std::nothrow) vrt_group();
    if (!c || !g) { delete c; delete g; return fail(nullptr, VRT_ERR_OOM, "host allocation failed"); }
    c->grp = g;
    g->texels = (cfg->flags & VRT_FLAG_TEXEL_MESSAGES) != 0u;
    g->poison = (cfg->flags & VRT_FLAG_POISON_MESSAGES) != 0u;
    g->staged.assign(n, (cfg->flags & VRT_FLAG_STAGED_MESSAGES) ? 1 : 0);
    g->stage.assign(n, {nullptr, nullptr});
    // the root's own tiles never cross a link, so it takes more of the frame; measured defaults
    const uint32_t w0 = cfg->shard_root_weight ? cfg->shard_root_weight : (n == 2u ? 4u : n <= 4u ? 3u : 2u);
    DeviceRestore restore;
    auto body = [&]() -> int {
        for (uint32_t r = 0; r < n; r++) {
            vrt_config sub = *cfg;
            sub.n_devices = 0;
            sub.device = cfg->device_ids[r];
            sub.shard_rank = r;
            sub.shard_count = n;
            sub.shard_root_weight = w0;
            sub.flags = r == 0u ? VRT_FLAG_ROW_MAJOR : (g->texels ? 0u : VRT_FLAG_COMPACT);   // (the group's own flags stay here)
            vrt_ctx *d = nullptr;
            const int rc = vrt_create(&sub, &d);
            if (rc) { c->err = g_create_err; return rc; }
            g->dev.push_back(d);
        }
        g->dev[0]->whole_frame_owner = true;
        g->done.resize(n);
        for (uint32_t r = 1; r < n; r++) {
            HIP_TRY(c, hipSetDevice(g->dev[r]->device));
            if (g->dev[r]->device != g->dev[0]->device && !g->staged[r]) {
                // no peer access: the message is rendered at home and copied over (a copy between two devices needs none)
                const hipError_t e = hipDeviceEnablePeerAccess(g->dev[0]->device, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) g->staged[r] = 1;
                (void)hipGetLastError();
            }
            for (auto &ev : g->done[r]) HIP_TRY(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        }
        HIP_TRY(c, hipSetDevice(g->dev[0]->device));
        for (auto &ev : g->consumed) HIP_TRY(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        // issuing threads pay off when the devices are different ones: launches to one device serialise inside the
        // runtime whichever thread makes them (measured with device_ids = {0, ...}: 168 us of host time per frame for 8
        // contexts with workers, 145 without).  VRT_GROUP_THREADS=1 / 0 forces them on / off.
        bool distinct = true;
        for (uint32_t a = 0; a < n; a++)
            for (uint32_t b = a + 1; b < n; b++)
                if (cfg->device_ids[a] == cfg->device_ids[b]) distinct = false;
        const char *e = getenv("VRT_GROUP_THREADS");
        if (e ? e[0] == '1' : distinct)
            for (uint32_t r = 1; r < n; r++) {
                g->workers.emplace_back(new GrpWorker());
                GrpWorker *w = g->workers.back().get();
                w->th = std::thread([w] { w->run(); });
            }
        return grp_alloc_messages(c);
    };
    const int rc = body();
    if (rc) {
        g_create_err = c->err;
        grp_destroy(c);
        return rc;
    }
    *out = c;
    return VRT_OK;
}

static void grp_destroy(vrt_ctx *c) {
    vrt_group *g = c->grp;
    DeviceRestore restore;
    for (auto &w : g->workers) w->stop();
    for (vrt_ctx *d : g->dev) {
        (void)hipSetDevice(d->device);
        (void)vrt_synchronize(d);
    }
    for (size_t r = 1; r < g->dev.size() && r < g->stage.size(); r++) {
        (void)hipSetDevice(g->dev[r]->device);
        for (auto p : g->stage[r]) (void)hipFree(p);
    }
    if (!g->dev.empty()) (void)hipSetDevice(g->dev[0]->device);
    for (auto p : g->recv) (void)hipFree(p);
    for (auto ev : g->consumed)
        if (ev) (void)hipEventDestroy(ev);
    for (auto &evs : g->done)
        for (auto ev : evs)
            if (ev) (void)hipEventDestroy(ev);
    for (vrt_ctx *d : g->dev) vrt_destroy(d);
    delete g;
    delete c;
}

static int grp_synchronize(vrt_ctx *c) { return grp_each(c, [](vrt_ctx *d) { return vrt_synchronize(d); }); }

static int grp_set_frames_in_flight(vrt_ctx *c, uint32_t n) {
    if (n < 1u || n > vrt_group::kSlots) return fail(c, VRT_ERR_INVALID_ARG, "vrt_set_frames_in_flight: 1..%u on a multi-device context", vrt_group::kSlots);
    int rc = grp_synchronize(c);
    if (rc) return rc;
    c->grp->in_flight = n;
    return grp_each(c, [&](vrt_ctx *d) { return vrt_set_frames_in_flight(d, n); });
}

static int grp_resize_output(vrt_ctx *c, uint32_t w, uint32_t h) {
    DeviceRestore restore;
    int rc = grp_synchronize(c);
    if (rc) return rc;
    rc = grp_each(c, [&](vrt_ctx *d) { return vrt_resize_output(d, w, h); });
    if (rc) return rc;
    return grp_alloc_messages(c);
}

static int grp_render(vrt_ctx *c, const vrt_render_opts *opts) {
    vrt_group *g = c->grp;
    vrt_ctx *root = g->dev[0];
    vrt_render_opts o;
    memset(&o, 0, sizeof o);
    if (opts) o = *opts;
    if (!g->texels && (o.mode == VRT_MODE_PATH || (o.variant != 0u && o.variant != 2u)))
        return fail(c, VRT_ERR_STATE, "vrt_render: this multi-device context exchanges 8-byte records (primary(+shadow) frames of the default "
                    "march); create it with VRT_FLAG_TEXEL_MESSAGES for the path trace and the other marches");
    if (o.stats == 2u || o.variant == 4u) return fail(c, VRT_ERR_INVALID_ARG, "vrt_render: no clock probe / persistent grid on a multi-device context");
    const uint32_t n = (uint32_t)g->dev.size();
    const bool plain = o.stats == 0u && root->settings.show_step_count != 1u;
    if (!plain || g->in_flight == 1u || g->last_was_stats) {   // a stats frame (counters are read back) stands alone
        const int rc = grp_synchronize(c);
        if (rc) return rc;
    }
    g->last_was_stats = !plain;
    const uint32_t k = g->slot;
    g->slot = (g->slot + 1u) % (g->in_flight > 1u ? vrt_group::kSlots : 1u);
    o.flags |= VRT_RENDER_OWN_STREAMS;   // every device's frame runs on that context's in-flight streams, into the buffer bound here
    // what is issued to device r >= 1 for this frame — by its worker thread, or here
    auto issue = [g, k, o, root](uint32_t r) -> int {
        vrt_ctx *d = g->dev[r];
        if (hipSetDevice(d->device) != hipSuccess) return fail(d, VRT_ERR_DEVICE, "hipSetDevice(%d) failed", d->device);
        void *slot = (uint8_t *)g->recv[k] + (size_t)r * g->rank_stride;
        int rc = vrt_bind_output(d, g->staged[r] ? g->stage[r][k] : slot);
        // the slot's previous message must have been consumed by device 0 before this frame overwrites it
        d->wait_before_frame = g->consumed_used[k] ? g->consumed[k] : nullptr;
        if (!rc) rc = vrt_render(d, &o);
        d->wait_before_frame = nullptr;
        if (rc) return rc;
        // (a staged message: behind the frame on its stream — which waited for the slot to be consumed — over to device 0)
        if (d->tiles_local && g->staged[r] &&
            hipMemcpyPeerAsync(slot, root->device, g->stage[r][k], d->device, g->rank_stride, d->last_stream) != hipSuccess)
            return fail(d, VRT_ERR_DEVICE, "hipMemcpyPeerAsync from device %d to device %d failed", d->device, root->device);
        if (d->tiles_local && hipEventRecord(g->done[r][k], d->last_stream) != hipSuccess)
            return fail(d, VRT_ERR_DEVICE, "hipEventRecord failed on device %d", d->device);
        return VRT_OK;
    };
    DeviceRestore restore;
    if (!g->workers.empty()) {
        for (uint32_t r = 1; r < n; r++) g->workers[r - 1]->post([issue, r] { return issue(r); });
    } else {
        for (uint32_t r = 1; r < n; r++) {
            const int rc = issue(r);
            if (rc) { c->err = g->dev[r]->err; return rc; }
        }
    }
    int rc = hipSetDevice(root->device) == hipSuccess ? VRT_OK : fail(root, VRT_ERR_DEVICE, "hipSetDevice(%d) failed", root->device);
    if (!rc) rc = vrt_render(root, &o);
    // the workers have *enqueued* their frames (their done events are recorded) before the root's stream is told to wait —
    // and they are joined on every path out of here: nothing of a context is ever touched by two threads
    int wrc = VRT_OK;
    for (uint32_t r = 1; r < n && !g->workers.empty(); r++) {
        const int one = g->workers[r - 1]->join();
        if (one && !wrc) { wrc = one; c->err = g->dev[r]->err; }
    }
    if (rc) { c->err = root->err; return rc; }
    if (wrc) return wrc;
    hipStream_t X = root->last_stream ? root->last_stream : root->stream;
    for (uint32_t r = 1; r < n; r++)
        if (g->dev[r]->tiles_local) HIP_TRY(c, hipStreamWaitEvent(X, g->done[r][k], 0));
    // shade / scatter the other devices' messages into the frame the root has just rendered its own tiles into
    vrt::Texel *frame = root->last_out;
    if (g->texels) {
        vrt::launch_assemble((const vrt::Texel *)g->recv[k], frame, root->width, root->tiles_x, root->tiles_total, root->shard_w0,
                             root->shard_period, true, g->rank_stride / 16u, X);
    } else {
        vrt::FrameParams P;
        memset(&P, 0, sizeof P);
        P.mats = root->d_mats;
        fill_uniforms(root, P);
        vrt::launch_assemble_shade(P, g->recv[k], frame, root->shard_w0, root->shard_period, g->rank_stride / 8u, X);
    }
    HIP_TRY(c, hipGetLastError());
    // (testing: a consumed slot holds nothing a later frame could pass for its own)
    if (g->poison) HIP_TRY(c, hipMemsetAsync((uint8_t *)g->recv[k] + g->rank_stride, 0xFF, g->rank_stride * (n - 1u), X));
    HIP_TRY(c, hipEventRecord(g->consumed[k], X));
    g->consumed_used[k] = true;
    return VRT_OK;
}

static int grp_get_stats(vrt_ctx *c, vrt_stats *out) {
    if (!out) return fail(c, VRT_ERR_INVALID_ARG, "vrt_get_stats: null argument");
    DeviceRestore restore;
    int rc = grp_synchronize(c);
    if (rc) return rc;
    vrt_stats acc;
    memset(&acc, 0, sizeof acc);
    bool first = true;
    for (vrt_ctx *d : c->grp->dev) {
        vrt_stats s;
        (void)hipSetDevice(d->device);
        rc = vrt_get_stats(d, &s);
        if (rc) { c->err = d->err; return rc; }
        acc.primary_rays += s.primary_rays; acc.secondary_rays += s.secondary_rays; acc.hits += s.hits;
        acc.steps += s.steps; acc.node_visits += s.node_visits; acc.primary_steps += s.primary_steps;
        acc.primary_node_visits += s.primary_node_visits;
        if (first) {   // kernel times: the root's own launches
            acc.ms_total = s.ms_total; acc.ms_primary = s.ms_primary; acc.ms_secondary = s.ms_secondary; acc.frames = s.frames;
            acc.sum_ms_primary = s.sum_ms_primary; acc.sum_ms_secondary = s.sum_ms_secondary; acc.sum_ms_total = s.sum_ms_total;
            first = false;
        }
    }
    *out = acc;
    return VRT_OK;
}
