// vrt_device.h — shared device-side definitions for the gfx950 SVO ray-march kernels.
//
// Float contract: every expression that feeds control flow is strict IEEE
// binary32 in the evaluation order the reference shader text gives
// (clientdesktop/src/graphics/ray_tracer.wgsl).  The translation unit is built with
// -ffp-contract=off and hipcc's default correctly-rounded f32 divide/sqrt; no fast-math.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vrt.h"

namespace vrt {

constexpr float kShadowBias = 0.002f;   // DESIGN.md §Shadow rays (build-defined)
constexpr float kShadowFactor = 0.35f;
constexpr uint32_t kMaxSteps = 500u;    // ray_tracer.wgsl:220
// compact records only: norm.y == -1 (the one thing face shading needs beyond the id word's norm flags, :298-306)
constexpr uint32_t kIdNormYNeg = 1u << 23;

// The cell grid's entry of an air leaf (vrt_accel.hip): lo = leaf size - 1 under nine set bits.  The march inserts a direction
// mask into the position under the entry as the bit selector (vrt_march.h (h)): with the top nine bits set, the insert takes
// them from the mask — which carries the exponent of 2^23 there — and the exit plane arrives as the float 2^23 + plane
// without an instruction of its own ((q) of vrt_march.h).  The march cells' first word keeps the plain form (lo alone).
constexpr uint32_t kCamNotFinite = 1u, kCamOnPlane = 2u, kCamOutside = 4u;   // FrameParams.cam_origin_facts

constexpr uint32_t kAirLeaf = 0xFF800000u;
// ... of a cell split at depth 3: 0x80000000 | brick * 64 — below the air leaves: the pool holds fewer than 0x1FE0000 bricks
__host__ __device__ inline bool is_split_entry(uint32_t e) { return (int32_t)e < 0 && e < kAirLeaf; }
// a march cell's first word (lo of an air leaf, voxel << 16 | lo, 0x80000000 | brick * 64) as the cell grid holds it
__host__ __device__ inline uint32_t grid_entry(uint32_t x) { return x - 1u < 31u ? (kAirLeaf | x) : x; }

// One output texel: {r, g, b as f32 bits, id word}. 16 B so that a wave stores 1 KiB contiguously.
using Texel = uint4;

// Everything a frame's kernels read, passed by value (kernarg -> SGPRs).
struct FrameParams {
    const uint16_t *nodes;   // flat node pool, little-endian u16 == the reference's packed u32 pairs
    const uint32_t *roots;   // chunk_roots, S^3
    const vrt_material *mats;  // 256 x 32 B
    // derived lookup tables of the default march (vrt_accel.hip): one u32 per depth-3 cell of the world, x-major,
    // and 64 u16 per split cell
    const uint32_t *grid;
    const uint16_t *bricks;
    uint32_t grid_dim;       // cells per world axis = 8 * size_in_chunks
    uint32_t grid_bytes, brick_bytes;
    // the march cells (vrt_accel.hip): a chunk directory u32[S][S+1][S+1] -> 8-KiB blocks of 512 cells x 16 bytes (the cell's
    // entry, which of its 2^3 sub-blocks are depth-4 leaves, which of its 64 voxels a ray passes); null when not kept
    const uint32_t *cdir;
    const uint4 *mblk;
    uint32_t cdir_bytes, mblk_bytes;
    uint32_t march_direct;   // 1: the block of directory position i is 2 + i (small worlds; no directory load in the march)
    Texel *out;              // one texel per pixel slot
    uint4 *hits;             // hit buffer {slot, origin.xyz bits}: 256 records per primary workgroup, compacted per workgroup
    uint32_t *blk_counts;    // records appended by primary workgroup b
    uint32_t *seg_counts;    // path mode: records in segment s of the path buffer at seg_counts[s * kSegStride]
    unsigned long long *counters;  // see Counter
    uint32_t *steps;         // optional per-slot step counts (stats frames only), may be null
    unsigned long long *clock;  // clock-probe frames only (vrt_render_opts.stats = 2): {shader-clock ticks, 100 MHz reference ticks}
    // path-trace mode: wavefront of live paths, ping-pong between bounces. A record is three uint4 planes
    // {slot, origin.xyz} {dir.xyz, rng} {throughput.rgb, 0} of path_cap entries each, segmented like `hits`.
    const uint4 *path_in;
    uint4 *path_out;
    const uint32_t *seg_in;  // segment counters of path_in
    uint32_t *seg_clear;     // the counter set the NEXT launch of the frame appends to: this launch zeroes it (nobody reads it meanwhile)
    uint32_t path_cap;       // entries per plane = kHitSegments * hit_seg_cap
    uint32_t in_seg_cap;     // capacity of one segment of path_in (hit_seg_cap)
    uint32_t in_cap;         // entries per plane of path_in; path_cap is path_out's
    uint32_t spp, sample, seed;
    // several samples per launch chain (plain frames with spp > 1): the primary launch traces samples sample .. sample +
    // chain - 1 of every pixel, sample s into its own plane of `acc` ({light, id} at [s - sample][slot]); the bounce launches
    // see `out` = acc and slots that carry the plane; path_chain_finish_kernel adds the planes to the frame in sample order
    Texel *acc;              // null: one sample per chain, radiance accumulates in `out` itself
    uint32_t chain, acc_slots;
    uint32_t last_bounce;    // 1: paths that hit on this segment end (max_ray_bounces reached)
    // the window bounce launch (vrt_path_window.hip): the primary launch's workgroup b compacts its survivors into records
    // [b * grp_cap, b * grp_cap + grp_counts[b]) of path_out instead of appending to a segment
    uint32_t *grp_counts;
    uint32_t grp_cap;
    uint32_t blk_w, blk_h;   // ... and takes the tiles in blocks of blk_w x blk_h (a bounce workgroup's regions: one block)
    uint32_t n_nodes, n_roots;
    uint32_t width, height;
    uint32_t tiles_x, tiles_total;
    uint32_t tiles_x_magic;  // ceil(2^32 / tiles_x) when tile / tiles_x = umulhi(tile, magic) for every tile of the frame, else 0 (tile_pixel)
    // this context's t_local-th tile is screen tile (t_local / shard_run) * shard_period + shard_first + t_local % shard_run
    uint32_t shard_first, shard_run, shard_period, tiles_local;
    uint32_t hit_seg_cap;    // capacity of one hit-buffer segment, a multiple of 256
    // Longest tiles first (vrt_kernels.hip: tile_order_*): a wave notes how many trips its two march loops took and a
    // later frame launches its tiles in descending order of that, so the launch's last waves are its cheapest
    const uint32_t *tile_order;  // null, or a permutation of [0, tiles_local): the t-th wave of the launch takes tile tile_order[t]
    uint32_t *tile_cost;         // null, or [tiles_local]: what this frame's waves note
    uint32_t tile_major;     // output slots are [t_local][64] (sharded, or VRT_FLAG_TILE_MAJOR) instead of row-major
    uint32_t compact;        // VRT_FLAG_COMPACT: a slot is an 8-byte record {id word | kIdNormYNeg, water_dist} instead of a texel
    uint32_t finite_settings;  // 1: every Settings float is finite (lets hits skip the sky term exactly)
    // vrt_set_presentation: the frame is presented 1:1 — its march kernel also stores the window's rgba8 pixel (vrt_tile.h:
    // store_screen) into the frame set's screen buffer; null: the blit is its own launch (vrt_present*)
    uint32_t *screen;
    uint32_t screen_only;    // VRT_PRESENT_SKIP_TEXELS: ... and no 16-byte texel
    uint32_t present_box[4]; // [x0, x1) x [y0, y1): the pixels the crosshair's mask can reach (x0 >= x1: none)
    vrt_crosshair crosshair;
    vrt_cam_data cam;
    vrt_settings settings;
    vrt_world_data world;
    uint32_t liquid[8];      // bit v set <=> materials[v].is_liquid == 1, v < 256
    // the same set as an id range when it is one ([liquid_lo, liquid_lo + liquid_span], all below 255 — the standard pack's
    // lava 2, water 3 — or empty): the march then asks with a subtract and a compare instead of the mask lookup in LDS
    uint32_t liquid_is_range, liquid_lo, liquid_span;
    // frame-uniform subexpressions of the shader, evaluated once on the host in the same IEEE binary32 operations
    // (the host half of vrt_frames.hip is built with -ffp-contract=off like the kernels):
    const float *ndc_x;      // [width]  ((float)px * 2) / proj_size.x - 1        (create_ray_from_screen :160)
    const float *ndc_y;      // [height] ((float)py * 2) / proj_size.y - 1        (:161)
    float cam_sun_dir[3];    // normalize(sun_pos - world.min - (cam.pos - world.min)): ray_sky's sun_dir for primary rays (:149)
    // A primary ray's origin is the camera — the same for every pixel — and so is what the march asks of it before its first step: the
    // host evaluates both once per frame, in the kernels' own binary32 operations (march_grid<.., CAMERA>)
    float cam_origin[3];     // cam.pos - f32(world.min) (:169)
    float sun_local[3];      // settings.sun_pos - f32(world.min): the sun as the shadow ray's and the sky's direction subtract from it
    float world_max;         // 0.0 + f32(world.size) (:285)
    uint32_t cam_origin_facts;   // kCamNotFinite | kCamOnPlane (the start nudge :188-190 applies) | kCamOutside (:285, asked of the origin as it is)
};

// The path-trace buffers are compacted per segment, not globally: one device-scope counter saturates at ~88
// returning atomics per microsecond (MI355X_MICROARCH.md "dequeue"), which made 32 400 per-wave atomics the
// whole 0.37 ms of the first primary kernel.  Workgroup b appends to segment b % kHitSegments; each counter
// sits on its own 64-byte line so the adds spread over the L2 channels.  (The shadow hit buffer is compacted
// per workgroup in LDS instead: vrt_kernels.hip.)
constexpr uint32_t kHitSegments = 256;
constexpr uint32_t kSegStride = 16;  // u32 words between counters (64 B)

enum Counter : int {
    kCtrHitCount = 0,       // unused by the kernels (the host sums the segment counters)
    kCtrSteps = 1,
    kCtrVisits = 2,
    kCtrPrimarySteps = 3,
    kCtrPrimaryVisits = 4,
    kCtrHits = 5,
    kCtrSecondary = 6,      // path mode, stats frames: bounce segments traced
    kCtrCount = 8
};

// screen tile of the context's t_local-th tile (vrt_config: shard_rank / shard_count / shard_root_weight)
// A lane's pixel of its wave's tile.  The tile is the WAVE's: its row and column are three instructions of the scalar unit (a multiply-high
// by the host's magic number, a multiply, a subtraction) — as a per-lane division by a run-time value they were 27 vector instructions
// of every tile, five of them 32 x 32 multiplies.
__device__ __forceinline__ void tile_pixel(const FrameParams &P, uint32_t tile, uint32_t lane, uint32_t &px, uint32_t &py) {
    const uint32_t t = __builtin_amdgcn_readfirstlane(tile);
    const uint32_t row = P.tiles_x_magic ? __umulhi(t, P.tiles_x_magic) : t / P.tiles_x;
    px = (t - row * P.tiles_x) * 8u + (lane & 7u);
    py = row * 8u + (lane >> 3);
}

__host__ __device__ __forceinline__ uint32_t shard_tile(uint32_t t_local, uint32_t first, uint32_t run, uint32_t period) {
    return run == 1u ? t_local * period + first : (t_local / run) * period + first + t_local % run;
}

// Pieces of staged uploads (vrt_accel.hip: upload_batch_kernel): words of the pinned ring -> words of destination 0 (the node
// pool) or, with bit 31 of n_words set, destination 1 (chunk_roots)
constexpr uint32_t kUploadPieceWords = 4096, kUploadBatchPieces = 128;
struct UploadPiece { uint32_t dst_word, src_word, n_words; };
struct UploadBatch { UploadPiece piece[kUploadBatchPieces]; };

// A finished pixel is stored NON-TEMPORALLY (global_store ... nt): nothing on this device reads a frame's texels while the frame is
// traced, and 33 MB of them per 1080p frame written through the L2 with the default policy push the derived tables' lines out of
// it — the march is bound by instruction issue only while its table loads hit.  Measured on C2, same box, the kernels otherwise
// identical (profiles/r06_nt_stores_ab.txt): lone launch 96.0 -> 93.3 us, 43.4 -> 44.1 Grays/s with two frames in flight, the
// client's frame (30^3 chunks: larger tables) 117.8 -> 110.4 us one at a time.  The same bytes, whatever the policy.
// (The path trace's texels are read again — a path that ends adds its light to its pixel — and are better left to the caches:
// the same policy there costs 3 % of C4, profiles/r06_path_nt_ab.txt.)
__device__ __forceinline__ void store_streaming(Texel *p, Texel v) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<u32x4 *>(p));
}

struct V3 { float x, y, z; };

// WGSL min() with a NaN operand is implementation-defined; choice: a NaN operand is ignored, ties
// return b (same text as oracle/vrt_oracle.c:orc_min — restated, not shared).
__device__ __forceinline__ float vmin(float a, float b) { return (a < b || b != b) ? a : b; }
__device__ __forceinline__ float vclamp(float e, float lo, float hi) {
    float m = (e > lo) ? e : lo;
    return vmin(m, hi);
}
__device__ __forceinline__ float vsign(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : x); }
__device__ __forceinline__ float vsmoothstep(float e0, float e1, float x) {
    float t = vclamp((x - e0) / (e1 - e0), 0.0f, 1.0f);
    return t * t * (3.0f - 2.0f * t);
}
__device__ __forceinline__ float vmix(float a, float b, float t) { return a * (1.0f - t) + b * t; }
__device__ __forceinline__ float vdot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 vnormalize(V3 v) {
    float len = sqrtf(vdot(v, v));
    return V3{v.x / len, v.y / len, v.z / len};
}
// WGSL i32(f32): NaN -> 0 (the only out-of-range case reachable: positions are range-checked).
__device__ __forceinline__ int f2i(float x) { return (x != x) ? 0 : (int)x; }

struct MarchResult {
    bool hit;
    V3 pos, norm;
    float water_dist;
    uint32_t voxel;
    uint32_t iters;
    uint32_t visits;
    uint32_t trips;     // grid march: the march loop's trips this lane was in (the wave's trip count = the maximum over its lanes)
};

}  // namespace vrt
