// vrt_path.hip — wavefront path trace (VRT_MODE_PATH) for gfx950.
//
// Structure after the reference's stale, never-dispatched path tracer
// (clientdesktop/src/graphics/path_tracer.wgsl: rng_next* :56-76, ray_color :149-194, seed :328) on top of the
// live march of ray_tracer.wgsl; the deliberate differences (bounce origin outside the hit voxel, clamped
// log argument, no emission, water neither stops nor tints a segment) are DESIGN.md §Path trace and are the
// same in oracle/vrt_oracle.c:trace_path.  log and cos are spelled out in + - * / so that host and device
// agree to the bit: a one-ulp different bounce direction eventually hits a different voxel.
//
// One launch per bounce: bounce 0 traces the primary rays; every later bounce reads the compacted buffer of
// paths that are still alive (same per-segment ballot compaction as the shadow hit buffer), marches them and
// appends the survivors to the other buffer.  A pixel's path has exactly one owner lane per bounce, so
// radiance accumulates into the pixel's texel with plain read-modify-writes.
#include <cstdlib>

#include "vrt_path_common.h"

namespace vrt {

// One segment of a path: its march, then path_after_march.
template <int MARCH, bool LDS_ROOTS, bool STATS>
__device__ __forceinline__ bool path_segment(const FrameParams &P, const uint32_t *s_roots, const uint32_t *s_liquid,
                                             PathState &st, MarchResult &R, V3 &light, bool &missed) {
    R = march<MARCH, LDS_ROOTS, STATS>(P, s_roots, s_liquid, st.origin, st.dir);
    return path_after_march(P, st, R, light, missed);
}

// Append the wave's surviving paths to this workgroup's segment of the out buffer (one atomic per wave).
__device__ __forceinline__ void append_paths(const FrameParams &P, bool alive, const PathState &st, uint32_t lane) {
    const unsigned long long ballot = __ballot(alive);
    const uint32_t n = (uint32_t)__popcll(ballot);
    if (!n) return;
    const uint32_t seg = blockIdx.x % kHitSegments;
    const int leader = __ffsll((long long)ballot) - 1;
    uint32_t base = 0;
    if ((int)lane == leader) base = atomicAdd(&P.seg_counts[seg * kSegStride], n);
    base = __shfl(base, leader, 64) + seg * P.hit_seg_cap;
    if (alive) {
        const uint32_t i = base + (uint32_t)__popcll(ballot & ((1ull << lane) - 1ull));
        P.path_out[i] = make_uint4(st.slot, __float_as_uint(st.origin.x), __float_as_uint(st.origin.y), __float_as_uint(st.origin.z));
        P.path_out[P.path_cap + i] = make_uint4(__float_as_uint(st.dir.x), __float_as_uint(st.dir.y), __float_as_uint(st.dir.z), st.rng);
        P.path_out[2u * P.path_cap + i] = make_uint4(__float_as_uint(st.thr.x), __float_as_uint(st.thr.y), __float_as_uint(st.thr.z), 0u);
    }
}

// The same into the workgroup's own region (path_primary_kernel<GROUPED>): the cursor is a word of the workgroup's LDS.
__device__ __forceinline__ void append_paths_grouped(const FrameParams &P, bool alive, const PathState &st, uint32_t lane, uint32_t *s_count) {
    const unsigned long long ballot = __ballot(alive);
    const uint32_t n = (uint32_t)__popcll(ballot);
    if (!n) return;
    const int leader = __ffsll((long long)ballot) - 1;
    uint32_t base = 0;
    if ((int)lane == leader) base = atomicAdd(s_count, n);
    base = __shfl(base, leader, 64) + blockIdx.x * P.grp_cap;
    if (alive) {
        const uint32_t i = base + lanes_below(ballot);
        P.path_out[i] = make_uint4(st.slot, __float_as_uint(st.origin.x), __float_as_uint(st.origin.y), __float_as_uint(st.origin.z));
        P.path_out[P.path_cap + i] = make_uint4(__float_as_uint(st.dir.x), __float_as_uint(st.dir.y), __float_as_uint(st.dir.z), st.rng);
        P.path_out[2u * P.path_cap + i] = make_uint4(__float_as_uint(st.thr.x), __float_as_uint(st.thr.y), __float_as_uint(st.thr.z), 0u);
    }
}

// The launch's q-th tile when the tiles are taken in blocks of blk_w x blk_h (bands of blk_h tile rows, the blocks of a band
// from left to right, a block's tiles row by row: a workgroup's four waves are four tiles of a row, the workgroups of a block
// consecutive).  A ragged last band (or last block of a band) is shorter (narrower); the order stays a permutation of the
// frame's tiles.  Sharded frames keep their order (their tiles are interleaved with the other shards' anyway).
__device__ __forceinline__ uint32_t block_order_tile(uint32_t q, const FrameParams &P) {
    if (P.shard_period != 1u || P.tiles_local != P.tiles_total) return q;
    const uint32_t tx = P.tiles_x, ty = P.tiles_total / P.tiles_x, bw = P.blk_w, bh = P.blk_h;
    const uint32_t band = q / (bh * tx), r = q - band * bh * tx;
    const uint32_t h = min(bh, ty - bh * band), full = tx / bw;
    uint32_t x, y;
    if (r < full * bw * h) {
        const uint32_t cb = r / (bw * h), j = r - cb * bw * h;
        x = bw * cb + j % bw;
        y = j / bw;
    } else {
        const uint32_t w = tx - bw * full, rr = r - full * bw * h;
        x = bw * full + rr % w;
        y = rr / w;
    }
    return (bh * band + y) * tx + x;
}

static_assert(kHitSegments == 256, "a launch's first workgroup (256 threads) clears the next launch's 256 segment cursors");

// Bounce 0: primary rays of sample P.sample. Sample 0 initialises the texel {light, id}; later samples add.
// MULTI: the samples of a launch chain (P.acc, P.chain) share the primary march; otherwise one sample, straight into `out`
// GROUPED (the window bounce launch, vrt_path_window.hip): a workgroup's survivors are compacted — by the workgroup, through a
// counter in LDS — into the workgroup's own region of the path buffer (P.grp_cap records; the count into P.grp_counts), and
// the workgroups take the tiles in blocks of 4 x 4, so that four consecutive regions hold the paths of 32 x 32 pixels: a
// bounce workgroup's rays then start within a few voxels of each other.  No cursors, no global atomics.
template <int MARCH, bool LDS_ROOTS, bool STATS, bool MULTI = false, bool GROUPED = false>
__global__ void __launch_bounds__(256) path_primary_kernel(FrameParams P) {
    extern __shared__ uint32_t smem[];
    uint32_t *s_liquid = smem, *s_roots = smem + 24;
    unsigned long long *s_acc = reinterpret_cast<unsigned long long *>(smem + 8);
    if (STATS && threadIdx.x < 8) s_acc[threadIdx.x] = 0ull;
    if (GROUPED && threadIdx.x == 0) smem[8] = 0u;   // (no stats in a grouped launch: the word is free)
    stage_lds(P, s_roots, s_liquid, LDS_ROOTS);

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t t_launch = blockIdx.x * 4u + (threadIdx.x >> 6);
    const bool live = t_launch < P.tiles_local;
    const uint32_t t_local = (GROUPED && live) ? block_order_tile(t_launch, P) : t_launch;
    if (!GROUPED && blockIdx.x == 0 && P.seg_clear) P.seg_clear[threadIdx.x * kSegStride] = 0u;   // kHitSegments == blockDim.x cursors
    if (!STATS && !GROUPED && !live) return;
    MarchResult R;
    R.iters = 0; R.visits = 0; R.hit = false;
    if (live) {
        const uint32_t tile = shard_tile(t_local, P.shard_first, P.shard_run, P.shard_period);
        const uint32_t px = (tile % P.tiles_x) * 8u + (lane & 7u);
        const uint32_t py = (tile / P.tiles_x) * 8u + (lane >> 3);
        const uint32_t pixel_slot = P.tile_major ? t_local * 64u + lane : py * P.width + px;
        V3 origin, dir;
        create_ray(P, (int)px, (int)py, origin, dir);
        // every sample of a pixel starts with the same ray (the samples differ from their first bounce on: the RNG is not
        // asked before a hit), so the primary segment is marched once for all the samples of this launch chain
        R = march<MARCH, LDS_ROOTS, STATS>(P, s_roots, s_liquid, origin, dir);
        uint32_t id0 = R.voxel & VRT_ID_VOXEL_MASK;   // the id word of the primary segment, composed as shade() does
        if (R.hit) id0 |= VRT_ID_HIT;
        if (R.norm.x != 0.0f) id0 |= VRT_ID_NX;
        if (R.norm.y != 0.0f) id0 |= VRT_ID_NY;
        if (R.norm.z != 0.0f) id0 |= VRT_ID_NZ;
        if (R.water_dist != 0.0f) id0 |= VRT_ID_WATER;
        for (uint32_t s_local = 0; s_local < (MULTI ? P.chain : 1u); s_local++) {
            const uint32_t sample = P.sample + s_local;
            PathState st;
            st.slot = pixel_slot;
            st.origin = origin;
            st.dir = dir;
            st.thr = V3{1.0f, 1.0f, 1.0f};
            // seed: path_tracer.wgsl:328 (y*W + x) + the per-sample stride and frame seed of SURVEY §8d
            st.rng = py * P.width + px + sample * (P.width * P.height) + P.seed * 0x9E3779B9u;
            V3 light{0.f, 0.f, 0.f};
            bool missed;
            const bool alive = path_after_march(P, st, R, light, missed) && !P.last_bounce;
            if (MULTI) {
                // this sample's own plane: its light so far and, for the frame's first sample, the id word (0 otherwise);
                // the path's later segments find the plane through the slot
                st.slot += s_local * P.acc_slots;
                P.acc[st.slot] = make_uint4(__float_as_uint(light.x), __float_as_uint(light.y), __float_as_uint(light.z), sample == 0u ? id0 : 0u);
            } else if (P.sample == 0u) {
                P.out[st.slot] = make_uint4(__float_as_uint(light.x), __float_as_uint(light.y), __float_as_uint(light.z), id0);
            } else if (missed) {
                uint4 t = P.out[st.slot];
                t.x = __float_as_uint(__uint_as_float(t.x) + light.x);
                t.y = __float_as_uint(__uint_as_float(t.y) + light.y);
                t.z = __float_as_uint(__uint_as_float(t.z) + light.z);
                P.out[st.slot] = t;
            }
            if (GROUPED) append_paths_grouped(P, alive, st, lane, &smem[8]);
            else append_paths(P, alive, st, lane);
        }
        if (STATS && P.steps && P.sample == 0u) P.steps[pixel_slot] = R.iters;
    }
    if (GROUPED) {
        __syncthreads();
        if (threadIdx.x == 0) P.grp_counts[blockIdx.x] = smem[8];
    }
    if (STATS) {
        block_add(s_acc, 0, R.iters);
        block_add(s_acc, 1, R.visits);
        block_add(s_acc, 2, (R.hit && P.sample == 0u) ? 1ull : 0ull);
        __syncthreads();
        if (threadIdx.x == 0) {
            atomicAdd(&P.counters[kCtrSteps], s_acc[0]);
            atomicAdd(&P.counters[kCtrVisits], s_acc[1]);
            atomicAdd(&P.counters[kCtrPrimarySteps], s_acc[0]);
            atomicAdd(&P.counters[kCtrPrimaryVisits], s_acc[1]);
            atomicAdd(&P.counters[kCtrHits], s_acc[2]);
        }
    }
}

// Bounce b >= 1: lane = one live path of the in buffer.
template <int MARCH, bool LDS_ROOTS, bool STATS>
__global__ void __launch_bounds__(256) path_bounce_kernel(FrameParams P) {
    extern __shared__ uint32_t smem[];
    uint32_t *s_liquid = smem, *s_roots = smem + 24;
    unsigned long long *s_acc = reinterpret_cast<unsigned long long *>(smem + 8);
    if (STATS && threadIdx.x < 8) s_acc[threadIdx.x] = 0ull;
    stage_lds(P, s_roots, s_liquid, LDS_ROOTS);

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t seg = blockIdx.x % kHitSegments, part = blockIdx.x / kHitSegments;
    if (blockIdx.x == 0 && P.seg_clear) P.seg_clear[threadIdx.x * kSegStride] = 0u;
    const uint32_t count = P.seg_in[seg * kSegStride];
    const uint32_t j = part * blockDim.x + threadIdx.x;
    const bool active = j < count;
    if (!STATS && part * blockDim.x >= count) return;
    MarchResult R;
    R.iters = 0; R.visits = 0; R.hit = false;
    bool alive = false;
    PathState st;
    st.slot = 0; st.rng = 0;
    st.origin = st.dir = st.thr = V3{0.f, 0.f, 0.f};
    if (active) {
        const uint32_t i = seg * P.hit_seg_cap + j;
        const uint4 a = P.path_in[i], b = P.path_in[P.path_cap + i], c = P.path_in[2u * P.path_cap + i];
        st.slot = a.x;
        st.origin = V3{__uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w)};
        st.dir = V3{__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z)};
        st.rng = b.w;
        st.thr = V3{__uint_as_float(c.x), __uint_as_float(c.y), __uint_as_float(c.z)};
        V3 light{0.f, 0.f, 0.f};
        bool missed;
        alive = path_segment<MARCH, LDS_ROOTS, STATS>(P, s_roots, s_liquid, st, R, light, missed) && !P.last_bounce;
        if (missed) {
            uint4 t = P.out[st.slot];
            t.x = __float_as_uint(__uint_as_float(t.x) + light.x);
            t.y = __float_as_uint(__uint_as_float(t.y) + light.y);
            t.z = __float_as_uint(__uint_as_float(t.z) + light.z);
            P.out[st.slot] = t;
        }
        if (STATS && P.steps && P.sample == 0u) P.steps[st.slot] += R.iters << 16;
    }
    append_paths(P, alive, st, lane);
    if (STATS) {
        block_add(s_acc, 0, active ? R.iters : 0u);
        block_add(s_acc, 1, active ? R.visits : 0u);
        block_add(s_acc, 2, active ? 1ull : 0ull);
        __syncthreads();
        if (threadIdx.x == 0 && s_acc[2]) {
            atomicAdd(&P.counters[kCtrSteps], s_acc[0]);
            atomicAdd(&P.counters[kCtrVisits], s_acc[1]);
            atomicAdd(&P.counters[kCtrSecondary], s_acc[2]);
        }
    }
}

// A wave's pool of rays (the bounce launches below): K batches of 64 paths, 4 words each in the wave's own LDS — unit[3]
// (phase A -> the hand-out), then pos[3] + the packed end state (the march -> phase C): 4 KiB per wave
#ifndef VRT_POOL_K
#define VRT_POOL_K 4
#endif
constexpr uint32_t kPoolBatches = VRT_POOL_K;            // K
constexpr uint32_t kPoolEntries = kPoolBatches * 64u;
constexpr uint32_t kPoolWords = 4u * kPoolEntries;
constexpr uint32_t kPoolRefillAt = 16u;                  // idle lanes (of 64) that send the wave back to the pool

#ifdef VRT_EXPERIMENTS
// ------------------------------------------------------------------------------------------------
// The path trace as ONE launch (VRT_PATH_PERSISTENT=1; built and measured, not the default): persistent waves, lanes
// refilled in batches.
//
// The wavefront kernels above run one launch per bounce; a bounce launch marches rays whose directions were just
// randomised, so a wave's lanes finish after very different numbers of steps and the wave runs as long as its slowest
// lane (measured: 36-39 % lane utilisation).  Here a lane owns a *pixel* — all its samples, all their segments, in the
// order the oracle traces them, the radiance summed in a register — and a wave keeps marching with the lanes it has:
//   * the march loop is resumable (its state lives in the lane's registers across the phases below);
//   * when kRefillAt lanes have finished their segment, the wave leaves the march loop once, and those lanes do what
//     comes next together: shade, draw the bounce direction, or end the sample / the pixel and take the next pixel of
//     the wave's tile queue — then all lanes re-enter the march loop;
//   * tiles come from eight per-XCD ticket counters (one atomic per 64 pixels, any wave steals from any queue).
// Every path executes exactly the instructions the wavefront kernels execute for it, so the frame is bit-identical to
// theirs (tests) — scheduling is the only difference.  No path buffers in HBM, one launch per frame whatever spp is.
// Measured on C4 (1080p, 4 bounces): 10.8 Grays/s at the best batch size (40 waiting lanes; 6.2 at 8, 8.1 at 64) against
// 13.0 for the launch-per-bounce kernels: what a lane does between two segments — shading, six RNG draws with three
// logarithms and cosines, two normalisations, the nine divides and four square roots of a ray's set-up, ~800 VALU
// instructions — costs as much as marching the segment, and here it is issued for a batch of 16-40 lanes where the
// launch-per-bounce kernels issue it for 64.  What the batches win in the march they lose between the segments.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kRefillAt = 40;   // lanes that must be waiting before the wave leaves the march loop for them (measured optimum)

struct Segment {      // one ray being marched (march_grid's loop state, vrt_march.h)
    V3 pos, dir;
    float ux, uy, uz;
    uint32_t mxm, mym, mzm;
    int vx, vy, vz;
    float step, adx, ady, adz, dew, total_len, water_dist;
    uint32_t slow_bit, voxel, iter;
    bool careful;
};

// march_grid's prologue.  false: the ray starts outside the world (a miss before any lookup).
__device__ __forceinline__ bool segment_begin(const FrameParams &P, V3 origin, V3 dir, Segment &m) {
    m.dir = dir;
    m.careful = !(finite3(origin) && finite3(dir));
    V3 pos = origin;
    if (pos.x - floorf(pos.x) < 0.001f || pos.y - floorf(pos.y) < 0.001f || pos.z - floorf(pos.z) < 0.001f) {
        pos.x += 0.001f * dir.x;
        pos.y += 0.001f * dir.y;
        pos.z += 0.001f * dir.z;
    }
    m.pos = pos;
    m.water_dist = 0.0f;
    m.voxel = 0u;
    m.step = -1.0f;
    m.adx = m.ady = m.adz = 0.0f;
    m.dew = -1.0f;
    m.total_len = 0.0f;
    m.iter = 0u;
    m.slow_bit = m.careful ? 0x80000000u : 0u;
    const float world_max = 0.0f + (float)P.world.size;
    if ((pos.x <= 0.0f || pos.y <= 0.0f || pos.z <= 0.0f) || (pos.x >= world_max || pos.y >= world_max || pos.z >= world_max)) return false;
    const V3 unit = unit_steps(dir);
    m.ux = fabsf(unit.x); m.uy = fabsf(unit.y); m.uz = fabsf(unit.z);
    m.mxm = dir.x >= 0.0f ? ~0u : 0u; m.mym = dir.y >= 0.0f ? ~0u : 0u; m.mzm = dir.z >= 0.0f ? ~0u : 0u;
    m.vx = trunc2i(pos.x); m.vy = trunc2i(pos.y); m.vz = trunc2i(pos.z);
    return true;
}

// One trip of march_grid's loop.  true: the segment is over (solid hit, left the world, or kMaxSteps lookups).
__device__ __forceinline__ bool segment_trip(const FrameParams &P, const uint32_t *s_liquid, TableBuf gb, TableBuf bb, uint32_t row_bytes,
                                             uint32_t slab_bytes, Segment &m) {
    m.iter += 1u;
    uint32_t e = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(
        gb, mad_i24(m.vz >> 2, slab_bytes, mad_i24(m.vy >> 2, row_bytes, (uint32_t)m.vx & ~3u)), 0, 0);
    uint32_t lo = e;
    if ((e | m.slow_bit) - 1u >= 31u) {
        if (m.careful) {
            m.vx = trunc2i(m.pos.x);
            m.vy = trunc2i(m.pos.y);
            m.vz = trunc2i(m.pos.z);
            e = 0u;
            if (!(min3_nan_ignoring(m.pos.x, m.pos.y, m.pos.z) < 0.0f ||
                  max(max((uint32_t)m.vx, (uint32_t)m.vy), (uint32_t)m.vz) >= P.world.size))
                e = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(
                    gb, mad_i24(m.vz >> 2, slab_bytes, mad_i24(m.vy >> 2, row_bytes, (uint32_t)m.vx & ~3u)), 0, 0);
            lo = e;
        }
        if (e == 0u) return true;   // outside the world
        m.voxel = 0u;
        if ((int)e < 0) {
            const uint32_t u = ((uint32_t)m.vx & 3u) | (((uint32_t)m.vy & 3u) << 2) | (((uint32_t)m.vz & 3u) << 4);
            const uint32_t b = (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(bb, (e + u) << 1, 0, 0);
            lo = b & 1u;
            m.voxel = b >> 1;
        } else if (e > 31u) {
            lo = e & 31u;
            m.voxel = e >> 16;
        }
        if (m.voxel != 0u) {
            if (!is_liquid_ranged(P, s_liquid, m.voxel)) return true;   // solid: the hit
            if (m.dew == -1.0f) { m.dew = m.total_len; m.slow_bit = 0x80000000u; }
        } else if (m.dew != -1.0f) {
            m.water_dist += m.total_len - m.dew;
            m.dew = -1.0f;
            if (!m.careful) m.slow_bit = 0u;
        }
    }
    const float tx = (float)(int)(bfi(lo, m.mxm, (uint32_t)m.vx) - m.mxm) - m.pos.x;
    const float ty = (float)(int)(bfi(lo, m.mym, (uint32_t)m.vy) - m.mym) - m.pos.y;
    const float tz = (float)(int)(bfi(lo, m.mzm, (uint32_t)m.vz) - m.mzm) - m.pos.z;
    m.adx = abs_mul(tx, m.ux);
    m.ady = abs_mul(ty, m.uy);
    m.adz = abs_mul(tz, m.uz);
    m.step = min3_f32(m.adx, m.ady, m.adz);   // (p) of vrt_march.h: the float minimum when no lane's is zero or NaN
    if (__ballot(!(m.step > 0.0f)) != 0ull)
        m.step = __uint_as_float(min3_u32(__float_as_uint(m.adx) - 1u, __float_as_uint(m.ady) - 1u, __float_as_uint(m.adz) - 1u) + 1u);
    m.total_len += m.step;
    const float sp = m.step + 0.001f;
    m.pos.x += m.dir.x * (m.step == m.adx ? sp : m.step);
    m.pos.y += m.dir.y * (m.step == m.ady ? sp : m.step);
    m.pos.z += m.dir.z * (m.step == m.adz ? sp : m.step);
    m.vx = flr2i(m.pos.x);
    m.vy = flr2i(m.pos.y);
    m.vz = flr2i(m.pos.z);
    return m.iter >= kMaxSteps;
}

// march_grid's epilogue: the MarchResult of a finished segment (`started` false: segment_begin said miss).
__device__ __forceinline__ MarchResult segment_end(const FrameParams &P, const Segment &m, bool started) {
    MarchResult R;
    R.hit = false;
    R.pos = V3{0.f, 0.f, 0.f};
    R.norm = V3{0.f, 0.f, 0.f};
    R.water_dist = 0.0f;
    R.voxel = 0u;
    R.iters = 0u;
    R.visits = 0u;
    if (!started) return R;
    R.water_dist = m.water_dist;
    if (m.dew != -1.0f) R.water_dist += m.total_len - m.dew;
    if (min3_nan_ignoring(m.pos.x, m.pos.y, m.pos.z) < 0.0f ||
        max(max((uint32_t)trunc2i(m.pos.x), (uint32_t)trunc2i(m.pos.y)), (uint32_t)trunc2i(m.pos.z)) >= P.world.size)
        return R;
    R.hit = true;
    R.pos = m.pos;
    if (m.step != -1.0f)
        R.norm = V3{(m.step == m.adx ? 1.0f : 0.0f) * -vsign(m.dir.x), (m.step == m.ady ? 1.0f : 0.0f) * -vsign(m.dir.y),
                    (m.step == m.adz ? 1.0f : 0.0f) * -vsign(m.dir.z)};
    R.voxel = m.voxel;
    return R;
}

// ------------------------------------------------------------------------------------------------
// Bounce b >= 1 with a wave-local ray pool (the default for plain frames over the derived tables).
//
// A bounce launch marches rays whose directions were just drawn at random: most end within a few steps on the terrain
// next to their origin, a few graze it for a hundred.  With lane = path for the whole kernel a wave runs as long as its
// longest ray: 65 wave-steps for a mean of 12 per ray — 19 % lane utilisation inside the march loop, which is three
// quarters of the kernel's instructions (profiles/r02_path_pmc_summary.txt).  Here a wave owns K x 64 paths and works
// in three phases, the two arithmetic-heavy ones at full width:
//   A  K batches, lane = path: load the record, do the march's prologue (the nudge off a voxel face, the nine divides
//      and three square roots of the unit steps), park {pos, dir, unit} in the wave's own LDS pool;
//   B  march: a lane takes the next ray of the pool when it has none; when `refill_at` lanes have finished theirs (each
//      parks its end state in the pool entry it came from) the wave leaves the loop once and those lanes take the
//      next ones — a dozen LDS reads, not the ~800 instructions of shading and set-up that made the persistent kernel
//      below lose what it won;
//   C  K batches, lane = path again: the record once more, the march's end state from the pool, then exactly what
//      path_bounce_kernel does after its march (shade, draw the bounce, accumulate a miss, append the survivor).
// Every ray executes the instructions the other kernels execute for it: bit-identical frames (tests).  The pool is
// wave-local: no barriers between the phases, a wave that has nothing left leaves.
// ------------------------------------------------------------------------------------------------
#ifndef VRT_POOL_RAYS
#define VRT_POOL_RAYS 1
#endif
// rays a lane marches at once.  2 and 3 were measured (their loads in flight together, their arithmetic interleaved): 117 and
// more registers instead of 60, half the waves per SIMD, 12.4 and 10.4 Grays/s on C4 against 14.5 — profiles/r02_path_pool_sweeps.txt
constexpr uint32_t kPoolRays = VRT_POOL_RAYS;
constexpr uint32_t kPoolEjectAt = 16;
constexpr uint32_t kPoolEjected = 0xFFFFFFFFu;   // a pool entry's packed end state: the ray went on to the continuation launch


#ifdef VRT_EXP_POOLDBG
__device__ unsigned long long g_pool_dbg[16384 * 8];   // experiment: per wave {n, A, B, C cycles, wave-steps, refills, start, end (100 MHz)}
#define POOLDBG_T(x) const unsigned long long x = __builtin_amdgcn_s_memtime()
#else
#define POOLDBG_T(x)
#endif

template <bool CONT>
__global__ void __launch_bounds__(256) path_bounce_pool_kernel(FrameParams P, uint32_t refill_at, uint32_t eject_at) {
    extern __shared__ uint32_t smem[];
    uint32_t *s_liquid = smem;
    if (threadIdx.x < 8) s_liquid[threadIdx.x] = P.liquid[threadIdx.x];
    if (blockIdx.x == 0 && P.seg_clear) P.seg_clear[threadIdx.x * kSegStride] = 0u;   // kHitSegments == blockDim.x cursors
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    float *pool = reinterpret_cast<float *>(smem + 8) + wave * kPoolWords;
    constexpr uint32_t E = kPoolEntries;

    // this wave's paths: the workgroup takes up to 4 E records of its segment, split evenly over its waves
    const uint32_t seg = blockIdx.x % kHitSegments, part = blockIdx.x / kHitSegments;
    const uint32_t count = P.seg_in[seg * kSegStride];
    const uint32_t wg_begin = part * 4u * E;
    if (wg_begin >= count) return;
    const uint32_t n_wg = min(4u * E, count - wg_begin), per = (n_wg + 3u) / 4u;
    if (wave * per >= n_wg) return;
    const uint32_t n = min(per, n_wg - wave * per);   // <= E
    const uint32_t base = seg * P.in_seg_cap + wg_begin + wave * per;
    const float world_max = 0.0f + (float)P.world.size;
#ifdef VRT_EXP_POOLDBG
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    uint32_t dbg_steps = 0, dbg_refills = 0, dbg_bricks = 0, dbg_dry_steps = 0, dbg_dry_lanes = 0, dbg_wet_lanes = 0;
    unsigned long long dbg_lat_grid = 0, dbg_lat_brick = 0;
#endif
    POOLDBG_T(t0);

    // ---- A: the unit steps of every ray (nine divides, three square roots), full width ----
    for (uint32_t k = 0; k * 64u < n; k++) {
        const uint32_t i = k * 64u + lane;
        if (i < n) {
            const uint4 b = P.path_in[P.in_cap + base + i];
            const V3 dir{__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z)};
            const V3 unit = unit_steps(dir);
            pool[0u * E + i] = unit.x; pool[1u * E + i] = unit.y; pool[2u * E + i] = unit.z;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    POOLDBG_T(t1);
    // ---- B: the marches, lanes refilled from the pool.  march_grid's loop (vrt_march.h) made resumable: a ray that has
    // stopped keeps its end state in its registers until the wave's next refill parks it; water is not tracked (no
    // output of a path segment depends on it), so a liquid voxel is simply not a hit.
    // (Written for kPoolRays rays per lane — independent rays, their loads in flight together; one is what is built.) ----
    {
        const TableBuf gb = table_buffer(P.grid, P.grid_bytes), bb = table_buffer(P.bricks, P.brick_bytes);
        const uint32_t row_bytes = (P.grid_dim + 1u) * 4u, slab_bytes = (P.grid_dim + 1u) * row_bytes;
        const uint32_t wsize = P.world.size;
        struct Ray {
            V3 pos, dir;
            float ux, uy, uz, step, adx, ady, adz;
            uint32_t mxm, mym, mzm, voxel, iter, idx;
            int vx, vy, vz;
            bool marching, parked, not_finite;
        };
        Ray ray[kPoolRays];
#pragma unroll
        for (uint32_t r = 0; r < kPoolRays; r++) {
            Ray &q = ray[r];
            q.pos = q.dir = V3{0.f, 0.f, 0.f};
            q.ux = q.uy = q.uz = 0.f; q.step = -1.f; q.adx = q.ady = q.adz = 0.f;
            q.mxm = q.mym = q.mzm = 0u; q.voxel = 0u; q.iter = 0u; q.idx = 0u;
            q.vx = q.vy = q.vz = 0;
            q.marching = false; q.parked = true; q.not_finite = false;
        }
        uint32_t next = 0u;   // wave-uniform: the pool's first ray not handed out yet
        const unsigned long long below = (1ull << lane) - 1ull;
        auto cell_offset = [&](const Ray &q) __attribute__((always_inline)) {
            return mad_i24(q.vz >> 2, slab_bytes, mad_i24(q.vy >> 2, row_bytes, (uint32_t)q.vx & ~3u));
        };
        auto park = [&](Ray &q) __attribute__((always_inline)) {   // the end state segment_end needs: where, through which faces, on what
            uint32_t packed = q.voxel << 8;
            if (q.step != -1.0f) packed |= (q.step == q.adx ? 1u : 0u) | (q.step == q.ady ? 2u : 0u) | (q.step == q.adz ? 4u : 0u);
            pool[0u * E + q.idx] = q.pos.x; pool[1u * E + q.idx] = q.pos.y; pool[2u * E + q.idx] = q.pos.z;
            pool[3u * E + q.idx] = __uint_as_float(packed);
            q.parked = true;
        };
        auto take = [&](Ray &q, uint32_t idx) __attribute__((always_inline)) {
            // the rest of segment_begin: consecutive records for the lanes that refill, so the loads coalesce
            q.idx = idx;
            const uint32_t rec = base + idx;
            const uint4 a = P.path_in[rec], b = P.path_in[P.in_cap + rec];
            const V3 origin{__uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w)};
            q.dir = V3{__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z)};
            q.not_finite = !(finite3(origin) && finite3(q.dir));
            q.ux = pool[0u * E + idx]; q.uy = pool[1u * E + idx]; q.uz = pool[2u * E + idx];
            q.mxm = q.dir.x >= 0.0f ? ~0u : 0u; q.mym = q.dir.y >= 0.0f ? ~0u : 0u; q.mzm = q.dir.z >= 0.0f ? ~0u : 0u;
            q.voxel = 0u;
            q.marching = true;
            q.parked = false;
            uint4 d = make_uint4(0u, 0u, 0u, kContFresh);
            if (CONT) d = P.path_in[3u * P.in_cap + rec];
            if (CONT && !(d.w & kContFresh)) {
                // a ray the bounce launch handed on: where it stood, how many lookups it has had, and through which
                // faces its last step left (as a step / distances triple that compares the same way)
                q.pos = V3{__uint_as_float(d.x), __uint_as_float(d.y), __uint_as_float(d.z)};
                q.iter = d.w & 0xFFFFu;
                q.step = 1.0f;
                q.adx = (d.w & 0x10000u) ? 1.0f : 2.0f; q.ady = (d.w & 0x20000u) ? 1.0f : 2.0f; q.adz = (d.w & 0x40000u) ? 1.0f : 2.0f;
                q.vx = flr2i(q.pos.x); q.vy = flr2i(q.pos.y); q.vz = flr2i(q.pos.z);   // as take_step left them
            } else {
                q.pos = nudged(origin, q.dir);
                q.step = -1.0f; q.adx = q.ady = q.adz = 0.0f;
                q.iter = 0u;
                if ((q.pos.x <= 0.0f || q.pos.y <= 0.0f || q.pos.z <= 0.0f) || (q.pos.x >= world_max || q.pos.y >= world_max || q.pos.z >= world_max)) {
                    // starts outside the world: a miss before any lookup.  Its end state says so (a position outside)
                    q.marching = false;
                    q.pos = V3{-1.0f, -1.0f, -1.0f};
                }
                q.vx = trunc2i(q.pos.x); q.vy = trunc2i(q.pos.y); q.vz = trunc2i(q.pos.z);
            }
        };
        // the step to the leaf's exit face for a leaf of size lo + 1 (take_step of march_grid), then the lookup limit
        auto step_or_stop = [&](Ray &q, uint32_t lo, bool stop) __attribute__((always_inline)) {
#ifdef VRT_EXP_POOL_VALU   // tools/ab experiments only: the marginal cost of extra instructions per step
#pragma unroll
            for (int k_ = 0; k_ < VRT_EXP_POOL_VALU; k_++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(q.idx) : "v"(0u));
#endif
#ifdef VRT_EXP_POOL_SALU
#pragma unroll
            for (int k_ = 0; k_ < VRT_EXP_POOL_SALU; k_++) asm volatile("s_mov_b32 vcc_lo, 0" ::: "vcc");
#endif
            if (!stop) {
                const float tx = (float)(int)(bfi(lo, q.mxm, (uint32_t)q.vx) - q.mxm) - q.pos.x;
                const float ty = (float)(int)(bfi(lo, q.mym, (uint32_t)q.vy) - q.mym) - q.pos.y;
                const float tz = (float)(int)(bfi(lo, q.mzm, (uint32_t)q.vz) - q.mzm) - q.pos.z;
                q.adx = abs_mul(tx, q.ux);
                q.ady = abs_mul(ty, q.uy);
                q.adz = abs_mul(tz, q.uz);
                q.step = min3_f32(q.adx, q.ady, q.adz);   // (p) of vrt_march.h
                if (__ballot(!(q.step > 0.0f)) != 0ull)
                    q.step = __uint_as_float(min3_u32(__float_as_uint(q.adx) - 1u, __float_as_uint(q.ady) - 1u, __float_as_uint(q.adz) - 1u) + 1u);
                const float sp = q.step + 0.001f;
                q.pos.x += q.dir.x * (q.step == q.adx ? sp : q.step);
                q.pos.y += q.dir.y * (q.step == q.ady ? sp : q.step);
                q.pos.z += q.dir.z * (q.step == q.adz ? sp : q.step);
                q.vx = flr2i(q.pos.x);
                q.vy = flr2i(q.pos.y);
                q.vz = flr2i(q.pos.z);
                stop = q.iter >= kMaxSteps;
            }
            q.marching = !stop;
        };
        // (l) of vrt_march.h: the general step as march_grid has it for a wave with a ray that is not finite — the shader's
        // own bounds test, its lookup at i32(f32) coordinates
        auto careful_step = [&](Ray &q) __attribute__((always_inline)) {
            q.iter += 1u;
            q.vx = trunc2i(q.pos.x);
            q.vy = trunc2i(q.pos.y);
            q.vz = trunc2i(q.pos.z);
            uint32_t e = 0u;
            if (!(min3_nan_ignoring(q.pos.x, q.pos.y, q.pos.z) < 0.0f || max(max((uint32_t)q.vx, (uint32_t)q.vy), (uint32_t)q.vz) >= wsize))
                e = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(gb, cell_offset(q), 0, 0);
            uint32_t lo = e;
            bool stop = e == 0u;   // border, or past either end of the grid: the position is outside the world
            if (!stop) {
                q.voxel = 0u;
                if ((int)e < 0) {
                    const uint32_t u = ((uint32_t)q.vx & 3u) | (((uint32_t)q.vy & 3u) << 2) | (((uint32_t)q.vz & 3u) << 4);
                    const uint32_t b = (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(bb, (e + u) << 1, 0, 0);
                    lo = b & 1u;
                    q.voxel = b >> 1;
                } else if (e > 31u) {
                    lo = e & 31u;
                    q.voxel = e >> 16;
                }
                stop = q.voxel != 0u && !is_liquid_ranged(P, s_liquid, q.voxel);   // solid: the hit
            }
            step_or_stop(q, lo, stop);
        };
        for (;;) {
            // ---- refill: park what has stopped, hand out the pool's next rays (ray 0 of every lane first) ----
            uint32_t handed = 0u;
#pragma unroll
            for (uint32_t r = 0; r < kPoolRays; r++) {
                Ray &q = ray[r];
                if (!q.marching && !q.parked) park(q);
                const unsigned long long idle = __ballot(!q.marching);
                const uint32_t at = next + handed + (uint32_t)__popcll(idle & below);
                if (!q.marching && at < n) take(q, at);
                handed += (uint32_t)__popcll(idle);
            }
            next = min(n, next + handed);
#ifdef VRT_EXP_POOLDBG
            dbg_refills++;
#endif
            bool any = false, nf = false;
#pragma unroll
            for (uint32_t r = 0; r < kPoolRays; r++) { any |= ray[r].marching; nf |= ray[r].marching && ray[r].not_finite; }
            if (__ballot(any) == 0ull) {
                if (next >= n) break;   // the pool is empty and nobody marches
                continue;               // (every ray handed out started outside the world)
            }
            const bool careful = __ballot(nf) != 0ull;   // per refill round
            for (;;) {
                if (careful) {   // wave-uniform, rare
#pragma unroll
                    for (uint32_t r = 0; r < kPoolRays; r++)
                        if (ray[r].marching) careful_step(ray[r]);
                } else if (kPoolRays == 1u) {
                    // one ray per lane (what is built): everything inside one region of marching lanes — the scalar unit's
                    // time shows in this kernel (61 % of it, against 42 % of the VALU's: profiles/r02_path_pool_sweeps.txt),
                    // and every region of lanes is four or five scalar instructions
                    Ray &q = ray[0];
                    if (q.marching) {
                        const uint32_t e = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(gb, cell_offset(q), 0, 0);
#ifdef VRT_EXP_POOL_LOAD   // tools/ab experiments only: one more load per step, 1 = the line just read, 2 = a line nobody shares
                        {
                            const uint32_t off_ = VRT_EXP_POOL_LOAD == 1 ? cell_offset(q) : ((q.iter * 0x9E3779B9u + q.idx * 0x85EBCA6Bu + lane * 0xC2B2AE35u) % (P.grid_bytes / 4u)) * 4u;
                            const uint32_t x_ = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(gb, off_, 0, 0);
                            asm volatile("" :: "v"(x_));
                        }
#endif
                        q.iter += 1u;
                        const bool brick = (int)e < 0;
                        uint32_t b = 0u;
                        if (brick) {
                            const uint32_t u = ((uint32_t)q.vx & 3u) | (((uint32_t)q.vy & 3u) << 2) | (((uint32_t)q.vz & 3u) << 4);
                            b = (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(bb, (e + u) << 1, 0, 0);  // the shift drops bit 31
                        }
                        const uint32_t lo = brick ? (b & 1u) : (e & 31u);
                        q.voxel = brick ? (b >> 1) : (e >> 16);
                        const bool liquid = P.liquid_is_range ? (q.voxel - P.liquid_lo <= P.liquid_span) : is_liquid(s_liquid, q.voxel);
                        step_or_stop(q, lo, (e == 0u) | ((q.voxel != 0u) & !liquid));
                    }
                } else {
                    // the same decisions without a branch per case — a bounce wave has a ray in every case on nearly every
                    // step, and each divergent branch is half a dozen scalar instructions of exec-mask bookkeeping: an air
                    // leaf is the e <= 31 instance of "leaf" (lo = e & 31, voxel = e >> 16 = 0), the border (e = 0) is a
                    // leaf of nothing.  First all the grid loads, then all the brick loads, then the arithmetic
                    uint32_t e[kPoolRays], b[kPoolRays];
                    bool brick[kPoolRays], any_brick = false;
#pragma unroll
                    for (uint32_t r = 0; r < kPoolRays; r++) {
                        e[r] = 0u;
                        if (ray[r].marching) e[r] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(gb, cell_offset(ray[r]), 0, 0);
                    }
#pragma unroll
                    for (uint32_t r = 0; r < kPoolRays; r++) {
                        brick[r] = ray[r].marching && (int)e[r] < 0;
                        any_brick |= brick[r];
                        b[r] = 0u;
                    }
                    if (__ballot(any_brick) != 0ull) {   // wave-uniform: the second, dependent load only if some ray needs it
#pragma unroll
                        for (uint32_t r = 0; r < kPoolRays; r++) {
                            const Ray &q = ray[r];
                            const uint32_t u = ((uint32_t)q.vx & 3u) | (((uint32_t)q.vy & 3u) << 2) | (((uint32_t)q.vz & 3u) << 4);
                            if (brick[r]) b[r] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(bb, (e[r] + u) << 1, 0, 0);  // the shift drops bit 31
                        }
                    }
#pragma unroll
                    for (uint32_t r = 0; r < kPoolRays; r++) {
                        Ray &q = ray[r];
                        if (q.marching) {
                            q.iter += 1u;
                            const uint32_t lo = brick[r] ? (b[r] & 1u) : (e[r] & 31u);
                            q.voxel = brick[r] ? (b[r] >> 1) : (e[r] >> 16);
                            const bool liquid = P.liquid_is_range ? (q.voxel - P.liquid_lo <= P.liquid_span) : is_liquid(s_liquid, q.voxel);
                            step_or_stop(q, lo, (e[r] == 0u) | ((q.voxel != 0u) & !liquid));
                        }
                    }
                }
                uint32_t n_march = 0u;
#pragma unroll
                for (uint32_t r = 0; r < kPoolRays; r++) n_march += (uint32_t)__popcll(__ballot(ray[r].marching));
#ifdef VRT_EXP_POOLDBG
                dbg_steps++;
                if (next >= n) { dbg_dry_steps++; dbg_dry_lanes += n_march; } else dbg_wet_lanes += n_march;
#endif
                if (n_march == 0u || (next < n && 64u * kPoolRays - n_march >= refill_at)) break;
                if (!CONT && next >= n && n_march <= eject_at) break;   // the pool is dry and few rays are left: hand them on
            }
            if (!CONT && next >= n && P.cont_out) {
                // ---- the stragglers go to the straggler chain: their path record and where they stand ----
                uint32_t n_left = 0u;
#pragma unroll
                for (uint32_t r = 0; r < kPoolRays; r++) n_left += (uint32_t)__popcll(__ballot(ray[r].marching));
                if (n_left != 0u && n_left <= eject_at) {
                    uint32_t at0 = 0;
                    if (lane == 0) at0 = atomicAdd(&P.cont_counts[seg * kSegStride], n_left);
                    at0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)at0);
#pragma unroll
                    for (uint32_t r = 0; r < kPoolRays; r++) {
                        Ray &q = ray[r];
                        const unsigned long long ballot = __ballot(q.marching);
                        if (q.marching) {   // (a segment of the straggler records is as large as a segment of the paths)
                            const uint32_t rec = base + q.idx, o = seg * P.hit_seg_cap + at0 + (uint32_t)__popcll(ballot & below);
                            uint32_t w = q.iter;
                            if (q.step != -1.0f) w |= (q.step == q.adx ? 0x10000u : 0u) | (q.step == q.ady ? 0x20000u : 0u) | (q.step == q.adz ? 0x40000u : 0u);
                            P.cont_out[o] = P.path_in[rec];
                            P.cont_out[P.path_cap + o] = P.path_in[P.in_cap + rec];
                            P.cont_out[2u * P.path_cap + o] = P.path_in[2u * P.in_cap + rec];
                            P.cont_out[3u * P.path_cap + o] = make_uint4(__float_as_uint(q.pos.x), __float_as_uint(q.pos.y), __float_as_uint(q.pos.z), w);
                            pool[3u * E + q.idx] = __uint_as_float(kPoolEjected);
                            q.marching = false;
                            q.parked = true;
                        }
                        at0 += (uint32_t)__popcll(ballot);
                    }
                }
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    POOLDBG_T(t2);

    // ---- C: what follows the march, full width ----
    for (uint32_t k = 0; k * 64u < n; k++) {
        const uint32_t i = k * 64u + lane;
        bool alive = false;
        PathState st;
        st.slot = 0; st.rng = 0;
        st.origin = st.dir = st.thr = V3{0.f, 0.f, 0.f};
        if (i < n) {
            const uint32_t rec = base + i;
            const uint4 a = P.path_in[rec], b = P.path_in[P.in_cap + rec], c = P.path_in[2u * P.in_cap + rec];
            st.slot = a.x;
            st.origin = V3{__uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w)};
            st.dir = V3{__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z)};
            st.rng = b.w;
            st.thr = V3{__uint_as_float(c.x), __uint_as_float(c.y), __uint_as_float(c.z)};
            // segment_end on the parked end state (a path segment's water is nobody's business: DESIGN.md, path trace)
            const V3 pos{pool[0u * E + i], pool[1u * E + i], pool[2u * E + i]};
            const uint32_t packed = __float_as_uint(pool[3u * E + i]);
            if (packed != kPoolEjected) {
            MarchResult R;
            R.hit = false;
            R.pos = V3{0.f, 0.f, 0.f};
            R.norm = V3{0.f, 0.f, 0.f};
            R.water_dist = 0.0f;
            R.voxel = 0u;
            R.iters = 0u;
            R.visits = 0u;
            if (!(min3_nan_ignoring(pos.x, pos.y, pos.z) < 0.0f ||
                  max(max((uint32_t)trunc2i(pos.x), (uint32_t)trunc2i(pos.y)), (uint32_t)trunc2i(pos.z)) >= P.world.size)) {
                R.hit = true;
                R.pos = pos;
                R.norm = V3{((packed & 1u) ? 1.0f : 0.0f) * -vsign(st.dir.x), ((packed & 2u) ? 1.0f : 0.0f) * -vsign(st.dir.y),
                            ((packed & 4u) ? 1.0f : 0.0f) * -vsign(st.dir.z)};
                R.voxel = packed >> 8;
            }
            V3 light{0.f, 0.f, 0.f};
            bool missed;
            alive = path_after_march(P, st, R, light, missed) && !P.last_bounce;
            if (missed) {
                uint4 t = P.out[st.slot];
                t.x = __float_as_uint(__uint_as_float(t.x) + light.x);
                t.y = __float_as_uint(__uint_as_float(t.y) + light.y);
                t.z = __float_as_uint(__uint_as_float(t.z) + light.z);
                P.out[st.slot] = t;
            }
            }
        }
        if (!CONT) {
            append_paths(P, alive, st, lane);
        } else {
            // a straggler's next segment stays with the stragglers: the bounce launch that marches its generation is
            // already running (or done)
            const unsigned long long ballot = __ballot(alive);
            const uint32_t n_alive = (uint32_t)__popcll(ballot);
            if (n_alive != 0u) {   // (alive implies !P.last_bounce, and then the host gave a cont_out)
                const int leader = __ffsll((long long)ballot) - 1;
                uint32_t at = 0;
                if ((int)lane == leader) at = atomicAdd(&P.cont_counts[seg * kSegStride], n_alive);
                at = (uint32_t)__shfl((int)at, leader, 64) + (uint32_t)__popcll(ballot & ((1ull << lane) - 1ull));
                if (alive) {
                    const uint32_t o = seg * P.hit_seg_cap + at;
                    P.cont_out[o] = make_uint4(st.slot, __float_as_uint(st.origin.x), __float_as_uint(st.origin.y), __float_as_uint(st.origin.z));
                    P.cont_out[P.path_cap + o] = make_uint4(__float_as_uint(st.dir.x), __float_as_uint(st.dir.y), __float_as_uint(st.dir.z), st.rng);
                    P.cont_out[2u * P.path_cap + o] = make_uint4(__float_as_uint(st.thr.x), __float_as_uint(st.thr.y), __float_as_uint(st.thr.z), 0u);
                    P.cont_out[3u * P.path_cap + o] = make_uint4(0u, 0u, 0u, kContFresh);
                }
            }
        }
    }
#ifdef VRT_EXP_POOLDBG
    {
        POOLDBG_T(t3);
        const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0 && blockIdx.x * 4u + wave < 16384u) {
            unsigned long long *d = &g_pool_dbg[(blockIdx.x * 4u + wave) * 8u];
            d[0] = n | (CONT ? (1ull << 32) : 0ull); d[1] = t1 - t0; d[2] = t2 - t1; d[3] = t3 - t2; d[4] = dbg_steps | ((unsigned long long)dbg_dry_steps << 32); d[5] = dbg_wet_lanes | ((unsigned long long)dbg_dry_lanes << 32); d[6] = r0; d[7] = r1;
        }
    }
#endif
}

#ifdef VRT_EXP_POOLDBG
extern "C" void vrt_exp_pool_dbg(unsigned long long *out) {   // read (16384 x 8 words) and reset
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pool_dbg), sizeof(unsigned long long) * 16384 * 8);
    void *p = nullptr;
    (void)hipGetSymbolAddress(&p, HIP_SYMBOL(g_pool_dbg));
    (void)hipMemset(p, 0, sizeof(unsigned long long) * 16384 * 8);
}
#endif
#endif  // VRT_EXPERIMENTS

// ------------------------------------------------------------------------------------------------
// Bounce b >= 1 over the MARCH CELLS (vrt_accel.hip; the default for plain frames of worlds that have them): the pool
// kernel above with a march loop that has ONE load per step and no dependent load at all.
//
// What held the pool kernel at a quarter of its issue rate was the pair of dependent loads of a step in a split cell —
// cell entry, then the voxel's brick entry: 290 + 515 cycles of a 2 200-cycle wave-step, half of them L1 misses
// (profiles/r02_path_pmc_summary.txt) — on the critical path of every ray, and a launch lasts as long as its longest
// chain of steps.  A march cell answers both questions of a step from one 16-byte entry: the leaf's size (a leaf
// cell's lo, or the split cell's size-2 mask) and whether the voxel stops the ray (64 bits; liquids are transparent to
// a path segment, so they count as air — the tables are built with the material table's liquid set).  Which voxel it
// stopped on is only asked after the march, at full width, from the brick (phase C).
// Also new here: phase C takes the rays that hit and the rays that missed in separate batches (a wave executes both sides of that branch otherwise: ~510 + ~130 vector
// instructions per ray), and on the last bounce the rays that hit are not shaded at all (their bounce would be dropped).
// Every ray executes the arithmetic the other kernels execute for it: bit-identical frames (tests).
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kCellsPoolBytesPerWave = kPoolWords * 4u + kPoolEntries * 2u;   // the pool + a u16 order per entry

#ifdef VRT_EXP_CELLDBG
// experiment (tools/ab build, tools/cells_probe.py): per wave of the launch {rays, start, end (100 MHz), wave-steps with rays left
// in the pool | after it ran dry << 32, lanes marching in them likewise, segments done, -, -}
__device__ unsigned long long g_cells_dbg[16384 * 8];
// launch totals of the lookups by what they find and whether the lane's previous lookup was in the same 128-byte line (the
// misses a ray cannot avoid), for this layout and for denser ones that are not built: see tools/cells_probe.py
__device__ unsigned long long g_cells_tot[32];
extern "C" void vrt_exp_cells_tot(unsigned long long *out) {   // read and reset
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cells_tot), sizeof(unsigned long long) * 32);
    void *p = nullptr;
    (void)hipGetSymbolAddress(&p, HIP_SYMBOL(g_cells_tot));
    (void)hipMemset(p, 0, sizeof(unsigned long long) * 32);
}
extern "C" void vrt_exp_cells_dbg(unsigned long long *out) {   // read and reset
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cells_dbg), sizeof(unsigned long long) * 16384 * 8);
    void *p = nullptr;
    (void)hipGetSymbolAddress(&p, HIP_SYMBOL(g_cells_dbg));
    (void)hipMemset(p, 0, sizeof(unsigned long long) * 16384 * 8);
}
#endif

// what a launch of path_bounce_cells_kernel is given (one argument: the kernel reads it again for every segment)
struct CellsLaunch {
    FrameParams P;
    uint32_t refill_at;   // a wave takes rays from its pool when this many of its lanes are idle
    uint32_t segments;    // bounce segments in this launch: all that the frame's paths have left
};

#ifndef VRT_CELLS_LOAD_AUX
#define VRT_CELLS_LOAD_AUX 0   // cache policy of the march-cell load (A/B builds: 1 sc0, 2 nt, 16 sc1)
#endif
template <bool DIRECT>
#ifndef VRT_CELLS_NO_WAVES_ATTR
__attribute__((amdgpu_waves_per_eu(8, 8)))
#endif
__global__ void __launch_bounds__(256) path_bounce_cells_kernel(CellsLaunch L) {
    const FrameParams &K = L.P;
    extern __shared__ uint32_t smem[];
    uint32_t *s_liquid = smem;
    if (threadIdx.x < 8) s_liquid[threadIdx.x] = K.liquid[threadIdx.x];
    if (blockIdx.x == 0 && K.seg_clear) K.seg_clear[threadIdx.x * kSegStride] = 0u;   // kHitSegments == blockDim.x cursors
    __syncthreads();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr uint32_t E = kPoolEntries;

    // this wave's paths: the workgroup takes up to 4 E records of its segment, split evenly over its waves
    const uint32_t seg = blockIdx.x % kHitSegments, part = blockIdx.x / kHitSegments;
    const uint32_t count = K.seg_in[seg * kSegStride];
    const uint32_t wg_begin = part * 4u * E;
    if (wg_begin >= count) return;
    const uint32_t n_wg = min(4u * E, count - wg_begin), per = (n_wg + 3u) / 4u;
    if (wave * per >= n_wg) return;
    uint32_t n = __builtin_amdgcn_readfirstlane(min(per, n_wg - wave * per));   // <= E
#ifdef VRT_EXP_CELLDBG
    const unsigned long long dbg_t0 = __builtin_amdgcn_s_memrealtime();
    const uint32_t dbg_n0 = n;
    uint32_t dbg_wet = 0, dbg_wet_lanes = 0, dbg_dry = 0, dbg_dry_lanes = 0, dbg_segments = 0, dbg_air4 = 0, dbg_air8 = 0, dbg_air16 = 0;
    uint32_t dbg_refill_ticks = 0, dbg_refills = 0, dbg_park_ticks = 0, dbg_load_ticks = 0;
    uint32_t dbg_t_loaded = 0;   // hand-out rounds that took rays: their time in 16-cycle ticks of the shader clock, their number
#ifdef VRT_EXP_CELLDBG_FULL   // (the lookup classes: 30 registers more — a build of its own, 4 waves per SIMD)
    uint32_t dbg_tot[20], dbg_lines = 0, dbg_line_steps = 0;   // (per lane; summed at the wave's end)
    for (int q = 0; q < 20; q++) dbg_tot[q] = 0;
    // per lane: the previous lookup's lines under the layouts compared, its class, and the ray's origin voxel
    uint32_t dbg_l8 = ~0u, dbg_l16 = ~0u, dbg_lc = ~0u, dbg_l4 = ~0u, dbg_prev_big = 0u;
    int dbg_ox = 0, dbg_oy = 0, dbg_oz = 0;
#endif
#endif
    const uint32_t base = __builtin_amdgcn_readfirstlane(seg * K.in_seg_cap + wg_begin + wave * per);
    // The wave keeps its paths for ALL the segments that are left (`segments` of them): the survivors of one segment are
    // compacted — by the wave alone, no cursor, no atomic — into the same index range of the other path buffer and are the
    // wave's pool for the next.  A launch per bounce ends when its slowest wave does (221 wave-steps against 81 on average),
    // three times per frame; here a wave that is done with one segment starts the next, and the launch waits for the slowest
    // SUM.  (The pools shrink — C4: 243, 198, 146 paths a wave — and phases A and C run their last batch partly empty.)
    // (as offsets from one pointer, so that the address of a record stays "scalar base + lane index")
    uint4 *const recs = K.path_in < K.path_out ? const_cast<uint4 *>(K.path_in) : K.path_out;
    uint32_t in_at = __builtin_amdgcn_readfirstlane((uint32_t)(K.path_in - recs));
    uint32_t out_at = __builtin_amdgcn_readfirstlane((uint32_t)(K.path_out - recs));
    for (uint32_t left = L.segments;; left--) {   // (left: segments still to do, this one included)
    // What does not change from one segment to the next is made anew for every one of them — the launch's parameters read
    // again from the kernel-argument segment (scalar loads), the lane's number and what follows from it computed again —
    // and not kept in registers around the whole loop: that is 15 VGPRs and 30 SGPRs too many for eight waves a SIMD.
    typedef const __attribute__((address_space(4))) CellsLaunch *KernArgs;
    KernArgs kargs = (KernArgs)__builtin_amdgcn_kernarg_segment_ptr();   // (L is the kernel's only argument)
    asm volatile("" : "+s"(kargs));
    const FrameParams &P = ((const CellsLaunch *)kargs)->P;
    const uint32_t refill_at = ((const CellsLaunch *)kargs)->refill_at;
    uint32_t none = 0u;
    asm volatile("" : "+s"(none));
    const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, none));
    float *pool = reinterpret_cast<float *>(smem + 8) + (wave + none) * kPoolWords;
    uint16_t *order = reinterpret_cast<uint16_t *>(smem + 8 + 4u * kPoolWords) + (wave + none) * E;
    const float world_max = 0.0f + (float)P.world.size;
    const bool last_bounce = left == 1u;   // (the launch's last segment is the paths' last)

    // ---- A: the unit steps of every ray (nine divides, three square roots), full width ----
#pragma unroll
    for (uint32_t k = 0; k < kPoolBatches; k++) {
        const uint32_t i = k * 64u + lane;
        if (i < n) {
            const uint4 b = recs[in_at + P.in_cap + base + i];
            {   // the origin plane is touched too: the hand-outs then find both planes of the record in L2 (the previous launch
                // wrote them, 69 MB ago; + 1 %.  Touching the next hand-outs' records at every refill instead: - 2 %)
                const uint4 a_ = recs[in_at + base + i];
                asm volatile("" :: "v"(a_.x));
            }
            const V3 unit = unit_steps(V3{__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z)});
            pool[0u * E + i] = unit.x; pool[1u * E + i] = unit.y; pool[2u * E + i] = unit.z;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- B: the marches, lanes refilled from the pool; a ray that has stopped keeps its end state in its registers until
    // the wave's next refill parks it.  Water is not tracked (no output of a path segment depends on it). ----
    {
        const TableBuf mb = table_buffer(P.mblk, P.mblk_bytes), db = table_buffer(P.cdir, P.cdir_bytes);
        const TableBuf bb = table_buffer(P.bricks, P.brick_bytes);
        // the chunk directory: [S][S+1][S+1] with a zero border; a direct world: [4S][4S+1][4S+1] lines of 128 bytes
        const uint32_t drow = (P.grid_dim / 8u + 1u) * 4u, dslab = (P.grid_dim / 8u + 1u) * drow;
        const uint32_t row128 = (P.grid_dim / 2u + 1u) * 128u, slab128 = (P.grid_dim / 2u + 1u) * row128;   // < 2^23: S <= 16
        const uint32_t wsize = P.world.size;
        V3 pos{0.f, 0.f, 0.f}, dir{0.f, 0.f, 0.f};
        float ux = 0.f, uy = 0.f, uz = 0.f, step = -1.f, adx = 0.f, ady = 0.f, adz = 0.f;
        // the direction masks with (q) of vrt_march.h riding on them: 0 or ~0, minus the bits of 2^23 — the bit-field insert
        // looks at their low five bits only (lo <= 31), which are the plain mask's
        uint32_t mxm = 0u, mym = 0u, mzm = 0u, ref = 0u, iter = 0u, idx = 0u;
        int vx = 0, vy = 0, vz = 0;
        // the chunk the ray is in — its coordinates as one number — and where that chunk's block of march cells begins
        constexpr uint32_t kNoChunk = 0x7FFFFFFFu;
        uint32_t ckey = kNoChunk, cblock = 0u;
        bool marching = false, parked = true, not_finite = false;
        uint32_t next = 0u;   // wave-uniform: the pool's first ray not handed out yet

        // the end state phase C needs: where, through which faces, on what (bit 3: `what` is the split cell's brick)
        auto park = [&]() __attribute__((always_inline)) {
            uint32_t packed = (int)ref < 0 ? (8u | ((ref & 0x7FFFFFC0u) >> 2)) : ((ref >> 16) << 4);
            if (step != -1.0f) packed |= (step == adx ? 1u : 0u) | (step == ady ? 2u : 0u) | (step == adz ? 4u : 0u);
            pool[0u * E + idx] = pos.x; pool[1u * E + idx] = pos.y; pool[2u * E + idx] = pos.z;
            pool[3u * E + idx] = __uint_as_float(packed);
            parked = true;
        };
        auto take = [&](uint32_t at) __attribute__((always_inline)) {
            idx = at;
            const uint32_t rec = base + idx;
            const uint4 a = recs[in_at + rec], b = recs[in_at + P.in_cap + rec];
#ifdef VRT_EXP_CELLDBG
            asm volatile("s_waitcnt vmcnt(0)" :: "v"(a.x), "v"(b.x) : "memory");
            dbg_t_loaded = (uint32_t)(__builtin_amdgcn_s_memtime() >> 4);
#endif
            const V3 origin{__uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w)};
            dir = V3{__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z)};
            not_finite = !(finite3(origin) && finite3(dir));
            ux = pool[0u * E + idx]; uy = pool[1u * E + idx]; uz = pool[2u * E + idx];
            constexpr uint32_t kTwo23 = 0x4B000000u;
            mxm = (dir.x >= 0.0f ? ~0u : 0u) - kTwo23; mym = (dir.y >= 0.0f ? ~0u : 0u) - kTwo23; mzm = (dir.z >= 0.0f ? ~0u : 0u) - kTwo23;
            ref = 0u;
            ckey = kNoChunk;
            marching = true;
            parked = false;
            pos = nudged(origin, dir);
            step = -1.0f; adx = ady = adz = 0.0f;
            iter = 0u;
            if ((pos.x <= 0.0f || pos.y <= 0.0f || pos.z <= 0.0f) || (pos.x >= world_max || pos.y >= world_max || pos.z >= world_max)) {
                // starts outside the world: a miss before any lookup.  Its end state says so (a position outside), and is
                // parked right here: the wave may find nothing left to march and never come back to the refill
                marching = false;
                pos = V3{-1.0f, -1.0f, -1.0f};
                park();
            }
            vx = trunc2i(pos.x); vy = trunc2i(pos.y); vz = trunc2i(pos.z);
#ifdef VRT_EXP_CELLDBG_FULL
            dbg_l8 = dbg_l16 = dbg_lc = dbg_l4 = ~0u; dbg_prev_big = 0u; dbg_ox = vx; dbg_oy = vy; dbg_oz = vz;
#endif
#ifdef VRT_EXP_NOMARCH   // counting experiment: every ray ends where it starts
            if (marching) { marching = false; park(); }
#endif
        };
        // the step to the leaf's exit face for a leaf of size lo + 1 (take_step of march_grid)
        auto take_step = [&](uint32_t lo) __attribute__((always_inline)) {
            // (h), (q) of vrt_march.h: the exit plane (v | lo) + 1 or v & ~lo, as a float without a conversion
            const float tx = (__uint_as_float(bfi(lo, mxm, (uint32_t)vx) - mxm) - 8388608.0f) - pos.x;
            const float ty = (__uint_as_float(bfi(lo, mym, (uint32_t)vy) - mym) - 8388608.0f) - pos.y;
            const float tz = (__uint_as_float(bfi(lo, mzm, (uint32_t)vz) - mzm) - 8388608.0f) - pos.z;
            adx = abs_mul(tx, ux);
            ady = abs_mul(ty, uy);
            adz = abs_mul(tz, uz);
            step = min3_f32(adx, ady, adz);   // (p) of vrt_march.h
            if (__ballot(!(step > 0.0f)) != 0ull)
                step = __uint_as_float(min3_u32(__float_as_uint(adx) - 1u, __float_as_uint(ady) - 1u, __float_as_uint(adz) - 1u) + 1u);
            const float sp = step + 0.001f;
            pos.x += dir.x * (step == adx ? sp : step);
            pos.y += dir.y * (step == ady ? sp : step);
            pos.z += dir.z * (step == adz ? sp : step);
            vx = flr2i(pos.x);
            vy = flr2i(pos.y);
            vz = flr2i(pos.z);
        };
        // (l) of vrt_march.h: the general step as march_grid has it for a wave with a ray that is not finite — the shader's
        // own bounds test, its lookup at i32(f32) coordinates; over the cell grid and the bricks (rare: NaN cameras)
        auto careful_step = [&]() __attribute__((always_inline)) {
            iter += 1u;
            vx = trunc2i(pos.x);
            vy = trunc2i(pos.y);
            vz = trunc2i(pos.z);
            uint32_t e = 0u;   // the cell's entry of the cell grid: the march cell's first word
            if (!(min3_nan_ignoring(pos.x, pos.y, pos.z) < 0.0f || max(max((uint32_t)vx, (uint32_t)vy), (uint32_t)vz) >= wsize)) {
                const uint32_t sub = ((((uint32_t)vz >> 2) & 1u) << 2) | ((((uint32_t)vy >> 2) & 1u) << 1) | (((uint32_t)vx >> 2) & 1u);
                uint32_t off;
                if (DIRECT) {
                    off = mad_i24(vz >> 3, slab128, mad_i24(vy >> 3, row128, ((uint32_t)(vx >> 3) << 7) + (sub << 4)));
                } else {   // (inside the world: the chunk has an entry in the directory)
                    const uint32_t block = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(db, mad_i24(vz >> 5, dslab, mad_i24(vy >> 5, drow, (uint32_t)(vx >> 5) << 2)), 0, 0) << 13;
                    const uint32_t line = ((((((uint32_t)vz >> 3) & 3u) << 2) | (((uint32_t)vy >> 3) & 3u)) << 2) | (((uint32_t)vx >> 3) & 3u);
                    off = block + (((line << 3) | sub) << 4);
                }
                e = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(mb, off, 0, 0);
            }
            uint32_t lo = e, voxel = 0u;
            bool stop = e == 0u;   // border, or past either end of the grid: the position is outside the world
            if (!stop) {
                if ((int)e < 0) {
                    const uint32_t u = ((uint32_t)vx & 3u) | (((uint32_t)vy & 3u) << 2) | (((uint32_t)vz & 3u) << 4);
                    const uint32_t b = (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(bb, (e + u) << 1, 0, 0);
                    lo = b & 1u;
                    voxel = b >> 1;
                } else if (e > 31u) {
                    lo = e & 31u;
                    voxel = e >> 16;
                }
                stop = voxel != 0u && !is_liquid_ranged(P, s_liquid, voxel);   // solid: the hit
            }
            ref = voxel << 16;   // (the voxel itself, as a leaf cell's entry has it)
            if (!stop) {
                take_step(lo);
                stop = iter >= kMaxSteps;
            }
            marching = !stop;
        };
        for (;;) {
            // ---- refill: park what has stopped, hand out the pool's next rays ----
#ifdef VRT_EXP_CELLDBG
            const unsigned long long dbg_r0 = __builtin_amdgcn_s_memtime();
            const uint32_t dbg_next0 = next;
#endif
            if (!marching && !parked) park();
#ifdef VRT_EXP_CELLDBG
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const unsigned long long dbg_r1 = __builtin_amdgcn_s_memtime();
            dbg_t_loaded = 0;
#endif
            {
                const unsigned long long idle = __ballot(!marching);
                const uint32_t at = next + lanes_below(idle);
                if (!marching && at < n) take(at);
                next = min(n, next + (uint32_t)__popcll(idle));
            }
#ifdef VRT_EXP_CELLDBG
            { const unsigned long long took_ = __ballot(dbg_t_loaded != 0u); if (took_) { dbg_park_ticks += (uint32_t)((dbg_r1 - dbg_r0) >> 4); dbg_load_ticks += (uint32_t)__builtin_amdgcn_readlane((int)dbg_t_loaded, (int)__builtin_ctzll(took_)) - (uint32_t)(dbg_r1 >> 4); } }
#endif
#ifdef VRT_EXP_CELLDBG
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            if (next != dbg_next0) { dbg_refill_ticks += (uint32_t)((__builtin_amdgcn_s_memtime() - dbg_r0) >> 4); dbg_refills++; }
#endif
            if (__ballot(marching) == 0ull) {
                if (next >= n) break;   // the pool is empty and nobody marches (every ray is parked: take() parks the ones that start outside)
                continue;
            }
            // (two loops, chosen per refill round: one loop with both bodies costs a register move per loop-carried value and
            // body on every trip — 22 of them, a quarter of the step)
            if (__ballot(marching && not_finite) != 0ull) {   // wave-uniform, rare
                for (;;) {
                    if (marching) careful_step();
                    const uint32_t n_march = (uint32_t)__popcll(__ballot(marching));
                    if (n_march == 0u || (next < n && 64u - n_march >= refill_at)) break;
                }
                continue;
            }
            for (;;) {
#ifdef VRT_EXP_CELLDBG
                {
                    const uint32_t m_ = (uint32_t)__popcll(__ballot(marching));
                    if (next < n) { dbg_wet++; dbg_wet_lanes += m_; } else { dbg_dry++; dbg_dry_lanes += m_; }
                }
#endif
                if (marching) {
                    // the chunk's block of march cells: looked up in the chunk directory when the ray has entered another chunk
                    // (coordinates -1 .. S: one voxel beyond the world at most; they make one number, base 128).  Outside the
                    // world the directory's border — or a load past either end of it — says block 0, whose cells are all zeros
                    // The cell inside its line of 2 x 2 x 2 (bits 2 of the coordinates), the line inside the chunk's block (bits 3, 4) —
                    // or, in a direct world, among the lines of the whole world (bits 3 and up; the border lines stay zero)
                    const uint32_t sub = ((((uint32_t)vz >> 2) & 1u) << 2) | ((((uint32_t)vy >> 2) & 1u) << 1) | (((uint32_t)vx >> 2) & 1u);
                    uint32_t off;
                    if (DIRECT) {
                        off = mad_i24(vz >> 3, slab128, mad_i24(vy >> 3, row128, ((uint32_t)(vx >> 3) << 7) + (sub << 4)));
                    } else {
                        const uint32_t key = (uint32_t)((((vz >> 5) << 7) + (vy >> 5)) << 7) + (uint32_t)(vx >> 5);
                        if (key != ckey) {   // (looking it up on every step instead: 1 % slower — the load is an L1 hit, but a dependent one)
                            ckey = key;
                            cblock = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(db, mad_i24(vz >> 5, dslab, mad_i24(vy >> 5, drow, (uint32_t)(vx >> 5) << 2)), 0, 0) << 13;
                        }
                        const uint32_t line = (((((uint32_t)vz >> 3) & 3u) << 2 | (((uint32_t)vy >> 3) & 3u)) << 2) | (((uint32_t)vx >> 3) & 3u);
                        off = cblock + (((line << 3) | sub) << 4);
                    }
                    // one 16-byte load answers the step: c.x the cell's entry, c.y the size-2 mask of a split cell, c.z / c.w
                    // the voxels a ray passes (zero stops it)
                    const uint4 c = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(mb, off, 0, VRT_CELLS_LOAD_AUX));
#ifdef VRT_EXP_CELLS_LOAD   // tools/ab experiments only: one more 16-byte load per step — 1: the cell just read (an L1 hit), 2: a line
                            // nobody shares (a miss of L1, a hit or miss of L2), 3: the line next to the cell's (L2-resident like it)
                    {
                        const uint32_t off_ = VRT_EXP_CELLS_LOAD == 1 ? off : VRT_EXP_CELLS_LOAD == 3 ? (off ^ 128u) :
                                              ((iter * 0x9E3779B9u + idx * 0x85EBCA6Bu + lane * 0xC2B2AE35u) % (P.mblk_bytes / 16u)) * 16u;
                        const uint4 x_ = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(mb, off_, 0, 0));
                        asm volatile("" :: "v"(x_.x), "v"(x_.y), "v"(x_.z), "v"(x_.w));
                    }
#endif
#ifdef VRT_EXP_CELLS_VALU   // ... extra full-rate vector instructions per step
#pragma unroll
                    for (int k_ = 0; k_ < VRT_EXP_CELLS_VALU; k_++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(idx) : "v"(0u));
#endif
#ifdef VRT_EXP_CELLS_SALU
#pragma unroll
                    for (int k_ = 0; k_ < VRT_EXP_CELLS_SALU; k_++) asm volatile("s_mov_b32 vcc_lo, 0" ::: "vcc");
#endif
                    iter += 1u;
                    // u = (x&3) | (y&3) << 2 | (z&3) << 4, with z's upper bits left on top: the shifts below use the low bits only
                    const uint32_t u = ((((uint32_t)vz << 2) | ((uint32_t)vy & 3u)) << 2) | ((uint32_t)vx & 3u);
                    const uint32_t passes = (uint32_t)((((unsigned long long)c.w << 32) | c.z) >> (u & 63u)) & 1u;
                    const uint32_t lo = (c.x & 31u) | __builtin_amdgcn_ubfe(c.y, (u >> 1) & 31u, 1u);
#ifdef VRT_EXP_CELLDBG
                    dbg_air4 += (uint32_t)__popcll(__ballot(c.x >= 3u && c.x <= 31u));     // lookups answered by an air leaf of the cell grid
                    dbg_air8 += (uint32_t)__popcll(__ballot(c.x >= 7u && c.x <= 31u));     // ... of 8 voxels or more: a whole line of cells
                    dbg_air16 += (uint32_t)__popcll(__ballot(c.x >= 15u && c.x <= 31u));
#endif
#ifdef VRT_EXP_CELLDBG_FULL
                    {
                        auto cnt = [&](int q, bool b) __attribute__((always_inline)) { dbg_tot[q] += b ? 1u : 0u; };
                        const uint32_t X = (uint32_t)(vx + 64), Y = (uint32_t)(vy + 64), Z = (uint32_t)(vz + 64);   // (coordinates -1 .. size)
                        const uint32_t l8 = ((Z >> 3) << 20) | ((Y >> 3) << 10) | (X >> 3);        // today's line: 8 x 8 x 8 voxels
                        const uint32_t l16 = ((Z >> 3) << 20) | ((Y >> 3) << 10) | (X >> 4);       // 8-byte cells: 16 x 8 x 8
                        const uint32_t l4 = ((Z >> 4) << 20) | ((Y >> 3) << 10) | (X >> 4);        // (4-byte cells: 16 x 8 x 16)
                        const uint32_t lc = ((Z >> 5) << 20) | ((Y >> 5) << 10) | (X >> 6);        // a byte per 8^3 voxels: 64 x 32 x 32
                        const bool big = c.x >= 7u && c.x <= 31u, air4 = c.x == 3u, split = (int)c.x < 0;
                        const bool n8 = l8 != dbg_l8, n16 = l16 != dbg_l16, n4 = l4 != dbg_l4, nc = lc != dbg_lc;
                        const bool near = max(max(abs(vx - dbg_ox), abs(vy - dbg_oy)), abs(vz - dbg_oz)) < 32;
                        cnt(0, true); cnt(1, n8); cnt(2, n16); cnt(3, n4);
                        cnt(4, big); cnt(5, big && n8); cnt(6, big && nc);
                        cnt(7, air4); cnt(8, air4 && n8); cnt(9, air4 && n16);
                        cnt(10, split); cnt(11, split && n8); cnt(12, split && n16);
                        cnt(13, !big && dbg_prev_big != 0u);            // a lane that was cruising in leaves of 8 or more finds something finer
                        cnt(14, near); cnt(15, near && n8);
                        cnt(16, !big && !air4 && !split); cnt(17, !big && !air4 && !split && n8);   // solid leaves of the grid, the border
                        cnt(18, big && dbg_prev_big != 0u);             // cruising goes on
                        cnt(19, big && dbg_prev_big != 0u && nc);       // ... into another line of the coarse table
                        // distinct lines among the wave's marching lanes in this wave-step (a leader loop over the first lanes' lines)
                        {
                            unsigned long long todo = __ballot(true);
                            uint32_t lines = 0;
                            while (todo) {
                                const uint32_t first = (uint32_t)__builtin_amdgcn_readlane((int)l8, (int)__builtin_ctzll(todo));
                                todo &= ~__ballot(l8 == first);
                                lines++;
                            }
                            dbg_lines += lines; dbg_line_steps += 1u;
                        }
                        dbg_l8 = l8; dbg_l16 = l16; dbg_l4 = l4; dbg_lc = lc; dbg_prev_big = big ? 1u : 0u;
                    }
#endif
                    bool stop = passes == 0u;
                    ref = c.x;
                    if (!stop) {
                        take_step(lo);
                        if (iter >= kMaxSteps) {
                            // out of lookups in air or in a liquid (:220, :293): the segment ends as a hit on the voxel of the last
                            // lookup — which for a split cell is in its brick, at the position that was looked up
                            stop = true;
                            ref = (int)c.x < 0 ? ((uint32_t)__builtin_amdgcn_raw_buffer_load_b16(bb, ((c.x & 0x7FFFFFFFu) + (u & 63u)) << 1, 0, 0) >> 1) << 16 : c.x;
                        }
                    }
                    marching = !stop;
                }
                const uint32_t n_march = (uint32_t)__popcll(__ballot(marching));
                if (n_march == 0u || (next < n && 64u - n_march >= refill_at)) break;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- between B and C: the rays that hit first, then the rays that missed (their end position is outside the world) ----
    uint32_t n_hit = 0u;
    {
        bool hit[kPoolBatches];
        uint32_t cnt[kPoolBatches];
#pragma unroll
        for (uint32_t k = 0; k < kPoolBatches; k++) {
            const uint32_t i = k * 64u + lane;
            hit[k] = false;
            if (i < n) {
                const V3 pos{pool[0u * E + i], pool[1u * E + i], pool[2u * E + i]};
                hit[k] = !(min3_nan_ignoring(pos.x, pos.y, pos.z) < 0.0f ||
                           max(max((uint32_t)trunc2i(pos.x), (uint32_t)trunc2i(pos.y)), (uint32_t)trunc2i(pos.z)) >= P.world.size);
            }
            cnt[k] = (uint32_t)__popcll(__ballot(hit[k]));
            n_hit += cnt[k];
        }
        uint32_t at_hit = 0u, at_miss = n_hit;
#pragma unroll
        for (uint32_t k = 0; k < kPoolBatches; k++) {
            const uint32_t i = k * 64u + lane;
            const unsigned long long mh = __ballot(hit[k]), mm = __ballot(i < n && !hit[k]);
            if (hit[k]) order[at_hit + lanes_below(mh)] = (uint16_t)i;
            else if (i < n) order[at_miss + lanes_below(mm)] = (uint16_t)i;
            at_hit += (uint32_t)__popcll(mh);
            at_miss += (uint32_t)__popcll(mm);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- C: what follows the march, full width.  On the last bounce a ray that hit has nothing left to do: its bounce
    // would be dropped and only a miss adds light ----
    const TableBuf bb = table_buffer(P.bricks, P.brick_bytes);
    uint32_t n_out = 0u;   // survivors so far: the next segment's pool
    for (uint32_t j0 = last_bounce ? n_hit & ~63u : 0u; j0 < n; j0 += 64u) {
        const uint32_t j = j0 + lane;
        bool alive = false;
        PathState st;
        st.slot = 0; st.rng = 0;
        st.origin = st.dir = st.thr = V3{0.f, 0.f, 0.f};
        if (j < n && !(last_bounce && j < n_hit)) {
            const uint32_t i = order[j];
            const uint32_t rec = base + i;
            const uint4 a = recs[in_at + rec], b = recs[in_at + P.in_cap + rec], c = recs[in_at + 2u * P.in_cap + rec];
            st.slot = a.x;
            st.origin = V3{__uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w)};
            st.dir = V3{__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z)};
            st.rng = b.w;
            st.thr = V3{__uint_as_float(c.x), __uint_as_float(c.y), __uint_as_float(c.z)};
            // segment_end on the parked end state (a path segment's water is nobody's business: DESIGN.md, path trace)
            const V3 pos{pool[0u * E + i], pool[1u * E + i], pool[2u * E + i]};
            const uint32_t packed = __float_as_uint(pool[3u * E + i]);
            MarchResult R;
            R.hit = false;
            R.pos = V3{0.f, 0.f, 0.f};
            R.norm = V3{0.f, 0.f, 0.f};
            R.water_dist = 0.0f;
            R.voxel = 0u;
            R.iters = 0u;
            R.visits = 0u;
            if (j < n_hit) {
                R.hit = true;
                R.pos = pos;
                R.norm = V3{((packed & 1u) ? 1.0f : 0.0f) * -vsign(st.dir.x), ((packed & 2u) ? 1.0f : 0.0f) * -vsign(st.dir.y),
                            ((packed & 4u) ? 1.0f : 0.0f) * -vsign(st.dir.z)};
                R.voxel = packed >> 4;
                if (packed & 8u) {   // stopped in a split cell: the voxel is in the cell's brick, at the end position
                    const uint32_t u = ((uint32_t)trunc2i(pos.x) & 3u) | (((uint32_t)trunc2i(pos.y) & 3u) << 2) | (((uint32_t)trunc2i(pos.z) & 3u) << 4);
                    R.voxel = (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(bb, (((packed >> 4) << 6) + u) << 1, 0, 0) >> 1;
                }
            }
            V3 light{0.f, 0.f, 0.f};
            bool missed;
            alive = path_after_march(P, st, R, light, missed) && !last_bounce;
            if (missed) {
                uint4 t = P.out[st.slot];
                t.x = __float_as_uint(__uint_as_float(t.x) + light.x);
                t.y = __float_as_uint(__uint_as_float(t.y) + light.y);
                t.z = __float_as_uint(__uint_as_float(t.z) + light.z);
                P.out[st.slot] = t;
            }
        }
        if (left != 1u) {   // the survivors, compacted into this wave's own range of the other buffer
            const unsigned long long m = __ballot(alive);
            if (alive) {
                const uint32_t o = out_at + base + n_out + lanes_below(m);
                recs[o] = make_uint4(st.slot, __float_as_uint(st.origin.x), __float_as_uint(st.origin.y), __float_as_uint(st.origin.z));
                recs[P.path_cap + o] = make_uint4(__float_as_uint(st.dir.x), __float_as_uint(st.dir.y), __float_as_uint(st.dir.z), st.rng);
                recs[2u * P.path_cap + o] = make_uint4(__float_as_uint(st.thr.x), __float_as_uint(st.thr.y), __float_as_uint(st.thr.z), 0u);
            }
            n_out += (uint32_t)__popcll(m);
        }
    }
#ifdef VRT_EXP_CELLDBG
    dbg_segments++;
    if (left == 1u || n_out == 0u) {
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0 && blockIdx.x * 4u + wave < 16384u) {
            unsigned long long *d = &g_cells_dbg[(blockIdx.x * 4u + wave) * 8u];
            d[0] = dbg_n0 | ((unsigned long long)dbg_park_ticks << 32); d[1] = dbg_t0; d[2] = t1; d[3] = dbg_wet | ((unsigned long long)dbg_dry << 32);
            d[4] = dbg_wet_lanes | ((unsigned long long)dbg_dry_lanes << 32); d[5] = (dbg_segments & 0xFFu) | ((unsigned long long)(dbg_load_ticks & 0xFFFFFFu) << 8) | ((unsigned long long)dbg_refills << 32);
            d[6] = dbg_air4 | ((unsigned long long)dbg_air8 << 32); d[7] = dbg_air16 | ((unsigned long long)dbg_refill_ticks << 32);
        }
#ifdef VRT_EXP_CELLDBG_FULL
#pragma unroll
        for (int q = 0; q < 20; q++) {
            const unsigned long long sum = wave_sum((unsigned long long)dbg_tot[q]);
            if (lane == 0) atomicAdd(&g_cells_tot[q], sum);
        }
        if (lane == 0) { atomicAdd(&g_cells_tot[20], (unsigned long long)dbg_lines); atomicAdd(&g_cells_tot[21], (unsigned long long)dbg_line_steps); }
#endif
    }
#endif
    if (left == 1u || n_out == 0u) break;
    // the next segment: the records just written are read back by other lanes of this wave (same CU, same L1)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    n = n_out;
    {
        const uint32_t t = in_at;
        in_at = out_at;
        out_at = t;
    }
    }
}

#ifdef VRT_EXPERIMENTS
__device__ __forceinline__ uint32_t path_xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u; }

// Next tile of the frame for this wave: its own XCD's queue first, then the others (tile i of queue x = x + 8 i).
// ~0u: the frame has no tiles left.  Wave-uniform.
__device__ __forceinline__ uint32_t next_tile(uint32_t *heads, uint32_t tiles, uint32_t &queue_round, uint32_t xcc, uint32_t lane) {
    while (queue_round < 8u) {
        const uint32_t q = (xcc + queue_round) & 7u;
        uint32_t i = 0;   // (called by all 64 lanes: lane 0 takes the ticket, everybody reads it)
        if (lane == 0) i = __hip_atomic_fetch_add(&heads[q * 16u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t t = q + 8u * (uint32_t)__builtin_amdgcn_readfirstlane((int)i);
        if (t < tiles) return t;
        queue_round += 1u;   // that queue is empty for good
    }
    return ~0u;
}

__global__ void __launch_bounds__(256) path_persistent_kernel(FrameParams P, uint32_t *heads, uint32_t refill_at) {
    extern __shared__ uint32_t smem[];
    uint32_t *s_liquid = smem;
    if (threadIdx.x < 8) s_liquid[threadIdx.x] = P.liquid[threadIdx.x];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t xcc = path_xcc_id();
    const TableBuf gb = table_buffer(P.grid, P.grid_bytes), bb = table_buffer(P.bricks, P.brick_bytes);
    const uint32_t row_bytes = (P.grid_dim + 1u) * 4u, slab_bytes = (P.grid_dim + 1u) * row_bytes;
    const uint32_t bounces = P.settings.max_ray_bounces;
    const float fspp = (float)P.spp;

    // the wave's queue: pixels q_pos .. 63 of tile q_tile are still to be handed out
    uint32_t queue_round = 0u;
    uint32_t q_tile = next_tile(heads, P.tiles_local, queue_round, xcc, lane), q_pos = 0u;

    enum : uint32_t { kMarching = 0u, kWaiting = 1u, kRetired = 2u };
    uint32_t state = kWaiting;   // kWaiting: the segment is over (or there is none yet) — something has to be decided
    bool have_pixel = false, started = false;
    Segment m;
    m.pos = m.dir = V3{0.f, 0.f, 0.f};
    m.ux = m.uy = m.uz = 0.f; m.mxm = m.mym = m.mzm = 0u; m.vx = m.vy = m.vz = 0;
    m.step = -1.f; m.adx = m.ady = m.adz = 0.f; m.dew = -1.f; m.total_len = 0.f; m.water_dist = 0.f;
    m.slow_bit = 0u; m.voxel = 0u; m.iter = 0u; m.careful = false;
    PathState st;
    st.slot = 0u; st.rng = 0u;
    st.origin = st.dir = st.thr = V3{0.f, 0.f, 0.f};
    V3 sum{0.f, 0.f, 0.f};
    uint32_t px = 0u, py = 0u, sample = 0u, bounce = 0u, id0 = 0u;

    for (;;) {
        // ---- 1. (waiting lanes) what comes after the segment this lane has just finished ----
        bool new_segment = false;
        if (state == kWaiting && have_pixel) {
            const MarchResult R = segment_end(P, m, started);
            if (sample == 0u && bounce == 0u) {   // the id word of the primary segment, composed as shade() does
                id0 = R.voxel & VRT_ID_VOXEL_MASK;
                if (R.hit) id0 |= VRT_ID_HIT;
                if (R.norm.x != 0.0f) id0 |= VRT_ID_NX;
                if (R.norm.y != 0.0f) id0 |= VRT_ID_NY;
                if (R.norm.z != 0.0f) id0 |= VRT_ID_NZ;
                if (R.water_dist != 0.0f) id0 |= VRT_ID_WATER;
            }
            bool path_over;
            if (!R.hit) {   // path_segment(): a miss adds the sky's light and ends the path
                const V3 sky = ray_sky(P, st.origin, st.dir);
                sum.x += sky.x * st.thr.x;
                sum.y += sky.y * st.thr.y;
                sum.z += sky.z * st.thr.z;
                path_over = true;
            } else {
                const V3 mc = hit_color(P, R);
                const float d = vdot(R.norm, st.dir);
                const V3 spec{st.dir.x - 2.0f * R.norm.x * d, st.dir.y - 2.0f * R.norm.y * d, st.dir.z - 2.0f * R.norm.z * d};
                const V3 rd = rng_next_dir(st.rng);
                const V3 sc = normalize_wave(V3{R.norm.x + rd.x, R.norm.y + rd.y, R.norm.z + rd.z});
                const float scatter = P.mats[min(R.voxel, 255u)].scatter;
                const V3 nd = normalize_wave(V3{vmix(spec.x, sc.x, scatter), vmix(spec.y, sc.y, scatter), vmix(spec.z, sc.z, scatter)});
                st.thr = V3{st.thr.x * mc.x, st.thr.y * mc.y, st.thr.z * mc.z};
                st.origin = V3{R.pos.x + R.norm.x * kShadowBias, R.pos.y + R.norm.y * kShadowBias, R.pos.z + R.norm.z * kShadowBias};
                st.dir = nd;
                bounce += 1u;
                path_over = bounce >= bounces;
                new_segment = !path_over;
            }
            if (path_over) {
                sample += 1u;
                if (sample < P.spp) {   // the pixel's next sample: its primary ray again, a fresh RNG stream
                    create_ray(P, (int)px, (int)py, st.origin, st.dir);
                    st.thr = V3{1.0f, 1.0f, 1.0f};
                    st.rng = py * P.width + px + sample * (P.width * P.height) + P.seed * 0x9E3779B9u;
                    bounce = 0u;
                    new_segment = true;
                } else {
                    P.out[st.slot] = make_uint4(__float_as_uint(sum.x / fspp), __float_as_uint(sum.y / fspp), __float_as_uint(sum.z / fspp), id0);
                    have_pixel = false;
                }
            }
        }
        // ---- 2. (the whole wave) lanes without a pixel take the next ones of the wave's queue.  The queue's position is
        // wave state: it is advanced here, outside any divergent branch, with every lane of the wave present ----
        for (uint32_t round = 0; round < 4u; round++) {   // (a hand-out spans at most two tiles; the bound is a belt)
            const unsigned long long want = __ballot(state == kWaiting && !have_pixel);
            if (want == 0ull || q_tile == ~0u) break;
            if (q_pos == 64u) {
                q_tile = next_tile(heads, P.tiles_local, queue_round, xcc, lane);
                q_pos = 0u;
                continue;
            }
            const uint32_t rank = (uint32_t)__popcll(want & ((1ull << lane) - 1ull));
            const uint32_t avail = 64u - q_pos;
            if (state == kWaiting && !have_pixel && rank < avail) {
                const uint32_t i = q_pos + rank;
                const uint32_t tile = shard_tile(q_tile, P.shard_first, P.shard_run, P.shard_period);
                px = (tile % P.tiles_x) * 8u + (i & 7u);
                py = (tile / P.tiles_x) * 8u + (i >> 3);
                st.slot = P.tile_major ? q_tile * 64u + i : py * P.width + px;
                create_ray(P, (int)px, (int)py, st.origin, st.dir);
                st.thr = V3{1.0f, 1.0f, 1.0f};
                st.rng = py * P.width + px + P.seed * 0x9E3779B9u;   // sample 0
                sum = V3{0.f, 0.f, 0.f};
                sample = 0u;
                bounce = 0u;
                have_pixel = true;
                new_segment = true;
            }
            const uint32_t n = (uint32_t)__popcll(want);
            q_pos += n < avail ? n : avail;
        }
        // ---- 3. (waiting lanes) the next segment's set-up, or retirement ----
        if (state == kWaiting) {
            if (new_segment) {
                started = segment_begin(P, st.origin, st.dir, m);
                if (started) state = kMarching;   // (a ray that starts outside the world is over at once: stays waiting)
            } else if (!have_pixel && q_tile == ~0u) {
                state = kRetired;   // nothing left to hand out: this lane is done for the frame
            }
        }
        const unsigned long long marching = __ballot(state == kMarching);
        if (marching == 0ull) {
            if (__ballot(state == kWaiting) == 0ull) break;   // every lane retired: the wave is done
            continue;
        }
        // ---- march: all lanes that have a ray, until enough of them are waiting again ----
        for (;;) {
            if (state == kMarching && segment_trip(P, s_liquid, gb, bb, row_bytes, slab_bytes, m)) state = kWaiting;
            const uint32_t n_march = (uint32_t)__popcll(__ballot(state == kMarching));
            const uint32_t n_wait = (uint32_t)__popcll(__ballot(state == kWaiting));
            if (n_march == 0u || n_wait >= refill_at) break;
        }
    }
}

void launch_path_persistent(const FrameParams &P, uint32_t *heads, uint32_t n_cus, hipStream_t st) {
    if (P.tiles_local == 0) return;
    // as many waves as the chip holds at this kernel's register count, but no more than there are tiles
    const uint32_t waves = min(n_cus * 4u * 8u, P.tiles_local);
    static uint32_t refill_at = 0;
    if (!refill_at) {
        const char *e = getenv("VRT_PATH_REFILL");   // experiments: how many waiting lanes end a march phase
        refill_at = e ? (uint32_t)atoi(e) : kRefillAt;
        if (refill_at < 1u || refill_at > 64u) refill_at = kRefillAt;
    }
    hipLaunchKernelGGL(path_persistent_kernel, dim3((waves + 3u) / 4u), dim3(256), 8u * 4u, st, P, heads, refill_at);
}
#endif  // VRT_EXPERIMENTS

// The end of a launch chain of several samples: the frame's running sum plus the chain's planes, in sample order — the
// order the one-sample-per-chain launches add them in and the oracle's — and the division once the last chain is in.
__global__ void path_chain_finish_kernel(Texel *out, const Texel *acc, uint32_t n, uint32_t chain, uint32_t first, uint32_t last, float spp) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint4 t = acc[i];   // the chain's first sample: the frame's first (light and id word as they are), or one more term
    if (!first) {
        const uint4 o = out[i];
        t.x = __float_as_uint(__uint_as_float(o.x) + __uint_as_float(t.x));
        t.y = __float_as_uint(__uint_as_float(o.y) + __uint_as_float(t.y));
        t.z = __float_as_uint(__uint_as_float(o.z) + __uint_as_float(t.z));
        t.w = o.w;
    }
    for (uint32_t s = 1; s < chain; s++) {
        const uint4 a = acc[(size_t)s * n + i];
        t.x = __float_as_uint(__uint_as_float(t.x) + __uint_as_float(a.x));
        t.y = __float_as_uint(__uint_as_float(t.y) + __uint_as_float(a.y));
        t.z = __float_as_uint(__uint_as_float(t.z) + __uint_as_float(a.z));
    }
    if (last) {
        t.x = __float_as_uint(__uint_as_float(t.x) / spp);
        t.y = __float_as_uint(__uint_as_float(t.y) / spp);
        t.z = __float_as_uint(__uint_as_float(t.z) / spp);
    }
    out[i] = t;
}

// rgb /= spp after the last sample
__global__ void path_finish_kernel(Texel *out, uint32_t n, float spp) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint4 t = out[i];
    t.x = __float_as_uint(__uint_as_float(t.x) / spp);
    t.y = __float_as_uint(__uint_as_float(t.y) / spp);
    t.z = __float_as_uint(__uint_as_float(t.z) / spp);
    out[i] = t;
}

static size_t lds_bytes_path(const FrameParams &P, bool lds_roots) { return (24u + (lds_roots ? P.n_roots : 0u)) * 4u; }

// The path trace marches with the grid march when the derived tables exist (P.grid), else with the ancestor-cache walk;
// `literal` (air flagged liquid, vrt_frames.hip) with the shader's text.
#define VRT_PATH_LAUNCH(kernel)                                                                                           \
    do {                                                                                                                  \
        const bool lds = (!P.grid || literal) && P.n_roots <= kLdsRootsMax;                                               \
        const size_t sh = lds_bytes_path(P, lds);                                                                         \
        if (literal) {                                                                                                    \
            if (lds) { if (stats) hipLaunchKernelGGL((kernel<1, true, true>), grid, block, sh, st, P); else hipLaunchKernelGGL((kernel<1, true, false>), grid, block, sh, st, P); } \
            else { if (stats) hipLaunchKernelGGL((kernel<1, false, true>), grid, block, sh, st, P); else hipLaunchKernelGGL((kernel<1, false, false>), grid, block, sh, st, P); } \
        } else if (P.grid) {                                                                                              \
            if (stats) hipLaunchKernelGGL((kernel<0, false, true>), grid, block, sh, st, P);                              \
            else hipLaunchKernelGGL((kernel<0, false, false>), grid, block, sh, st, P);                                   \
        } else if (lds) {                                                                                                 \
            if (stats) hipLaunchKernelGGL((kernel<2, true, true>), grid, block, sh, st, P);                               \
            else hipLaunchKernelGGL((kernel<2, true, false>), grid, block, sh, st, P);                                    \
        } else {                                                                                                          \
            if (stats) hipLaunchKernelGGL((kernel<2, false, true>), grid, block, sh, st, P);                              \
            else hipLaunchKernelGGL((kernel<2, false, false>), grid, block, sh, st, P);                                   \
        }                                                                                                                 \
    } while (0)

void launch_path_primary(const FrameParams &P, bool stats, bool literal, hipStream_t st) {
    if (P.tiles_local == 0) return;
    const dim3 grid((P.tiles_local + 3u) / 4u), block(256);
    if (P.acc) {   // several samples per launch chain: plain frames over the derived tables only (vrt_frames.hip)
        hipLaunchKernelGGL((path_primary_kernel<0, false, false, true>), grid, block, lds_bytes_path(P, false), st, P);
        return;
    }
    VRT_PATH_LAUNCH(path_primary_kernel);
}

#ifdef VRT_EXPERIMENTS
// the primary launch of a frame whose bounce launch is the window kernel (plain frames over the derived tables)
void launch_path_primary_grouped(const FrameParams &P, hipStream_t st) {
    if (P.tiles_local == 0) return;
    const dim3 grid((P.tiles_local + 3u) / 4u), block(256);
    if (P.acc) hipLaunchKernelGGL((path_primary_kernel<0, false, false, true, true>), grid, block, lds_bytes_path(P, false), st, P);
    else hipLaunchKernelGGL((path_primary_kernel<0, false, false, false, true>), grid, block, lds_bytes_path(P, false), st, P);
}

// `continuations`: a launch of the straggler chain (P.path_in = four-plane records: rays a bounce launch handed on and the
// next segments of the chain's own survivors), which marches every ray to its end.
void launch_path_bounce_pool(const FrameParams &P, bool continuations, uint32_t refill_at, uint32_t eject_at, hipStream_t st) {
    if (P.tiles_local == 0) return;
    // refill_at: idle ray slots that send a wave back to its pool (0: the default); eject_at: marching rays at or below which
    // a wave whose pool is dry hands them to the straggler chain (only with a cont_out)
    const uint32_t refill = refill_at >= 1u && refill_at <= 64u * kPoolRays ? refill_at : kPoolRefillAt;
    const uint32_t eject = eject_at <= 64u ? eject_at : kPoolEjectAt;
    const uint32_t parts = (P.in_seg_cap + 4u * kPoolEntries - 1u) / (4u * kPoolEntries);
    const dim3 grid(kHitSegments * parts), block(256);
    const size_t sh = (8u + 4u * kPoolWords) * 4u;
    if (continuations) hipLaunchKernelGGL(path_bounce_pool_kernel<true>, grid, block, sh, st, P, refill, 0u);
    else hipLaunchKernelGGL(path_bounce_pool_kernel<false>, grid, block, sh, st, P, refill, P.cont_out ? eject : 0u);
}
#endif

// the pool kernel over the march cells (P.mblk): `segments` bounce segments in this one launch (every wave carries its own
// survivors from one to the next; P.path_in / P.path_out are the two buffers it goes back and forth between)
void launch_path_bounce_cells(const FrameParams &P, uint32_t refill_at, uint32_t segments, uint32_t lds_pad, hipStream_t st) {
    if (P.tiles_local == 0 || segments == 0) return;
    const uint32_t refill = refill_at >= 1u && refill_at <= 64u ? refill_at : kPoolRefillAt;
    const uint32_t parts = (P.in_seg_cap + 4u * kPoolEntries - 1u) / (4u * kPoolEntries);
    const dim3 grid(kHitSegments * parts), block(256);
    const size_t sh = 8u * 4u + 4u * kCellsPoolBytesPerWave + lds_pad;   // (lds_pad: the occupancy sweep of profiles/r04_path_occupancy_sweep.txt)
    const CellsLaunch L{P, refill, segments};
    if (P.march_direct) hipLaunchKernelGGL(path_bounce_cells_kernel<true>, grid, block, sh, st, L);
    else hipLaunchKernelGGL(path_bounce_cells_kernel<false>, grid, block, sh, st, L);
}

void launch_path_bounce(const FrameParams &P, bool stats, bool literal, hipStream_t st) {
    if (P.tiles_local == 0) return;
    const dim3 grid(kHitSegments * (P.hit_seg_cap / 256u)), block(256);
    VRT_PATH_LAUNCH(path_bounce_kernel);
}
#undef VRT_PATH_LAUNCH

void launch_path_chain_finish(Texel *out, const Texel *acc, uint32_t n, uint32_t chain, bool first, bool last, uint32_t spp, hipStream_t st) {
    if (!n) return;
    hipLaunchKernelGGL(path_chain_finish_kernel, dim3((n + 255u) / 256u), dim3(256), 0, st, out, acc, n, chain, first ? 1u : 0u, last ? 1u : 0u,
                       (float)spp);
}

void launch_path_finish(Texel *out, uint32_t n, uint32_t spp, hipStream_t st) {
    if (!n) return;
    hipLaunchKernelGGL(path_finish_kernel, dim3((n + 255u) / 256u), dim3(256), 0, st, out, n, (float)spp);
}

}  // namespace vrt
