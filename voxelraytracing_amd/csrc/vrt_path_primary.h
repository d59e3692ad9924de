// vrt_path_primary.h — bounce 0 of the path trace (path_primary_kernel) for vrt_path.hip and, in its GROUPED form, for the window
// launch of experiments/vrt_path_window.hip.
#pragma once

#include "vrt_path_common.h"

namespace vrt {

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
        uint32_t px, py;
        tile_pixel(P, tile, lane, px, py);
        const uint32_t pixel_slot = P.tile_major ? t_local * 64u + lane : py * P.width + px;
        V3 origin, dir;
        create_ray(P, (int)px, (int)py, origin, dir);
        // every sample of a pixel starts with the same ray (the samples differ from their first bounce on: the RNG is not
        // asked before a hit), so the primary segment is marched once for all the samples of this launch chain
        R = march<MARCH, LDS_ROOTS, STATS, true>(P, s_roots, s_liquid, origin, dir);
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

static inline size_t lds_bytes_path(const FrameParams &P, bool lds_roots) { return (24u + (lds_roots ? P.n_roots : 0u)) * 4u; }

}  // namespace vrt
