// vrt_kernels.hip — gfx950 kernels of the per-pixel SVO ray-march.
//
// What the kernels compute is fixed by the reference shader
// (clientdesktop/src/graphics/ray_tracer.wgsl: update :173-180, create_ray_from_screen :159-171,
// ray_world :182-316, find_node/find_chunk_node :76-125, ray_color :131-142, ray_sky :144-157).
// How they compute it is not — see DESIGN.md §Kernels.  In short:
//   * one wave64 per 8x8 screen tile (the reference's @workgroup_size(8,8,1) is exactly one CDNA wave) and, since round 6, per workgroup;
//   * the default march (variant 0, vrt_march.h march_grid) does not walk the octree: the leaf under a position comes from
//     a cell grid + brick pool that vrt_accel.hip derives from the node pool on the device (at most two loads, no loop);
//     variants 1 (the shader's text) and 2 (walk resumed below the deepest shared ancestor) read the octree itself;
//   * the step arithmetic is branch-free and algebraically reduced where the reduction is bit-exact;
//   * primary + shadow is one launch with no cooperation between waves: a lane marches its pixel's shadow ray right
//     after its primary ray and stores one 16-byte texel {r,g,b,id} (a wave stores 1 KiB contiguously); the two-launch
//     form with a workgroup-compacted hit buffer in HBM is variant 3;
//   * the 256-bit liquid mask (and, for variants 1-2, the chunk-root table) is staged in LDS.
#include <hip/hip_ext.h>

#include <atomic>

#include "vrt_tile.h"
#include "vrt_path_common.h"   // (vrt_selftest_exact_math: the RNG's logarithm and direction, both forms)
#include "vrt_exp.h"

namespace vrt {


// ------------------------------------------------------------------------------------------------
// Primary rays: one wave per 8x8 tile; with the hit buffer (the two-launch variants) 4 adjacent tiles per 256-thread workgroup,
// without it one tile per workgroup.
// ------------------------------------------------------------------------------------------------
template <int MARCH, bool LDS_ROOTS, bool STATS, bool SHADOW>
__global__ void __launch_bounds__(256) primary_tile_kernel(FrameParams P) {
    extern __shared__ uint32_t smem[];  // [0,8) liquid mask, [8,24) stats scratch + hit count, [24, 24+n_roots) chunk roots
    uint32_t *s_liquid = smem, *s_roots = smem + 24;
    unsigned long long *s_acc = reinterpret_cast<unsigned long long *>(smem + 8);
    uint32_t *s_hits = smem + 22;  // records appended by this workgroup
    if (STATS && threadIdx.x < 6) s_acc[threadIdx.x] = 0ull;
    if (threadIdx.x == 0) *s_hits = 0u;
    stage_lds(P, s_roots, s_liquid, LDS_ROOTS);

    const uint32_t lane = threadIdx.x & 63u;
    uint32_t t_local = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);   // (four tiles a workgroup with the hit buffer, one without)
    const bool live = t_local < P.tiles_local;
    if (!SHADOW && P.tile_order && live) t_local = P.tile_order[t_local];   // longest tiles first (primary-only frames; below)
    MarchResult R;
    R.iters = 0; R.visits = 0; R.hit = false; R.trips = 0;
    if (live) {
        const uint32_t tile = shard_tile(t_local, P.shard_first, P.shard_run, P.shard_period);
        uint32_t px, py;
        tile_pixel(P, tile, lane, px, py);
        const uint32_t slot = P.tile_major ? t_local * 64u + lane : py * P.width + px;

        V3 origin, dir;
        create_ray(P, (int)px, (int)py, origin, dir);
        R = march<MARCH, LDS_ROOTS, STATS, true>(P, s_roots, s_liquid, origin, dir);
        V3 color;
        uint32_t id = shade<MARCH == 1>(P, R, origin, dir, color);

        bool launch = false;
        if (SHADOW) {
            launch = R.hit && R.voxel != 0u && !is_liquid(s_liquid, R.voxel);
            if (launch) id |= VRT_ID_SHADOW_RAY;
        }
        if (SHADOW) P.out[slot] = make_uint4(__float_as_uint(color.x), __float_as_uint(color.y), __float_as_uint(color.z), id);
        else store_pixel(P, slot, px, py, color, id, R);   // (the two-launch shadow kernel read-modify-writes texels: no compact form)

        if (SHADOW) {
            // Compaction of the solid hits into the workgroup's own 256-record slice of the hit buffer: ballot +
            // prefix popcount inside the wave, one LDS atomic per wave across the workgroup's four (adjacent) tiles.
            // Keeping a shadow wave's rays from one 32x8-pixel neighbourhood is worth more than packing every lane:
            // the oracle's step counts give 81 % lane utilisation for tile-ordered records against 69 % when the
            // records of unrelated tiles interleave (what a device-wide atomic cursor produces), and no global
            // atomic is left (one word sustains ~88 returning atomics/us: DESIGN.md §Measured decisions).
            const unsigned long long ballot = __ballot(launch);
            const uint32_t n = (uint32_t)__popcll(ballot);
            if (n) {
                const int leader = __ffsll((long long)ballot) - 1;
                uint32_t base = 0;
                if ((int)lane == leader) base = atomicAdd(s_hits, n);
                base = __shfl(base, leader, 64);
                if (launch) {
                    const uint32_t rank = (uint32_t)__popcll(ballot & ((1ull << lane) - 1ull));
                    const V3 so{R.pos.x + R.norm.x * kShadowBias, R.pos.y + R.norm.y * kShadowBias,
                                R.pos.z + R.norm.z * kShadowBias};
                    P.hits[blockIdx.x * 256u + base + rank] =
                        make_uint4(slot, __float_as_uint(so.x), __float_as_uint(so.y), __float_as_uint(so.z));
                }
            }
        }
        if (STATS && P.steps) P.steps[slot] = R.iters;
        if (!SHADOW && P.tile_cost) {
            uint32_t trips = R.trips;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) trips = max(trips, (uint32_t)__shfl_xor((int)trips, o, 64));
            if (lane == 0) P.tile_cost[t_local] = trips;
        }
    }
    if (STATS) {
        block_add(s_acc, 0, R.iters);
        block_add(s_acc, 1, R.visits);
        block_add(s_acc, 2, R.hit ? 1ull : 0ull);
    }
    if (SHADOW || STATS) {
        __syncthreads();
        if (threadIdx.x == 0) {
            if (SHADOW) P.blk_counts[blockIdx.x] = *s_hits;
            if (STATS) {
                atomicAdd(&P.counters[kCtrSteps], s_acc[0]);
                atomicAdd(&P.counters[kCtrVisits], s_acc[1]);
                atomicAdd(&P.counters[kCtrPrimarySteps], s_acc[0]);
                atomicAdd(&P.counters[kCtrPrimaryVisits], s_acc[1]);
                atomicAdd(&P.counters[kCtrHits], s_acc[2]);
            }
        }
    }
}

// Shadow rays: workgroup b marches the records primary workgroup b appended (lane = record).
template <int MARCH, bool LDS_ROOTS, bool STATS>
__global__ void __launch_bounds__(256) shadow_kernel(FrameParams P) {
    extern __shared__ uint32_t smem[];
    uint32_t *s_liquid = smem, *s_roots = smem + 24;
    unsigned long long *s_acc = reinterpret_cast<unsigned long long *>(smem + 8);
    const uint32_t count = P.blk_counts[blockIdx.x];
    if (!STATS && count == 0u) return;  // uniform: a workgroup of sky tiles
    if (STATS && threadIdx.x < 6) s_acc[threadIdx.x] = 0ull;
    stage_lds(P, s_roots, s_liquid, LDS_ROOTS);

    const bool active = threadIdx.x < count;
    if (!STATS && (threadIdx.x & ~63u) >= count) return;  // whole wave beyond the records
    MarchResult R;
    R.iters = 0; R.visits = 0; R.hit = false;
    uint32_t slot = 0;
    if (active) {
        const uint4 rec = P.hits[blockIdx.x * 256u + threadIdx.x];
        slot = rec.x;
        const V3 so{__uint_as_float(rec.y), __uint_as_float(rec.z), __uint_as_float(rec.w)};
        const V3 sd = normalize_wave(V3{P.sun_local[0] - so.x, P.sun_local[1] - so.y, P.sun_local[2] - so.z});   // (sun_pos - f32(world.min), the host's)
        R = march<MARCH, LDS_ROOTS, STATS>(P, s_roots, s_liquid, so, sd);
        if (R.hit) {
            uint4 t = P.out[slot];
            t.x = __float_as_uint(__uint_as_float(t.x) * kShadowFactor);
            t.y = __float_as_uint(__uint_as_float(t.y) * kShadowFactor);
            t.z = __float_as_uint(__uint_as_float(t.z) * kShadowFactor);
            t.w |= VRT_ID_SHADOWED;
            P.out[slot] = t;
        }
        if (STATS && P.steps) P.steps[slot] |= R.iters << 16;
    }
    if (STATS) {
        block_add(s_acc, 0, active ? R.iters : 0u);
        block_add(s_acc, 1, active ? R.visits : 0u);
        __syncthreads();
        if (threadIdx.x == 0 && s_acc[0]) {
            atomicAdd(&P.counters[kCtrSteps], s_acc[0]);
            atomicAdd(&P.counters[kCtrVisits], s_acc[1]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Primary + shadow in one launch (the default for VRT_MODE_PRIMARY_SHADOW), wave-local: each lane marches its pixel's
// primary ray and, if that ended on a solid voxel, its shadow ray right after, then stores the finished texel once, with
// the shadow factor applied.  No hit records, no barriers, no atomics, no second launch; a shadow wave is exactly one
// 8x8 tile's hits — the most coherent grouping there is — and its lane utilisation is the tile's hit fraction (1.0 for
// the terrain tiles that make up most of a frame).  (A workgroup-phase form — compact the four tiles' hits into an LDS
// hit buffer, lane = record, flags back through LDS — ran in exactly the same time and was dropped: DESIGN.md §5.)
// ------------------------------------------------------------------------------------------------

// PROBE (vrt_render_opts.stats = 2, a diagnostic build of the timed kernel — in the real one no stamp executes): one wave
// in sixteen stamps the shader clock (s_memtime) and the 100 MHz reference (s_memrealtime) around its work and adds the
// differences to P.clock[0], P.clock[1]; their ratio is the clock the march ran at (MI355X_MICROARCH.md, check 6).
template <int MARCH, bool LDS_ROOTS, bool STATS, int WAVES, bool PROBE = false>
__global__ void __launch_bounds__(64 * WAVES) primary_shadow_wave_kernel(FrameParams P) {
    extern __shared__ uint32_t smem[];  // [0,8) liquid mask, [8,24) stats scratch, [24, ...) chunk roots
    uint32_t *s_liquid = smem, *s_roots = smem + 24;
    unsigned long long *s_acc = reinterpret_cast<unsigned long long *>(smem + 8);
    if (STATS && threadIdx.x < 6) s_acc[threadIdx.x] = 0ull;
    unsigned long long t0 = 0, r0 = 0;
    if (PROBE) {
        t0 = __builtin_amdgcn_s_memtime();
        r0 = __builtin_amdgcn_s_memrealtime();
    }
    stage_lds(P, s_roots, s_liquid, LDS_ROOTS);

    const uint32_t lane = threadIdx.x & 63u;
    uint32_t t_local = blockIdx.x * (uint32_t)WAVES + (threadIdx.x >> 6);
    if (P.tile_order && t_local < P.tiles_local) t_local = P.tile_order[t_local];
    MarchResult R, S;
    R.iters = 0; R.visits = 0; R.hit = false;
    S.iters = 0; S.visits = 0; S.hit = false;
    R.trips = 0; S.trips = 0;
    if (t_local < P.tiles_local) {
        trace_tile<MARCH, LDS_ROOTS, STATS>(P, s_roots, s_liquid, t_local, lane, R, S);
        if (P.tile_cost) {   // the wave's trips through its two march loops: what the tile costs, whoever else is on the machine
            uint32_t trips = R.trips + S.trips;   // (S.trips: 0 for a lane without a shadow ray)
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) trips = max(trips, (uint32_t)__shfl_xor((int)trips, o, 64));
            if (lane == 0) P.tile_cost[t_local] = trips;
        }
    }
    if (PROBE) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        if ((blockIdx.x & 3u) == 0u && threadIdx.x == 0 && P.clock) {
            atomicAdd(&P.clock[0], t1 - t0);
            atomicAdd(&P.clock[1], r1 - r0);
        }
    }
    if (STATS) {
        block_add(s_acc, 0, R.iters);
        block_add(s_acc, 1, R.visits);
        block_add(s_acc, 2, R.hit ? 1ull : 0ull);
        block_add(s_acc, 3, S.iters);
        block_add(s_acc, 4, S.visits);
        __syncthreads();
        if (threadIdx.x == 0) {
            atomicAdd(&P.counters[kCtrSteps], s_acc[0] + s_acc[3]);
            atomicAdd(&P.counters[kCtrVisits], s_acc[1] + s_acc[4]);
            atomicAdd(&P.counters[kCtrPrimarySteps], s_acc[0]);
            atomicAdd(&P.counters[kCtrPrimaryVisits], s_acc[1]);
            atomicAdd(&P.counters[kCtrHits], s_acc[2]);
        }
    }
}

// ------------------------------------------------------------------------------------------------

// ------------------------------------------------------------------------------------------------
// Longest tiles first.  The dispatcher hands out workgroups in index order, and a launch ends when its slowest late
// workgroup does: with the tiles in screen order the expensive ones (terrain near the horizon) are spread through the
// launch and the last waves to start are as long as any — a lone launch spends its last fifth with the machine half
// empty.  Ordered by cost, the last waves are the cheapest (longest-processing-time-first scheduling): measured on C2, one
// launch at a time, 115.3 -> 100.6 us per frame with the order taken from the frame before (two frames in flight:
// 94.3 -> 93.7 us — the other frame already filled most of that tail).  A *random* order costs 133 us: consecutive tiles
// share lines of the tables, so the order is a stable counting sort into 64 cost classes — screen order within a class.
// Four small launches behind the frame that noted the trips: per-chunk class counts, the scan (class totals, then the rows),
// the scatter.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kCostClasses = 64;
__device__ __forceinline__ uint32_t cost_class(uint32_t trips, uint32_t shift) { return min(trips >> shift, kCostClasses - 1u); }

// counts[(kCostClasses - 1 - class) * chunks + chunk]: tiles of that class in that chunk of 64 tiles (descending classes first)
__global__ void __launch_bounds__(256) tile_order_count_kernel(const uint32_t *cost, uint32_t n, uint32_t chunks, uint32_t shift, uint32_t *counts) {
    __shared__ uint32_t s_h[4][kCostClasses];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t chunk = blockIdx.x * 4u + wave, t = chunk * 64u + lane;
    s_h[wave][lane] = 0u;
    __syncthreads();
    if (t < n) atomicAdd(&s_h[wave][cost_class(cost[t], shift)], 1u);
    __syncthreads();
    if (chunk < chunks) counts[(kCostClasses - 1u - lane) * chunks + chunk] = s_h[wave][lane];
}

// The exclusive scan of counts[class'][chunk] (class' descending) in two launches of one wave per class: the classes'
// totals, then every class's own chunks behind the totals of the classes before it.  Rows are read 64 consecutive words at
// a time.
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t x, uint32_t lane) {
#pragma unroll
    for (uint32_t o = 1; o < 64u; o <<= 1) {
        const uint32_t y = (uint32_t)__shfl_up((int)x, o, 64);
        if (lane >= o) x += y;
    }
    return x;
}
__global__ void __launch_bounds__(64) tile_order_totals_kernel(const uint32_t *counts, uint32_t chunks, uint32_t *totals) {
    const uint32_t cls = blockIdx.x, lane = threadIdx.x;
    uint32_t sum = 0;
    for (uint32_t i = lane; i < chunks; i += 64u) sum += counts[cls * chunks + i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += (uint32_t)__shfl_xor((int)sum, o, 64);
    if (lane == 0) totals[cls] = sum;
}
__global__ void __launch_bounds__(64) tile_order_scan_kernel(uint32_t *counts, uint32_t chunks, const uint32_t *totals) {
    const uint32_t cls = blockIdx.x, lane = threadIdx.x;
    const uint32_t t = totals[lane];   // kCostClasses == 64 == the wave
    uint32_t carry = (uint32_t)__shfl((int)(wave_inclusive_scan(t, lane) - t), (int)cls, 64);   // the classes before this one
    for (uint32_t i0 = 0; i0 < chunks; i0 += 64u) {
        const uint32_t i = i0 + lane;
        const uint32_t x = i < chunks ? counts[cls * chunks + i] : 0u;
        const uint32_t incl = wave_inclusive_scan(x, lane);
        if (i < chunks) counts[cls * chunks + i] = carry + incl - x;
        carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
}

// order[start of (class, chunk) + rank among the chunk's tiles of that class, in tile order] = tile
__global__ void __launch_bounds__(256) tile_order_scatter_kernel(const uint32_t *cost, uint32_t n, uint32_t chunks, uint32_t shift,
                                                                 const uint32_t *starts, uint32_t *order) {
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t chunk = blockIdx.x * 4u + wave, t = chunk * 64u + lane;
    const bool valid = t < n;
    const uint32_t cls = valid ? cost_class(cost[t], shift) : 0xFFFFFFFFu;
    unsigned long long todo = __ballot(valid);
    while (todo) {   // one round per class present in the wave
        const uint32_t pick = (uint32_t)__builtin_amdgcn_readlane((int)cls, __ffsll((long long)todo) - 1);
        const unsigned long long same = __ballot(cls == pick);
        if (cls == pick) order[starts[(kCostClasses - 1u - pick) * chunks + chunk] + (uint32_t)__popcll(same & ((1ull << lane) - 1ull))] = t;
        todo &= ~same;
    }
}

void launch_tile_order(const uint32_t *cost, uint32_t n, uint32_t shift, uint32_t *scratch, uint32_t *order, hipStream_t st) {
    if (!n) return;
    const uint32_t chunks = (n + 63u) / 64u;
    hipLaunchKernelGGL(tile_order_count_kernel, dim3((chunks + 3u) / 4u), dim3(256), 0, st, cost, n, chunks, shift, scratch);
    uint32_t *totals = scratch + (size_t)kCostClasses * chunks;   // (the scratch holds kCostClasses * (chunks + 1) words)
    hipLaunchKernelGGL(tile_order_totals_kernel, dim3(kCostClasses), dim3(64), 0, st, (const uint32_t *)scratch, chunks, totals);
    hipLaunchKernelGGL(tile_order_scan_kernel, dim3(kCostClasses), dim3(64), 0, st, scratch, chunks, (const uint32_t *)totals);
    hipLaunchKernelGGL(tile_order_scatter_kernel, dim3((chunks + 3u) / 4u), dim3(256), 0, st, cost, n, chunks, shift, (const uint32_t *)scratch, order);
}


// ------------------------------------------------------------------------------------------------
// The order for a view that MOVES, in ONE launch (round 5).  An order made from the frame before holds only near where its
// trips were noted — a silhouette that has moved into a tile the order starts last runs its whole length behind everything
// else (profiles/r02_tile_order_staleness.txt) — so every tile takes the largest cost within reach of the image's motion: the
// maximum over its block of 4 x 4 tiles and the `radius` blocks around it.  Round 4 built that as six small launches, which
// cost the stream 17 us for 6.4 us of shorter frame (profiles/r04_tile_order_moving.txt; experiments/vrt_kernels_experiments.hip
// keeps them).  All tiles of a block have the same dilated cost, so the sort is over BLOCKS (2 040 at 1080p) and fits one
// workgroup's LDS: block maxima, dilation, a stable counting sort of the blocks by cost class weighted with their tile counts,
// then every block's tiles row by row — screen order within a class at block granularity, which keeps what consecutive
// tiles share (a launch's workgroup takes four consecutive entries: one row of a block).
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kOrderBlocksMax = 8192u;   // 4K: 120 x 68 blocks
constexpr uint32_t kOrderRadiusMax = 6u;
__device__ __forceinline__ uint32_t lanes_below_mask(unsigned long long mask) {   // lanes of the mask below this one
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
// (bw_magic = ceil(2^32 / bw): a block's row is __umulhi(b, bw_magic) — a 32-bit division by a run-time value is ~ 25
// instructions, and the first forms of this kernel did four of them per tile: 13 of their 26 us)
__global__ void __launch_bounds__(1024) tile_order_blocks_kernel(const uint32_t *cost, uint32_t tiles_x, uint32_t tiles_y, uint32_t bw, uint32_t bh,
                                                                 uint32_t bw_magic, uint32_t shift, uint32_t radius, uint32_t *order) {
    extern __shared__ uint32_t s_mem[];
    const uint32_t nb = bw * bh, chunks = (nb + 63u) / 64u;
    uint32_t *s_blk = s_mem;                   // [nb] block maxima, then the blocks' classes
    uint32_t *s_dil = s_mem + nb;              // [nb] dilated maxima
    uint32_t *s_row = s_dil + nb;              // [nb] the maxima dilated along their rows
    uint32_t *s_cnt = s_row + nb;              // [chunks][64] tiles of class c' (descending) in chunk k of 64 blocks -> their start
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, nt = blockDim.x;   // (1024 threads alone on the stream, 256 beside a frame)
    auto block_at = [&](uint32_t b, uint32_t &bx, uint32_t &by) __attribute__((always_inline)) {
        by = bw == 1u ? b : __umulhi(b, bw_magic);   // (b < 2^13, bw <= 2^13: exact)
        bx = b - by * bw;
    };
    auto block_tiles = [&](uint32_t b, uint32_t &w, uint32_t &h) __attribute__((always_inline)) {
        uint32_t bx, by;
        block_at(b, bx, by);
        w = min(4u, tiles_x - bx * 4u);
        h = min(4u, tiles_y - by * 4u);
    };
    // block maxima.  A lone workgroup has nobody to hide a load behind: a block's rows are 16 contiguous bytes each (tiles_x a
    // multiple of four: aligned), and the rows of all the blocks a thread has are asked for before the first is used — one
    // round trip instead of thirty-two (the first form of this kernel: 38 us, 25 of them here)
    {
        const bool rows16 = (tiles_x & 3u) == 0u;
#pragma unroll 1
        for (uint32_t b0 = tid; b0 < nb; b0 += 2u * nt) {   // (two blocks a thread and round: 1080p is one round)
            uint4 r[2][4];
#pragma unroll
            for (uint32_t q = 0; q < 2u; q++) {
                const uint32_t b = b0 + q * nt;
                uint32_t bx, by;
                block_at(b, bx, by);
#pragma unroll
                for (uint32_t y = 0; y < 4u; y++) {
                    r[q][y] = make_uint4(0u, 0u, 0u, 0u);
                    const uint32_t ty = by * 4u + y;
                    if (b < nb && ty < tiles_y) {
                        const uint32_t *row = cost + ty * tiles_x + bx * 4u;
                        if (rows16) r[q][y] = *reinterpret_cast<const uint4 *>(row);
                        else {
                            const uint32_t w = min(4u, tiles_x - bx * 4u);
                            r[q][y].x = row[0];
                            if (w > 1u) r[q][y].y = row[1];
                            if (w > 2u) r[q][y].z = row[2];
                            if (w > 3u) r[q][y].w = row[3];
                        }
                    }
                }
            }
#pragma unroll
            for (uint32_t q = 0; q < 2u; q++) {
                const uint32_t b = b0 + q * nt;
                uint32_t m = 0u;
#pragma unroll
                for (uint32_t y = 0; y < 4u; y++) m = max(max(m, max(r[q][y].x, r[q][y].y)), max(r[q][y].z, r[q][y].w));
                if (b < nb) s_blk[b] = m;
            }
        }
    }
    for (uint32_t i = tid; i < kCostClasses * chunks; i += nt) s_cnt[i] = 0u;
    __syncthreads();
    // dilation — rows, then columns (clamped coordinates: a duplicate does not change a maximum; all reads of a pass in flight) —,
    // the class (descending: class' 0 is the most expensive), and the class's tiles per chunk of 64 blocks
#pragma unroll 1
    for (uint32_t b = tid; b < nb; b += nt) {
        uint32_t bx, by;
        block_at(b, bx, by);
        uint32_t m = 0u;
#pragma unroll
        for (int d = -(int)kOrderRadiusMax; d <= (int)kOrderRadiusMax; d++) {
            const int r = max(-(int)radius, min((int)radius, d));
            m = max(m, s_blk[by * bw + (uint32_t)min(max((int)bx + r, 0), (int)bw - 1)]);
        }
        s_row[b] = m;
    }
    __syncthreads();
#pragma unroll 1
    for (uint32_t b = tid; b < nb; b += nt) {
        uint32_t bx, by;
        block_at(b, bx, by);
        uint32_t m = 0u;
#pragma unroll
        for (int d = -(int)kOrderRadiusMax; d <= (int)kOrderRadiusMax; d++) {
            const int r = max(-(int)radius, min((int)radius, d));
            m = max(m, s_row[(uint32_t)min(max((int)by + r, 0), (int)bh - 1) * bw + bx]);
        }
        const uint32_t cls = kCostClasses - 1u - cost_class(m, shift);
        uint32_t w, h;
        block_tiles(b, w, h);
        s_dil[b] = cls;
        atomicAdd(&s_cnt[(b >> 6) * kCostClasses + cls], w * h);   // [chunk][class']: a wave's lanes on consecutive banks below
    }
    __syncthreads();
    // one wave, a lane per class: its tiles, the classes before it (the only cross-lane scan of the kernel), then its chunks'
    // starts — serial over <= 128 chunks, every lane on its own bank (a cross-lane scan per chunk piece was ds_bpermute latency)
    if (wave == 0u) {
        uint32_t total = 0u;
#pragma unroll 8
        for (uint32_t k = 0; k < chunks; k++) total += s_cnt[k * kCostClasses + lane];
        uint32_t carry = wave_inclusive_scan(total, lane) - total;
#pragma unroll 8
        for (uint32_t k = 0; k < chunks; k++) {
            const uint32_t x = s_cnt[k * kCostClasses + lane];
            s_cnt[k * kCostClasses + lane] = carry;
            carry += x;
        }
    }
    __syncthreads();
    // every block's place: the next free tiles of its class in its chunk (one LDS atomic; which of a chunk's blocks of one class
    // comes first is then up to the hardware — any order of the tiles is the same frame, and they are neighbours anyway)
#pragma unroll 1
    for (uint32_t b = tid; b < nb; b += nt) {
        uint32_t bx, by;
        block_at(b, bx, by);
        const uint32_t w = min(4u, tiles_x - bx * 4u), h = min(4u, tiles_y - by * 4u);
        const uint32_t at = atomicAdd(&s_cnt[(b >> 6) * kCostClasses + s_dil[b]], w * h);
        s_blk[b] = at;   // (the maxima and the classes are dead: the block's place and its shape)
        s_dil[b] = bx | (by << 8) | (w << 16) | (h << 20);
    }
    __syncthreads();
    // ... and its tiles row by row: sixteen lanes per block, so that a wave's store is four runs of 64 bytes (a lane per block and
    // sixteen stores each were 64 lines per store instruction: 14 of the first form's 26 us)
#pragma unroll 1
    for (uint32_t t = tid; t < nb * 16u; t += nt) {
        const uint32_t b = t >> 4, i = t & 15u;
        const uint32_t geo = s_dil[b], w = (geo >> 16) & 15u, h = geo >> 20;
        if (i < w * h) {
            const uint32_t y = w == 4u ? i >> 2 : w == 3u ? (i * 11u) >> 5 : w == 2u ? i >> 1 : i, x = i - y * w;   // (i / w for i < 16)
            order[s_blk[b] + i] = (((geo >> 8) & 255u) * 4u + y) * tiles_x + (geo & 255u) * 4u + x;
        }
    }
}

// cost: [tiles_x * tiles_y] trips in screen order; false: the frame has more blocks than the kernel's LDS holds (the caller then
// keeps screen order)
bool launch_tile_order_blocks(const uint32_t *cost, uint32_t tiles_x, uint32_t tiles_y, uint32_t shift, uint32_t radius, uint32_t *order, hipStream_t st, uint32_t threads) {
    const uint32_t bw = (tiles_x + 3u) / 4u, bh = (tiles_y + 3u) / 4u, nb = bw * bh;
    if (!nb || nb > kOrderBlocksMax || bw > 255u || bh > 255u || radius > kOrderRadiusMax) return false;
    const uint32_t chunks = (nb + 63u) / 64u;
    const size_t lds = ((size_t)3 * nb + (size_t)kCostClasses * chunks) * sizeof(uint32_t);   // 4K: 96 + 32 KiB
    // (> 48 KiB of dynamic LDS needs opting in, once per device — with the kernel's MAXIMUM, kOrderBlocksMax blocks, so that a later,
    // larger frame on the same device needs nothing more: 128 KiB of gfx950's 160)
    static std::atomic<uint64_t> opted_in{0};
    if (lds > 48u * 1024u) {
        constexpr size_t kLdsMax = ((size_t)3 * kOrderBlocksMax + (size_t)kCostClasses * (kOrderBlocksMax / 64u)) * sizeof(uint32_t);
        int dev = 0;
        (void)hipGetDevice(&dev);
        const uint64_t bit = (unsigned)dev < 64u ? 1ull << dev : 0ull;
        if (!bit || !(opted_in.load(std::memory_order_relaxed) & bit)) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(tile_order_blocks_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsMax) != hipSuccess) {
                (void)hipGetLastError();
                return false;
            }
            opted_in.fetch_or(bit, std::memory_order_relaxed);
        }
    }
    const uint32_t bw_magic = bw > 1u ? (uint32_t)((0x100000000ull + bw - 1u) / bw) : 0u;
    hipLaunchKernelGGL(tile_order_blocks_kernel, dim3(1), dim3(threads >= 64u && threads <= 1024u ? threads & ~63u : 1024u), lds, st, cost, tiles_x, tiles_y, bw, bh, bw_magic, shift, radius, order);
    return true;
}


// ------------------------------------------------------------------------------------------------
// Output helpers
// ------------------------------------------------------------------------------------------------

// textureStore to rgba8unorm (ray_tracer.wgsl:179): clamp to [0,1], scale by 255, round to nearest.  Texels beyond the
// dispatched workgroups (main.rs:452: tex_size / 8 of them per axis) are never stored to: they keep the fresh texture's zeros,
// alpha included.
__global__ void quantize_rgba8_kernel(const Texel *out, uint8_t *rgba8, uint32_t n, uint32_t w, uint32_t cov_w, uint32_t cov_h) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Texel t = out[i];
    const float c[3] = {__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z)};
    uint32_t q = (i % w < cov_w && i / w < cov_h) ? 0xFF000000u : 0u;
#pragma unroll
    for (int k = 0; k < 3; k++) q |= ((uint32_t)rintf(vclamp(c[k], 0.0f, 1.0f) * 255.0f) & 0xFFu) << (8 * k);
    reinterpret_cast<uint32_t *>(rgba8)[i] = q;
}

// Presentation: fs_main of screen_shader.wgsl:43-65 over the rgba8unorm result texture (ray_tracer.wgsl:179), one lane
// per screen pixel: the sampler's (bilinear) sample of the texture at the pixel centre, crosshair mask, blend, unorm8 store.

// decoded[q] = q / 255 as the sampler decodes an rgba8unorm texel, a table of the workgroup (256 threads, a correctly rounded divide each):
// a sample took twelve such divides — ~ 130 of the general blit's ~ 250 instructions per pixel, 68 us for a 4K window
// one tap of the sampler: the texel as the rgba8unorm texture holds it, decoded (alpha: 1 where the compute pass stored a texel,
// 0 beyond its workgroups — all of it when w and h are multiples of 8).  (Sampling a texture quantised once by its own launch
// instead — 4 bytes a tap, no quantisation per tap — was measured for windows larger than the texture: no faster, the blit is
// bound by its ~ 170 vector instructions per pixel, not by its loads: profiles/r04_present_cost.txt.)
__device__ __forceinline__ void present_tap(const Texel *tex, size_t i, bool covered, const float *decoded, float v[4]) {
    const Texel t = tex[i];
    v[0] = decoded[unorm8(__uint_as_float(t.x))]; v[1] = decoded[unorm8(__uint_as_float(t.y))]; v[2] = decoded[unorm8(__uint_as_float(t.z))];
    v[3] = covered ? 1.0f : 0.0f;
}

__device__ __forceinline__ uint32_t present_pixel(const Texel *tex, uint32_t w, uint32_t h, uint32_t cov_w, uint32_t cov_h, uint32_t screen_w, uint32_t screen_h,
                                                   const vrt_crosshair &ch, uint32_t sx, uint32_t sy, const float *decoded, bool in_box) {
    // (the sample's taps and weights, the crosshair's mask and the blend: vrt_tile.h, shared with the march kernels' own store)
    const PresentSample S = present_sample(w, h, screen_w, screen_h, ch, sx, sy, in_box);
    // alpha: 1 where the compute pass stored a texel, 0 beyond its workgroups (all of it when w and h are multiples of 8)
    float v00[4], v10[4] = {0.f, 0.f, 0.f, 0.f}, v01[4] = {0.f, 0.f, 0.f, 0.f}, v11[4] = {0.f, 0.f, 0.f, 0.f};
    present_tap(tex, (size_t)S.y0 * w + S.x0, (uint32_t)S.x0 < cov_w && (uint32_t)S.y0 < cov_h, decoded, v00);
    if (!(S.a == 0.0f && S.b == 0.0f)) {
        // (a sample at a texel's centre looks at one tap instead of four: every pixel of a window of the texture's size that
        // present_plain_kernel does not take)
        present_tap(tex, (size_t)S.y0 * w + S.x1, (uint32_t)S.x1 < cov_w && (uint32_t)S.y0 < cov_h, decoded, v10);
        present_tap(tex, (size_t)S.y1 * w + S.x0, (uint32_t)S.x0 < cov_w && (uint32_t)S.y1 < cov_h, decoded, v01);
        present_tap(tex, (size_t)S.y1 * w + S.x1, (uint32_t)S.x1 < cov_w && (uint32_t)S.y1 < cov_h, decoded, v11);
    }
    return present_blend(S, ch, v00, v10, v01, v11);
}


__global__ void present_kernel(const Texel *tex, uint32_t w, uint32_t h, uint32_t cov_w, uint32_t cov_h, uint32_t screen_w, uint32_t screen_h,
                               vrt_crosshair ch, uint32_t box_x0, uint32_t box_x1, uint32_t box_y0, uint32_t box_y1, uint8_t *rgba8) {
    __shared__ float s_decoded[256];
    s_decoded[threadIdx.x] = (float)threadIdx.x / 255.0f;   // (256 threads: the launcher's)
    __syncthreads();
    const uint32_t sx = blockIdx.x * blockDim.x + threadIdx.x, sy = blockIdx.y;
    if (sx >= screen_w) return;
    const bool in_box = sx >= box_x0 && sx < box_x1 && sy >= box_y0 && sy < box_y1;
    reinterpret_cast<uint32_t *>(rgba8)[(size_t)sy * screen_w + sx] = present_pixel(tex, w, h, cov_w, cov_h, screen_w, screen_h, ch, sx, sy, s_decoded, in_box);
}

// The blit of a window of the texture's size (the reference keeps its texture at 1080 rows and the window's aspect, main.rs:255-262:
// that is a window 1080 rows tall — full HD, full screen) when the HOST has
// found — in this kernel's own arithmetic, vrt_present.hip: present_is_one_to_one, where the proof is — that every pixel samples
// its own texel's centre to within 1e-4 of a texel: then a pixel outside the crosshair is its texel quantised, and the two IEEE divides,
// the floors and the clamps that only find that out again per pixel (~ 110 instructions, two thirds of the blit's issue slots
// beside a frame that is bound by exactly those) are not executed.  Four pixels per thread: 64 bytes of texels in, 16 bytes out.
// Inside the box around the crosshair (the host's, a pixel wider than the mask can reach) every pixel takes present_pixel.
__global__ void __launch_bounds__(256) present_plain_kernel(const Texel *out, uint32_t w, uint32_t h, uint32_t cov_w, uint32_t cov_h, vrt_crosshair ch,
                                                            uint32_t box_x0, uint32_t box_x1, uint32_t box_y0, uint32_t box_y1, uint8_t *rgba8) {
    __shared__ float s_decoded[256];
    const bool box_row = ch.style != 0u && blockIdx.y >= box_y0 && blockIdx.y < box_y1;   // (uniform)
    if (box_row) {
        s_decoded[threadIdx.x] = (float)threadIdx.x / 255.0f;
        __syncthreads();
    }
    const uint32_t x4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4u, sy = blockIdx.y;   // (w % 4 == 0: the launcher's condition)
    if (x4 >= w) return;
    uint32_t q[4];
    if (box_row && x4 + 4u > box_x0 && x4 < box_x1) {
#pragma unroll
        for (uint32_t k = 0; k < 4u; k++) q[k] = present_pixel(out, w, h, cov_w, cov_h, w, h, ch, x4 + k, sy, s_decoded, true);
    } else {
        Texel t[4];
#pragma unroll
        for (uint32_t k = 0; k < 4u; k++) t[k] = out[(size_t)sy * w + x4 + k];
#pragma unroll
        for (uint32_t k = 0; k < 4u; k++)
            q[k] = (x4 + k < cov_w && sy < cov_h ? 0xFF000000u : 0u) | unorm8(__uint_as_float(t[k].x)) | (unorm8(__uint_as_float(t[k].y)) << 8) |
                   (unorm8(__uint_as_float(t[k].z)) << 16);
    }
    reinterpret_cast<uint4 *>(rgba8)[((size_t)sy * w + x4) / 4u] = make_uint4(q[0], q[1], q[2], q[3]);
}

// Gather root: tile-major [rank][slots_per_rank] texels -> row-major frame of texels.  Tiles are dealt out in periods
// of `period` = root_weight + N - 1 (vrt_config); skip_root: the root rendered its own tiles in place.
__global__ void assemble_kernel(const Texel *gathered, Texel *dst, uint32_t width, uint32_t tiles_x, uint32_t tiles_total,
                                uint32_t root_weight, uint32_t period, uint32_t skip_root, uint64_t rank_stride) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t tile = gid >> 6, p = gid & 63u;
    if (tile >= tiles_total) return;
    const uint32_t q = tile / period, r = tile % period;
    uint32_t rank, t_local;
    if (r < root_weight) {
        if (skip_root) return;
        rank = 0u;
        t_local = q * root_weight + r;
    } else {
        rank = r - root_weight + 1u;
        t_local = q;
    }
    const uint32_t px = (tile % tiles_x) * 8u + (p & 7u), py = (tile / tiles_x) * 8u + (p >> 3);
    // the messages may have been stored by another device (a multi-device context's peers write them over xGMI, RCCL's
    // gather likewise): read them past this device's caches
    const unsigned long long *src = reinterpret_cast<const unsigned long long *>(gathered + rank * rank_stride + (uint64_t)t_local * 64u + p);
    const unsigned long long lo = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long hi = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    store_streaming(&dst[py * width + px], make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32)));   // (the frame: vrt_device.h)
}

// Gather root, compact messages: the other ranks sent 8 bytes per pixel {id word | kIdNormYNeg, water_dist}; the root
// re-creates the pixel's ray for the sky of a miss, runs the same shade() the sender would have run on the same
// operands, applies the shadow factor and writes the texel at its row-major position.  Bit-identical to the texel the
// sender would have stored (tests: the assembled frame equals the unsharded one).
__global__ void assemble_shade_kernel(FrameParams P, const uint2 *gathered, Texel *dst, uint32_t root_weight, uint32_t period,
                                      uint64_t rank_stride) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t tile = gid >> 6, p = gid & 63u;
    if (tile >= P.tiles_total) return;
    const uint32_t q = tile / period, r = tile % period;
    if (r < root_weight) return;  // the root's own tiles are already in dst
    const uint32_t rank = r - root_weight + 1u;
    // (read past this device's caches: another device may have stored the record, see assemble_kernel)
    const unsigned long long rw = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(gathered + rank * rank_stride + (uint64_t)q * 64u + p),
                                                    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const uint2 rec = make_uint2((uint32_t)rw, (uint32_t)(rw >> 32));
    const uint32_t px = (tile % P.tiles_x) * 8u + (p & 7u), py = (tile / P.tiles_x) * 8u + (p >> 3);

    MarchResult R;
    R.hit = (rec.x & VRT_ID_HIT) != 0u;
    R.voxel = rec.x & VRT_ID_VOXEL_MASK;
    R.norm = V3{(rec.x & VRT_ID_NX) ? 1.0f : 0.0f, (rec.x & VRT_ID_NY) ? ((rec.x & kIdNormYNeg) ? -1.0f : 1.0f) : 0.0f,
                (rec.x & VRT_ID_NZ) ? 1.0f : 0.0f};  // shade() only asks norm.x != 0, norm.z != 0, norm.y == -1
    R.water_dist = __uint_as_float(rec.y);
    R.pos = V3{0.f, 0.f, 0.f};
    R.iters = 0u;
    R.visits = 0u;
    V3 origin, dir, color;
    create_ray(P, (int)px, (int)py, origin, dir);
    uint32_t id = shade<false>(P, R, origin, dir, color);
    id |= rec.x & (VRT_ID_SHADOW_RAY | VRT_ID_SHADOWED);
    if (rec.x & VRT_ID_SHADOWED) {
        color.x *= kShadowFactor;
        color.y *= kShadowFactor;
        color.z *= kShadowFactor;
    }
    // (the frame's texels, non-temporal like the march's own: vrt_device.h — the root's assembly 17.0 -> 16.1 us in the N = 8 rehearsal)
    store_streaming(&dst[py * P.width + px], make_uint4(__float_as_uint(color.x), __float_as_uint(color.y), __float_as_uint(color.z), id));
}

// ------------------------------------------------------------------------------------------------
// Launchers (called from vrt_frames.hip, vrt_present.hip, vrt_group.hip)
// ------------------------------------------------------------------------------------------------

static size_t lds_bytes(const FrameParams &P, bool lds_roots) { return (24u + (lds_roots ? P.n_roots : 0u)) * 4u; }

template <int MARCH, bool LDS_ROOTS>
static void launch_primary_t(const FrameParams &P, bool stats, bool shadow, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
    // with the hit buffer a workgroup is four adjacent tiles (their records share its 256-record slice); primary rays alone: one tile =
    // one wave = one workgroup, like the one-launch kernel (launch_fused_t)
    const dim3 grid(shadow ? (P.tiles_local + 3u) / 4u : P.tiles_local), block(shadow ? 256 : 64);
    const size_t lds = lds_bytes(P, LDS_ROOTS);
    if (stats) {
        if (shadow) hipExtLaunchKernelGGL((primary_tile_kernel<MARCH, LDS_ROOTS, true, true>), grid, block, (uint32_t)lds, st, e0, e1, 0, P);
        else hipExtLaunchKernelGGL((primary_tile_kernel<MARCH, LDS_ROOTS, true, false>), grid, block, (uint32_t)lds, st, e0, e1, 0, P);
    } else {
        if (shadow) hipExtLaunchKernelGGL((primary_tile_kernel<MARCH, LDS_ROOTS, false, true>), grid, block, (uint32_t)lds, st, e0, e1, 0, P);
        else hipExtLaunchKernelGGL((primary_tile_kernel<MARCH, LDS_ROOTS, false, false>), grid, block, (uint32_t)lds, st, e0, e1, 0, P);
    }
}

template <int MARCH, bool LDS_ROOTS>
static void launch_shadow_t(const FrameParams &P, bool stats, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
    const dim3 grid((P.tiles_local + 3u) / 4u), block(256);  // one workgroup per primary workgroup
    const size_t lds = lds_bytes(P, LDS_ROOTS);
    if (stats) hipExtLaunchKernelGGL((shadow_kernel<MARCH, LDS_ROOTS, true>), grid, block, (uint32_t)lds, st, e0, e1, 0, P);
    else hipExtLaunchKernelGGL((shadow_kernel<MARCH, LDS_ROOTS, false>), grid, block, (uint32_t)lds, st, e0, e1, 0, P);
}

// variant 0: grid march over the derived cell grid / brick pool (needs P.grid), primary + shadow fused into one launch;
// 1: literal octree walk; 2: ancestor-cache octree walk; 3: grid march, shadow rays as a second launch from a hit buffer
// in HBM (the wavefront form the path trace is built from)
bool variant_supported(uint32_t variant) { return variant <= 3u; }


// One launch for primary + shadow; blk_counts gets one launched-ray count per tile.  march 0 = the grid march (variant 0);
// march 2 = the ancestor-cache octree walk, for contexts whose pixels are 8-byte records (VRT_FLAG_COMPACT: the
// two-launch kernels store and re-read whole texels) when the world is too large for the derived tables.

template <int MARCH, bool LDS_ROOTS>
static void launch_fused_t(const FrameParams &P, bool stats, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
    // ONE tile = one wave = one workgroup (round 6; four until then).  The waves do not cooperate, and a workgroup of four holds its
    // four wave slots until its slowest tile is done and is placed only where four slots are free at once: with single-wave workgroups
    // the dispatcher fills every slot as it frees.  Same box, alternating runs (profiles/r06_waves_ab.txt): C2 43.9 -> 45.3 Grays/s,
    // standing camera 43.0 -> 44.6, one frame at a time 40.3 -> 40.9 (orbit 35.9 -> 36.7), C3's shape 49.6 -> 50.6, the client's frame with
    // two in flight 91.5 -> 87.1 us (one at a time 110.5 -> 111.7: its four neighbouring tiles no longer share a CU's L1); two waves a
    // workgroup lie between, eight are worse than four (42.0).  (Every XCD taking whole tile rows — workgroup b runs on XCD b % 8 — on top
    // of it: - 1 % on C2 and C3's shape, - 2 % of the client's frame: not kept.)
    constexpr int WAVES = 1;
    const dim3 grid((P.tiles_local + WAVES - 1u) / WAVES), block(64 * WAVES);
    const uint32_t lds = (uint32_t)lds_bytes(P, LDS_ROOTS);
    if (P.clock && !stats) hipExtLaunchKernelGGL((primary_shadow_wave_kernel<MARCH, LDS_ROOTS, false, WAVES, true>), grid, block, lds, st, e0, e1, 0, P);
    else if (stats) hipExtLaunchKernelGGL((primary_shadow_wave_kernel<MARCH, LDS_ROOTS, true, WAVES>), grid, block, lds, st, e0, e1, 0, P);
    else hipExtLaunchKernelGGL((primary_shadow_wave_kernel<MARCH, LDS_ROOTS, false, WAVES>), grid, block, lds, st, e0, e1, 0, P);
}

void launch_primary_shadow_fused(const FrameParams &P, uint32_t march, bool stats, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
    if (march == 2u) {
        if (P.n_roots <= kLdsRootsMax) launch_fused_t<2, true>(P, stats, st, e0, e1);
        else launch_fused_t<2, false>(P, stats, st, e0, e1);
    } else {
        launch_fused_t<0, false>(P, stats, st, e0, e1);
    }
}

// e0 / e1: events the dispatch itself stamps with the kernel's begin and end (no separate marker packets on the stream)
void launch_primary(const FrameParams &P, uint32_t variant, bool stats, bool shadow, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
    const bool lds = P.n_roots <= kLdsRootsMax;
    if (variant == 1u) {
        if (lds) launch_primary_t<1, true>(P, stats, shadow, st, e0, e1);
        else launch_primary_t<1, false>(P, stats, shadow, st, e0, e1);
    } else if (variant == 2u) {
        if (lds) launch_primary_t<2, true>(P, stats, shadow, st, e0, e1);
        else launch_primary_t<2, false>(P, stats, shadow, st, e0, e1);
    } else {
        launch_primary_t<0, false>(P, stats, shadow, st, e0, e1);
    }
}

void launch_shadow(const FrameParams &P, uint32_t variant, bool stats, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
    const bool lds = P.n_roots <= kLdsRootsMax;
    if (variant == 1u) {
        if (lds) launch_shadow_t<1, true>(P, stats, st, e0, e1);
        else launch_shadow_t<1, false>(P, stats, st, e0, e1);
    } else if (variant == 2u) {
        if (lds) launch_shadow_t<2, true>(P, stats, st, e0, e1);
        else launch_shadow_t<2, false>(P, stats, st, e0, e1);
    } else {
        launch_shadow_t<0, false>(P, stats, st, e0, e1);
    }
}

void launch_quantize(const Texel *out, uint8_t *rgba8, uint32_t w, uint32_t h, hipStream_t st) {
    const uint32_t n = w * h;
    if (!n) return;
    hipLaunchKernelGGL(quantize_rgba8_kernel, dim3((n + 255u) / 256u), dim3(256), 0, st, out, rgba8, n, w, w & ~7u, h & ~7u);
}

void launch_assemble_shade(const FrameParams &P, const void *gathered, Texel *dst, uint32_t root_weight, uint32_t period,
                           uint64_t rank_stride, hipStream_t st) {
    if (!P.tiles_total) return;
    hipLaunchKernelGGL(assemble_shade_kernel, dim3((P.tiles_total * 64u + 255u) / 256u), dim3(256), 0, st, P,
                       (const uint2 *)gathered, dst, root_weight, period, rank_stride);
}

// one_to_one: the host's finding for this pair of sizes (screen == texture, every sample at its own texel's centre); box: the
// pixels the crosshair's mask can reach, [x0, x1) x [y0, y1)
void launch_present(const Texel *out, uint32_t w, uint32_t h, uint32_t screen_w, uint32_t screen_h, const vrt_crosshair &ch,
                    uint8_t *rgba8, bool one_to_one, const uint32_t box[4], hipStream_t st) {
    if (one_to_one && screen_w == w && screen_h == h && w % 4u == 0u) {
        hipLaunchKernelGGL(present_plain_kernel, dim3((w / 4u + 255u) / 256u, h), dim3(256), 0, st, out, w, h, w & ~7u, h & ~7u, ch, box[0], box[1], box[2], box[3], rgba8);
        return;
    }
    hipLaunchKernelGGL(present_kernel, dim3((screen_w + 255u) / 256u, screen_h), dim3(256), 0, st, out, w, h, w & ~7u, h & ~7u, screen_w,
                       screen_h, ch, box[0], box[1], box[2], box[3], rgba8);
}

void launch_assemble(const Texel *gathered, Texel *dst, uint32_t width, uint32_t tiles_x, uint32_t tiles_total,
                     uint32_t root_weight, uint32_t period, bool skip_root, uint64_t rank_stride, hipStream_t st) {
    if (!tiles_total) return;
    hipLaunchKernelGGL(assemble_kernel, dim3((tiles_total * 64u + 255u) / 256u), dim3(256), 0, st, gathered, dst, width,
                       tiles_x, tiles_total, root_weight, period, skip_root ? 1u : 0u, rank_stride);
}

// ------------------------------------------------------------------------------------------------
// vrt_selftest_exact_math: the banded division / square root of vrt_march.h against the compiler's general sequences
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t st_hash(uint32_t x) {
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}
// a float with a random sign and mantissa and an exponent field in [lo, hi]
__device__ __forceinline__ float st_float(uint32_t h, uint32_t lo, uint32_t hi) {
    const uint32_t e = lo + (h >> 9) % (hi - lo + 1u);
    return __uint_as_float((h & 0x807FFFFFu) | (e << 23));
}
__device__ __noinline__ float st_div_general(float n, float d) { return n / d; }
__device__ __noinline__ float st_sqrt_general(float x) { return sqrtf(x); }
// rng_next_dir (vrt_path_common.h) as the text has it: the general division inside the logarithm, sqrtf, the general normalise
__device__ __forceinline__ float st_rng_norm_general(uint32_t &state) {
    const float u1 = rng_next(state);
    float u2 = rng_next(state);
    if (u2 < 1.0e-10f) u2 = 1.0e-10f;
    return st_sqrt_general(-2.0f * vlog_t<false>(u2)) * vcos2pi(u1);
}
__device__ __forceinline__ V3 st_rng_dir_general(uint32_t &state) {
    const float x = st_rng_norm_general(state);
    const float y = st_rng_norm_general(state);
    const float z = st_rng_norm_general(state);
    const float len = st_sqrt_general(x * x + y * y + z * z);
    return V3{st_div_general(x, len), st_div_general(y, len), st_div_general(z, len)};
}

__global__ void selftest_exact_math_kernel(uint32_t n, uint32_t seed, unsigned long long *mismatches) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t bad = 0u;
    if (i < n) {
        const uint32_t h0 = st_hash(i * 4u + seed), h1 = st_hash(i * 4u + 1u + seed * 0x9E3779B9u), h2 = st_hash(i * 4u + 2u + seed), h3 = st_hash(i * 4u + 3u + seed);
        // exponent fields 97 .. 157 = the band [2^-30, 2^31); every 16th operand set sits on the band's edges
        const bool edge = (i & 15u) == 0u;
        const V3 v{st_float(h0, edge ? 97u : 97u, edge ? 97u : 157u), st_float(h1, edge ? 157u : 97u, 157u), st_float(h2, 97u, 157u)};
        // one division, bit for bit
        const float q_fast = div_refined(v.x, v.y, rcp_refined(v.y)), q_gen = st_div_general(v.x, v.y);
        bad += __float_as_uint(q_fast) != __float_as_uint(q_gen);
        // square root over its whole claimed range: exponent fields 31 (2^-96) .. 254
        const float x = fabsf(st_float(h3, 31u, 254u));
        bad += __float_as_uint(sqrt_banded(x)) != __float_as_uint(st_sqrt_general(x));
        // the two composites the kernels use, against their plain texts
        const V3 u = unit_steps(v);
        const V3 ug{fabsf(st_sqrt_general(1.0f + st_div_general(v.y, v.x) * st_div_general(v.y, v.x) + st_div_general(v.z, v.x) * st_div_general(v.z, v.x))),
                    fabsf(st_sqrt_general(1.0f + st_div_general(v.x, v.y) * st_div_general(v.x, v.y) + st_div_general(v.z, v.y) * st_div_general(v.z, v.y))),
                    fabsf(st_sqrt_general(1.0f + st_div_general(v.x, v.z) * st_div_general(v.x, v.z) + st_div_general(v.y, v.z) * st_div_general(v.y, v.z)))};
        bad += __float_as_uint(u.x) != __float_as_uint(ug.x) || __float_as_uint(u.y) != __float_as_uint(ug.y) || __float_as_uint(u.z) != __float_as_uint(ug.z);
        const V3 nn = normalize_wave(v);
        const float len = st_sqrt_general(vdot(v, v));
        const V3 ng{st_div_general(v.x, len), st_div_general(v.y, len), st_div_general(v.z, len)};
        bad += __float_as_uint(nn.x) != __float_as_uint(ng.x) || __float_as_uint(nn.y) != __float_as_uint(ng.y) || __float_as_uint(nn.z) != __float_as_uint(ng.z);
        // the path trace's logarithm, its division with and without the scaffolding: the first 2^23 indices run through every
        // mantissa (the division sees nothing else of x), the rest through the exponents a draw can have (1e-10 .. 1)
        const float xl = i < (1u << 23) ? __uint_as_float(0x3F000000u | i) : fabsf(st_float(h1, 93u, 126u));
        bad += __float_as_uint(vlog_t<true>(xl)) != __float_as_uint(vlog_t<false>(xl));
        bad += __float_as_uint(vlog_t<true>(1.0f)) != __float_as_uint(vlog_t<false>(1.0f));
        // ... and a whole direction draw from a random state
        uint32_t sa = h2, sb = h2;
        const V3 da = rng_next_dir(sa), db = st_rng_dir_general(sb);
        bad += sa != sb || __float_as_uint(da.x) != __float_as_uint(db.x) || __float_as_uint(da.y) != __float_as_uint(db.y) || __float_as_uint(da.z) != __float_as_uint(db.z);
    }
    if (bad) atomicAdd(mismatches, (unsigned long long)bad);
}

}  // namespace vrt

extern "C" int vrt_selftest_exact_math(int32_t device, uint32_t n, uint32_t seed, uint64_t *mismatches) {
    if (!mismatches || n == 0u) return VRT_ERR_INVALID_ARG;
    if (hipSetDevice(device) != hipSuccess) return VRT_ERR_DEVICE;
    unsigned long long *d = nullptr;
    if (hipMalloc(&d, sizeof *d) != hipSuccess) return VRT_ERR_DEVICE;
    int rc = VRT_OK;
    unsigned long long h = 0;
    if (hipMemset(d, 0, sizeof *d) != hipSuccess) rc = VRT_ERR_DEVICE;
    if (rc == VRT_OK) {
        hipLaunchKernelGGL(vrt::selftest_exact_math_kernel, dim3((n + 255u) / 256u), dim3(256), 0, nullptr, n, seed, d);
        if (hipGetLastError() != hipSuccess || hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) rc = VRT_ERR_DEVICE;
    }
    (void)hipFree(d);
    *mismatches = h;
    return rc;
}
