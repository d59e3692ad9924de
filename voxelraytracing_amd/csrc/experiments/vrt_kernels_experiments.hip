// experiments/vrt_kernels_experiments.hip — primary + shadow structures that were built, measured and NOT chosen (profiles/DECISIONS.md):
// the persistent grid over per-XCD tile queues (variant 4; north_star's "persistent-threads kernel": 156.8 us against 125.9 for the plain
// launch of the same code) and the tile order for a moving camera (profiles/r04_tile_order_moving.txt).  tools/ab/libvrt_exp.so only.
#include <hip/hip_ext.h>

#include <atomic>

#include "../vrt_tile.h"
#include "../vrt_exp.h"
#include "../vrt_ctx.h"

namespace vrt {

// The same work as a persistent grid (variant 4; north_star's "persistent-threads kernel", kept for the measurement):
// exactly as many workgroups as the chip holds (8 per CU), every wave pulls tiles from the queue of the XCD it runs on
// until that is empty.  Eight queue heads, one per XCD on its own 64-byte line: a single head saturates at ~88 returning
// atomics per microsecond (MI355X_MICROARCH.md "dequeue"), which 32 400 tiles per 0.1 ms frame would exceed.  Tile i of
// XCD x is this context's tile x + 8 i, so neighbouring tiles still spread over the XCDs as the plain launch spreads them.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u; }  // HW_REG_XCC_ID[3:0]

// Next tile of this XCD's queue (wave-uniform): lane 0 takes a ticket, everybody reads it.
__device__ __forceinline__ uint32_t pop_tile(uint32_t *heads, uint32_t xcc, uint32_t lane) {
    uint32_t i = 0;
    if (lane == 0) i = __hip_atomic_fetch_add(&heads[xcc * 16u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return xcc + 8u * (uint32_t)__builtin_amdgcn_readfirstlane((int)i);
}

template <int MARCH, bool LDS_ROOTS>
__global__ void __launch_bounds__(256) primary_shadow_persistent_kernel(FrameParams P, uint32_t *heads, uint32_t max_tiles_per_wave) {
    extern __shared__ uint32_t smem[];
    uint32_t *s_liquid = smem, *s_roots = smem + 24;
    stage_lds(P, s_roots, s_liquid, LDS_ROOTS);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t xcc = xcc_id();
    // Own queue first; once it is empty the wave goes round the other seven, so the frame is complete whatever set of
    // XCDs the grid landed on (a partitioned device or a CU-masked stream shows fewer than eight XCC ids).  Every trip
    // count is bounded by the host (no queue holds more than max_tiles_per_wave - 1 tiles): whatever the queues do, every
    // wave leaves the loops and the grid drains.
    for (uint32_t q = 0; q < 8u; q++) {
        const uint32_t queue = (xcc + q) & 7u;
        uint32_t t_local = pop_tile(heads, queue, lane);
        for (uint32_t k = 0; k < max_tiles_per_wave && t_local < P.tiles_local; k++) {
            MarchResult R, S;
            S.iters = 0; S.visits = 0; S.hit = false;
            trace_tile<MARCH, LDS_ROOTS, false>(P, s_roots, s_liquid, t_local, lane, R, S);
            t_local = pop_tile(heads, queue, lane);  // (taking the next ticket before tracing this tile was slower still: 183 us)
        }
    }
}

// The order for a view that MOVES: made from the frame before, whose trips are this frame's only near where they were
// noted — a silhouette that has moved into a tile the order starts last (it was sky) runs its whole length behind everything
// else (profiles/r02_tile_order_staleness.txt: an exact order one camera step old is 17 % worse than screen order).  So every
// tile takes the largest cost within reach of the image's motion: the maximum over its block of 4 x 4 tiles and the `radius`
// blocks around it (radius 2: 8-11 tiles each way, what the study's r = 8 covers), and the order is made from that.  Two
// launches in front of the four of launch_tile_order; the block maxima live in its scratch until it counts.
__global__ void __launch_bounds__(256) tile_block_max_kernel(const uint32_t *cost, uint32_t tiles_x, uint32_t tiles_y, uint32_t bw, uint32_t blocks, uint32_t *blk) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= blocks) return;
    const uint32_t bx = b % bw, by = b / bw;
    uint32_t m = 0u;
    for (uint32_t y = by * 4u; y < min(by * 4u + 4u, tiles_y); y++)
        for (uint32_t x = bx * 4u; x < min(bx * 4u + 4u, tiles_x); x++) m = max(m, cost[y * tiles_x + x]);
    blk[b] = m;
}
__global__ void __launch_bounds__(256) tile_dilate_kernel(const uint32_t *blk, uint32_t tiles_x, uint32_t n, uint32_t bw, uint32_t bh, uint32_t radius, uint32_t *cost) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const uint32_t bx = (t % tiles_x) / 4u, by = (t / tiles_x) / 4u;
    uint32_t m = 0u;
    for (uint32_t y = by - min(by, radius); y <= min(by + radius, bh - 1u); y++)
        for (uint32_t x = bx - min(bx, radius); x <= min(bx + radius, bw - 1u); x++) m = max(m, blk[y * bw + x]);
    cost[t] = m;
}

// cost: [tiles_x * tiles_y] trips in screen order (overwritten with the dilated ones); scratch as for launch_tile_order
void launch_tile_order_moving(uint32_t *cost, uint32_t tiles_x, uint32_t tiles_y, uint32_t shift, uint32_t radius, uint32_t *scratch, uint32_t *order, hipStream_t st) {
    const uint32_t n = tiles_x * tiles_y;
    if (!n) return;
    const uint32_t bw = (tiles_x + 3u) / 4u, bh = (tiles_y + 3u) / 4u;   // bw * bh <= n: fits the scratch
    hipLaunchKernelGGL(tile_block_max_kernel, dim3((bw * bh + 255u) / 256u), dim3(256), 0, st, (const uint32_t *)cost, tiles_x, tiles_y, bw, bw * bh, scratch);
    hipLaunchKernelGGL(tile_dilate_kernel, dim3((n + 255u) / 256u), dim3(256), 0, st, (const uint32_t *)scratch, tiles_x, n, bw, bh, radius, cost);
    launch_tile_order(cost, n, shift, scratch, order, st);
}

// Variant 4: `heads` = 8 queue heads 64 bytes apart, zeroed by the caller on the same stream.
void launch_primary_shadow_persistent(const FrameParams &P, uint32_t *heads, uint32_t n_cus, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
    hipExtLaunchKernelGGL((primary_shadow_persistent_kernel<0, false>), dim3(n_cus * 8u), dim3(256), 24u * 4u, st, e0, e1, 0, P, heads,
                          (P.tiles_local + 7u) / 8u + 1u);
}


// (as in vrt_kernels.hip)
constexpr uint32_t kCostClasses = 64;
__device__ __forceinline__ uint32_t cost_class(uint32_t trips, uint32_t shift) { return min(trips >> shift, kCostClasses - 1u); }
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t x, uint32_t lane) {
#pragma unroll
    for (uint32_t o = 1; o < 64u; o <<= 1) {
        const uint32_t y = (uint32_t)__shfl_up((int)x, o, 64);
        if (lane >= o) x += y;
    }
    return x;
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
__device__ __forceinline__ uint32_t lanes_below_mask(unsigned long long mask) {   // lanes of the mask below this one
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
#ifdef VRT_EXP_ORDDBG   // (tools/r5/ord_dbg.py: when thread 0 passes each phase, 100 MHz clock)
__device__ unsigned long long g_ord_dbg[8];
extern "C" void vrt_dbg_order(unsigned long long *out) { (void)hipDeviceSynchronize(); (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ord_dbg), 64); }
#define ORD_STAMP(i) do { if (threadIdx.x == 0) g_ord_dbg[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define ORD_STAMP(i) do { } while (0)
#endif
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
    ORD_STAMP(0);
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
    ORD_STAMP(1);
    // dilation — rows, then columns (clamped coordinates: a duplicate does not change a maximum; all reads of a pass in flight) —,
    // the class (descending: class' 0 is the most expensive), and the class's tiles per chunk of 64 blocks
#pragma unroll 1
    for (uint32_t b = tid; b < nb; b += nt) {
        uint32_t bx, by;
        block_at(b, bx, by);
        uint32_t m = 0u;
#pragma unroll
        for (int d = -4; d <= 4; d++) {
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
        for (int d = -4; d <= 4; d++) {
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
    ORD_STAMP(2);
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
    ORD_STAMP(3);
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
    ORD_STAMP(4);
}

// cost: [tiles_x * tiles_y] trips in screen order; false: the frame has more blocks than the kernel's LDS holds (the caller then
// keeps screen order)
bool launch_tile_order_blocks(const uint32_t *cost, uint32_t tiles_x, uint32_t tiles_y, uint32_t shift, uint32_t radius, uint32_t *order, hipStream_t st, uint32_t threads) {
    const uint32_t bw = (tiles_x + 3u) / 4u, bh = (tiles_y + 3u) / 4u, nb = bw * bh;
    if (!nb || nb > kOrderBlocksMax || bw > 255u || bh > 255u || radius > 4u) return false;
    const uint32_t chunks = (nb + 63u) / 64u;
    const size_t lds = ((size_t)3 * nb + (size_t)kCostClasses * chunks) * sizeof(uint32_t);   // 4K: 96 + 32 KiB
    static std::atomic<uint64_t> opted_in{0};   // (> 64 KiB of dynamic LDS needs opting in, once per device)
    if (lds > 48u * 1024u) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        const uint64_t bit = (unsigned)dev < 64u ? 1ull << dev : 0ull;
        if (!(opted_in.load(std::memory_order_relaxed) & bit)) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(tile_order_blocks_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return false;
            opted_in.fetch_or(bit, std::memory_order_relaxed);
        }
    }
    const uint32_t bw_magic = bw > 1u ? (uint32_t)((0x100000000ull + bw - 1u) / bw) : 0u;
    hipLaunchKernelGGL(tile_order_blocks_kernel, dim3(1), dim3(threads >= 64u && threads <= 1024u ? threads & ~63u : 1024u), lds, st, cost, tiles_x, tiles_y, bw, bh, bw_magic, shift, radius, order);
    return true;
}


}  // namespace vrt
