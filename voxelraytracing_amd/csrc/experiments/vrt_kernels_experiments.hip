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


}  // namespace vrt
