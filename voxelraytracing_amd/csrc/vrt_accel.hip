// vrt_accel.hip — device-side re-layout of the reference's node pool into the march's lookup structure.
//
// The reference finds the leaf under a position by walking the chunk's octree from its root on every march
// step (find_node / find_chunk_node, ray_tracer.wgsl:76-125: up to 1 chunk_roots load + 6 dependent 2-byte
// node loads).  The backend owns the device layout (SURVEY.md §7 "may re-lay-out on upload"), so it keeps the
// pool byte-identical to the host's (uploads stay range writes) and *derives* from it, on the GPU, a two-level
// table that answers the same question in at most two loads and no loop:
//
//   cell grid   u32[(8S)^3], x-major over the whole world: one entry per depth-3 octree cell (4^3 voxels).
//               leaf at depth d <= 3 covering the cell ->  voxel | (32 >> d) << 15      (bit 31 clear)
//               split at depth 3                        ->  0x80000000 | brick * 64
//   brick pool  u16[bricks][64], index (x&3) | (y&3) << 2 | (z&3) << 4 inside the cell:
//               leaf at depth 4 -> voxel | 0x8000 (size 2), leaf at depth 5 -> voxel (size 1)
//
// Every entry is exactly what find_node would return for any position inside it (same node word, same depth;
// a node read past the end of the pool is 0, a missing chunk is root 0), so the march visits the same leaves
// and produces the same frame bit for bit; variant 1 (the literal walk) and variant 2 (ancestor cache) still
// read the octree itself and are compared against it in the tests.  Rebuilt lazily before the next frame
// whenever nodes, chunk_roots or the world size changed; never inside the kernels' timed region.
#include "vrt_device.h"

namespace vrt {

namespace {

__device__ __forceinline__ uint32_t pool_node(const uint16_t *nodes, uint32_t n_nodes, uint32_t idx) {
    return idx < n_nodes ? (uint32_t)nodes[idx] : 0u;  // past the end: an air leaf (what the march's buffer loads return)
}

// Walks the three levels above a cell. Returns the node word at the stop depth (<= 3) and that depth.
__device__ __forceinline__ uint32_t descend3(const uint16_t *nodes, uint32_t n_nodes, uint32_t root, uint32_t cx, uint32_t cy,
                                             uint32_t cz, uint32_t &depth) {
    uint32_t node = pool_node(nodes, n_nodes, root);
    depth = 0u;
    while ((node & 0x8000u) && depth < 3u) {
        const uint32_t sh = 2u - depth;  // cell coordinates carry bits 4..2 of the voxel coordinate
        const uint32_t sel = ((cx >> sh) & 1u) | (((cy >> sh) & 1u) << 1) | (((cz >> sh) & 1u) << 2);
        node = pool_node(nodes, n_nodes, root + (node & 0x7FFFu) + sel);
        depth += 1u;
    }
    return node;
}

// One workgroup per chunk slot, one thread per depth-3 cell. Writes leaf entries, ranks the split cells
// inside the chunk (their bricks are laid out contiguously per chunk, cells in x-major order) and the
// chunk's brick count.
__global__ void __launch_bounds__(512) accel_cells_kernel(const uint16_t *nodes, uint32_t n_nodes, const uint32_t *roots,
                                                          uint32_t S, uint32_t *grid, uint32_t *chunk_bricks) {
    __shared__ uint32_t s_wave[8];
    const uint32_t chunk = blockIdx.x;
    const uint32_t t = threadIdx.x, cx = t & 7u, cy = (t >> 3) & 7u, cz = t >> 6;
    const uint32_t root = roots[chunk];
    uint32_t depth;
    const uint32_t node = descend3(nodes, n_nodes, root, cx, cy, cz, depth);
    const bool split = (node & 0x8000u) != 0u;  // only possible at depth 3

    const unsigned long long ballot = __ballot(split);
    const uint32_t lane = t & 63u, wave = t >> 6;
    if (lane == 0) s_wave[wave] = (uint32_t)__popcll(ballot);
    __syncthreads();
    uint32_t before = 0, total = 0;
    for (uint32_t w = 0; w < 8u; w++) {
        if (w < wave) before += s_wave[w];
        total += s_wave[w];
    }
    const uint32_t rank = before + (uint32_t)__popcll(ballot & ((1ull << lane) - 1ull));

    const uint32_t G = S * 8u;
    const uint32_t chx = chunk % S, chy = (chunk / S) % S, chz = chunk / (S * S);
    const size_t cell = ((size_t)(chz * 8u + cz) * G + (chy * 8u + cy)) * G + (chx * 8u + cx);
    grid[cell] = split ? (0x80000000u | rank) : ((node & 0x7FFFu) | ((32u >> depth) << 15));
    if (t == 0) chunk_bricks[chunk] = total;
}

// Exclusive scan of the per-chunk brick counts (one workgroup; S^3 <= 10^6 entries, off the frame path).
__global__ void __launch_bounds__(1024) accel_scan_kernel(const uint32_t *counts, uint32_t *offsets, uint32_t n, uint32_t *total) {
    __shared__ uint32_t s_part[1024];
    const uint32_t t = threadIdx.x;
    const uint32_t per = (n + 1023u) / 1024u;
    const uint32_t lo = min(t * per, n), hi = min(lo + per, n);
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; i++) sum += counts[i];
    s_part[t] = sum;
    __syncthreads();
    for (uint32_t o = 1; o < 1024u; o <<= 1) {  // Hillis-Steele inclusive scan of the partials
        const uint32_t v = t >= o ? s_part[t - o] : 0u;
        __syncthreads();
        s_part[t] += v;
        __syncthreads();
    }
    uint32_t run = s_part[t] - sum;
    for (uint32_t i = lo; i < hi; i++) {
        offsets[i] = run;
        run += counts[i];
    }
    if (t == 1023u) *total = s_part[1023];
}

// Fills the bricks of the split cells and replaces their in-chunk rank by the pool position.
__global__ void __launch_bounds__(512) accel_bricks_kernel(const uint16_t *nodes, uint32_t n_nodes, const uint32_t *roots,
                                                           uint32_t S, uint32_t *grid, const uint32_t *chunk_offsets,
                                                           uint16_t *bricks, uint32_t brick_cap) {
    const uint32_t chunk = blockIdx.x;
    const uint32_t t = threadIdx.x, cx = t & 7u, cy = (t >> 3) & 7u, cz = t >> 6;
    const uint32_t G = S * 8u;
    const uint32_t chx = chunk % S, chy = (chunk / S) % S, chz = chunk / (S * S);
    const size_t cell = ((size_t)(chz * 8u + cz) * G + (chy * 8u + cy)) * G + (chx * 8u + cx);
    const uint32_t e = grid[cell];
    if (!(e & 0x80000000u)) return;
    const uint32_t brick = chunk_offsets[chunk] + (e & 0x7FFFFFFFu);
    if (brick >= brick_cap) return;  // cannot happen: the pool was sized from the scan's total
    const uint32_t root = roots[chunk];
    uint32_t depth;
    const uint32_t n3 = descend3(nodes, n_nodes, root, cx, cy, cz, depth);
    uint16_t *b = bricks + (size_t)brick * 64u;
    for (uint32_t c = 0; c < 8u; c++) {
        const uint32_t n4 = pool_node(nodes, n_nodes, root + (n3 & 0x7FFFu) + c);
        const uint32_t x1 = (c & 1u) * 2u, y1 = ((c >> 1) & 1u) * 2u, z1 = ((c >> 2) & 1u) * 2u;
        for (uint32_t g = 0; g < 8u; g++) {
            const uint32_t x = x1 + (g & 1u), y = y1 + ((g >> 1) & 1u), z = z1 + ((g >> 2) & 1u);
            uint32_t word;
            if (n4 & 0x8000u) word = pool_node(nodes, n_nodes, root + (n4 & 0x7FFFu) + g) & 0x7FFFu;  // depth 5: the walk stops here
            else word = (n4 & 0x7FFFu) | 0x8000u;                                                      // depth-4 leaf, size 2
            b[x | (y << 2) | (z << 4)] = (uint16_t)word;
        }
    }
    grid[cell] = 0x80000000u | (brick * 64u);
}

}  // namespace

void launch_accel_cells(const uint16_t *nodes, uint32_t n_nodes, const uint32_t *roots, uint32_t S, uint32_t *grid,
                        uint32_t *chunk_bricks, uint32_t *chunk_offsets, uint32_t *total, hipStream_t st) {
    const uint32_t n = S * S * S;
    hipLaunchKernelGGL(accel_cells_kernel, dim3(n), dim3(512), 0, st, nodes, n_nodes, roots, S, grid, chunk_bricks);
    hipLaunchKernelGGL(accel_scan_kernel, dim3(1), dim3(1024), 0, st, (const uint32_t *)chunk_bricks, chunk_offsets, n, total);
}

void launch_accel_bricks(const uint16_t *nodes, uint32_t n_nodes, const uint32_t *roots, uint32_t S, uint32_t *grid,
                         const uint32_t *chunk_offsets, uint16_t *bricks, uint32_t brick_cap, hipStream_t st) {
    hipLaunchKernelGGL(accel_bricks_kernel, dim3(S * S * S), dim3(512), 0, st, nodes, n_nodes, roots, S, grid, chunk_offsets, bricks,
                       brick_cap);
}

}  // namespace vrt
