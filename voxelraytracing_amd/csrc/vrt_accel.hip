// vrt_accel.hip — device-side re-layout of the reference's node pool into the march's lookup structure.
//
// The reference finds the leaf under a position by walking the chunk's octree from its root on every march
// step (find_node / find_chunk_node, ray_tracer.wgsl:76-125: up to 1 chunk_roots load + 6 dependent 2-byte
// node loads).  The backend owns the device layout (SURVEY.md §7 "may re-lay-out on upload"), so it keeps the
// pool byte-identical to the host's (uploads stay range writes) and *derives* from it, on the GPU, a two-level
// table that answers the same question in at most two loads and no loop:
//
//   cell grid   u32[8S][8S+1][8S+1] = [z][y][x]: one entry per depth-3 octree cell (4^3 voxels); the last row of
//               every z slab and the last entry of every row are a border that stays 0.  The entry *is* the march's
//               next decision (lo = leaf size - 1):
//               air leaf at depth d <= 3     ->  0xFF800000 | lo          (lo 3, 7, 15 or 31: "step through, nothing to do"; the
//                                                                          set bits make the entry the march's bit selector as
//                                                                          it is, vrt_device.h kAirLeaf)
//               other leaf at depth d <= 3   ->  voxel << 16 | lo
//               cell split at depth 3        ->  0x80000000 | brick * 64
//               border / beyond the grid     ->  0                        (a raw buffer load past the end returns 0 too:
//                                                                          "the ray has left the world")
//   brick pool  u16[bricks][64], index (x&3) | (y&3) << 2 | (z&3) << 4 inside the cell:
//               air leaf -> lo (1 at depth 4, 0 at depth 5); other leaf -> voxel << 1 | lo
//
//   march cells (the path trace's bounce launches, vrt_path.hip): everything a march step has to know about a cell in ONE
//               16-byte load, so that a split cell costs no second, dependent load —
//               .x  the cell grid's entry — an air leaf's as lo alone — (lo of a leaf in its low 5 bits, 0x80000000 | brick * 64
//                   for a split cell)
//               .y  split cell: bit (u >> 1) & 31, u = (x&3) | (y&3) << 2 | (z&3) << 4, set <=> the voxel's 2^3 sub-block is
//                   one depth-4 leaf (the index is what a shift of u gives: every sub-block owns four of the 32 bits)
//               .z .w  64 bits, bit u set <=> a ray PASSES the voxel: it is air, or a liquid of the material table the tables
//                   were built with; a leaf cell is all ones or all zeros.  Zero stops the ray.
//               The voxel a ray stopped on is read from the brick afterwards, at full width, not inside the loop.
//               Sixteen bytes for every cell of the world would be 273 MB for a 32^3-chunk world — beyond the 256 MB
//               memory-side cache, where the 67 MB cell grid sits inside it — and nine chunks in ten are one leaf.  So the
//               cells come in BLOCKS of one chunk (512 cells x-major, 8 KiB), found through a chunk directory
//               u32[S][S+1][S+1] (zero border like the grid's) that a ray re-reads only when it enters another chunk:
//               0 = outside the world -> block 0, all zeros: every cell stops the ray, "outside" needs no test;
//               1 = a chunk that is one air leaf (or missing) -> block 1, shared: 512 x {lo 31, passes};
//               b >= 2 = the chunk's own block.  23 MB instead of 273 for that world.  Inside a block the cells stand in lines
//               of 2 x 2 x 2 (a 128-byte cache line is a cube of 8^3 voxels).  Worlds of S <= 16 skip the directory: the same
//               lines, x-major over the whole world with a zero border ("direct": 4.5 MB for 8^3 chunks, 35 MB for 16^3).
//
// Every entry is exactly what find_node would return for any position inside it (same node word, same depth;
// a node read past the end of the pool is 0, a missing chunk is root 0), so the march visits the same leaves
// and produces the same frame bit for bit; variant 1 (the literal walk) and variant 2 (ancestor cache) still
// read the octree itself and are compared against it in the tests.  Rebuilt lazily before the next frame
// whenever nodes, chunk_roots or the world size changed; never inside the kernels' timed region.
#include <atomic>
#include <hip/hip_ext.h>

#include "vrt_device.h"

namespace vrt {

namespace {

__device__ __forceinline__ uint32_t pool_node(const uint16_t *nodes, uint32_t n_nodes, uint32_t idx) {
    return idx < n_nodes ? (uint32_t)nodes[idx] : 0u;  // past the end: an air leaf (what the march's buffer loads return)
}

// Position of a chunk's cell (cx, cy, cz) in the bordered grid [8S][8S+1][8S+1].
__device__ __forceinline__ size_t cell_index(uint32_t S, uint32_t chunk, uint32_t cx, uint32_t cy, uint32_t cz) {
    const uint32_t G = S * 8u, G1 = G + 1u;
    const uint32_t chx = chunk % S, chy = (chunk / S) % S, chz = chunk / (S * S);
    return ((size_t)(chz * 8u + cz) * G1 + (chy * 8u + cy)) * G1 + (chx * 8u + cx);
}

// What the kernels below need of the march cells (null blocks: not kept for this world).
struct MarchCells {
    uint32_t *dir;      // the chunk directory, [S][S+1][S+1] (null: direct)
    uint4 *blocks;      // [cap][512]; direct: [4S][4S+1][4S+1][8]
    uint32_t *tail;     // the first block not handed out yet (single-chunk rebuilds take theirs from here)
    uint32_t cap;
    uint32_t direct;    // small worlds: no directory, the cells of the whole world in lines of 2 x 2 x 2 cells, the lines x-major
                        // over the world with one border line per row and one border row of lines per slab
};
// A cell inside its chunk's block: lines of 2 x 2 x 2 cells (8^3 voxels, 128 bytes: one cache line), the lines x-major.  A ray
// reads the cells it walks through: a line that is a cube holds two or three of them, a row of eight cells along x hardly
// ever two (measured on C4: 17.5 Grays/s against 15.4).
__host__ __device__ __forceinline__ uint32_t cell_in_block(uint32_t cx, uint32_t cy, uint32_t cz) {
    return ((((cz >> 1) << 2 | (cy >> 1)) << 2 | (cx >> 1)) << 3) | (cx & 1u) | ((cy & 1u) << 1) | ((cz & 1u) << 2);
}
// ... and in a direct world
__device__ __forceinline__ size_t direct_cell_index(uint32_t S, uint32_t chunk, uint32_t cx, uint32_t cy, uint32_t cz) {
    const uint32_t chx = chunk % S, chy = (chunk / S) % S, chz = chunk / (S * S);
    const uint32_t gx = chx * 8u + cx, gy = chy * 8u + cy, gz = chz * 8u + cz, B1 = S * 4u + 1u;
    return (((size_t)(gz >> 1) * B1 + (gy >> 1)) * B1 + (gx >> 1)) * 8u + ((gx & 1u) | ((gy & 1u) << 1) | ((gz & 1u) << 2));
}
__device__ __forceinline__ size_t chunk_dir_index(uint32_t S, uint32_t chunk) {
    const uint32_t chx = chunk % S, chy = (chunk / S) % S, chz = chunk / (S * S);
    return ((size_t)chz * (S + 1u) + chy) * (S + 1u) + chx;
}
// does a chunk with this root node need a block of its own?  (one air leaf — a missing chunk is node 0 — shares block 1)
__device__ __forceinline__ bool chunk_needs_block(uint32_t root_node) { return (root_node & 0x8000u) != 0u || (root_node & 0x7FFFu) != 0u; }

// A leaf's entry as the march cells hold it: air -> lo, anything else -> voxel << 16 | lo  (lo = leaf size - 1); the cell grid
// holds grid_entry() of it.
__device__ __forceinline__ uint32_t leaf_entry(uint32_t node, uint32_t lo) { return ((node & 0x7FFFu) << 16) | lo; }

// voxel_mats[voxel].is_liquid == 1 with ids >= 256 clamped to material 255 (ray_tracer.wgsl:226), as a 256-bit mask
struct LiquidMask { uint32_t w[8]; };
__device__ __forceinline__ bool stops_a_ray(const LiquidMask &lq, uint32_t voxel) {
    const uint32_t v = min(voxel, 255u);
    return voxel != 0u && !((lq.w[v >> 5] >> (v & 31u)) & 1u);
}
// The march-cell entry of a leaf cell: a ray passes all of it, or none of it.
__device__ __forceinline__ uint4 leaf_march_cell(const LiquidMask &lq, uint32_t node, uint32_t lo) {
    const uint32_t m = stops_a_ray(lq, node & 0x7FFFu) ? 0u : 0xFFFFFFFFu;
    return make_uint4(leaf_entry(node, lo), 0u, m, m);
}
// ... of a split cell, from its brick's 64 entries (voxel << 1 | lo) packed two per word
__device__ __forceinline__ uint4 split_march_cell(const LiquidMask &lq, uint32_t brick, const uint32_t w[32]) {
    uint32_t pass[2] = {0u, 0u}, size2 = 0u;
#pragma unroll
    for (uint32_t e = 0; e < 64u; e++) {
        const uint32_t b = (w[e >> 1] >> ((e & 1u) * 16u)) & 0xFFFFu;
        if (!stops_a_ray(lq, b >> 1)) pass[e >> 5] |= 1u << (e & 31u);
        if (b & 1u) size2 |= 1u << (e >> 1);   // (every voxel of a depth-4 leaf carries lo = 1)
    }
    return make_uint4(0x80000000u | (brick * 64u), size2, pass[0], pass[1]);
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

// In-chunk rank of this thread's cell among the chunk's split cells (x-major order) and the chunk's split-cell count.
__device__ __forceinline__ uint32_t rank_split_cells(bool split, uint32_t *s_wave, uint32_t &total) {
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const unsigned long long ballot = __ballot(split);
    if (lane == 0) s_wave[wave] = (uint32_t)__popcll(ballot);
    __syncthreads();
    uint32_t before = 0;
    total = 0;
    for (uint32_t w = 0; w < 8u; w++) {
        if (w < wave) before += s_wave[w];
        total += s_wave[w];
    }
    return before + (uint32_t)__popcll(ballot & ((1ull << lane) - 1ull));
}

// The 64 entries of one split cell's brick — 8 depth-4 children x 8 depth-5 grandchildren — assembled in registers, two
// entries per word.  node_at(i): node i of the chunk, relative to its root.
template <typename NodeAt>
__device__ __forceinline__ void assemble_brick(const NodeAt &node_at, uint32_t n3, uint32_t w[32]) {
#pragma unroll
    for (uint32_t k = 0; k < 32u; k++) w[k] = 0u;
#pragma unroll
    for (uint32_t c = 0; c < 8u; c++) {
        const uint32_t n4 = node_at((n3 & 0x7FFFu) + c);
        const uint32_t x1 = (c & 1u) * 2u, y1 = ((c >> 1) & 1u) * 2u, z1 = ((c >> 2) & 1u) * 2u;
#pragma unroll
        for (uint32_t g = 0; g < 8u; g++) {
            const uint32_t x = x1 + (g & 1u), y = y1 + ((g >> 1) & 1u), z = z1 + ((g >> 2) & 1u);
            uint32_t word;
            if (n4 & 0x8000u) word = (node_at((n4 & 0x7FFFu) + g) & 0x7FFFu) << 1;  // depth 5: the walk stops here, size 1
            else word = ((n4 & 0x7FFFu) << 1) | 1u;                                 // depth-4 leaf, size 2
            const uint32_t e = x | (y << 2) | (z << 4);
            w[e >> 1] |= word << ((e & 1u) * 16u);
        }
    }
}
// ... and stored as eight 16-byte vectors (a brick is 128-byte aligned: hipMalloc + brick * 128)
__device__ __forceinline__ void store_brick(uint16_t *bricks, uint32_t brick, const uint32_t w[32]) {
    uint4 *dst = reinterpret_cast<uint4 *>(bricks + (size_t)brick * 64u);
#pragma unroll
    for (uint32_t k = 0; k < 8u; k++) dst[k] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
}

// Whole-world build, pass 1.  One workgroup per chunk slot, one thread per depth-3 cell: writes leaf entries, ranks the
// split cells inside the chunk (their bricks are laid out contiguously per chunk, cells in x-major order) and the
// chunk's brick count.
__global__ void __launch_bounds__(512) accel_cells_kernel(const uint16_t *nodes, uint32_t n_nodes, const uint32_t *roots,
                                                          uint32_t S, uint32_t *grid, uint32_t *chunk_bricks, uint32_t *chunk_needs) {
    __shared__ uint32_t s_wave[8];
    const uint32_t chunk = blockIdx.x;
    const uint32_t t = threadIdx.x, cx = t & 7u, cy = (t >> 3) & 7u, cz = t >> 6;
    const uint32_t root = roots[chunk];
    uint32_t depth;
    const uint32_t node = descend3(nodes, n_nodes, root, cx, cy, cz, depth);
    const bool split = (node & 0x8000u) != 0u;  // only possible at depth 3
    uint32_t total;
    const uint32_t rank = rank_split_cells(split, s_wave, total);
    const size_t cell = cell_index(S, chunk, cx, cy, cz);
    grid[cell] = split ? (0x80000000u | rank) : grid_entry(leaf_entry(node, (32u >> depth) - 1u));
    if (t == 0) {
        chunk_bricks[chunk] = total;
        if (chunk_needs) chunk_needs[chunk] = chunk_needs_block(pool_node(nodes, n_nodes, root)) ? 1u : 0u;
    }
}

// Pass 2b: the chunk directory.  Chunks that need a block of their own get 2, 3, ... in chunk order; block 1 is shared by the
// chunks that are one air leaf, block 0 (outside the world: the directory's zero border) stays all zeros.  One workgroup.
__global__ void __launch_bounds__(1024) accel_dir_kernel(const uint32_t *needs, uint32_t S, MarchCells mc, uint32_t *total_blocks) {
    __shared__ uint32_t s_part[1024];
    const uint32_t t = threadIdx.x, n = S * S * S;
    const uint32_t per = (n + 1023u) / 1024u;
    const uint32_t lo = min(t * per, n), hi = min(lo + per, n);

    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; i++) sum += needs[i];
    s_part[t] = sum;
    __syncthreads();
    for (uint32_t o = 1; o < 1024u; o <<= 1) {
        const uint32_t v = t >= o ? s_part[t - o] : 0u;
        __syncthreads();
        s_part[t] += v;
        __syncthreads();
    }
    uint32_t run = 2u + s_part[t] - sum;
    for (uint32_t i = lo; i < hi; i++) {
        mc.dir[chunk_dir_index(S, i)] = needs[i] ? run : 1u;
        run += needs[i];
    }
    if (t == 1023u) { *total_blocks = s_part[1023]; *mc.tail = 2u + s_part[1023]; }
}

// How many bricks a chunk's region of the pool holds beyond what it needs now: room for the cells an edit splits, so
// that a chunk can be rebuilt in place (accel_chunks_kernel).
// (A chunk without split cells — uniform, or missing — gets none: its first brick moves it to the tail.)
__host__ __device__ __forceinline__ uint32_t brick_slack(uint32_t count) { return count ? 8u + count / 8u : 0u; }

// Pass 2: exclusive scan of the per-chunk region sizes (count + slack) — one workgroup; S^3 <= 10^6 entries, off the
// frame path.  `total` = bricks in all regions = where the relocation tail starts.
__global__ void __launch_bounds__(1024) accel_scan_kernel(const uint32_t *counts, uint32_t *bases, uint32_t *caps, uint32_t n,
                                                          uint32_t *total, uint32_t *tail) {
    __shared__ uint32_t s_part[1024];
    const uint32_t t = threadIdx.x;
    const uint32_t per = (n + 1023u) / 1024u;
    const uint32_t lo = min(t * per, n), hi = min(lo + per, n);
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; i++) sum += min(512u, counts[i] + brick_slack(counts[i]));
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
        const uint32_t cap = min(512u, counts[i] + brick_slack(counts[i]));  // a chunk has 512 cells: never more bricks
        bases[i] = run;
        caps[i] = cap;
        run += cap;
    }
    if (t == 1023u) { *total = s_part[1023]; *tail = s_part[1023]; }
}

// Pass 3: fills the bricks of the split cells and replaces their in-chunk rank by the pool position.
__global__ void __launch_bounds__(512) accel_bricks_kernel(const uint16_t *nodes, uint32_t n_nodes, const uint32_t *roots,
                                                           uint32_t S, uint32_t *grid, const uint32_t *chunk_bases,
                                                           uint16_t *bricks, uint32_t brick_cap, MarchCells mc, LiquidMask lq) {
    const uint32_t chunk = blockIdx.x;
    const uint32_t t = threadIdx.x, cx = t & 7u, cy = (t >> 3) & 7u, cz = t >> 6;
    const size_t cell = cell_index(S, chunk, cx, cy, cz);
    const uint32_t e = grid[cell];
    uint4 *mcell = nullptr;
    if (mc.blocks && mc.direct) {
        mcell = mc.blocks + direct_cell_index(S, chunk, cx, cy, cz);
    } else if (mc.blocks) {
        const uint32_t blk = mc.dir[chunk_dir_index(S, chunk)];
        if (blk >= 2u && blk < mc.cap) mcell = mc.blocks + (size_t)blk * 512u + cell_in_block(cx, cy, cz);
    }
    const uint32_t root = roots[chunk];
    if (!is_split_entry(e)) {
        if (!mcell) return;
        uint32_t depth;
        const uint32_t node = descend3(nodes, n_nodes, root, cx, cy, cz, depth);
        *mcell = leaf_march_cell(lq, node, (32u >> depth) - 1u);
        return;
    }
    const uint32_t brick = chunk_bases[chunk] + (e & 0x7FFFFFFFu);
    if (brick >= brick_cap) return;  // cannot happen: the pool was sized from the scan's total
    uint32_t depth;
    const uint32_t n3 = descend3(nodes, n_nodes, root, cx, cy, cz, depth);
    uint32_t w[32];
    assemble_brick([&](uint32_t i) { return pool_node(nodes, n_nodes, root + i); }, n3, w);
    store_brick(bricks, brick, w);
    grid[cell] = 0x80000000u | (brick * 64u);
    if (mcell) *mcell = split_march_cell(lq, brick, w);
}

// Rebuild of single chunks (a voxel edit, a chunk that arrived): one workgroup per listed chunk does all three passes
// for it.  One workgroup has nobody to hide its load latency behind, and a cell's walk is a chain of dependent 2-byte
// reads (3 levels to the cell, then 72 more for a brick), so the chunk's nodes — at most 32 767 + 8 of them, 64 KiB, a
// chunk addresses no more (common/src/world/mod.rs:416) — are first copied into LDS with coalesced reads and every walk
// reads LDS.  The bricks go back into the chunk's own region when they fit (they do, unless an edit burst outgrew the
// slack); otherwise the chunk moves to a fresh 512-brick region — the most a chunk can ever need — taken from the tail of
// the pool with one atomic.  The host keeps the tail from overflowing: it counts the chunks that may have moved since
// the last whole-world build and asks for one of those instead when the tail could run out (vrt_uploads.hip).
struct ChunkList { uint32_t chunk[64]; uint32_t extent[64]; uint32_t root[64]; };   // extent: nodes from the chunk's root up to the next chunk's (host's estimate); root: chunk_roots[chunk] as uploaded (the host's mirror: one memory round trip less on a lone workgroup's critical path)

constexpr uint32_t kChunkNodesMax = 0x7FFFu + 8u;  // child_idx <= 0x7FFF, + 8 children

// Node `idx` of the chunk rooted at `root`: from LDS if it was staged, else from the pool (a pool whose child indices reach
// past the chunk's own extent — no world the host mirror builds; garbage pools in the tests — is still read as the march
// reads it); an index the 15-bit child field cannot form, or one past the end of the pool, is an air leaf.
struct ChunkNodes {
    const uint16_t *lds;     // staged words, relative to the root
    const uint16_t *pool;
    uint32_t staged, root, n_nodes;
    __device__ __forceinline__ uint32_t operator()(uint32_t idx) const {
        if (idx < staged) return (uint32_t)lds[idx];
        if (idx >= kChunkNodesMax) return 0u;
        const uint64_t g = (uint64_t)root + idx;
        return g < n_nodes ? (uint32_t)pool[g] : 0u;
    }
};

__global__ void __launch_bounds__(512) accel_chunks_kernel(const uint16_t *nodes, uint32_t n_nodes, const uint32_t *roots, uint32_t S,
                                                           uint32_t *grid, uint32_t *chunk_bricks, uint32_t *chunk_bases,
                                                           uint32_t *chunk_caps, uint32_t *tail, uint16_t *bricks, uint32_t brick_cap,
                                                           MarchCells mc, LiquidMask lq, ChunkList list) {
    extern __shared__ __attribute__((aligned(16))) uint16_t s_raw[];  // the chunk's node words from the 16-byte boundary at or below its root
    __shared__ uint32_t s_wave[8];
    __shared__ uint32_t s_base, s_blk;
    __shared__ uint32_t s_split[512];
    __shared__ uint32_t s_lq[8];
    // (The kernel is written for SIZE: a lone workgroup runs it once, cold — the frame before it owned the instruction cache — and
    // straight-line code is then fetched a cache line per memory round trip.  The 11.7 KB of the first wave-per-cell version ran in
    // 14 us whatever its loops did; see tools/chunk_probe.py.  Hence buffer loads instead of a tail path per staged vector, the
    // liquid mask in LDS instead of a select chain per use, one place that reads a node, and no unrolling but the four cells the
    // last loop keeps in flight for its latency chain: 7.6 KB.)
    // (512 threads, not more: sixteen waves halve the lone kernel — 9.3 us, a lone edit 162.7 — but a workgroup that needs a
    // whole CU at once waits for one while frames are in flight: an edit before every frame 97 -> 107 us per frame, measured)
    const uint32_t chunk = list.chunk[blockIdx.x];
    const uint32_t t = threadIdx.x, cx = t & 7u, cy = (t >> 3) & 7u, cz = t >> 6;
    const uint32_t root = list.root[blockIdx.x];
    (void)roots;
    // the region's base and size are wanted after two barriers: asked for now, beside the staging loads
    uint32_t base0 = 0u, cap0 = 0u, blk0 = 0u;
    if (t == 0) {
        base0 = chunk_bases[chunk]; cap0 = chunk_caps[chunk];
        if (mc.blocks && !mc.direct) blk0 = mc.dir[chunk_dir_index(S, chunk)];
    }
    if (t < 8u) s_lq[t] = lq.w[t];
    // stage the chunk's own extent, eight nodes (16 bytes) per load; the pool is 16-byte aligned (hipMalloc).  Range-checked
    // loads: what lies past the end of the pool reads as air leaves, as the march's own loads have it (whole words of two
    // nodes are checked: the last node of a pool of odd size is put in place below)
    const uint32_t head = root & 7u, first = root - head;
    const uint32_t staged = min(list.extent[blockIdx.x], kChunkNodesMax);
    const uint32_t vecs = (head + staged + 7u) / 8u;
    // (n_nodes <= 2^31 - 2^17: vrt_create — so that the byte offset of the extent's last vector, (first + 8 vecs) * 2 with at most
    // kChunkNodesMax + 16 nodes behind `first`, cannot wrap around 2^32 and read the start of the pool instead of zeros)
    const __amdgpu_buffer_rsrc_t nb = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(nodes), 0, (n_nodes & ~1u) * 2u, 0x00020000);
    // (a lone workgroup: all of a thread's loads are issued before the first of them is stored — one load per loop trip
    // was up to nine memory round trips in a row, 15 of the kernel's 17 us for a chunk of 40 000 nodes)
    constexpr uint32_t kVecsPerThread = (kChunkNodesMax + 16u + 8u * 512u - 1u) / (8u * 512u);
    uint4 sw[kVecsPerThread];
#pragma unroll
    for (uint32_t j = 0; j < kVecsPerThread; j++) {
        const uint32_t v = t + 512u * j;
        sw[j] = make_uint4(0u, 0u, 0u, 0u);
        if (v < vecs) sw[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(nb, (first + v * 8u) * 2u, 0, 0));
    }
#pragma unroll
    for (uint32_t j = 0; j < kVecsPerThread; j++) {
        const uint32_t v = t + 512u * j;
        if (v < vecs) reinterpret_cast<uint4 *>(s_raw)[v] = sw[j];
    }
    __syncthreads();
    if (t == 0 && (n_nodes & 1u) && n_nodes - 1u >= first && n_nodes - 1u - first < vecs * 8u) s_raw[n_nodes - 1u - first] = nodes[n_nodes - 1u];
    if (n_nodes & 1u) __syncthreads();   // (uniform)
    // node `idx` of the chunk, relative to its root: the one place that reads one (ChunkNodes above, spelled out once)
    const uint16_t *lds = s_raw + head;
    auto node_at = [&](uint32_t idx) __attribute__((always_inline)) -> uint32_t {
        if (idx < staged) return (uint32_t)lds[idx];
        if (idx >= kChunkNodesMax) return 0u;
        const uint64_t g = (uint64_t)root + idx;
        return g < n_nodes ? (uint32_t)nodes[g] : 0u;
    };
    // the three levels above this thread's cell
    const uint32_t root_node = node_at(0u);
    uint32_t node = root_node, depth = 0u;
    while ((node & 0x8000u) && depth < 3u) {
        const uint32_t sh = 2u - depth;
        const uint32_t sel = ((cx >> sh) & 1u) | (((cy >> sh) & 1u) << 1) | (((cz >> sh) & 1u) << 2);
        node = node_at((node & 0x7FFFu) + sel);
        depth += 1u;
    }
    const bool split = (node & 0x8000u) != 0u;
    uint32_t total;
    const uint32_t rank = rank_split_cells(split, s_wave, total);
    if (split) s_split[rank] = t | (node << 16);   // the split cells in rank order: which cell, and its depth-3 node
    if (t == 0) {
        uint32_t base = base0;
        if (total > cap0) {
            base = atomicAdd(tail, 512u);
            chunk_bases[chunk] = base;
            chunk_caps[chunk] = 512u;
        }
        chunk_bricks[chunk] = total;
        s_base = base;
        // the chunk's block of march cells: its own if it has one (it keeps it, whatever the chunk turns into), a new one
        // from the tail when a chunk that was one air leaf is that no longer
        uint32_t blk = 0u;
        if (mc.blocks && !mc.direct) {
            blk = blk0;
            if (blk < 2u && chunk_needs_block(root_node)) {
                blk = atomicAdd(mc.tail, 1u);
                mc.dir[chunk_dir_index(S, chunk)] = blk < mc.cap ? blk : 1u;   // (cannot overflow: the host accounts for every possible new block)
            }
        }
        s_blk = blk;
    }
    __syncthreads();
    const bool own_block = s_blk >= 2u && s_blk < mc.cap;
    // what a cell writes — its entry of the cell grid and its march cell — in one place: a leaf cell's thread, or lane 0 of the
    // wave that did a split cell's brick
    auto stops = [&](uint32_t voxel) __attribute__((always_inline)) {   // stops_a_ray on the mask in LDS
        const uint32_t v = min(voxel, 255u);
        return voxel != 0u && !((s_lq[v >> 5] >> (v & 31u)) & 1u);
    };
    auto write_cell = [&](uint32_t ct, uint4 mcell_value) __attribute__((always_inline)) {
        const uint32_t x = ct & 7u, y = (ct >> 3) & 7u, z = ct >> 6;
        grid[cell_index(S, chunk, x, y, z)] = grid_entry(mcell_value.x);
        uint4 *mcell = nullptr;
        if (mc.blocks && mc.direct) mcell = mc.blocks + direct_cell_index(S, chunk, x, y, z);
        else if (own_block) mcell = mc.blocks + (size_t)s_blk * 512u + cell_in_block(x, y, z);
        if (mcell) *mcell = mcell_value;
    };
    if (!split) {   // a leaf cell: its thread writes it (a march cell's first word is the cell's entry of the grid)
        const uint32_t m = stops(node & 0x7FFFu) ? 0u : 0xFFFFFFFFu;
        write_cell(t, make_uint4(leaf_entry(node, (32u >> depth) - 1u), 0u, m, m));
    }
    // The split cells: A WAVE PER CELL, A LANE PER VOXEL of its brick.  (One thread per cell — round 3 — had the threads of the
    // split cells walk 72 nodes and classify 64 voxels each, alone in their waves, while the threads of the leaf cells were
    // done.)  Lane e = x | y << 2 | z << 4 reads its depth-4 node — eight different ones per wave — and, below a split one, its
    // depth-5 node; the brick is one 128-byte store per wave; which voxels a ray passes is a ballot; the size-2 bit of an entry
    // pair is that of its even lane (both are under the same depth-4 node).  Same words in the same places as assemble_brick /
    // split_march_cell produce (the whole-world build still uses those: tests/test_gpu_accel.py compares the two).
    const uint32_t lane = t & 63u, wave = t >> 6;
    const uint32_t c4 = ((lane >> 1) & 1u) | (((lane >> 3) & 1u) << 1) | (((lane >> 5) & 1u) << 2);   // the depth-4 child the voxel is in
    const uint32_t g5 = (lane & 1u) | (((lane >> 2) & 1u) << 1) | (((lane >> 4) & 1u) << 2);          // the voxel inside it
    // Four cells at a time, every step for all four before the next: a cell is a chain of six LDS round trips (its list entry,
    // the depth-4 node, the depth-5 node, the liquid mask, the shuffle, ~ 150 cycles each with two waves on a SIMD:
    // tools/chunk_probe.py) and the lone workgroup has nothing else to put between them.  The reads are branch-free — an index
    // beyond what was staged (no world the host mirror builds) is patched afterwards, under a branch the whole wave skips.
    constexpr uint32_t kAtOnce = 4u;
    const uint32_t last = staged ? staged - 1u : 0u;
#pragma clang loop unroll(disable)
    for (uint32_t r0 = wave; r0 < total; r0 += 8u * kAtOnce) {
        uint32_t ent[kAtOnce], i4[kAtOnce], n4[kAtOnce], i5[kAtOnce], w[kAtOnce], lq_word[kAtOnce], lo_even[kAtOnce];
#pragma unroll
        for (uint32_t k = 0; k < kAtOnce; k++) ent[k] = s_split[min(r0 + 8u * k, 511u)];   // (past the list: whatever is there, unused)
        bool beyond = false;
#pragma unroll
        for (uint32_t k = 0; k < kAtOnce; k++) {
            i4[k] = ((ent[k] >> 16) & 0x7FFFu) + c4;
            n4[k] = (uint32_t)lds[min(i4[k], last)];
            beyond = beyond || i4[k] >= staged;
        }
        if (__ballot(beyond) != 0ull) {
#pragma unroll
            for (uint32_t k = 0; k < kAtOnce; k++) n4[k] = node_at(i4[k]);
        }
        beyond = false;
#pragma unroll
        for (uint32_t k = 0; k < kAtOnce; k++) {
            i5[k] = (n4[k] & 0x7FFFu) + g5;
            w[k] = (uint32_t)lds[min(i5[k], last)];   // (read whether the depth-4 node is split or not)
            beyond = beyond || ((n4[k] & 0x8000u) != 0u && i5[k] >= staged);
        }
        if (__ballot(beyond) != 0ull) {
#pragma unroll
            for (uint32_t k = 0; k < kAtOnce; k++) w[k] = node_at(i5[k]);
        }
#pragma unroll
        for (uint32_t k = 0; k < kAtOnce; k++) {
            w[k] = (n4[k] & 0x8000u) ? (w[k] & 0x7FFFu) << 1         // depth 5: the walk stops here, size 1
                                     : ((n4[k] & 0x7FFFu) << 1) | 1u;   // depth-4 leaf, size 2
            lq_word[k] = s_lq[min(w[k] >> 1, 255u) >> 5];
        }
#pragma unroll
        for (uint32_t k = 0; k < kAtOnce; k++) lo_even[k] = (uint32_t)__shfl((int)(w[k] & 1u), (int)((2u * lane) & 63u), 64);
        // the four cells' entries are written by lanes 0..3, one cell each
        uint32_t my_ct = 0u, my_brick = 0u, my_size2 = 0u, my_pass_lo = 0u, my_pass_hi = 0u;
        bool mine = false;
#pragma unroll
        for (uint32_t k = 0; k < kAtOnce; k++) {
            const uint32_t r = r0 + 8u * k, brick = s_base + r;
            if (r >= total || brick >= brick_cap) break;  // (the second cannot happen: the host accounts for every possible move)
            bricks[(size_t)brick * 64u + lane] = (uint16_t)w[k];
            const uint32_t voxel = w[k] >> 1, v = min(voxel, 255u);
            const unsigned long long pass = __ballot(!(voxel != 0u && !((lq_word[k] >> (v & 31u)) & 1u)));   // stops_a_ray
            const uint32_t size2 = (uint32_t)__ballot(lane < 32u && lo_even[k] != 0u);
            if (lane == k) { mine = true; my_ct = ent[k] & 0x1FFu; my_brick = brick; my_size2 = size2; my_pass_lo = (uint32_t)pass; my_pass_hi = (uint32_t)(pass >> 32); }
        }
        // (no fence between the brick and the entry that names it: nothing reads this table set while the kernel runs — its
        // frames are later on this very stream, and frames of other sets that shared it were waited for, update_tables)
        if (mine) write_cell(my_ct, make_uint4(0x80000000u | (my_brick * 64u), my_size2, my_pass_lo, my_pass_hi));
    }
}

// Upload of a staged range: the pinned ring is mapped into the device's address space, so a kernel reads it over PCIe and
// writes the resident buffer — a launch like any other on the stream, where hipMemcpyAsync makes the host wait for the
// work the stream still has queued (measured: vrt_write_nodes returned after the frames in flight, 235 us, instead of 15).
__global__ void __launch_bounds__(256) upload_words_kernel(uint32_t *dst, const uint32_t *src, uint32_t n_words) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += gridDim.x * blockDim.x) dst[i] = src[i];
}

// A batch of staged ranges — everything vrt_write_nodes / vrt_write_chunk_roots staged since the last frame — in ONE launch:
// a workgroup per piece of at most kUploadPieceWords words (the host cuts the ranges up; the pieces ride in the kernarg).
// The source is host memory: a load takes a PCIe round trip (~1.5 us), so every thread has ALL its loads in flight before
// it stores anything — four 16-byte loads cover a 16-KiB piece (one word at a time the piece was sixteen round trips in a
// row: 20 us for a chunk's 79 KB, most of a lone edit's latency; ~3 us now).  The ring side of a piece is 64-byte aligned,
// the destination only word aligned (NodeBuffer::write widens to even nodes, not to 16 bytes): word stores.
__global__ void __launch_bounds__(256) upload_batch_kernel(uint32_t *dst0, uint32_t *dst1, const uint32_t *ring, UploadBatch batch) {
    const UploadPiece p = batch.piece[blockIdx.x];
    uint32_t *dst = ((p.n_words >> 31) ? dst1 : dst0) + p.dst_word;
    const uint32_t *src = ring + p.src_word;
    const uint32_t n = p.n_words & 0x7FFFFFFFu, n4 = n / 4u;
    static_assert(kUploadPieceWords == 4u * 4u * 256u, "four 16-byte loads per thread cover a piece");
    const uint4 *src4 = reinterpret_cast<const uint4 *>(src);
    uint4 v[4];
#pragma unroll
    for (uint32_t j = 0; j < 4u; j++) {
        const uint32_t i = threadIdx.x + 256u * j;
        v[j] = i < n4 ? src4[i] : make_uint4(0u, 0u, 0u, 0u);
    }
    const uint32_t tail = n4 * 4u + threadIdx.x;
    const uint32_t last = tail < n ? src[tail] : 0u;   // (a piece that is not whole 16-byte vectors: up to three words more)
#pragma unroll
    for (uint32_t j = 0; j < 4u; j++) {
        const uint32_t i = threadIdx.x + 256u * j;
        if (i < n4) { dst[4u * i] = v[j].x; dst[4u * i + 1u] = v[j].y; dst[4u * i + 2u] = v[j].z; dst[4u * i + 3u] = v[j].w; }
    }
    if (tail < n) dst[tail] = last;
}

}  // namespace

void launch_upload_batch(void *dst0, void *dst1, const void *pinned_ring, const UploadBatch &batch, uint32_t n_pieces, hipStream_t st) {
    if (!n_pieces) return;
    hipLaunchKernelGGL(upload_batch_kernel, dim3(n_pieces), dim3(256), 0, st, (uint32_t *)dst0, (uint32_t *)dst1, (const uint32_t *)pinned_ring, batch);
}

void launch_upload_words(void *dst, const void *pinned_src, uint32_t n_words, hipStream_t st) {
    if (!n_words) return;
    const uint32_t blocks = (n_words + 255u) / 256u;
    hipLaunchKernelGGL(upload_words_kernel, dim3(blocks < 1024u ? blocks : 1024u), dim3(256), 0, st, (uint32_t *)dst,
                       (const uint32_t *)pinned_src, n_words);
}

static LiquidMask liquid_mask(const uint32_t liquid[8]) {
    LiquidMask lq;
    for (int i = 0; i < 8; i++) lq.w[i] = liquid[i];
    return lq;
}

// march cells: dir / blocks / tail / cap (blocks null: not kept for this world; dir null: a direct world), liquid: the 256-bit is_liquid mask they are
// built with.  launch_accel_cells leaves the number of blocks the chunks need in *total_blocks (+ 2: blocks 0 and 1).
void launch_accel_cells(const uint16_t *nodes, uint32_t n_nodes, const uint32_t *roots, uint32_t S, uint32_t *grid,
                        uint32_t *chunk_bricks, uint32_t *chunk_bases, uint32_t *chunk_caps, uint32_t *total, uint32_t *tail,
                        uint32_t *chunk_needs, uint32_t *dir, uint32_t *block_tail, uint32_t *total_blocks, hipStream_t st) {
    const uint32_t n = S * S * S;
    hipLaunchKernelGGL(accel_cells_kernel, dim3(n), dim3(512), 0, st, nodes, n_nodes, roots, S, grid, chunk_bricks, chunk_needs);
    hipLaunchKernelGGL(accel_scan_kernel, dim3(1), dim3(1024), 0, st, (const uint32_t *)chunk_bricks, chunk_bases, chunk_caps, n, total, tail);
    if (chunk_needs) {   // (the directory: block 1 is filled by the bricks pass' launcher once the blocks exist)
        const MarchCells mc{dir, nullptr, block_tail, 0u, 0u};
        hipLaunchKernelGGL(accel_dir_kernel, dim3(1), dim3(1024), 0, st, (const uint32_t *)chunk_needs, S, mc, total_blocks);
    }
}

__global__ void __launch_bounds__(512) accel_air_block_kernel(uint4 *blocks) {
    blocks[512u + threadIdx.x] = make_uint4(31u, 0u, 0xFFFFFFFFu, 0xFFFFFFFFu);   // block 1: one air leaf
}

void launch_accel_bricks(const uint16_t *nodes, uint32_t n_nodes, const uint32_t *roots, uint32_t S, uint32_t *grid,
                         const uint32_t *chunk_bases, uint16_t *bricks, uint32_t brick_cap, uint32_t *dir, uint4 *blocks, uint32_t *block_tail,
                         uint32_t block_cap, const uint32_t liquid[8], hipStream_t st) {
    if (blocks && dir) hipLaunchKernelGGL(accel_air_block_kernel, dim3(1), dim3(512), 0, st, blocks);
    const MarchCells mc{dir, blocks, block_tail, block_cap, dir ? 0u : 1u};
    hipLaunchKernelGGL(accel_bricks_kernel, dim3(S * S * S), dim3(512), 0, st, nodes, n_nodes, roots, S, grid, chunk_bases, bricks,
                       brick_cap, mc, liquid_mask(liquid));
}

void launch_accel_chunks(const uint16_t *nodes, uint32_t n_nodes, const uint32_t *roots, uint32_t S, uint32_t *grid,
                         uint32_t *chunk_bricks, uint32_t *chunk_bases, uint32_t *chunk_caps, uint32_t *tail, uint16_t *bricks,
                         uint32_t brick_cap, uint32_t *dir, uint4 *blocks, uint32_t *block_tail, uint32_t block_cap, const uint32_t liquid[8],
                         const uint32_t *chunks, const uint32_t *extents, const uint32_t *chunk_roots_host, uint32_t n, hipStream_t st, hipEvent_t done) {
    // 64 KiB + of dynamic LDS needs opting in (the CU has 160 KiB); per device, and any thread may be the first
    const size_t lds = (size_t)(kChunkNodesMax + 16u) * sizeof(uint16_t);
    {   // (once per device: the call is a few microseconds of every edit's frame otherwise)
        // (a device is marked only once its opt-in succeeded — a failed call is tried again by the next rebuild, whose launch then
        // reports it; device ordinals beyond the mask's 64 bits opt in every time)
        static std::atomic<uint64_t> opted_in{0};
        int dev = 0;
        (void)hipGetDevice(&dev);
        const uint64_t bit = (unsigned)dev < 64u ? 1ull << dev : 0ull;
        if (!(opted_in.load(std::memory_order_relaxed) & bit) &&
            hipFuncSetAttribute(reinterpret_cast<const void *>(accel_chunks_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess)
            opted_in.fetch_or(bit, std::memory_order_relaxed);
    }
    const MarchCells mc{dir, blocks, block_tail, block_cap, dir ? 0u : 1u};
    for (uint32_t i = 0; i < n; i += 64u) {
        ChunkList list;
        const uint32_t m = n - i < 64u ? n - i : 64u;
        for (uint32_t k = 0; k < m; k++) { list.chunk[k] = chunks[i + k]; list.extent[k] = extents[i + k]; list.root[k] = chunk_roots_host[i + k]; }
        // `done` (may be null): an event the LAST launch completes — the launch's own completion signal, no marker packet behind it
        if (done && i + 64u >= n)
            hipExtLaunchKernelGGL(accel_chunks_kernel, dim3(m), dim3(512), (uint32_t)lds, st, nullptr, done, 0, nodes, n_nodes, roots, S, grid, chunk_bricks,
                                  chunk_bases, chunk_caps, tail, bricks, brick_cap, mc, liquid_mask(liquid), list);
        else
            hipLaunchKernelGGL(accel_chunks_kernel, dim3(m), dim3(512), lds, st, nodes, n_nodes, roots, S, grid, chunk_bricks, chunk_bases,
                               chunk_caps, tail, bricks, brick_cap, mc, liquid_mask(liquid), list);
    }
}

}  // namespace vrt
