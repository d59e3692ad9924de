// worldgen.hpp — deterministic world generator and bottom-up SVO builder.
//
// Stands where the reference's server-side generator does (server/src/world/gen.rs:171-286: per-column
// terrain layers, water up to sea level, tree features), but is build-defined: the reference's generator
// cannot be reproduced even by itself (third-party Perlin + an unseeded global fastrand, gen.rs:11,263,274).
// Everything here is integer arithmetic on (seed, x, y, z), so every platform produces the same world.
// The SVO is built bottom-up from a dense 32^3 block and laid out breadth-first: the root, its 8
// children and their 64 children occupy the first 73 node slots (146 B, two cache lines), which is what
// every ray's descent touches first.
#pragma once

#include <cstdint>
#include <vector>

#include "world.hpp"

namespace vrt {

// voxel ids from stdrespack/voxels.ron (index in the list = id)
namespace vox {
constexpr uint16_t AIR = 0, LAVA = 2, WATER = 3, LIMESTONE = 4, SLATE = 5, DIRT = 39, GRASS = 40, SNOW = 45,
                   SAND = 47, OAK_WOOD = 53, OAK_LEAVES = 62;
}

struct WorldGen {
    uint32_t seed = 1;
    int32_t h_min = 40, h_max = 200, sea_level = 70, snow_line = 172;

    static uint32_t mix(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
        // PCG-style output permutation over a 4-word key
        uint32_t h = a * 747796405u + 2891336453u;
        h = (h ^ b) * 277803737u; h ^= h >> 15;
        h = (h ^ c) * 2246822519u; h ^= h >> 13;
        h = (h ^ d) * 3266489917u; h ^= h >> 16;
        return h;
    }
    // lattice value in [0, 65535]
    uint32_t lattice(int32_t ix, int32_t iz, uint32_t octave) const { return mix(seed, (uint32_t)ix, (uint32_t)iz, octave) >> 16; }

    // value noise at (x,z) with cell size `cell` (power of two), 16.16 fixed point result in [0, 65536)
    uint32_t value_noise(int32_t x, int32_t z, uint32_t cell_log2, uint32_t octave) const {
        const int32_t ix = x >> cell_log2, iz = z >> cell_log2;  // floor for negatives too
        const uint32_t m = (1u << cell_log2) - 1u;
        const uint64_t tx = ((uint64_t)((uint32_t)x & m) << 16) >> cell_log2, tz = ((uint64_t)((uint32_t)z & m) << 16) >> cell_log2;
        const uint64_t sx = (tx * tx * (3u * 65536u - 2u * tx)) >> 32, sz = (tz * tz * (3u * 65536u - 2u * tz)) >> 32;  // smoothstep, 0..65536
        const uint64_t v00 = lattice(ix, iz, octave), v10 = lattice(ix + 1, iz, octave), v01 = lattice(ix, iz + 1, octave),
                       v11 = lattice(ix + 1, iz + 1, octave);
        const uint64_t a = (v00 * (65536u - sx) + v10 * sx) >> 16, b = (v01 * (65536u - sx) + v11 * sx) >> 16;
        return (uint32_t)((a * (65536u - sz) + b * sz) >> 16);
    }

    // terrain surface height at (x,z): y <= height is ground
    int32_t height(int32_t x, int32_t z) const {
        const uint64_t f = (8ull * value_noise(x, z, 7, 0) + 4ull * value_noise(x, z, 6, 1) + 2ull * value_noise(x, z, 5, 2) +
                            1ull * value_noise(x, z, 4, 3)) / 15ull;  // 0..65535
        // contrast stretch around the middle (x2.25), clamped
        int64_t g = ((int64_t)f - 32768) * 9 / 4 + 32768;
        if (g < 0) g = 0;
        if (g > 65535) g = 65535;
        return h_min + (int32_t)(((int64_t)(h_max - h_min) * g) >> 16);
    }

    // terrain + water only (no trees)
    uint16_t terrain_at(int32_t h, int32_t y) const {
        if (y > h) return y <= sea_level ? vox::WATER : vox::AIR;
        const int32_t layer = h - y;
        if (layer == 0) return h <= sea_level + 1 ? vox::SAND : (h >= snow_line ? vox::SNOW : vox::GRASS);
        if (layer <= 4) return h <= sea_level + 1 ? vox::SAND : vox::DIRT;
        return vox::SLATE;
    }

    struct Tree { bool present; int32_t x, z, base, trunk; };
    // one candidate tree per 16x16 cell; offsets 3..12 keep the radius-3 crown inside its cell
    Tree tree_in_cell(int32_t cx16, int32_t cz16) const {
        const uint32_t h = mix(seed ^ 0x9E3779B9u, (uint32_t)cx16, (uint32_t)cz16, 77u);
        Tree t;
        t.x = cx16 * 16 + 3 + (int32_t)((h >> 4) % 10u);
        t.z = cz16 * 16 + 3 + (int32_t)((h >> 12) % 10u);
        t.trunk = 5 + (int32_t)((h >> 20) & 3u);
        t.base = height(t.x, t.z);
        t.present = (h & 3u) != 0u && t.base > sea_level + 1 && t.base < snow_line - 8;
        return t;
    }

    // dense[x + 32*(y + 32*z)] for chunk cp; returns true if the block is uniform (value in dense[0])
    bool fill_dense(ChunkPos cp, uint16_t *dense) const {
        const int32_t x0 = cp.x * 32, y0 = cp.y * 32, z0 = cp.z * 32;
        for (int32_t z = 0; z < 32; z++)
            for (int32_t x = 0; x < 32; x++) {
                const int32_t h = height(x0 + x, z0 + z);
                for (int32_t y = 0; y < 32; y++) dense[x + 32 * (y + 32 * z)] = terrain_at(h, y0 + y);
            }
        for (int32_t cz = 0; cz < 2; cz++)
            for (int32_t cx = 0; cx < 2; cx++) {
                const Tree t = tree_in_cell((x0 >> 4) + cx, (z0 >> 4) + cz);
                if (!t.present) continue;
                const int32_t top = t.base + t.trunk;
                for (int32_t dy = -3; dy <= 3; dy++)
                    for (int32_t dz = -3; dz <= 3; dz++)
                        for (int32_t dx = -3; dx <= 3; dx++) {
                            if (dx * dx + dy * dy + dz * dz > 11) continue;
                            const int32_t x = t.x + dx - x0, y = top + dy - y0, z = t.z + dz - z0;
                            if (x < 0 || y < 0 || z < 0 || x >= 32 || y >= 32 || z >= 32) continue;
                            uint16_t &v = dense[x + 32 * (y + 32 * z)];
                            if (v == vox::AIR) v = vox::OAK_LEAVES;
                        }
                for (int32_t y = t.base + 1; y <= top; y++) {
                    const int32_t ly = y - y0;
                    if (ly < 0 || ly >= 32) continue;
                    dense[(t.x - x0) + 32 * (ly + 32 * (t.z - z0))] = vox::OAK_WOOD;
                }
            }
        bool uniform = true;
        for (int i = 1; i < 32 * 32 * 32 && uniform; i++) uniform = dense[i] == dense[0];
        return uniform;
    }
};

// Minimal octree of a dense 32^3 block (dense[x + 32*(y + 32*z)]), breadth-first node order, 8-blocks
// starting at index 1 exactly where NodeAlloc::new(0..1, 1..cap) would hand out its first block
// (server/src/world/gen.rs:177).  Returns false if the tree needs more than 32767 addressable nodes
// (the 15-bit child index, common/src/world/mod.rs:416).
inline bool build_svo_bottom_up(const uint16_t *dense, std::vector<Node> &out) {
    // level L has (1<<L)^3 cells; val >= 0: uniform voxel id, -1: mixed
    std::vector<int32_t> lv[6];
    lv[5].resize(32768);
    for (int i = 0; i < 32768; i++) lv[5][i] = dense[i];
    for (int L = 4; L >= 0; L--) {
        const int n = 1 << L, c = n * 2;
        lv[L].resize((size_t)n * n * n);
        for (int z = 0; z < n; z++)
            for (int y = 0; y < n; y++)
                for (int x = 0; x < n; x++) {
                    const int32_t first = lv[L + 1][(2 * x) + c * ((2 * y) + c * (2 * z))];
                    int32_t v = first;
                    for (int k = 1; k < 8 && v >= 0; k++) {
                        const int32_t ch = lv[L + 1][(2 * x + (k & 1)) + c * ((2 * y + ((k >> 1) & 1)) + c * (2 * z + (k >> 2)))];
                        if (ch != first) v = -1;
                    }
                    lv[L][x + n * (y + n * z)] = v;
                }
    }
    struct Item { uint8_t L, x, y, z; uint32_t idx; };
    std::vector<Item> q;
    q.push_back({0, 0, 0, 0, 0});
    out.assign(1, Node());
    for (size_t head = 0; head < q.size(); head++) {
        const Item it = q[head];
        const int n = 1 << it.L;
        const int32_t v = lv[it.L][it.x + n * (it.y + n * it.z)];
        if (v >= 0) { out[it.idx] = Node::make(Voxel((uint16_t)v)); continue; }
        const uint32_t base = (uint32_t)out.size();
        if (base + 8 > 32767u) return false;
        out.resize(base + 8);
        out[it.idx] = Node::new_split((uint16_t)base);
        for (uint32_t k = 0; k < 8; k++)
            q.push_back({(uint8_t)(it.L + 1), (uint8_t)(2 * it.x + (k & 1)), (uint8_t)(2 * it.y + ((k >> 1) & 1)),
                         (uint8_t)(2 * it.z + (k >> 2)), base + k});
    }
    return true;
}

// The reference's way (gen.rs:171-286 reduced to its SVO loop): NodeAlloc::new(0..1, 1..cap), then
// Svo::set_node for x, then z, then y ascending, skipping air. Returns nodes in use (last_used_addr+1), 0 on OOM.
inline uint32_t build_svo_by_set_node(const uint16_t *dense, Node *nodes, uint32_t cap) {
    NodeAlloc alloc({0, 1}, {1, cap});
    for (uint32_t i = 0; i < cap; i++) nodes[i] = Node();
    const Svo svo{0, CHUNK_SIZE};
    for (uint32_t x = 0; x < 32; x++)
        for (uint32_t z = 0; z < 32; z++)
            for (uint32_t y = 0; y < 32; y++) {
                const uint16_t v = dense[x + 32 * (y + 32 * z)];
                if (v == 0) continue;
                if (svo.set_node(nodes, {x, y, z}, Voxel(v), CHUNK_DEPTH, alloc) != SetVoxelErr::Ok) return 0;
            }
    return alloc.last_used_addr + 1;
}

// "Superflat" rule (stdrespack/world_gen.ron:214-249: height 12, layers grass x1 / dirt x3, earth
// limestone): y <= 8 limestone, 9..11 dirt, 12 grass, in world coordinates.
inline void fill_dense_superflat(ChunkPos cp, uint16_t *dense) {
    for (int32_t z = 0; z < 32; z++)
        for (int32_t y = 0; y < 32; y++) {
            const int32_t wy = cp.y * 32 + y;
            const uint16_t v = wy <= 8 ? vox::LIMESTONE : (wy <= 11 ? vox::DIRT : (wy == 12 ? vox::GRASS : vox::AIR));
            for (int32_t x = 0; x < 32; x++) dense[x + 32 * (y + 32 * z)] = v;
        }
}

}  // namespace vrt
