// world.hpp — C++ host mirror of the reference's SVO data model and client world.
//
// Same names, argument meaning and error behaviour as the Rust types so that code written against the
// reference reads the same here (the image has no Rust toolchain; INTEGRATION.md has the extern "C"
// block for a Rust host):
//   Voxel, Node, NodeAlloc, Svo            <- common/src/world/mod.rs:137-194, 213-313, 323-471
//   Chunk, ChunkGrid, ChunkAlloc, ClientWorld <- client/src/world.rs:6-42, 44-201, 203-257, 259-367
// The implementation is not a translation: the octree walk runs on the integer voxel coordinate with an
// explicit path (no float centres, no re-walk from the root to find a parent); node arrays and
// allocator state come out identical to the reference's algorithm (tests/test_host_world.py checks
// this against the oracle's restatement).
#pragma once

#include <array>
#include <atomic>
#include <cstdint>
#include <cstring>
#include <optional>
#include <utility>
#include <vector>

namespace vrt {

using NodeAddr = uint32_t;
struct NodeRange {
    NodeAddr start = 0, end = 0;
    uint32_t len() const { return end > start ? end - start : 0; }
};

constexpr uint32_t CHUNK_SIZE = 32;          // common/src/world/mod.rs:10
constexpr uint32_t CHUNK_DEPTH = 5;          // :14
constexpr uint32_t NODES_PER_CHUNK = 37449;  // :18
constexpr uint32_t CHUNK_INIT_FREE_MEM = 2048;  // :23
constexpr uint32_t REGION_SIZE = 16;         // :25

struct IVec3 {
    int32_t x = 0, y = 0, z = 0;
    bool operator==(const IVec3 &o) const { return x == o.x && y == o.y && z == o.z; }
};
struct UVec3 { uint32_t x = 0, y = 0, z = 0; };

using VoxelPos = IVec3;  // GlobalPos<1>
using ChunkPos = IVec3;  // GlobalPos<CHUNK_SIZE>

inline int32_t div_euclid(int32_t a, int32_t b) { int32_t q = a / b; return (a % b < 0) ? q - 1 : q; }

// mod.rs:129-135, plus BadChunkData (build-defined): a chunk payload whose child indices leave its own node array — the
// reference panics on the slice bound when such a chunk is walked; here it is refused when it arrives (create_chunk)
enum class SetVoxelErr : int { Ok = 0, PosOutOfBounds = 1, OutOfMemory = 2, NoChunk = 3, NoChange = 4, BadChunkData = 5 };

// Voxel, mod.rs:137-148
struct Voxel {
    uint16_t v = 0;
    static constexpr uint16_t MAX_VALUE = 0xFFFF / 2;
    constexpr Voxel() = default;
    constexpr explicit Voxel(uint16_t d) : v(d) {}
    static constexpr Voxel from_data(uint16_t d) { return Voxel(d); }
    constexpr uint16_t as_data() const { return v; }
    constexpr bool is_empty() const { return v == 0; }
    constexpr bool operator==(Voxel o) const { return v == o.v; }
};

// Node, mod.rs:150-194: bit 15 = split; low 15 bits = voxel id or first-child index (chunk-relative)
struct Node {
    uint16_t w = 0;
    static constexpr uint16_t SPLIT_MASK = 0x8000, DATA_MASK = 0x7FFF;
    constexpr Node() = default;
    static constexpr Node make(Voxel vox) { Node n; n.w = vox.as_data() & DATA_MASK; return n; }
    static constexpr Node new_split(uint16_t child_idx) { Node n; n.w = child_idx | SPLIT_MASK; return n; }
    constexpr Voxel voxel() const { return Voxel(w & DATA_MASK); }
    constexpr bool is_split() const { return (w & SPLIT_MASK) != 0; }
    constexpr uint16_t child_idx() const { return w & DATA_MASK; }
    constexpr bool operator==(Node o) const { return w == o.w; }
};
static_assert(sizeof(Node) == 2, "Node is a transparent u16");

// NodeAlloc, mod.rs:213-313: hands out 8-node blocks, lowest address first.
class NodeAlloc {
public:
    NodeRange range;
    std::vector<NodeRange> free_mem;
    NodeAddr last_used_addr = 0;

    NodeAlloc() = default;
    NodeAlloc(NodeRange used, NodeRange free) {  // :224-231 (used.end == free.start)
        range = {used.start, free.end};
        free_mem.push_back(free);
        last_used_addr = used.end - 1;
    }
    void move_end(NodeAddr new_end) {  // :234-242
        for (auto &f : free_mem)
            if (f.end == range.end) { f.end = new_end; break; }
        range.end = new_end;
    }
    uint32_t total_free_mem() const { uint32_t t = 0; for (auto &f : free_mem) t += f.len(); return t; }
    uint32_t total_used_mem() const { return range.end - total_free_mem(); }
    std::optional<NodeAddr> next() {  // :275-286
        int i = find_next();
        if (i < 0) return std::nullopt;
        NodeRange &f = free_mem[(size_t)i];
        const NodeAddr result = f.start;
        f.start += 8;
        if (f.start + 1 == f.end) free_mem.erase(free_mem.begin() + i);
        if (result + 7 > last_used_addr) last_used_addr = result + 7;
        return result;
    }
    std::optional<NodeAddr> peek() const {  // :288-291
        int i = find_next();
        if (i < 0) return std::nullopt;
        return free_mem[(size_t)i].start;
    }
    void free(NodeAddr addr) {  // :293-307
        for (auto &f : free_mem) {
            if (f.start == addr + 8) { f.start -= 8; return; }
            if (f.end == addr) { f.end += 8; return; }
        }
        free_mem.push_back({addr, addr + 8});
    }

private:
    int find_next() const {  // :255-273
        int best = -1;
        NodeAddr best_addr = UINT32_MAX;
        for (size_t i = 0; i < free_mem.size(); i++) {
            if (free_mem[i].len() < 8) continue;
            if (free_mem[i].start < best_addr) { best_addr = free_mem[i].start; best = (int)i; }
        }
        return best;
    }
};

struct FoundNode {  // mod.rs:315-321 (center kept as the integer min corner + size)
    NodeAddr idx = 0;
    uint32_t depth = 0;
    uint32_t size = 0;
    UVec3 min;
};

// Svo, mod.rs:323-460.  The reference's `pos >= center` compares are bits of the integer coordinate:
// at depth d (node size 32>>d) the octant is bit (log2(size)-1) of each axis.
struct Svo {
    NodeAddr root = 0;
    uint32_t size = CHUNK_SIZE;  // power of two

    static uint32_t log2u(uint32_t s) { uint32_t l = 0; while ((1u << l) < s) l++; return l; }

    // find_node, :366-395; optionally records the path (node index at every depth walked)
    FoundNode find_node(const Node *nodes, UVec3 pos, uint32_t max_depth, NodeAddr *path = nullptr) const {
        uint32_t sz = size, depth = 0;
        NodeAddr idx = root;
        uint32_t sh = log2u(size);
        for (;;) {
            if (path) path[depth] = idx;
            const Node n = nodes[idx];
            if (!n.is_split() || depth == max_depth) {
                FoundNode f;
                f.idx = idx; f.depth = depth; f.size = sz;
                const uint32_t m = ~(sz - 1u);
                f.min = {pos.x & m, pos.y & m, pos.z & m};
                return f;
            }
            sz >>= 1; sh -= 1;
            const uint32_t child = ((pos.x >> sh) & 1u) | (((pos.y >> sh) & 1u) << 1) | (((pos.z >> sh) & 1u) << 2);
            idx = (NodeAddr)n.child_idx() + child;
            depth += 1;
        }
    }

    // set_node, :397-459: split down to target_depth, write, then merge identical siblings upwards.
    SetVoxelErr set_node(Node *nodes, UVec3 pos, Voxel voxel, uint32_t target_depth, NodeAlloc &alloc) const {
        NodeAddr path[32];
        FoundNode node = find_node(nodes, pos, target_depth, path);
        const Voxel parent_voxel = nodes[node.idx].voxel();
        if (parent_voxel == voxel) return SetVoxelErr::Ok;

        uint32_t sh = log2u(size) - node.depth;
        while (node.depth < target_depth) {
            // `alloc.next().ok_or(OutOfMemory)?` (:414-415): running out of blocks returns mid-way, leaving the splits made
            // so far (a split node whose 8 children repeat its voxel: the same voxels).  The reference then asserts
            // first_child < Voxel::MAX_VALUE (:416; a chunk addresses < 32767 nodes) — a panic there, an error here, and
            // checked before the block is taken so that nothing leaks.
            const auto peeked = alloc.peek();
            if (!peeked || *peeked >= Voxel::MAX_VALUE) return SetVoxelErr::OutOfMemory;
            const auto first = alloc.next();
            for (int i = 0; i < 8; i++) nodes[*first + i] = Node::make(parent_voxel);
            nodes[node.idx] = Node::new_split((uint16_t)*first);
            sh -= 1;
            const uint32_t child = ((pos.x >> sh) & 1u) | (((pos.y >> sh) & 1u) << 1) | (((pos.z >> sh) & 1u) << 2);
            node.idx = *first + child;
            node.depth += 1;
            path[node.depth] = node.idx;
        }
        nodes[node.idx] = Node::make(voxel);

        for (uint32_t d = node.depth; d > 0; d--) {
            const NodeAddr parent_idx = path[d - 1];
            const NodeAddr base = nodes[parent_idx].child_idx();
            const Node *ch = nodes + base;
            bool eq = true;
            for (int i = 1; i < 8; i++) eq = eq && (ch[0] == ch[i]);
            if (!eq) break;
            alloc.free(base);
            nodes[parent_idx] = Node::make(voxel);
        }
        return SetVoxelErr::Ok;
    }
};

// Chunk, client/src/world.rs:6-42
struct Chunk {
    NodeRange range;  // absolute range in the pool
    NodeAlloc alloc;  // addresses relative to range.start

    static Chunk empty() { Chunk c; c.range = {0, 1}; c.alloc = NodeAlloc({0, 1}, {1, 1}); return c; }
    static Chunk make(NodeAddr root, NodeRange used, NodeRange free) {
        Chunk c;
        c.range = {root + used.start, root + free.end};
        c.alloc = NodeAlloc(used, free);
        return c;
    }
    SetVoxelErr set_voxel(Node *pool, UVec3 pos, Voxel v) { return Svo{0, CHUNK_SIZE}.set_node(pool + range.start, pos, v, CHUNK_DEPTH, alloc); }
    Voxel get_voxel(const Node *pool, UVec3 pos) const {
        const Node *n = pool + range.start;
        return n[Svo{0, CHUNK_SIZE}.find_node(n, pos, CHUNK_DEPTH).idx].voxel();
    }
};

// ChunkGrid, client/src/world.rs:44-201
class ChunkGrid {
public:
    ChunkGrid(ChunkPos center, uint32_t size_in_chunks) : size_(size_in_chunks) {
        chunks_.assign((size_t)size_ * size_ * size_, std::nullopt);
        roots_.assign(chunks_.size(), 0u);
        const int32_t h = (int32_t)size_ / 2;
        min_ = {center.x - h, center.y - h, center.z - h};
    }
    static size_t local_pos_to_idx(UVec3 p, uint32_t s) { return (size_t)p.x + (size_t)p.y * s + (size_t)p.z * s * s; }  // :94-97

    ChunkPos center_chunk() const { const int32_t h = (int32_t)size_ / 2; return {min_.x + h, min_.y + h, min_.z + h}; }
    std::optional<UVec3> local_pos_for(ChunkPos p) const {  // :99-105
        const ChunkPos mx = max_chunk();
        if (p.x < min_.x || p.y < min_.y || p.z < min_.z || p.x >= mx.x || p.y >= mx.y || p.z >= mx.z) return std::nullopt;
        return UVec3{(uint32_t)(p.x - min_.x), (uint32_t)(p.y - min_.y), (uint32_t)(p.z - min_.z)};
    }
    ChunkPos unlocal_pos_for(UVec3 p) const { return {(int32_t)p.x + min_.x, (int32_t)p.y + min_.y, (int32_t)p.z + min_.z}; }
    size_t chunk_count() const { return chunks_.size(); }
    uint32_t size_in_voxels() const { return size_ * CHUNK_SIZE; }
    uint32_t size_in_chunks() const { return size_; }
    VoxelPos min_voxel() const { return {min_.x * (int32_t)CHUNK_SIZE, min_.y * (int32_t)CHUNK_SIZE, min_.z * (int32_t)CHUNK_SIZE}; }
    // max_chunk().max() — the LAST voxel of the chunk one past the grid (:115), kept as the reference has it
    VoxelPos max_voxel() const {
        const ChunkPos m = max_chunk();
        const int32_t c = (int32_t)CHUNK_SIZE;
        return {m.x * c + c - 1, m.y * c + c - 1, m.z * c + c - 1};
    }
    ChunkPos min_chunk() const { return min_; }
    ChunkPos max_chunk() const { return {min_.x + (int32_t)size_, min_.y + (int32_t)size_, min_.z + (int32_t)size_}; }

    void resize(uint32_t size_in_chunks) {  // :58-88
        if (size_in_chunks == size_) return;
        ChunkGrid g(center_chunk(), size_in_chunks);
        for (uint32_t x = 0; x < size_; x++)
            for (uint32_t y = 0; y < size_; y++)
                for (uint32_t z = 0; z < size_; z++) {
                    auto &src = chunks_[local_pos_to_idx({x, y, z}, size_)];
                    if (!src) continue;
                    auto lp = g.local_pos_for(unlocal_pos_for({x, y, z}));
                    if (!lp) continue;
                    g.chunks_[local_pos_to_idx(*lp, size_in_chunks)] = std::move(src);
                }
        g.rebuild_roots();
        *this = std::move(g);
    }

    void shift_chunks(IVec3 off, std::vector<std::pair<ChunkPos, Chunk>> &removed) {  // :126-152
        std::vector<std::optional<Chunk>> nc(chunks_.size());
        const int32_t s = (int32_t)size_;
        for (uint32_t x = 0; x < size_; x++)
            for (uint32_t y = 0; y < size_; y++)
                for (uint32_t z = 0; z < size_; z++) {
                    auto &src = chunks_[local_pos_to_idx({x, y, z}, size_)];
                    const int32_t dx = (int32_t)x - off.x, dy = (int32_t)y - off.y, dz = (int32_t)z - off.z;
                    if (dx < 0 || dy < 0 || dz < 0 || dx >= s || dy >= s || dz >= s) {
                        if (src) removed.emplace_back(unlocal_pos_for({x, y, z}), std::move(*src));
                        continue;
                    }
                    nc[local_pos_to_idx({(uint32_t)dx, (uint32_t)dy, (uint32_t)dz}, size_)] = std::move(src);
                }
        chunks_ = std::move(nc);
        rebuild_roots();
    }

    // chunk_roots, :154-159: root of every cell, 0 (the permanent air leaf) for a missing chunk.  The reference builds a
    // fresh Vec every frame (main.rs:446: 27 000 entries at its 30^3 chunks); here the table is kept up to date by the calls
    // that change the grid (set_chunk: one entry; shift_chunks, resize: all of them), and roots_generation() changes with
    // every one of them — the tag of vrt_write_chunk_roots_tagged.  A plain read: nothing is built behind a const call.
    // The reference stays valid until resize() replaces the grid; its contents are the grid's as of the last mutating call.
    const std::vector<NodeAddr> &chunk_roots() const { return roots_; }
    uint64_t roots_generation() const { return roots_gen_; }
    size_t populated_count() const { size_t r = 0; for (auto &c : chunks_) r += c.has_value(); return r; }
    std::vector<ChunkPos> empty_chunks() const {  // :169-183
        std::vector<ChunkPos> out;
        for (uint32_t x = 0; x < size_; x++)
            for (uint32_t y = 0; y < size_; y++)
                for (uint32_t z = 0; z < size_; z++)
                    if (!chunks_[local_pos_to_idx({x, y, z}, size_)]) out.push_back(unlocal_pos_for({x, y, z}));
        return out;
    }
    bool set_chunk(ChunkPos p, Chunk c) {
        auto lp = local_pos_for(p);
        if (!lp) return false;
        const size_t i = local_pos_to_idx(*lp, size_);
        roots_[i] = c.range.start;
        chunks_[i] = std::move(c);
        touch_roots();
        return true;
    }
    const Chunk *get_chunk(ChunkPos p) const {
        auto lp = local_pos_for(p);
        if (!lp) return nullptr;
        auto &c = chunks_[local_pos_to_idx(*lp, size_)];
        return c ? &*c : nullptr;
    }
    Chunk *get_chunk_mut(ChunkPos p) { return const_cast<Chunk *>(static_cast<const ChunkGrid *>(this)->get_chunk(p)); }

protected:
    friend class ClientWorld;
    // (values never repeat, whichever grid object hands them out: resize() replaces the grid by a new one)
    static uint64_t next_generation() { static std::atomic<uint64_t> g{1}; return g.fetch_add(1) + 1; }
    void touch_roots() { roots_gen_ = next_generation(); }
    void rebuild_roots() {
        roots_.resize(chunks_.size());
        for (size_t i = 0; i < chunks_.size(); i++) roots_[i] = chunks_[i] ? chunks_[i]->range.start : 0u;
        touch_roots();
    }
    ChunkPos min_;
    std::vector<std::optional<Chunk>> chunks_;
    uint32_t size_;
    std::vector<NodeAddr> roots_;
    uint64_t roots_gen_ = next_generation();
};

// ChunkAlloc, client/src/world.rs:203-257: first-fit over spans of the flat pool; slot 0 is reserved.
class ChunkAlloc {
public:
    explicit ChunkAlloc(uint32_t max_nodes) : max_nodes_(max_nodes) { free_mem_.push_back({1, max_nodes}); }
    std::pair<uint32_t, uint32_t> status() const { uint32_t t = 0; for (auto &f : free_mem_) t += f.len(); return {t, max_nodes_}; }
    void free_chunk(uint32_t root, uint32_t size) {  // :223-237
        for (auto &f : free_mem_) {
            if (f.start == root + size) { f.start -= size; return; }
            if (f.end == root) { f.end += size; return; }
        }
        free_mem_.push_back({root, root + size});
    }
    // :239-256; the reference panics when nothing fits — here the caller gets nullopt
    std::optional<Chunk> alloc_chunk(uint32_t size) {
        const uint32_t req = size + CHUNK_INIT_FREE_MEM;
        for (auto &f : free_mem_) {
            if (f.end - f.start >= req) {
                const NodeAddr s = f.start;
                f.start = s + req;
                return Chunk::make(s, {0, size}, {size, req});
            }
        }
        return std::nullopt;
    }
    const std::vector<NodeRange> &free_mem() const { return free_mem_; }

private:
    std::vector<NodeRange> free_mem_;
    uint32_t max_nodes_;
};

// ClientWorld, client/src/world.rs:259-367 (derefs to its ChunkGrid in the reference; inherits here)
class ClientWorld : public ChunkGrid {
public:
    ClientWorld(ChunkPos center, uint32_t max_nodes, uint32_t size)
        : ChunkGrid(center, size), nodes_(max_nodes), chunk_alloc_(max_nodes) {
        nodes_[0] = Node::make(Voxel(0));  // 0 = air (:274)
    }
    void free_chunk(const Chunk &c) { chunk_alloc_.free_chunk(c.range.start, c.range.len()); }
    std::pair<uint32_t, uint32_t> chunk_alloc_status() const { return chunk_alloc_.status(); }
    const Node *nodes() const { return nodes_.data(); }
    Node *nodes_mut() { return nodes_.data(); }
    uint32_t max_nodes() const { return (uint32_t)nodes_.size(); }

    // :297-308
    void center_chunks(ChunkPos anchor, std::vector<std::pair<ChunkPos, Chunk>> &removed) {
        const int32_t h = (int32_t)size_in_chunks() / 2;
        const ChunkPos nm{anchor.x - h, anchor.y - h, anchor.z - h};
        if (nm == min_) return;
        const IVec3 off{nm.x - min_.x, nm.y - min_.y, nm.z - min_.z};
        min_ = nm;
        shift_chunks(off, removed);
    }

    // A chunk payload is walked with its own child indices (Svo::find_node / set_node index nodes[child_idx + octant]):
    // every split node must point at 8 nodes inside the payload, and a payload is at least its root.  Payloads come
    // from the network (GiveChunkData) and from region files, i.e. they are untrusted.
    static bool chunk_payload_ok(const Node *src, uint32_t n) {
        if (n == 0 || n > (uint32_t)Node::DATA_MASK + 8u) return false;
        for (uint32_t i = 0; i < n; i++)
            if (src[i].is_split() && (uint32_t)src[i].child_idx() + 8u > n) return false;
        return true;
    }

    // :310-335. Returns the absolute root; err set on failure.
    NodeAddr create_chunk(ChunkPos pos, const Node *src, uint32_t n, SetVoxelErr &err) {
        err = SetVoxelErr::Ok;
        if (!local_pos_for(pos)) { err = SetVoxelErr::PosOutOfBounds; return 0; }
        if (!chunk_payload_ok(src, n)) { err = SetVoxelErr::BadChunkData; return 0; }
        if (Chunk *c = get_chunk_mut(pos)) {
            if (c->range.len() >= n) {
                std::memcpy(nodes_.data() + c->range.start, src, (size_t)n * sizeof(Node));
                c->alloc = NodeAlloc({0, n}, {n, c->range.len()});
                return c->range.start;
            }
        }
        auto chunk = chunk_alloc_.alloc_chunk(n);
        if (!chunk) { err = SetVoxelErr::OutOfMemory; return 0; }  // reference: panic (:251)
        const NodeAddr start = chunk->range.start;
        std::memcpy(nodes_.data() + start, src, (size_t)n * sizeof(Node));
        set_chunk(pos, std::move(*chunk));
        return start;
    }

    SetVoxelErr check_bounds(VoxelPos p) const {  // :337-342
        const VoxelPos a = min_voxel(), b = max_voxel();
        if (p.x < a.x || p.y < a.y || p.z < a.z || p.x >= b.x || p.y >= b.y || p.z >= b.z) return SetVoxelErr::PosOutOfBounds;
        return SetVoxelErr::Ok;
    }
    static std::pair<ChunkPos, UVec3> split_pos(VoxelPos p) {  // VoxelPos::chunk, mod.rs:82-89
        const int32_t c = (int32_t)CHUNK_SIZE;
        const ChunkPos cp{div_euclid(p.x, c), div_euclid(p.y, c), div_euclid(p.z, c)};
        return {cp, {(uint32_t)(p.x - cp.x * c), (uint32_t)(p.y - cp.y * c), (uint32_t)(p.z - cp.z * c)}};
    }
    // :344-350; *chunk_out is the chunk the edit went to (its whole range is what the caller re-uploads) — also when the
    // edit ran out of memory half-way: the splits made until then are in the pool (Svo::set_node)
    SetVoxelErr set_voxel(VoxelPos p, Voxel v, const Chunk **chunk_out = nullptr) {
        if (auto e = check_bounds(p); e != SetVoxelErr::Ok) return e;
        auto [cp, lp] = split_pos(p);
        Chunk *c = get_chunk_mut(cp);
        if (!c) return SetVoxelErr::NoChunk;
        if (chunk_out) *chunk_out = c;
        return c->set_voxel(nodes_.data(), lp, v);
    }
    SetVoxelErr get_voxel(VoxelPos p, Voxel &out) const {  // :352-357
        if (auto e = check_bounds(p); e != SetVoxelErr::Ok) return e;
        auto [cp, lp] = split_pos(p);
        const Chunk *c = get_chunk(cp);
        if (!c) return SetVoxelErr::NoChunk;
        out = c->get_voxel(nodes_.data(), lp);
        return SetVoxelErr::Ok;
    }
    std::optional<int32_t> highest_vox_at(VoxelPos p) const {  // :359-366
        for (int32_t y = max_voxel().y - 1; y >= min_voxel().y; y--) {
            Voxel v;
            if (get_voxel({p.x, y, p.z}, v) == SetVoxelErr::Ok && !v.is_empty()) return y;
        }
        return std::nullopt;
    }

private:
    std::vector<Node> nodes_;
    ChunkAlloc chunk_alloc_;
};

}  // namespace vrt
