// regionfile.hpp — the reference server's on-disk region format, so that worlds saved by
// the reference game can be loaded into a ClientWorld and traced by this backend.
//
// servercli/src/main.rs:25-73: a file `regions/r_{x}_{y}_{z}_.data` (REGION_SIZE = 16 chunks per axis,
// common/src/world/mod.rs:25) is
//     bincode(RegionFileHeader { chunks: HashMap<[u32;3], Range<u32>> })  ++  raw little-endian u16 nodes
// with `bincode::config::standard()` of bincode 2.0.1 (Cargo.lock:201-203), a crates.io dependency that is not
// under /root/reference.  Its published encoding, restated: integers are varints — u < 251 one byte; 251 +
// u16 LE; 252 + u32 LE; 253 + u64 LE — a map is its length (u64 varint) then key/value pairs, a fixed-size
// array [u32;3] is its elements with no length prefix, a Range<u32> is `start` then `end`.  The key is the
// chunk's position inside the region, the range indexes the node array that follows the header.
// Parity is unpinned: the reference ships no region file or test vector; tests/test_regionfile.py pins the
// byte layout with hand-assembled headers and round trips.
#pragma once

#include <cstdint>
#include <map>
#include <optional>
#include <string>
#include <vector>

#include "world.hpp"

namespace vrt {

namespace bincode {
inline void put_varint(std::vector<uint8_t> &out, uint64_t v) {
    auto le = [&](int n) { for (int i = 0; i < n; i++) out.push_back((uint8_t)(v >> (8 * i))); };
    if (v < 251) out.push_back((uint8_t)v);
    else if (v <= 0xFFFF) { out.push_back(251); le(2); }
    else if (v <= 0xFFFFFFFFull) { out.push_back(252); le(4); }
    else { out.push_back(253); le(8); }
}
inline bool get_varint(const uint8_t *p, size_t n, size_t &pos, uint64_t &v) {
    if (pos >= n) return false;
    const uint8_t tag = p[pos++];
    int len;
    if (tag < 251) { v = tag; return true; }
    else if (tag == 251) len = 2;
    else if (tag == 252) len = 4;
    else if (tag == 253) len = 8;
    else return false;  // 254 = u128, 255 reserved: never produced for these fields
    if (pos + (size_t)len > n) return false;
    v = 0;
    for (int i = 0; i < len; i++) v |= (uint64_t)p[pos + i] << (8 * i);
    pos += (size_t)len;
    return true;
}
}  // namespace bincode

// RegionFile, servercli/src/main.rs:45-73
struct RegionFile {
    std::map<std::array<uint32_t, 3>, NodeRange> chunks;  // header (a HashMap in the reference: order-free)
    std::vector<Node> nodes;

    void append_chunk(std::array<uint32_t, 3> pos_in_region, const Node *chunk, uint32_t n) {  // :51-55
        const uint32_t s = (uint32_t)nodes.size();
        chunks[pos_in_region] = {s, s + n};
        nodes.insert(nodes.end(), chunk, chunk + n);
    }
    // read_chunk_data, :56-59
    const Node *read_chunk_data(std::array<uint32_t, 3> pos_in_region, uint32_t &n) const {
        auto it = chunks.find(pos_in_region);
        if (it == chunks.end() || it->second.end > nodes.size() || it->second.end < it->second.start) return nullptr;
        n = it->second.end - it->second.start;
        return nodes.data() + it->second.start;
    }
    // from_file, :65-69
    static std::optional<RegionFile> from_file(const uint8_t *bytes, size_t n) {
        RegionFile r;
        size_t pos = 0;
        uint64_t count;
        if (!bincode::get_varint(bytes, n, pos, count) || count > (1u << 20)) return std::nullopt;
        for (uint64_t i = 0; i < count; i++) {
            uint64_t k[3], s, e;
            for (auto &c : k)
                if (!bincode::get_varint(bytes, n, pos, c) || c > 0xFFFFFFFFull) return std::nullopt;
            if (!bincode::get_varint(bytes, n, pos, s) || !bincode::get_varint(bytes, n, pos, e)) return std::nullopt;
            if (s > 0xFFFFFFFFull || e > 0xFFFFFFFFull) return std::nullopt;
            r.chunks[{(uint32_t)k[0], (uint32_t)k[1], (uint32_t)k[2]}] = {(uint32_t)s, (uint32_t)e};
        }
        if ((n - pos) % sizeof(Node)) return std::nullopt;  // node_slice_from_bytes asserts this (:30)
        r.nodes.resize((n - pos) / sizeof(Node));
        for (size_t i = 0; i < r.nodes.size(); i++) r.nodes[i].w = (uint16_t)(bytes[pos + 2 * i] | (bytes[pos + 2 * i + 1] << 8));
        return r;
    }
    // to_file, :70-75
    std::vector<uint8_t> to_file() const {
        std::vector<uint8_t> out;
        bincode::put_varint(out, chunks.size());
        for (auto &kv : chunks) {
            for (uint32_t c : kv.first) bincode::put_varint(out, c);
            bincode::put_varint(out, kv.second.start);
            bincode::put_varint(out, kv.second.end);
        }
        for (const Node &nd : nodes) { out.push_back((uint8_t)(nd.w & 0xFF)); out.push_back((uint8_t)(nd.w >> 8)); }
        return out;
    }
};

// region_path_by_pos, :25-27
inline std::string region_file_name(IVec3 region_pos) {
    return "regions/r_" + std::to_string(region_pos.x) + "_" + std::to_string(region_pos.y) + "_" + std::to_string(region_pos.z) + "_.data";
}

// ChunkPos::region, common/src/world/mod.rs:90-96
inline std::pair<IVec3, std::array<uint32_t, 3>> chunk_region(ChunkPos p) {
    const int32_t r = (int32_t)REGION_SIZE;
    const IVec3 rp{div_euclid(p.x, r), div_euclid(p.y, r), div_euclid(p.z, r)};
    return {rp, {(uint32_t)(p.x - rp.x * r), (uint32_t)(p.y - rp.y * r), (uint32_t)(p.z - rp.z * r)}};
}

}  // namespace vrt
