// netmsg.hpp — the one network payload that feeds the path: `ClientCmd::GiveChunkData(ChunkPos, Cow<[Node]>,
// NodeAlloc)` (common/src/net.rs:46-55), the server's answer to LoadChunks (server/src/lib.rs:229,292) that the
// client turns into `world.create_chunk(pos, &nodes)` + a range upload of the node buffer
// (client/src/lib.rs:110-118, clientdesktop/src/main.rs:280-295).  SURVEY.md §8f N3.
//
// Wire form: `bincode::serde::encode_to_vec(&cmd, bincode::config::standard())` of bincode 2.0.1 (client/src/net.rs:38,
// server/src/net.rs:60-64; Cargo.lock:201-203), a crates.io dependency that is not under /root/reference.  Its
// published serde encoding, restated: an enum is its variant index as a u32 varint (GiveChunkData is the sixth
// variant: 5) followed by the fields; unsigned integers are varints (u < 251 one byte; 251 + u16 LE; 252 + u32 LE;
// 253 + u64 LE), signed ones zig-zag first; a newtype struct is its field (ChunkPos = GlobalPos(IVec3), Node(u16));
// glam's IVec3 serialises as a 3-tuple of i32; a slice / Vec is its length as a u64 varint then the elements; a
// struct is its fields in declaration order (NodeAlloc: range, free_mem, last_used_addr — common/src/world/mod.rs:
// 213-222); a Range<u32> is start, end.  No length prefix around a message: the receiver decodes from its byte
// queue and drops what was consumed (client/src/net.rs:52-57, UnexpectedEnd = wait for more).
// Parity is unpinned: the reference holds no captured traffic; tests/test_netmsg.py pins the bytes with
// hand-assembled messages and round trips.
#pragma once

#include <cstdint>
#include <optional>
#include <vector>

#include "regionfile.hpp"  // bincode::put_varint / get_varint
#include "world.hpp"

namespace vrt {

namespace bincode {
inline void put_zigzag(std::vector<uint8_t> &out, int64_t v) { put_varint(out, ((uint64_t)v << 1) ^ (uint64_t)(v >> 63)); }
inline bool get_zigzag(const uint8_t *p, size_t n, size_t &pos, int64_t &v) {
    uint64_t u;
    if (!get_varint(p, n, pos, u)) return false;
    v = (int64_t)(u >> 1) ^ -(int64_t)(u & 1);
    return true;
}
}  // namespace bincode

constexpr uint32_t kClientCmdGiveChunkData = 5;  // index of the variant in `enum ClientCmd` (common/src/net.rs:46-55)

struct GiveChunkData {
    ChunkPos pos{0, 0, 0};
    std::vector<Node> nodes;
    // NodeAlloc as sent (addresses relative to the chunk root); the client ignores it (lib.rs:112 `_node_alloc`)
    NodeRange range{0, 0};
    std::vector<NodeRange> free_mem;
    NodeAddr last_used_addr = 0;

    std::vector<uint8_t> encode() const {
        std::vector<uint8_t> out;
        bincode::put_varint(out, kClientCmdGiveChunkData);
        bincode::put_zigzag(out, pos.x);
        bincode::put_zigzag(out, pos.y);
        bincode::put_zigzag(out, pos.z);
        bincode::put_varint(out, nodes.size());
        for (const Node &nd : nodes) bincode::put_varint(out, nd.w);
        bincode::put_varint(out, range.start);
        bincode::put_varint(out, range.end);
        bincode::put_varint(out, free_mem.size());
        for (const NodeRange &r : free_mem) { bincode::put_varint(out, r.start); bincode::put_varint(out, r.end); }
        bincode::put_varint(out, last_used_addr);
        return out;
    }

    enum class Status { Ok, NeedMore, NotChunkData, Malformed };

    // Decode one message from the front of a byte queue. `consumed` is set on Ok.
    static Status decode(const uint8_t *p, size_t n, GiveChunkData &m, size_t &consumed) {
        size_t pos = 0;
        uint64_t u;
        // the tag byte says how many bytes a varint needs; running out of them is "wait for more", not an error
        auto var = [&](uint64_t &v) -> Status {
            if (pos >= n) return Status::NeedMore;
            const uint8_t tag = p[pos];
            const size_t need = tag < 251 ? 1 : tag == 251 ? 3 : tag == 252 ? 5 : tag == 253 ? 9 : 0;
            if (!need) return Status::Malformed;
            if (pos + need > n) return Status::NeedMore;
            return bincode::get_varint(p, n, pos, v) ? Status::Ok : Status::Malformed;
        };
#define VRT_VAR(v, limit)                                     \
    if (Status s_ = var(v); s_ != Status::Ok) return s_;      \
    if ((v) > (limit)) return Status::Malformed;
        VRT_VAR(u, 0xFFFFFFFFull)
        if (u != kClientCmdGiveChunkData) return u <= 6 ? Status::NotChunkData : Status::Malformed;
        int32_t xyz[3];
        for (int i = 0; i < 3; i++) {
            VRT_VAR(u, 0xFFFFFFFFull)  // a zig-zagged i32
            xyz[i] = (int32_t)((uint32_t)(u >> 1) ^ (0u - (uint32_t)(u & 1)));
        }
        m.pos = ChunkPos{xyz[0], xyz[1], xyz[2]};
        VRT_VAR(u, (uint64_t)NODES_PER_CHUNK + 4096)  // a chunk addresses <= 32 767 nodes (+ slack the server may send)
        const size_t count = (size_t)u;
        m.nodes.resize(count);
        for (size_t i = 0; i < count; i++) {
            VRT_VAR(u, 0xFFFFull)
            m.nodes[i].w = (uint16_t)u;
        }
        VRT_VAR(u, 0xFFFFFFFFull)
        m.range.start = (uint32_t)u;
        VRT_VAR(u, 0xFFFFFFFFull)
        m.range.end = (uint32_t)u;
        VRT_VAR(u, 1u << 20)
        m.free_mem.resize((size_t)u);
        for (NodeRange &r : m.free_mem) {
            VRT_VAR(u, 0xFFFFFFFFull)
            r.start = (uint32_t)u;
            VRT_VAR(u, 0xFFFFFFFFull)
            r.end = (uint32_t)u;
        }
        VRT_VAR(u, 0xFFFFFFFFull)
        m.last_used_addr = (uint32_t)u;
#undef VRT_VAR
        consumed = pos;
        return Status::Ok;
    }
};

}  // namespace vrt
