// host_capi.cpp — include/vrt_host.h over the C++ host mirror (world.hpp, graphics.hpp, worldgen.hpp).
#include <algorithm>
#include <atomic>
#include <system_error>
#include <thread>

#include "../../../include/vrt_host.h"
#include "graphics.hpp"
#include "materials.hpp"
#include "netmsg.hpp"
#include "regionfile.hpp"
#include "worldgen.hpp"

using namespace vrt;

struct vrth_world {
    ClientWorld w;
    vrth_world(ChunkPos c, uint32_t max_nodes, uint32_t size) : w(c, max_nodes, size) {}
};

static ChunkPos cp3(const int32_t p[3]) { return {p[0], p[1], p[2]}; }

extern "C" {

vrth_world *vrth_world_new(const int32_t center_chunk[3], uint32_t max_nodes, uint32_t size_in_chunks) {
    if (!center_chunk || max_nodes < 2 || size_in_chunks == 0) return nullptr;
    try {
        return new vrth_world(cp3(center_chunk), max_nodes, size_in_chunks);
    } catch (...) {
        return nullptr;
    }
}

void vrth_world_free(vrth_world *w) { delete w; }

int vrth_world_create_chunk(vrth_world *w, const int32_t chunk_pos[3], const uint16_t *nodes, uint32_t n, uint32_t *root_out) {
    SetVoxelErr err;
    const NodeAddr root = w->w.create_chunk(cp3(chunk_pos), reinterpret_cast<const Node *>(nodes), n, err);
    if (root_out) *root_out = root;
    return (int)err;
}

int vrth_world_set_voxel(vrth_world *w, const int32_t p[3], uint16_t voxel, uint32_t *range_start, uint32_t *range_len) {
    // GameState::set_voxel (client/src/lib.rs:67-76): NoChange short-circuit, then ClientWorld::set_voxel
    Voxel cur;
    if (auto e = w->w.get_voxel(cp3(p), cur); e != SetVoxelErr::Ok) return (int)e;
    if (cur == Voxel(voxel)) return (int)SetVoxelErr::NoChange;
    const Chunk *c = nullptr;
    const SetVoxelErr e = w->w.set_voxel(cp3(p), Voxel(voxel), &c);
    if ((e == SetVoxelErr::Ok || e == SetVoxelErr::OutOfMemory) && c) {  // OutOfMemory: the partial split is in the pool
        if (range_start) *range_start = c->range.start;
        if (range_len) *range_len = c->range.len();
    }
    return (int)e;
}

int vrth_world_get_voxel(const vrth_world *w, const int32_t p[3], uint16_t *voxel_out) {
    Voxel v;
    const SetVoxelErr e = w->w.get_voxel(cp3(p), v);
    if (e == SetVoxelErr::Ok && voxel_out) *voxel_out = v.as_data();
    return (int)e;
}

uint32_t vrth_world_center_chunks(vrth_world *w, const int32_t anchor_chunk[3]) {
    std::vector<std::pair<ChunkPos, Chunk>> removed;
    w->w.center_chunks(cp3(anchor_chunk), removed);
    for (auto &rc : removed) w->w.free_chunk(rc.second);
    return (uint32_t)removed.size();
}

void vrth_world_resize(vrth_world *w, uint32_t size_in_chunks) { w->w.resize(size_in_chunks); }

const uint16_t *vrth_world_nodes(const vrth_world *w) { return reinterpret_cast<const uint16_t *>(w->w.nodes()); }
uint32_t vrth_world_max_nodes(const vrth_world *w) { return w->w.max_nodes(); }

uint32_t vrth_world_chunk_roots(const vrth_world *w, uint32_t *out, uint32_t cap) {
    if (!out) return (uint32_t)w->w.chunk_count();
    const std::vector<NodeAddr> &r = w->w.chunk_roots();
    std::copy_n(r.begin(), std::min<size_t>(cap, r.size()), out);
    return (uint32_t)r.size();
}

const uint32_t *vrth_world_chunk_roots_ptr(const vrth_world *w) { return w->w.chunk_roots().data(); }
uint64_t vrth_world_chunk_roots_generation(const vrth_world *w) { return w->w.roots_generation(); }

void vrth_world_info(const vrth_world *w, int32_t min_voxel[3], uint32_t *size_in_voxels, uint32_t *size_in_chunks, uint32_t *populated) {
    const VoxelPos m = w->w.min_voxel();
    if (min_voxel) { min_voxel[0] = m.x; min_voxel[1] = m.y; min_voxel[2] = m.z; }
    if (size_in_voxels) *size_in_voxels = w->w.size_in_voxels();
    if (size_in_chunks) *size_in_chunks = w->w.size_in_chunks();
    if (populated) *populated = (uint32_t)w->w.populated_count();
}

void vrth_world_alloc_status(const vrth_world *w, uint32_t *free_nodes, uint32_t *max_nodes) {
    const auto s = w->w.chunk_alloc_status();
    if (free_nodes) *free_nodes = s.first;
    if (max_nodes) *max_nodes = s.second;
}

int vrth_world_chunk_state(const vrth_world *w, const int32_t chunk_pos[3], uint32_t *range_start, uint32_t *range_end,
                           uint32_t *last_used_addr, uint32_t *spans, uint32_t cap_spans) {
    const Chunk *c = w->w.get_chunk(cp3(chunk_pos));
    if (!c) return -1;
    if (range_start) *range_start = c->range.start;
    if (range_end) *range_end = c->range.end;
    if (last_used_addr) *last_used_addr = c->alloc.last_used_addr;
    const auto &fm = c->alloc.free_mem;
    for (size_t i = 0; i < fm.size() && i < cap_spans; i++) {
        spans[2 * i] = fm[i].start;
        spans[2 * i + 1] = fm[i].end;
    }
    return (int)fm.size();
}

int vrth_world_highest_vox_at(const vrth_world *w, int32_t x, int32_t z, int32_t *y_out) {
    const auto y = w->w.highest_vox_at({x, 0, z});
    if (!y) return 0;
    if (y_out) *y_out = *y;
    return 1;
}

void vrth_world_data_from(const vrth_world *w, vrt_world_data *out) { *out = WorldData::from(w->w); }

void vrth_cam_data_create(const float rot_deg[3], const float eye[3], float fov_deg, const float proj_size[2], vrt_cam_data *out) {
    *out = CamData::create({rot_deg[0], rot_deg[1], rot_deg[2]}, {eye[0], eye[1], eye[2]}, fov_deg, {proj_size[0], proj_size[1]});
}

void vrth_axis_rot_to_ray(const float rot_rad[3], float out[3]) {
    const float r = std::cos(rot_rad[0]);
    out[0] = r * -std::sin(rot_rad[1]);
    out[2] = r * -std::cos(rot_rad[1]);
    out[1] = -std::sin(rot_rad[0]);
}

void vrth_std_materials(vrt_material *out256) {
    std::memset(out256, 0, 256 * sizeof(vrt_material));  // Material::ZERO for unnamed slots
    for (int i = 0; i < kStdVoxelCount; i++) {
        out256[i].color[0] = kStdVoxels[i].r;
        out256[i].color[1] = kStdVoxels[i].g;
        out256[i].color[2] = kStdVoxels[i].b;
        out256[i].is_empty = kStdVoxels[i].is_empty;
        out256[i].is_liquid = kStdVoxels[i].is_liquid;
        out256[i].scatter = 0.0f;
    }
}

const char *vrth_std_voxel_name(uint32_t id) { return id < (uint32_t)kStdVoxelCount ? kStdVoxels[id].name : nullptr; }

uint32_t vrth_svo_build_by_set_node(const uint16_t *dense, uint16_t *nodes, uint32_t cap) {
    return build_svo_by_set_node(dense, reinterpret_cast<Node *>(nodes), cap);
}

uint32_t vrth_svo_build_bottom_up(const uint16_t *dense, uint16_t *nodes, uint32_t cap) {
    std::vector<Node> out;
    if (!build_svo_bottom_up(dense, out) || out.size() > cap) return 0;
    std::memcpy(nodes, out.data(), out.size() * sizeof(Node));
    return (uint32_t)out.size();
}

void vrth_svo_to_dense(const uint16_t *nodes, uint16_t *dense) {
    const Node *n = reinterpret_cast<const Node *>(nodes);
    const Svo svo{0, CHUNK_SIZE};
    for (uint32_t z = 0; z < 32; z++)
        for (uint32_t y = 0; y < 32; y++)
            for (uint32_t x = 0; x < 32; x++) dense[x + 32 * (y + 32 * z)] = n[svo.find_node(n, {x, y, z}, CHUNK_DEPTH).idx].voxel().as_data();
}

int32_t vrth_gen_height(uint32_t seed, int32_t x, int32_t z) {
    WorldGen g;
    g.seed = seed;
    return g.height(x, z);
}

int vrth_gen_dense(uint32_t seed, const int32_t chunk_pos[3], uint16_t *dense) {
    WorldGen g;
    g.seed = seed;
    return g.fill_dense(cp3(chunk_pos), dense) ? 1 : 0;
}

void vrth_gen_dense_superflat(const int32_t chunk_pos[3], uint16_t *dense) { fill_dense_superflat(cp3(chunk_pos), dense); }

// Shared by vrth_world_generate (every cell of the grid) and vrth_world_generate_missing (the empty cells only: what the
// server sends back for request_missing_chunks, client/src/lib.rs:80-108, after the grid moved).  `ranges`: (root, count)
// of every chunk created, in grid order — the ranges GameState::process_cmd hands to NodeBuffer::write (main.rs:289-295).
static int generate_impl(vrth_world *w, uint32_t kind, uint32_t seed, int threads, bool only_missing, uint32_t *ranges, uint32_t cap,
                         uint32_t *n_ranges) {
    const uint32_t S = w->w.size_in_chunks();
    const size_t total = (size_t)S * S * S;
    std::vector<std::vector<Node>> built(total);
    std::vector<uint8_t> skip(total, 0);
    if (only_missing) {
        const ChunkPos mn0 = w->w.min_chunk();
        for (size_t i = 0; i < total; i++) {
            const ChunkPos cp{mn0.x + (int32_t)(i % S), mn0.y + (int32_t)((i / S) % S), mn0.z + (int32_t)(i / ((size_t)S * S))};
            skip[i] = w->w.get_chunk(cp) != nullptr;
        }
    }
    std::atomic<size_t> next{0};
    std::atomic<int> failed{0};
    // (at most 16 workers unless asked for more: a chunk is ~0.1 ms of work, and a host with hundreds of hardware threads —
    // or a limit on tasks per process — gains nothing from one thread per core here)
    unsigned nt = threads > 0 ? (unsigned)threads : std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    nt = (unsigned)std::min<size_t>(nt, total);
    WorldGen g;
    g.seed = seed;
    const ChunkPos mn = w->w.min_chunk();
    auto work = [&]() {
        std::vector<uint16_t> dense(32768);
        std::vector<Node> scratch(NODES_PER_CHUNK + 64);
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= total) return;
            if (skip[i]) continue;
            const ChunkPos cp{mn.x + (int32_t)(i % S), mn.y + (int32_t)((i / S) % S), mn.z + (int32_t)(i / ((size_t)S * S))};
            if (kind == 1) {
                fill_dense_superflat(cp, dense.data());
                // C1 is built the reference's way (set_node, x -> z -> y), holes and all
                const uint32_t used = build_svo_by_set_node(dense.data(), scratch.data(), (uint32_t)scratch.size());
                if (!used) { failed = (int)SetVoxelErr::OutOfMemory; return; }
                built[i].assign(scratch.begin(), scratch.begin() + used);
            } else {
                const bool uniform = g.fill_dense(cp, dense.data());
                if (uniform) built[i].assign(1, Node::make(Voxel(dense[0])));
                else if (!build_svo_bottom_up(dense.data(), built[i])) { failed = (int)SetVoxelErr::OutOfMemory; return; }
            }
        }
    };
    std::vector<std::thread> pool;
    try {
        for (unsigned t = 1; t < nt; t++) pool.emplace_back(work);
    } catch (const std::system_error &) {
        // no more threads to be had: the ones that started (and this one) do the work
    }
    work();
    for (auto &t : pool) t.join();
    if (failed) return failed;
    // create_chunk in grid order (x fastest), so pool layout is deterministic regardless of threads
    uint32_t count = 0;
    for (size_t i = 0; i < total; i++) {
        const ChunkPos cp{mn.x + (int32_t)(i % S), mn.y + (int32_t)((i / S) % S), mn.z + (int32_t)(i / ((size_t)S * S))};
        // an all-air chunk needs no storage: leaving the cell empty resolves to pool[0] (world.rs:154-159)
        if (skip[i] || (built[i].size() == 1 && built[i][0].w == 0)) continue;
        SetVoxelErr err;
        const NodeAddr root = w->w.create_chunk(cp, built[i].data(), (uint32_t)built[i].size(), err);
        if (err != SetVoxelErr::Ok) return (int)err;
        if (ranges && count < cap) { ranges[2 * count] = root; ranges[2 * count + 1] = (uint32_t)built[i].size(); }
        count++;
    }
    if (n_ranges) *n_ranges = count;
    return 0;
}

int vrth_world_generate(vrth_world *w, uint32_t kind, uint32_t seed, int threads) {
    return generate_impl(w, kind, seed, threads, false, nullptr, 0, nullptr);
}

int vrth_world_generate_missing(vrth_world *w, uint32_t kind, uint32_t seed, int threads, uint32_t *ranges, uint32_t cap, uint32_t *n_ranges) {
    return generate_impl(w, kind, seed, threads, true, ranges, cap, n_ranges);
}


// ---- region files (servercli/src/main.rs:25-73) ----

int vrth_region_load_into_world(vrth_world *w, const uint8_t *bytes, uint64_t n, const int32_t region_pos[3], uint32_t *chunks_loaded) {
    auto rf = RegionFile::from_file(bytes, (size_t)n);
    if (!rf) return -1;
    uint32_t loaded = 0;
    for (auto &kv : rf->chunks) {
        uint32_t cn = 0;
        const Node *nodes = rf->read_chunk_data(kv.first, cn);
        if (!nodes) return -1;
        const ChunkPos cp{region_pos[0] * (int32_t)REGION_SIZE + (int32_t)kv.first[0], region_pos[1] * (int32_t)REGION_SIZE + (int32_t)kv.first[1],
                          region_pos[2] * (int32_t)REGION_SIZE + (int32_t)kv.first[2]};
        SetVoxelErr err;
        w->w.create_chunk(cp, nodes, cn, err);
        if (err == SetVoxelErr::PosOutOfBounds) continue;  // outside the client's grid, like received_oob_chunks (lib.rs:116)
        if (err != SetVoxelErr::Ok) return (int)err;
        loaded++;
    }
    if (chunks_loaded) *chunks_loaded = loaded;
    return 0;
}

uint64_t vrth_region_save_from_world(const vrth_world *w, const int32_t region_pos[3], uint8_t *out, uint64_t cap) {
    RegionFile rf;
    const int32_t r = (int32_t)REGION_SIZE;
    for (int32_t z = 0; z < r; z++)
        for (int32_t y = 0; y < r; y++)
            for (int32_t x = 0; x < r; x++) {
                const ChunkPos cp{region_pos[0] * r + x, region_pos[1] * r + y, region_pos[2] * r + z};
                const Chunk *c = w->w.get_chunk(cp);
                if (!c) continue;
                // what the server stores per chunk is its used node prefix (ServerChunk::used_nodes, server/src/world/mod.rs:109-112)
                rf.append_chunk({(uint32_t)x, (uint32_t)y, (uint32_t)z}, w->w.nodes() + c->range.start, c->alloc.last_used_addr + 1);
            }
    const std::vector<uint8_t> bytes = rf.to_file();
    if (out && bytes.size() <= cap) std::memcpy(out, bytes.data(), bytes.size());
    return bytes.size();
}

void vrth_region_of_chunk(const int32_t chunk_pos[3], int32_t region_pos[3], uint32_t pos_in_region[3]) {
    const auto rp = chunk_region(cp3(chunk_pos));
    region_pos[0] = rp.first.x; region_pos[1] = rp.first.y; region_pos[2] = rp.first.z;
    for (int i = 0; i < 3; i++) pos_in_region[i] = rp.second[i];
}

uint32_t vrth_region_file_name(const int32_t region_pos[3], char *out, uint32_t cap) {
    const std::string s = region_file_name(cp3(region_pos));
    if (out && cap > s.size()) std::memcpy(out, s.c_str(), s.size() + 1);
    return (uint32_t)s.size();
}

int vrth_chunk_msg_ingest(vrth_world *w, const uint8_t *bytes, uint64_t n, uint64_t *consumed, int32_t chunk_pos[3], uint32_t *root,
                          uint32_t *node_count) {
    GiveChunkData m;
    size_t used = 0;
    switch (GiveChunkData::decode(bytes, (size_t)n, m, used)) {
        case GiveChunkData::Status::NeedMore: return -2;
        case GiveChunkData::Status::NotChunkData: return -3;
        case GiveChunkData::Status::Malformed: return -1;
        case GiveChunkData::Status::Ok: break;
    }
    if (consumed) *consumed = used;
    if (chunk_pos) { chunk_pos[0] = m.pos.x; chunk_pos[1] = m.pos.y; chunk_pos[2] = m.pos.z; }
    // GameState::process_cmd, client/src/lib.rs:112-118
    SetVoxelErr err;
    const NodeAddr addr = w->w.create_chunk(m.pos, m.nodes.data(), (uint32_t)m.nodes.size(), err);
    if (root) *root = addr;
    if (node_count) *node_count = (uint32_t)m.nodes.size();
    return (int)err;
}

uint64_t vrth_chunk_msg_encode(const vrth_world *w, const int32_t chunk_pos[3], uint8_t *out, uint64_t cap) {
    const Chunk *c = w->w.get_chunk(cp3(chunk_pos));
    if (!c) return 0;
    // server/src/lib.rs:229-233, 292-296: GiveChunkData(pos, Cow::Borrowed(chunk.used_nodes()), NodeAlloc::new(0..1, 1..2))
    // — the allocator field is a placeholder on the wire; the client rebuilds its own (world.rs:323)
    GiveChunkData m;
    m.pos = cp3(chunk_pos);
    const uint32_t used = c->alloc.last_used_addr + 1;
    m.nodes.assign(w->w.nodes() + c->range.start, w->w.nodes() + c->range.start + used);
    m.range = {0, 2};
    m.free_mem = {NodeRange{1, 2}};
    m.last_used_addr = 0;
    const std::vector<uint8_t> bytes = m.encode();
    if (out && bytes.size() <= cap) std::memcpy(out, bytes.data(), bytes.size());
    return bytes.size();
}

}  // extern "C"
