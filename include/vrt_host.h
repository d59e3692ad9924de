/*
 * vrt_host.h — C ABI of the host-side mirror (libvrt_host.so): the reference's world / camera API
 * surface that feeds the GPU seam, flattened for FFI (cgo, ctypes, or a Rust extern "C" block).
 * The C++ types behind it keep the reference's names (the .hpp files under voxelraytracing_amd/csrc/host):
 *   ClientWorld, ChunkGrid, ChunkAlloc, Chunk   client/src/world.rs:6-367
 *   Node, Voxel, NodeAlloc, Svo                 common/src/world/mod.rs:137-471
 *   CamData::create, WorldData::from            clientdesktop/src/graphics/mod.rs:92-130
 *   axis_rot_to_ray                             common/src/math.rs:131-146
 * plus the build's deterministic world generator (no reference counterpart can be
 * reproduced: server/src/world/gen.rs uses an unseeded global RNG).
 * No GPU is needed for anything here.  Error codes are SetVoxelErr (common/src/world/mod.rs:129-135):
 * 0 ok, 1 PosOutOfBounds, 2 OutOfMemory, 3 NoChunk, 4 NoChange; 5 BadChunkData (build-defined): a chunk payload whose
 * child indices leave its own node array (untrusted network / file input) is refused by create_chunk.
 */
#ifndef VRT_HOST_H
#define VRT_HOST_H

#include <stdint.h>

#include "vrt.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vrth_world vrth_world; /* ClientWorld */

/* ClientWorld::new(center, max_nodes, size) — client/src/world.rs:272-280 */
vrth_world *vrth_world_new(const int32_t center_chunk[3], uint32_t max_nodes, uint32_t size_in_chunks);
void vrth_world_free(vrth_world *w);

/* ClientWorld::create_chunk(pos, &nodes) -> root — :310-335. Pool exhaustion is OutOfMemory here
 * (the reference panics, :251). */
int vrth_world_create_chunk(vrth_world *w, const int32_t chunk_pos[3], const uint16_t *nodes, uint32_t n, uint32_t *root_out);

/* GameState::set_voxel / ClientWorld::set_voxel — client/src/lib.rs:67-76, world.rs:344-350.
 * On success range_start/len is the edited chunk's whole pool range, which the caller re-uploads
 * (main.rs:352,356-362). NoChange when the voxel already has that value.  OutOfMemory (2) also fills the range: the
 * reference's set_node returns mid-way (mod.rs:414-415), leaving the splits it had made — same voxels, other nodes. */
int vrth_world_set_voxel(vrth_world *w, const int32_t voxel_pos[3], uint16_t voxel, uint32_t *range_start, uint32_t *range_len);
int vrth_world_get_voxel(const vrth_world *w, const int32_t voxel_pos[3], uint16_t *voxel_out);

/* GameState::center_chunks — client/src/lib.rs:55-65: recentre the grid on `anchor`, free the chunks
 * that fell out. Returns how many were removed. */
uint32_t vrth_world_center_chunks(vrth_world *w, const int32_t anchor_chunk[3]);

/* ChunkGrid::resize — world.rs:58-88 */
void vrth_world_resize(vrth_world *w, uint32_t size_in_chunks);

/* ClientWorld::nodes — the whole flat pool (max_nodes u16 words) */
const uint16_t *vrth_world_nodes(const vrth_world *w);
uint32_t vrth_world_max_nodes(const vrth_world *w);

/* ChunkGrid::chunk_roots — world.rs:154-159. Writes min(cap, S^3) entries, returns S^3. */
uint32_t vrth_world_chunk_roots(const vrth_world *w, uint32_t *out, uint32_t cap);
/* The table itself (S^3 entries) and a number that changes whenever its contents may have: the tag of
 * vrt_write_chunk_roots_tagged, which lets the per-frame rewrite of an unchanged table (main.rs:446) cost nothing.  (The mirror
 * keeps the table up to date inside the calls that change the grid; the reference builds a fresh Vec per frame.)
 * Lifetime: the pointer stays valid until vrth_world_resize (which replaces the grid); the entries change under it with
 * every create_chunk / chunk message / center_chunks, and the generation with them.  Threading: a world belongs to one
 * thread at a time, as the reference's ClientWorld does (main.rs: everything on the event-loop thread) — nothing here locks. */
const uint32_t *vrth_world_chunk_roots_ptr(const vrth_world *w);
uint64_t vrth_world_chunk_roots_generation(const vrth_world *w);

/* min_voxel, size_in_voxels, size_in_chunks, populated_count */
void vrth_world_info(const vrth_world *w, int32_t min_voxel[3], uint32_t *size_in_voxels, uint32_t *size_in_chunks, uint32_t *populated);
/* chunk_alloc_status -> (free, max) — world.rs:288-290 */
void vrth_world_alloc_status(const vrth_world *w, uint32_t *free_nodes, uint32_t *max_nodes);
/* One chunk's pool range and NodeAlloc state: spans[2*i], spans[2*i+1] = free span i (chunk-relative).
 * Returns the number of free spans (writes at most cap_spans), or -1 if there is no chunk. */
int vrth_world_chunk_state(const vrth_world *w, const int32_t chunk_pos[3], uint32_t *range_start, uint32_t *range_end,
                           uint32_t *last_used_addr, uint32_t *spans, uint32_t cap_spans);
/* highest_vox_at — world.rs:359-366; returns 1 and *y_out when a non-empty voxel exists */
int vrth_world_highest_vox_at(const vrth_world *w, int32_t x, int32_t z, int32_t *y_out);

/* WorldData::from(&world) — graphics/mod.rs:121-130 */
void vrth_world_data_from(const vrth_world *w, vrt_world_data *out);
/* CamData::create(rot_deg, eye, fov_deg, proj_size) — graphics/mod.rs:92-111 */
void vrth_cam_data_create(const float rot_deg[3], const float eye[3], float fov_deg, const float proj_size[2], vrt_cam_data *out);
/* axis_rot_to_ray(rot in radians) — common/src/math.rs:131-146 */
void vrth_axis_rot_to_ray(const float rot_rad[3], float out[3]);
/* Material::construct_arr for the standard data pack — graphics/mod.rs:38-60; out[256] */
void vrth_std_materials(vrt_material *out256);
/* Name of standard voxel id, or NULL */
const char *vrth_std_voxel_name(uint32_t id);

/* ---- SVO construction ---- */
/* Svo::set_node driven the way server/src/world/gen.rs:171-286 drives it (x, z, y ascending, air
 * skipped) from dense[x + 32*(y + 32*z)]. Returns nodes in use (last_used_addr + 1), 0 on OOM. */
uint32_t vrth_svo_build_by_set_node(const uint16_t *dense, uint16_t *nodes, uint32_t cap);
/* Minimal octree, breadth-first layout. Returns node count, 0 if > 32767 nodes or cap too small. */
uint32_t vrth_svo_build_bottom_up(const uint16_t *dense, uint16_t *nodes, uint32_t cap);
/* Expand a chunk's SVO back to dense[32^3] through Svo::find_node. */
void vrth_svo_to_dense(const uint16_t *nodes, uint16_t *dense);

/* ---- deterministic world generator (build-defined) ---- */
int32_t vrth_gen_height(uint32_t seed, int32_t x, int32_t z);
/* dense[32^3] of chunk (cx,cy,cz); returns 1 if uniform */
int vrth_gen_dense(uint32_t seed, const int32_t chunk_pos[3], uint16_t *dense);
void vrth_gen_dense_superflat(const int32_t chunk_pos[3], uint16_t *dense);
/* Generate every chunk of the world's grid and create_chunk it. kind 0 = procedural (seed),
 * 1 = superflat built by set_node (config C1). threads <= 0: all cores. Returns 0 or a SetVoxelErr. */
int vrth_world_generate(vrth_world *w, uint32_t kind, uint32_t seed, int threads);
/* The same for the grid's EMPTY cells only — what arrives after request_missing_chunks (client/src/lib.rs:80-108) once
 * center_chunks moved the grid: ranges[2 i], ranges[2 i + 1] = (root, node count) of the i-th chunk created (grid order;
 * at most cap pairs are written), *n_ranges = how many were created — the ranges to hand to vrt_write_nodes
 * (main.rs:289-295).  All-air chunks stay empty cells. */
int vrth_world_generate_missing(vrth_world *w, uint32_t kind, uint32_t seed, int threads, uint32_t *ranges, uint32_t cap, uint32_t *n_ranges);

/* ---- region files of the reference server (servercli/src/main.rs:25-73; format in csrc/host/regionfile.hpp) ---- */
/* Parse one `regions/r_X_Y_Z_.data` image and create_chunk every chunk of it that lies inside the world's grid.
 * Returns 0, -1 for a malformed file, or a SetVoxelErr. */
int vrth_region_load_into_world(vrth_world *w, const uint8_t *bytes, uint64_t n, const int32_t region_pos[3], uint32_t *chunks_loaded);
/* Serialise the world's chunks of one region in the same format. Returns the byte count (writes if it fits cap). */
uint64_t vrth_region_save_from_world(const vrth_world *w, const int32_t region_pos[3], uint8_t *out, uint64_t cap);
/* ChunkPos::region (common/src/world/mod.rs:90-96) and region_path_by_pos (servercli/src/main.rs:25-27). */
void vrth_region_of_chunk(const int32_t chunk_pos[3], int32_t region_pos[3], uint32_t pos_in_region[3]);
uint32_t vrth_region_file_name(const int32_t region_pos[3], char *out, uint32_t cap);

/* ---- the chunk payload of the reference's wire protocol (common/src/net.rs:46-55; encoding in csrc/host/netmsg.hpp) ---- */
/* Decode one `ClientCmd::GiveChunkData(pos, nodes, node_alloc)` from the front of a received-byte queue and do what
 * GameState::process_cmd does with it (client/src/lib.rs:110-118): create_chunk(pos, nodes).  On 0 the caller uploads
 * pool[root, root + node_count) (clientdesktop/src/main.rs:289-295) and drops `consumed` bytes.
 * Returns 0, a SetVoxelErr (> 0; the message is consumed: 1 = PosOutOfBounds is `received_oob_chunks`), -1 malformed,
 * -2 incomplete (bincode's UnexpectedEnd: wait for more bytes), -3 another ClientCmd variant. */
int vrth_chunk_msg_ingest(vrth_world *w, const uint8_t *bytes, uint64_t n, uint64_t *consumed, int32_t chunk_pos[3], uint32_t *root,
                          uint32_t *node_count);
/* What the server sends for a chunk of this world (server/src/lib.rs:229-233: its used node prefix + the placeholder
 * NodeAlloc::new(0..1, 1..2)).
 * Returns the byte count (writes if it fits cap), 0 if the world has no such chunk. */
uint64_t vrth_chunk_msg_encode(const vrth_world *w, const int32_t chunk_pos[3], uint8_t *out, uint64_t cap);

#ifdef __cplusplus
}
#endif
#endif
