// vrt_uploads.hip — the write half of the C ABI (NodeBuffer::write, ArrayBuffer::write, SimpleBuffer::write_slice:
// clientdesktop/src/graphics/shader.rs:22-40,101-142) and the derived tables that follow the writes (vrt_accel.hip).
#include "vrt_ctx.h"

size_t chunk_dir_entries(uint32_t S) { return (size_t)S * (S + 1u) * (S + 1u); }
size_t direct_cell_entries(uint32_t S) { const size_t B = (size_t)S * 4u; return B * (B + 1u) * (B + 1u) * 8u; }

int alloc_roots(vrt_ctx *c, uint32_t world_size) {
    const uint64_t n = (uint64_t)world_size * world_size * world_size;
    if (world_size == 0 || n > (1ull << 28)) return fail(c, VRT_ERR_INVALID_ARG, "world_size_chunks %u out of range", world_size);
    {   // uploads into the table that goes away: launched, then waited for
        const int rc = flush_staged(c);
        if (rc) return rc;
    }
    if (c->up_stream) HIP_TRY(c, hipStreamSynchronize(c->up_stream));
    c->roots_tag = 0;
    (void)hipFree(c->d_roots);
    c->d_roots = nullptr;
    HIP_TRY(c, hipMalloc(&c->d_roots, n * sizeof(uint32_t)));
    // A fresh wgpu buffer is zero-initialised: every chunk resolves to pool[0], the air leaf.
    HIP_TRY(c, hipMemsetAsync(c->d_roots, 0, n * sizeof(uint32_t), c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));   // (later uploads run on their own stream)
    c->world_size = world_size;
    c->n_roots = (uint32_t)n;
    c->h_roots.assign((size_t)n, 0u);
    for (auto &T : c->tabs) {
        T.dirty_chunks.clear();
        T.chunk_is_dirty.assign((size_t)n, 0);
        T.chunk_may_have_moved.assign((size_t)n, 0);
        T.chunks_moved = 0;
    }
    c->roots_index_stale = true;
    c->accel_dirty = true;
    return VRT_OK;
}

// ---- ordering without draining -------------------------------------------------------------------------------------
// Frames in flight run on the context's own streams; uploads and table rebuilds run on c->stream.  An upload must come
// after every frame enqueued before it (they read what it overwrites) and before every frame enqueued after it: both are
// stream waits on events, the host never blocks on the device here.

// `target` waits for everything enqueued so far on the frame streams other than itself.
int order_after_frames(vrt_ctx *c, hipStream_t target) {
    auto wait_for = [&](hipStream_t st) -> int {
        if (!st || st == target) return VRT_OK;
        if (!c->ev_frames) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_frames, hipEventDisableTiming));
        HIP_TRY(c, hipEventRecord(c->ev_frames, st));
        HIP_TRY(c, hipStreamWaitEvent(target, c->ev_frames, 0));
        return VRT_OK;
    };
    if (c->alt_pending)
        for (hipStream_t st : c->extra_stream) { const int rc = wait_for(st); if (rc) return rc; }
    if (c->own_pending) { const int rc = wait_for(c->own_stream); if (rc) return rc; }
    if (target != c->stream) { const int rc = wait_for(c->stream); if (rc) return rc; }
    return VRT_OK;
}
int order_after_frames(vrt_ctx *c) { return order_after_frames(c, c->stream); }

// Everything enqueued on c->stream so far (an upload, a table rebuild) happens before later frames on other streams.
int publish_upload(vrt_ctx *c) {
    if (!c->ev_upload) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_upload, hipEventDisableTiming));
    HIP_TRY(c, hipEventRecord(c->ev_upload, c->stream));
    c->upload_gen += 1;
    return VRT_OK;
}

// `st` (a frame stream's slot, or kMaxInFlight for c->stream) waits for the node-pool / chunk_roots uploads so far.
int wait_for_pool_uploads(vrt_ctx *c, hipStream_t st, uint32_t slot) {
    {
        const int rc = flush_staged(c);   // (what vrt_write_nodes / vrt_write_chunk_roots staged since the last flush: one launch)
        if (rc) return rc;
    }
    if (!c->ev_pool_upload || c->seen_pool_gen[slot] == c->pool_gen) return VRT_OK;
    HIP_TRY(c, hipStreamWaitEvent(st, c->ev_pool_upload, 0));
    c->seen_pool_gen[slot] = c->pool_gen;
    return VRT_OK;
}

// Called before a frame is enqueued on frame stream `st` (slot 0 = own_stream, k = extra_stream[k - 1]).
int frame_waits_for_uploads(vrt_ctx *c, hipStream_t st, uint32_t slot) {
    const int rc = wait_for_pool_uploads(c, st, st == c->stream ? vrt_ctx::kMaxInFlight : slot);
    if (rc) return rc;
    if (st == c->stream || c->seen_gen[slot] == c->upload_gen) return VRT_OK;
    HIP_TRY(c, hipStreamWaitEvent(st, c->ev_upload, 0));
    c->seen_gen[slot] = c->upload_gen;
    return VRT_OK;
}

// Copy `bytes` of host memory to the device with wgpu's write_buffer semantics — the caller may reuse `src` as soon as
// this returns, the data is visible to the next frame — without waiting for the device: the bytes are copied into a
// pinned ring now, the ring feeds an asynchronous copy kernel.  `pool`: the destination is the node pool or chunk_roots —
// the copy runs on the upload stream behind the readers of those two buffers (the table updates; every frame only if one
// that walks the octree is in flight); otherwise on c->stream behind the frames in flight.  Transfers larger than a ring
// segment (the initial pool upload) take the synchronous route.
// `bytes` of the pinned ring (64-byte aligned), valid until the segment comes round again (eight segments on)
static int ring_place(vrt_ctx *c, size_t bytes, size_t *at) {
    if (!c->h_ring) {
        HIP_TRY(c, hipHostMalloc((void **)&c->h_ring, vrt_ctx::kRingSegBytes * vrt_ctx::kRingSegs, hipHostMallocMapped));
        HIP_TRY(c, hipHostGetDevicePointer((void **)&c->d_ring, c->h_ring, 0));
    }
    const size_t need = (bytes + 63u) & ~(size_t)63u;
    if (c->ring_off + need > vrt_ctx::kRingSegBytes) {
        const uint32_t next = (c->ring_seg + 1u) % vrt_ctx::kRingSegs;
        // staged ranges have no event until they are flushed: a segment that still holds some (only possible when uploads that
        // are not staged — materials, ndc tables — have walked the ring round since) is flushed before it is written again
        if (c->staged_seg[next]) {
            const int rc = flush_staged(c);
            if (rc) return rc;
        }
        c->ring_seg = next;
        c->ring_off = 0;
        // the segment's previous copies must have left it (seven segments ago: practically always long done)
        for (int k = 0; k < 2; k++)
            if (c->ring_ev_used[c->ring_seg][k]) HIP_TRY(c, hipEventSynchronize(c->ring_ev[c->ring_seg][k]));
    }
    *at = (size_t)c->ring_seg * vrt_ctx::kRingSegBytes + c->ring_off;
    c->ring_off += need;
    return VRT_OK;
}

// The node-pool / chunk_roots uploads staged so far, as one launch on the upload stream: behind the readers of those two
// buffers (whole-world builds and the other uploads so far, every table set's last update — those that are not over yet: a
// wait is a barrier packet on the stream, a query is a load; every frame only if one that walks the octree is in flight).
int flush_staged(vrt_ctx *c) {
    if (c->staged.empty()) return VRT_OK;
    hipStream_t st = c->up_stream;
    if (c->ev_upload && hipEventQuery(c->ev_upload) != hipSuccess) HIP_TRY(c, hipStreamWaitEvent(st, c->ev_upload, 0));
    for (auto &T : c->tabs)
        if (T.update_pending) {
            if (hipEventQuery(T.ev_updated) == hipSuccess) T.update_pending = false;
            else HIP_TRY(c, hipStreamWaitEvent(st, T.ev_updated, 0));
        }
    (void)hipGetLastError();   // (hipErrorNotReady from the queries is not an error)
    if (c->walkers_in_flight) {
        const int rc = order_after_frames(c, st);
        if (rc) return rc;
    }
    vrt::UploadBatch batch;
    uint32_t n = 0;
    auto launch = [&]() -> int {
        vrt::launch_upload_batch(c->d_nodes, c->d_roots, c->d_ring, batch, n, st);
        HIP_TRY(c, hipGetLastError());
        n = 0;
        return VRT_OK;
    };
    for (const auto &s : c->staged)
        for (uint32_t done = 0; done < s.n_words; done += vrt::kUploadPieceWords) {
            const uint32_t words = s.n_words - done < vrt::kUploadPieceWords ? s.n_words - done : vrt::kUploadPieceWords;
            batch.piece[n++] = vrt::UploadPiece{s.dst_word + done, (uint32_t)(s.ring_at / 4u) + done, words | (s.buf ? 0x80000000u : 0u)};
            if (n == vrt::kUploadBatchPieces) { const int rc = launch(); if (rc) return rc; }
        }
    if (n) { const int rc = launch(); if (rc) return rc; }
    for (uint32_t k = 0; k < vrt_ctx::kRingSegs; k++)
        if (c->staged_seg[k]) {
            hipEvent_t &rev = c->ring_ev[k][1];
            if (!rev) HIP_TRY(c, hipEventCreateWithFlags(&rev, hipEventDisableTiming));
            HIP_TRY(c, hipEventRecord(rev, st));
            c->ring_ev_used[k][1] = true;
            c->staged_seg[k] = false;
        }
    c->staged.clear();
    c->staged_bytes = 0;
    HIP_TRY(c, hipEventRecord(c->ev_pool_upload, st));
    c->pool_gen += 1;
    return VRT_OK;
}

// Stage `bytes` for words [dst_word, ...) of the node pool (buf 0) or chunk_roots (buf 1): copied now, launched at the flush.
static int stage_pool_upload(vrt_ctx *c, uint32_t buf, uint32_t dst_word, const void *src, size_t bytes) {
    // The batch's pieces run side by side, so the staged ranges of a buffer are kept disjoint.  A range INSIDE one staged earlier
    // overwrites that one's bytes in the ring (several edits of one chunk before a frame: main.rs:352-362 uploads the chunk's
    // range after every edit — a launch per edit was 10 us per edit, tools/edit_burst.py); one that COVERS staged ranges takes their
    // place; any other overlap must land after what is staged: that goes out first.
    {
        const uint32_t n_words = (uint32_t)(bytes / 4u), end_word = dst_word + n_words;
        bool partial = false;
        for (auto &s : c->staged) {
            if (s.buf != buf || !s.n_words || !(dst_word < s.dst_word + s.n_words && s.dst_word < end_word)) continue;
            if (dst_word >= s.dst_word && end_word <= s.dst_word + s.n_words) {   // inside: the staged copy becomes this one's data
                memcpy(c->h_ring + s.ring_at + (size_t)(dst_word - s.dst_word) * 4u, src, bytes);
                return VRT_OK;
            }
            if (!(s.dst_word >= dst_word && s.dst_word + s.n_words <= end_word)) partial = true;
        }
        if (partial) {
            const int rc = flush_staged(c);
            if (rc) return rc;
        } else {
            for (auto &s : c->staged)   // covered ones: dropped (their ring bytes are simply not copied)
                if (s.buf == buf && s.n_words && s.dst_word >= dst_word && s.dst_word + s.n_words <= end_word) { c->staged_bytes -= (size_t)s.n_words * 4u; s.n_words = 0u; }
        }
    }
    // (half the ring: staged data is never overwritten by what follows; and no more ranges than the scan above looks through in a
    // microsecond — a host that writes node by node would otherwise pay for every range staged before it, quadratically)
    if (c->staged_bytes + bytes > 4u * vrt_ctx::kRingSegBytes || c->staged.size() >= 512u) {
        const int rc = flush_staged(c);
        if (rc) return rc;
    }
    size_t at = 0;
    const int rc = ring_place(c, bytes, &at);
    if (rc) return rc;
    memcpy(c->h_ring + at, src, bytes);
    c->staged.push_back({buf, dst_word, (uint32_t)(bytes / 4u), at});
    c->staged_bytes += bytes;
    c->staged_seg[at / vrt_ctx::kRingSegBytes] = true;
    // Batching pays while frames are in flight (the launches would queue up behind them anyway).  With the device idle — a
    // lone edit with a synchronise behind its frame, main.rs:352-362 — the copy goes out now and runs while the host is
    // still on its way to vrt_render, instead of at the head of that call (170 -> 200 us in round 3's edit_cost).
    // ONE range per frame goes out like that.  A burst — chunks arriving at a join, main.rs:289-295: hundreds of ranges before one
    // frame — outlasts the frames in flight, and from then on every call found the device idle and launched its own copy:
    // 1.5 us per call became 6 (256 uploads per frame: 2.0 ms per frame, tools/stream_burst.py); the rest of a burst is batched.
    if (c->flushed_at_call || !c->rendered || !c->last_stream) return VRT_OK;
    if (hipStreamQuery(c->last_stream) == hipSuccess) {
        c->flushed_at_call = true;   // (until the next vrt_render)
        return flush_staged(c);
    }
    (void)hipGetLastError();   // (hipErrorNotReady is not an error)
    return VRT_OK;
}

int stage_upload(vrt_ctx *c, void *dst, const void *src, size_t bytes, bool pool) {
    if (bytes == 0) return VRT_OK;
    hipStream_t st = c->stream;
    if (pool) {
        if (!c->up_stream) HIP_TRY(c, hipStreamCreateWithFlags(&c->up_stream, hipStreamNonBlocking));
        if (!c->ev_pool_upload) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_pool_upload, hipEventDisableTiming));
        st = c->up_stream;
        const bool is_roots = dst >= (void *)c->d_roots && dst < (void *)(c->d_roots + c->n_roots);
        if (bytes <= vrt_ctx::kRingSegBytes && !(bytes & 3u) && !((uintptr_t)dst & 3u))
            return stage_pool_upload(c, is_roots ? 1u : 0u,
                                     (uint32_t)(((uintptr_t)dst - (uintptr_t)(is_roots ? (void *)c->d_roots : (void *)c->d_nodes)) / 4u), src, bytes);
        {   // (the whole pool at join time: the synchronous route below, behind what is staged)
            const int rc = flush_staged(c);
            if (rc) return rc;
        }
        // whole-world builds and the other uploads so far, every set's last update — those that are not over yet (a wait is
        // a barrier packet on the stream, a query is a load)
        if (c->ev_upload && hipEventQuery(c->ev_upload) != hipSuccess) HIP_TRY(c, hipStreamWaitEvent(st, c->ev_upload, 0));
        for (auto &T : c->tabs)
            if (T.update_pending) {
                if (hipEventQuery(T.ev_updated) == hipSuccess) T.update_pending = false;
                else HIP_TRY(c, hipStreamWaitEvent(st, T.ev_updated, 0));
            }
        (void)hipGetLastError();   // (hipErrorNotReady from the queries is not an error)
        if (c->walkers_in_flight) {
            const int rc = order_after_frames(c, st);
            if (rc) return rc;
        }
    } else {
        const int rc = order_after_frames(c);
        if (rc) return rc;
    }
    auto publish = [&]() -> int {
        if (!pool) return publish_upload(c);
        HIP_TRY(c, hipEventRecord(c->ev_pool_upload, st));
        c->pool_gen += 1;
        return VRT_OK;
    };
    if (bytes > vrt_ctx::kRingSegBytes || (bytes & 3u) || ((uintptr_t)dst & 3u)) {
        HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st));
        HIP_TRY(c, hipStreamSynchronize(st));
        return publish();
    }
    size_t at = 0;
    {
        const int rc = ring_place(c, bytes, &at);
        if (rc) return rc;
    }
    memcpy(c->h_ring + at, src, bytes);
    vrt::launch_upload_words(dst, c->d_ring + at, (uint32_t)(bytes / 4u), st);
    HIP_TRY(c, hipGetLastError());
    const size_t seg = at / vrt_ctx::kRingSegBytes;
    hipEvent_t &rev = c->ring_ev[seg][pool ? 1 : 0];
    if (!rev) HIP_TRY(c, hipEventCreateWithFlags(&rev, hipEventDisableTiming));
    HIP_TRY(c, hipEventRecord(rev, st));
    c->ring_ev_used[seg][pool ? 1 : 0] = true;
    return publish();
}

// ---- which chunks a write touched ----------------------------------------------------------------------------------
void mark_all_dirty(vrt_ctx *c) {
    c->view_gen++;
    c->accel_dirty = true;
    for (auto &T : c->tabs) {
        for (uint32_t ch : T.dirty_chunks) T.chunk_is_dirty[ch] = 0;
        T.dirty_chunks.clear();
    }
}

// every table set in use hears of it (a set's list: what changed since *that set* was last brought up to date)
static void mark_chunk_dirty(vrt_ctx *c, uint32_t chunk) {
    // A chunk's edit moves the cost of the few tiles that see it by a few steps: the tile order of the view stays the
    // better order for the edit's own frame (screen order costs a lone frame 15 us), and the next frame without an edit
    // notes its trips again (vrt_render).  Whole-world changes (mark_all_dirty) drop the order as a camera move does.
    c->tile_order_stale = true;
    if (c->accel_dirty) return;
    c->tables_split = true;   // (from the next frame on; see vrt_render)
    c->quiet_frames = 0;
    for (uint32_t k = 0; k < vrt_ctx::kMaxInFlight; k++) {
        auto &T = c->tabs[k];
        if ((k && !T.live) || T.chunk_is_dirty[chunk]) continue;
        if (T.dirty_chunks.size() >= kMaxDirtyChunks) { mark_all_dirty(c); return; }
        T.chunk_is_dirty[chunk] = 1;
        T.dirty_chunks.push_back(chunk);
    }
}

static void refresh_roots_index(vrt_ctx *c) {   // (root, chunk) of every present chunk, sorted
    if (!c->roots_index_stale) return;
    c->roots_index.clear();
    for (uint32_t i = 0; i < c->n_roots; i++)
        if (c->h_roots[i]) c->roots_index.emplace_back(c->h_roots[i], i);
    std::sort(c->roots_index.begin(), c->roots_index.end());
    c->roots_index_stale = false;
}

// Nodes [start, end) were overwritten: the chunks whose octrees may have changed are the ones whose root lies in the range
// and the one whose root precedes it (a chunk's nodes follow its root up to the next chunk's root: ChunkAlloc hands out
// disjoint ranges, client/src/world.rs:239-256).  Node 0 is the root of every missing chunk: a write to it is everything.
static void mark_node_range_dirty(vrt_ctx *c, uint32_t start, uint32_t end) {
    if (c->accel_dirty) return;
    if (start == 0u) { mark_all_dirty(c); return; }
    refresh_roots_index(c);
    auto it = std::upper_bound(c->roots_index.begin(), c->roots_index.end(), std::make_pair(start, 0xFFFFFFFFu));
    // chunks sharing the root that precedes the range (normally one), then every chunk rooted inside it
    if (it != c->roots_index.begin()) {
        const uint32_t r = std::prev(it)->first;
        for (auto k = std::prev(it);; --k) {
            if (k->first != r) break;
            mark_chunk_dirty(c, k->second);
            if (c->accel_dirty || k == c->roots_index.begin()) break;
        }
    }
    for (; it != c->roots_index.end() && it->first < end && !c->accel_dirty; ++it) mark_chunk_dirty(c, it->second);
}

// The cell grid and brick pool (vrt_accel.hip) follow the node pool and chunk_roots in two steps.
//   ensure_accel_world  before a frame picks its frame set: the whole-world build when it is due (first frame, resized or
//                       recentred grid, too many single-chunk updates since the last one) — on c->stream with the frames
//                       in flight waited for, into tabs[0], copied to the other sets in use;
//   update_tables       once the frame has its set and stream: the chunks dirtied since *that set* was last brought up to
//                       date, rebuilt alone on the frame's own stream — nothing waits for the frames in flight, which read
//                       other sets (or are earlier on this very stream).
int free_tables(vrt_ctx *c, vrt_ctx::Tables &T) {
    (void)c;
    (void)hipFree(T.d_grid); (void)hipFree(T.d_bricks); (void)hipFree(T.d_chunk_bricks); (void)hipFree(T.d_chunk_bases);
    (void)hipFree(T.d_chunk_caps); (void)hipFree(T.d_brick_tail); (void)hipFree(T.d_cdir); (void)hipFree(T.d_mblk); (void)hipFree(T.d_mblk_tail);
    T.d_grid = nullptr; T.d_bricks = nullptr; T.d_chunk_bricks = T.d_chunk_bases = T.d_chunk_caps = T.d_brick_tail = nullptr;
    T.d_cdir = nullptr; T.d_mblk = nullptr; T.d_mblk_tail = nullptr;
    T.grid_cap = 0; T.brick_cap = 0; T.chunk_cap = 0; T.cdir_cap = 0; T.mblk_cap = 0;
    T.live = false;
    return VRT_OK;
}

// tabs[k] becomes a copy of tabs[0] (device tables and host bookkeeping); everything on c->stream, the caller has waited
// for the frames in flight.
static int alloc_tables_like_first(vrt_ctx *c, uint32_t k) {
    auto &A = c->tabs[0];
    auto &T = c->tabs[k];
    const uint32_t S = c->accel_S, n_chunks = S * S * S;
    const size_t G = (size_t)S * 8u, entries = G * (G + 1u) * (G + 1u);
    if (T.grid_cap < entries) {
        (void)hipFree(T.d_grid); T.d_grid = nullptr; T.grid_cap = 0;
        HIP_TRY(c, hipMalloc(&T.d_grid, entries * sizeof(uint32_t)));
        T.grid_cap = entries;
    }
    if (A.d_mblk) {
        if (T.cdir_cap < chunk_dir_entries(S)) {
            (void)hipFree(T.d_cdir); T.d_cdir = nullptr; T.cdir_cap = 0;
            HIP_TRY(c, hipMalloc(&T.d_cdir, chunk_dir_entries(S) * sizeof(uint32_t)));
            T.cdir_cap = chunk_dir_entries(S);
        }
        if (T.mblk_cap != A.mblk_cap || !T.d_mblk) {
            (void)hipFree(T.d_mblk); T.d_mblk = nullptr; T.mblk_cap = 0;
            HIP_TRY(c, hipMalloc(&T.d_mblk, (size_t)A.mblk_cap * 512u * sizeof(uint4)));
            T.mblk_cap = A.mblk_cap;
        }
        if (!T.d_mblk_tail) HIP_TRY(c, hipMalloc(&T.d_mblk_tail, sizeof(uint32_t)));
    } else if (T.d_mblk) {
        (void)hipFree(T.d_mblk); T.d_mblk = nullptr; T.mblk_cap = 0;
    }
    if (T.chunk_cap < n_chunks) {
        (void)hipFree(T.d_chunk_bricks); (void)hipFree(T.d_chunk_bases); (void)hipFree(T.d_chunk_caps);
        T.d_chunk_bricks = T.d_chunk_bases = T.d_chunk_caps = nullptr; T.chunk_cap = 0;
        HIP_TRY(c, hipMalloc(&T.d_chunk_bricks, (size_t)n_chunks * sizeof(uint32_t)));
        HIP_TRY(c, hipMalloc(&T.d_chunk_bases, (size_t)n_chunks * sizeof(uint32_t)));
        HIP_TRY(c, hipMalloc(&T.d_chunk_caps, (size_t)n_chunks * sizeof(uint32_t)));
        T.chunk_cap = n_chunks;
    }
    if (T.brick_cap != A.brick_cap || !T.d_bricks) {
        (void)hipFree(T.d_bricks); T.d_bricks = nullptr; T.brick_cap = 0;
        HIP_TRY(c, hipMalloc(&T.d_bricks, (size_t)A.brick_cap * 64u * sizeof(uint16_t)));
        T.brick_cap = A.brick_cap;
    }
    if (!T.d_brick_tail) HIP_TRY(c, hipMalloc(&T.d_brick_tail, sizeof(uint32_t)));
    return VRT_OK;
}

static int copy_tables_from_first(vrt_ctx *c, uint32_t k) {
    auto &A = c->tabs[0];
    auto &T = c->tabs[k];
    const uint32_t S = c->accel_S, n_chunks = S * S * S;
    const size_t G = (size_t)S * 8u, entries = G * (G + 1u) * (G + 1u);
    {
        const int rc = alloc_tables_like_first(c, k);
        if (rc) return rc;
    }
    HIP_TRY(c, hipMemcpyAsync(T.d_grid, A.d_grid, entries * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
    if (A.d_mblk) {
        HIP_TRY(c, hipMemcpyAsync(T.d_cdir, A.d_cdir, chunk_dir_entries(S) * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(T.d_mblk, A.d_mblk, (size_t)A.mblk_cap * 512u * sizeof(uint4), hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(T.d_mblk_tail, A.d_mblk_tail, sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
    }
    HIP_TRY(c, hipMemcpyAsync(T.d_chunk_bricks, A.d_chunk_bricks, (size_t)n_chunks * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(T.d_chunk_bases, A.d_chunk_bases, (size_t)n_chunks * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(T.d_chunk_caps, A.d_chunk_caps, (size_t)n_chunks * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(T.d_brick_tail, A.d_brick_tail, sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(T.d_bricks, A.d_bricks, (size_t)A.brick_cap * 64u * sizeof(uint16_t), hipMemcpyDeviceToDevice, c->stream));
    T.chunk_may_have_moved = A.chunk_may_have_moved;
    T.chunks_moved = A.chunks_moved;
    T.chunk_builds = A.chunk_builds;
    T.dirty_chunks = A.dirty_chunks;   // what tabs[0] has not caught up with yet, this copy has not either
    T.chunk_is_dirty = A.chunk_is_dirty;
    T.update_pending = false;
    T.live = true;
    return VRT_OK;
}

int ensure_accel_world(vrt_ctx *c) {
    const uint32_t S = c->world.size_in_chunks;
    if (S > c->accel_max_s) {  // too large for the tables: nothing to keep up to date, the octree walk reads the pool itself
        if (c->accel_dirty || c->accel_S != S) {
            mark_all_dirty(c);
            c->accel_ok = false;
            c->accel_S = S;
            c->accel_dirty = false;
        }
        return VRT_OK;
    }
    if (!c->accel_dirty && c->accel_S == S) {
        if (!c->accel_ok) return VRT_OK;   // (the brick pool would be too large: stays off until something changes)
        // single chunks; a chunk that outgrew its region moves to the tail once — every set's tail must have room for all of them
        bool room = true;
        for (uint32_t k = 0; k < vrt_ctx::kMaxInFlight; k++) {
            const auto &T = c->tabs[k];
            if (k && !T.live) continue;
            uint32_t fresh = 0;
            for (uint32_t ch : T.dirty_chunks) fresh += T.chunk_may_have_moved[ch] ? 0u : 1u;
            room = room && T.chunks_moved + fresh <= kTailChunks;
        }
        if (room) return VRT_OK;
    }
    mark_all_dirty(c);   // (clears the chunk lists: a whole-world build covers them)
    QUIESCE(c);  // frames on the other streams may still be reading the old tables; this path reads a count back anyway
    {   // ... and the node pool and chunk_roots as uploaded so far
        const int rc = wait_for_pool_uploads(c, c->stream, vrt_ctx::kMaxInFlight);
        if (rc) return rc;
    }
    auto &A = c->tabs[0];
    c->accel_ok = false;
    c->accel_S = S;
    c->accel_dirty = false;
    const uint32_t n_chunks = S * S * S;
    const size_t G = (size_t)S * 8u;
    const size_t entries = G * (G + 1u) * (G + 1u);
    if (entries > A.grid_cap) {
        (void)hipFree(A.d_grid);
        A.d_grid = nullptr; A.grid_cap = 0;
        HIP_TRY(c, hipMalloc(&A.d_grid, entries * sizeof(uint32_t)));
        A.grid_cap = entries;
    }
    // the border rows / entries are never written by the kernels: zero = "outside the world"
    HIP_TRY(c, hipMemsetAsync(A.d_grid, 0, entries * sizeof(uint32_t), c->stream));
    if (chunk_dir_entries(S) > A.cdir_cap) {
        (void)hipFree(A.d_cdir);
        A.d_cdir = nullptr; A.cdir_cap = 0;
        HIP_TRY(c, hipMalloc(&A.d_cdir, chunk_dir_entries(S) * sizeof(uint32_t)));
        A.cdir_cap = chunk_dir_entries(S);
    }
    HIP_TRY(c, hipMemsetAsync(A.d_cdir, 0, chunk_dir_entries(S) * sizeof(uint32_t), c->stream));   // the border: outside the world
    if (!A.d_mblk_tail) HIP_TRY(c, hipMalloc(&A.d_mblk_tail, sizeof(uint32_t)));
    if (n_chunks > c->chunk_needs_cap) {
        (void)hipFree(c->d_chunk_needs);
        c->d_chunk_needs = nullptr; c->chunk_needs_cap = 0;
        HIP_TRY(c, hipMalloc(&c->d_chunk_needs, (size_t)n_chunks * sizeof(uint32_t)));
        c->chunk_needs_cap = n_chunks;
    }
    if (n_chunks > A.chunk_cap) {
        (void)hipFree(A.d_chunk_bricks); (void)hipFree(A.d_chunk_bases); (void)hipFree(A.d_chunk_caps);
        A.d_chunk_bricks = A.d_chunk_bases = A.d_chunk_caps = nullptr; A.chunk_cap = 0;
        HIP_TRY(c, hipMalloc(&A.d_chunk_bricks, (size_t)n_chunks * sizeof(uint32_t)));
        HIP_TRY(c, hipMalloc(&A.d_chunk_bases, (size_t)n_chunks * sizeof(uint32_t)));
        HIP_TRY(c, hipMalloc(&A.d_chunk_caps, (size_t)n_chunks * sizeof(uint32_t)));
        A.chunk_cap = n_chunks;
    }
    if (!c->d_brick_total) HIP_TRY(c, hipMalloc(&c->d_brick_total, 2 * sizeof(uint32_t)));   // [0] bricks, [1] march-cell blocks
    if (!A.d_brick_tail) HIP_TRY(c, hipMalloc(&A.d_brick_tail, sizeof(uint32_t)));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIP_TRY(c, hipEventCreate(&e0));
    HIP_TRY(c, hipEventCreate(&e1));
    const bool direct = S <= c->march_direct_max_s;
    c->march_direct = direct;
    auto body = [&]() -> int {
        HIP_TRY(c, hipEventRecord(e0, c->stream));
        vrt::launch_accel_cells(c->d_nodes, c->max_nodes, c->d_roots, S, A.d_grid, A.d_chunk_bricks, A.d_chunk_bases, A.d_chunk_caps,
                                c->d_brick_total, A.d_brick_tail, direct ? nullptr : c->d_chunk_needs, A.d_cdir, A.d_mblk_tail, c->d_brick_total + 1, c->stream);
        HIP_TRY(c, hipGetLastError());
        uint32_t totals[2] = {0, 0};  // bricks in all chunk regions (counts + slack); chunks that need a block of march cells
        HIP_TRY(c, hipMemcpyAsync(totals, c->d_brick_total, sizeof totals, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        const uint32_t total = totals[0];
        // the march cells: blocks 0 and 1, one per chunk that needs its own, and room for the chunks that may come to need one
        // before the next whole-world build (every one of them is a chunk rebuilt alone: at most kTailChunks)
        const uint64_t want_blocks = direct ? (direct_cell_entries(S) + 511u) / 512u : 2ull + totals[1] + kTailChunks;
        if (want_blocks > kMarchBlocksMax) {
            for (auto &T : c->tabs) { (void)hipFree(T.d_mblk); T.d_mblk = nullptr; T.mblk_cap = 0; }
        } else {
            if (want_blocks > A.mblk_cap || !A.d_mblk) {
                (void)hipFree(A.d_mblk);
                A.d_mblk = nullptr; A.mblk_cap = 0;
                uint64_t cap = want_blocks + (direct ? 0u : totals[1] / 8u);
                if (cap > kMarchBlocksMax) cap = kMarchBlocksMax;
                HIP_TRY(c, hipMalloc(&A.d_mblk, (size_t)cap * 512u * sizeof(uint4)));
                A.mblk_cap = (uint32_t)cap;
            }
            // block 0: every cell stops the ray (direct: so do the blocks of the directory's border)
            HIP_TRY(c, hipMemsetAsync(A.d_mblk, 0, (direct ? (size_t)want_blocks : (size_t)1) * 512u * sizeof(uint4), c->stream));
        }
        const uint64_t want = (uint64_t)total + (uint64_t)kTailChunks * 512u;
        if (want > kAccelMaxBricks) return VRT_OK;  // accel_ok stays false
        if (want > A.brick_cap || !A.d_bricks) {
            (void)hipFree(A.d_bricks);
            A.d_bricks = nullptr; A.brick_cap = 0;
            uint64_t cap = want + total / 4u;  // room to grow before the next reallocation
            if (cap > kAccelMaxBricks) cap = kAccelMaxBricks;
            HIP_TRY(c, hipMalloc(&A.d_bricks, (size_t)cap * 64u * sizeof(uint16_t)));
            A.brick_cap = (uint32_t)cap;
        }
        vrt::launch_accel_bricks(c->d_nodes, c->max_nodes, c->d_roots, S, A.d_grid, A.d_chunk_bases, A.d_bricks, A.brick_cap, direct ? nullptr : A.d_cdir, A.d_mblk,
                                 A.d_mblk_tail, A.mblk_cap, c->liquid_mask, c->stream);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipEventRecord(e1, c->stream));
        HIP_TRY(c, hipEventSynchronize(e1));
        HIP_TRY(c, hipEventElapsedTime(&c->accel_last_ms, e0, e1));
        c->n_bricks = total;
        c->accel_builds += 1;
        c->accel_ok = true;
        std::fill(A.chunk_may_have_moved.begin(), A.chunk_may_have_moved.end(), (uint8_t)0);
        A.chunks_moved = 0;
        A.update_pending = false;
        A.live = true;
        for (uint32_t k = 1; k < vrt_ctx::kMaxInFlight; k++) {   // the other sets in use start over as copies
            if (c->tabs[k].live) {
                const int rc = copy_tables_from_first(c, k);
                if (rc) return rc;
            } else if (k < c->in_flight) {   // (memory for the sets the first edit will want: an allocation is milliseconds)
                const int rc = alloc_tables_like_first(c, k);
                if (rc) return rc;
            }
        }
        return publish_upload(c);
    };
    const int rc = body();
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc) c->accel_dirty = true;
    return rc;
}

// The frame about to be enqueued on `st` uses table set `k`: bring it up to date there.  `st` has been made to wait for the
// uploads so far (frame_waits_for_uploads).
int update_tables(vrt_ctx *c, uint32_t k, hipStream_t st) {
    if (!c->accel_ok || c->accel_dirty) return VRT_OK;
    auto &T = c->tabs[k];
    if (!T.live) {   // this frame set's first frame since the last whole-world build of a smaller crowd: a copy of tabs[0]
        // (wait for the frames in flight; the frame about to be enqueued has already been announced on its stream)
        const bool alt = c->alt_pending, own = c->own_pending;
        QUIESCE(c);
        c->alt_pending = alt;
        c->own_pending = own;
        {
            const int rc = wait_for_pool_uploads(c, c->stream, vrt_ctx::kMaxInFlight);
            if (rc) return rc;
        }
        if (c->tabs[0].update_pending) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->tabs[0].ev_updated, 0));
        int rc = copy_tables_from_first(c, k);
        if (rc) return rc;
        rc = publish_upload(c);
        if (rc) return rc;
        if (st != c->stream) HIP_TRY(c, hipStreamWaitEvent(st, c->ev_upload, 0));
    }
    if (T.dirty_chunks.empty()) return VRT_OK;
    if (k == 0u && c->shared_readers_in_flight) {   // frames of other frame sets still read this set (it was shared until now)
        const bool alt = c->alt_pending, own = c->own_pending;
        QUIESCE(c);
        c->alt_pending = alt;
        c->own_pending = own;
    }
    // how far each chunk's nodes can reach: up to the next chunk's root (ChunkAlloc's ranges are disjoint); the kernel
    // stages that much of the pool and reads anything beyond from the pool itself
    refresh_roots_index(c);
    std::vector<uint32_t> extents(T.dirty_chunks.size()), roots_now(T.dirty_chunks.size());
    for (size_t i = 0; i < extents.size(); i++) {
        const uint32_t r = T.dirty_chunks[i] < c->n_roots ? c->h_roots[T.dirty_chunks[i]] : 0u;
        roots_now[i] = r;
        auto nx = std::upper_bound(c->roots_index.begin(), c->roots_index.end(), std::make_pair(r, 0xFFFFFFFFu));
        const uint32_t end = nx != c->roots_index.end() ? nx->first : c->max_nodes;
        extents[i] = r ? (end > r ? end - r : 0u) : 1u;   // (a missing chunk is node 0 alone: one air leaf)
    }
    if (!T.ev_updated) HIP_TRY(c, hipEventCreateWithFlags(&T.ev_updated, hipEventDisableTiming));
    // (the next upload of nodes or roots waits for this reader: ev_updated is the completion signal of the launch itself — a
    // record behind it is one more packet between the rebuild and the frame that waits for it)
    vrt::launch_accel_chunks(c->d_nodes, c->max_nodes, c->d_roots, c->accel_S, T.d_grid, T.d_chunk_bricks, T.d_chunk_bases, T.d_chunk_caps,
                             T.d_brick_tail, T.d_bricks, T.brick_cap, c->march_direct ? nullptr : T.d_cdir, T.d_mblk, T.d_mblk_tail, T.mblk_cap, c->liquid_mask,
                             T.dirty_chunks.data(), extents.data(), roots_now.data(), (uint32_t)T.dirty_chunks.size(), st, T.ev_updated);
    HIP_TRY(c, hipGetLastError());
    T.update_pending = true;
    for (uint32_t ch : T.dirty_chunks) {
        if (!T.chunk_may_have_moved[ch]) { T.chunk_may_have_moved[ch] = 1; T.chunks_moved += 1; }
        T.chunk_is_dirty[ch] = 0;
    }
    T.chunk_builds += (uint32_t)T.dirty_chunks.size();
    T.dirty_chunks.clear();
    return VRT_OK;
}

// tabs[0] is what vrt_get_accel_info / vrt_read_accel report: apply what it has not caught up with (the frames since may
// have run on other frame sets).  Waits for the frames in flight.
static int first_tables_up_to_date(vrt_ctx *c) {
    if (!c->accel_ok || c->accel_dirty || !c->tabs[0].live) return VRT_OK;
    QUIESCE(c);
    int rc = wait_for_pool_uploads(c, c->stream, vrt_ctx::kMaxInFlight);
    if (rc) return rc;
    if (c->ev_upload && c->stream != c->own_stream) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_upload, 0));
    rc = update_tables(c, 0, c->stream);
    if (rc) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return VRT_OK;
}

// Bricks of the pool in use: the chunks' regions plus the tail regions of chunks that moved.
static int bricks_in_use(vrt_ctx *c, const vrt_ctx::Tables &T, uint32_t *n) {
    *n = 0;
    if (!c->accel_ok || !T.live || !T.d_brick_tail) return VRT_OK;
    QUIESCE(c);   // (the set's last update may be on another frame stream)
    HIP_TRY(c, hipMemcpyAsync(n, T.d_brick_tail, sizeof *n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (*n > T.brick_cap) *n = T.brick_cap;
    return VRT_OK;
}

extern "C" {

int vrt_write_nodes(vrt_ctx *c, const uint16_t *pool, uint32_t start, uint32_t end) {
    GRP_EACH(c, vrt_write_nodes(d, pool, start, end));
    if (!c || !pool) return fail(c, VRT_ERR_INVALID_ARG, "vrt_write_nodes: null argument");
    if (end < start) return fail(c, VRT_ERR_INVALID_ARG, "vrt_write_nodes: end %u < start %u", end, start);
    // NodeBuffer::write, shader.rs:24-33: widen to even bounds
    uint32_t root = start, count = end - start;
    if (root % 2 == 1) { root -= 1; count += 1; }
    if (count % 2 == 1) count += 1;
    if (count == 0) return VRT_OK;
    if ((uint64_t)root + count > c->max_nodes)
        return fail(c, VRT_ERR_OUT_OF_RANGE, "vrt_write_nodes: [%u,%u) exceeds the %u-node buffer", start, end, c->max_nodes);
    HIP_TRY(c, hipSetDevice(c->device));
    // copy-at-call-time (write_buffer semantics: the caller may reuse `pool` as soon as this returns) through the pinned
    // ring, ordered after the frames in flight without waiting for them
    const int rc = stage_upload(c, c->d_nodes + root, pool + root, (size_t)count * sizeof(uint16_t), true);
    if (rc) return rc;
    mark_node_range_dirty(c, start, end);   // (the widening repeats a neighbour's node: nothing of its octree changes)
    return VRT_OK;
}

int vrt_write_chunk_roots(vrt_ctx *c, uint32_t offset, const uint32_t *roots, uint32_t n) { return vrt_write_chunk_roots_tagged(c, offset, roots, n, 0); }

int vrt_write_chunk_roots_tagged(vrt_ctx *c, uint32_t offset, const uint32_t *roots, uint32_t n, uint64_t tag) {
    GRP_EACH(c, vrt_write_chunk_roots_tagged(d, offset, roots, n, tag));
    if (!c || (!roots && n)) return fail(c, VRT_ERR_INVALID_ARG, "vrt_write_chunk_roots: null argument");
    // the reference rewrites the whole table every frame (main.rs:446).  A caller that can say "nothing changed since the
    // write I tagged like this" is believed: the 128 KB compare of a 32^3 table is half a frame's host time
    if (tag != 0 && tag == c->roots_tag && offset == c->roots_tag_offset && n == c->roots_tag_n) return VRT_OK;
    c->roots_tag = 0;
    if (offset > c->n_roots) return fail(c, VRT_ERR_OUT_OF_RANGE, "vrt_write_chunk_roots: offset %u > %u", offset, c->n_roots);
    // ArrayBuffer::write truncates to capacity (shader.rs:134-135)
    const uint32_t cut = n < c->n_roots - offset ? n : c->n_roots - offset;
    if (cut == 0) return VRT_OK;
    auto remember = [&]() { c->roots_tag = tag; c->roots_tag_offset = offset; c->roots_tag_n = n; };
    // ... an identical rewrite changes nothing
    if (memcmp(c->h_roots.data() + offset, roots, (size_t)cut * sizeof(uint32_t)) == 0) { remember(); return VRT_OK; }
    HIP_TRY(c, hipSetDevice(c->device));
    const int rc = stage_upload(c, c->d_roots + offset, roots, (size_t)cut * sizeof(uint32_t), true);
    if (rc) return rc;
    // the slots whose root changed are the chunks to rebuild (a chunk arrived or was dropped); a recentred grid changes
    // nearly all of them and becomes a whole-world build
    for (uint32_t i = 0; i < cut && !c->accel_dirty; i++)
        if (c->h_roots[offset + i] != roots[i]) mark_chunk_dirty(c, offset + i);
    memcpy(c->h_roots.data() + offset, roots, (size_t)cut * sizeof(uint32_t));
    c->roots_index_stale = true;
    remember();
    return VRT_OK;
}

int vrt_resize_world(vrt_ctx *c, uint32_t world_size_chunks) {
    GRP_EACH(c, vrt_resize_world(d, world_size_chunks));
    if (!c) return VRT_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    QUIESCE(c);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return alloc_roots(c, world_size_chunks);
}

int vrt_write_materials(vrt_ctx *c, uint32_t first, const vrt_material *mats, uint32_t n) {
    GRP_EACH(c, vrt_write_materials(d, first, mats, n));
    if (!c || (!mats && n)) return fail(c, VRT_ERR_INVALID_ARG, "vrt_write_materials: null argument");
    if ((uint64_t)first + n > 256) return fail(c, VRT_ERR_OUT_OF_RANGE, "vrt_write_materials: %u+%u > 256", first, n);
    if (n == 0) return VRT_OK;
    c->view_gen++;
    memcpy(c->h_mats + first, mats, (size_t)n * sizeof(vrt_material));
    uint32_t old_mask[8];
    memcpy(old_mask, c->liquid_mask, sizeof old_mask);
    memset(c->liquid_mask, 0, sizeof c->liquid_mask);
    int lo = -1, hi = -1, n_liquid = 0;
    for (int v = 0; v < 256; v++)
        if (c->h_mats[v].is_liquid == 1u) {
            c->liquid_mask[v >> 5] |= 1u << (v & 31);
            if (lo < 0) lo = v;
            hi = v;
            n_liquid++;
        }
    // one contiguous id range, and material 255 (which every id >= 255 clamps to) not in it
    c->liquid_is_range = n_liquid == 0 || (hi - lo + 1 == n_liquid && hi < 255);
    c->liquid_lo = n_liquid ? (uint32_t)lo : 0x80000000u;
    c->liquid_span = n_liquid ? (uint32_t)(hi - lo) : 0u;
    // the march cells say which voxels stop a ray: another set of liquids is another set of tables (a join-time event)
    if (memcmp(old_mask, c->liquid_mask, sizeof old_mask) != 0) mark_all_dirty(c);
    HIP_TRY(c, hipSetDevice(c->device));
    return stage_upload(c, c->d_mats + first, mats, (size_t)n * sizeof(vrt_material));
}

int vrt_get_accel_info(vrt_ctx *c, vrt_accel_info *out) {
    GRP_ROOT(c, vrt_get_accel_info(d, out));
    if (!c || !out) return fail(c, VRT_ERR_INVALID_ARG, "vrt_get_accel_info: null argument");
    memset(out, 0, sizeof *out);
    HIP_TRY(c, hipSetDevice(c->device));
    // (as of the table set the last frame used; the other sets catch up when their frame set renders next)
    const vrt_ctx::Tables &L = c->tabs[c->last_tab];
    const bool up_to_date = c->accel_ok && !c->accel_dirty && L.live && L.dirty_chunks.empty();
    out->available = up_to_date ? 1u : 0u;
    out->world_size_chunks = c->accel_S;
    out->cells = (uint64_t)c->accel_S * c->accel_S * c->accel_S * 512u;
    uint32_t used = 0;
    const int rc = bricks_in_use(c, L, &used);
    if (rc) return rc;
    out->bricks = used;
    const uint64_t G = (uint64_t)c->accel_S * 8u;
    out->bytes = G * (G + 1u) * (G + 1u) * sizeof(uint32_t) + (uint64_t)used * 64u * sizeof(uint16_t);
    out->builds = c->accel_builds;
    out->last_build_ms = c->accel_last_ms;
    out->ordered_frames = c->ordered_frames;
    for (const auto &T : c->tabs)
        if (T.live && T.chunk_builds > out->chunk_builds) out->chunk_builds = T.chunk_builds;   // every set rebuilds every dirty chunk once
    return VRT_OK;
}

int vrt_read_accel(vrt_ctx *c, uint32_t *grid, uint16_t *bricks) {
    GRP_ROOT(c, vrt_read_accel(d, grid, bricks));
    if (!c) return VRT_ERR_INVALID_ARG;
    if (!c->accel_ok || c->accel_dirty)
        return fail(c, VRT_ERR_STATE, "vrt_read_accel: the tables are not up to date (render a frame first)");
    HIP_TRY(c, hipSetDevice(c->device));
    {
        const int rc = first_tables_up_to_date(c);   // (the last frame may have used another frame set's tables)
        if (rc) return rc;
    }
    const size_t G = (size_t)c->accel_S * 8u, G1 = G + 1u;
    if (grid) {  // the device layout carries a zero border row / entry ([G][G+1][G+1]); the caller gets the G^3 cells
        std::vector<uint32_t> t(G * G1 * G1);
        HIP_TRY(c, hipMemcpyAsync(t.data(), c->tabs[0].d_grid, t.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        for (size_t z = 0; z < G; z++)
            for (size_t y = 0; y < G; y++) memcpy(grid + (z * G + y) * G, t.data() + (z * G1 + y) * G1, G * sizeof(uint32_t));
    }
    uint32_t used = 0;
    const int rc = bricks_in_use(c, c->tabs[0], &used);
    if (rc) return rc;
    if (bricks && used)
        HIP_TRY(c, hipMemcpyAsync(bricks, c->tabs[0].d_bricks, (size_t)used * 64u * sizeof(uint16_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return VRT_OK;
}

int vrt_read_march_cells(vrt_ctx *c, uint32_t *cells, uint32_t *direct) {
    GRP_ROOT(c, vrt_read_march_cells(d, cells, direct));
    if (!c || !cells) return fail(c, VRT_ERR_INVALID_ARG, "vrt_read_march_cells: null argument");
    if (!c->accel_ok || c->accel_dirty || !c->tabs[0].d_mblk)
        return fail(c, VRT_ERR_STATE, "vrt_read_march_cells: no march cells, or not up to date (render a frame first)");
    HIP_TRY(c, hipSetDevice(c->device));
    {
        const int rc = first_tables_up_to_date(c);   // (the last frame may have used another frame set's tables)
        if (rc) return rc;
    }
    const auto &T = c->tabs[0];
    const uint32_t S = c->accel_S;
    const size_t G = (size_t)S * 8u;
    if (direct) *direct = c->march_direct ? 1u : 0u;
    std::vector<uint4> blocks(c->march_direct ? direct_cell_entries(S) : (size_t)T.mblk_cap * 512u);
    std::vector<uint32_t> dir(chunk_dir_entries(S));
    HIP_TRY(c, hipMemcpyAsync(blocks.data(), T.d_mblk, blocks.size() * sizeof(uint4), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(dir.data(), T.d_cdir, dir.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const size_t B1 = (size_t)S * 4u + 1u;
    for (size_t z = 0; z < G; z++)
        for (size_t y = 0; y < G; y++)
            for (size_t x = 0; x < G; x++) {
                const size_t sub = (x & 1u) | ((y & 1u) << 1) | ((z & 1u) << 2);
                size_t at;
                if (c->march_direct) {
                    at = (((z >> 1) * B1 + (y >> 1)) * B1 + (x >> 1)) * 8u + sub;
                } else {
                    const uint32_t blk = dir[((z >> 3) * (S + 1u) + (y >> 3)) * (S + 1u) + (x >> 3)];
                    if (blk >= T.mblk_cap) return fail(c, VRT_ERR_DEVICE, "vrt_read_march_cells: the directory names block %u of %u", blk, T.mblk_cap);
                    const size_t line = ((((z >> 1) & 3u) << 2 | ((y >> 1) & 3u)) << 2) | ((x >> 1) & 3u);
                    at = (size_t)blk * 512u + line * 8u + sub;
                }
                memcpy(cells + ((z * G + y) * G + x) * 4u, &blocks[at], sizeof(uint4));
            }
    return VRT_OK;
}

}  // extern "C"
