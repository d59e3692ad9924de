// vrt_order.hip — the order a one-frame-at-a-time context launches its 8 x 8 tiles in (include/vrt.h: vrt_set_frames_in_flight).
// The dispatcher hands out workgroups in index order and a lone launch ends when its slowest late workgroup does: launched
// longest first, the last waves to start are the cheapest.  A view at rest gets the exact order of the trips its second frame
// noted (vrt_kernels.hip: launch_tile_order); a view that MOVES gets, once in a while, a block order dilated over the image's
// motion (launch_tile_order_blocks) that is KEPT while the camera stays within what the dilation covers.  Every order is a
// permutation of the same tiles: the frame is the same.  vrt_render calls tile_order_before_frame ahead of the frame's launch
// and tile_order_after_frame behind it.  profiles/r05_tile_order_moving.txt, DESIGN.md section 5.
#include "vrt_ctx.h"

#include <cmath>
#include <cstring>

// Whether camera b is near camera a: the same projection, the eye within `voxels`, every axis of the view within the angle whose
// cosine is `cos_angle` (a NaN camera is near nothing).
static bool cameras_within(const vrt_cam_data &a, const vrt_cam_data &b, float voxels, float cos_angle) {
    if (memcmp(a.inv_proj_mat, b.inv_proj_mat, sizeof a.inv_proj_mat) != 0 || memcmp(a.proj_size, b.proj_size, sizeof a.proj_size) != 0) return false;
    float d2 = 0.f;
    for (int k = 0; k < 3; k++) d2 += (a.pos[k] - b.pos[k]) * (a.pos[k] - b.pos[k]);
    if (!(d2 <= voxels * voxels)) return false;
    for (int col = 0; col < 3; col++) {
        float dot = 0.f, na = 0.f, nb = 0.f;
        for (int k = 0; k < 3; k++) {
            const float x = a.inv_view_mat[4 * col + k], y = b.inv_view_mat[4 * col + k];
            dot += x * y; na += x * x; nb += y * y;
        }
        if (!(dot >= cos_angle * sqrtf(na * nb))) return false;
    }
    return true;
}
// Whether a tile order made from camera a's frame, dilated over ~ 10 tiles each way, still serves camera b `steps` camera steps on:
// the eye within a voxel and a half per step, every axis of the view within two (four) degrees.
static bool cameras_close(const vrt_cam_data &a, const vrt_cam_data &b, float steps = 1.0f) {
    return cameras_within(a, b, 1.5f * steps, steps > 1.5f ? 0.99756f : 0.99939f);
}
// ... and the kept order of the default: a block order dilated over `radius` blocks of 32 pixels each way serves the views whose
// image has moved by less — every axis of the view within three quarters of the angle that many pixels are (1080p at 70 degrees,
// radius 5: 8.9 degrees), the eye within 1.3 voxels per block (6.5) — and is made again by the frame that passes three quarters
// of that.  The bench's orbit (0.8 voxels and ~ 1 degree a step) is served for 8 steps: the best of the sweep in
// profiles/r05_tile_order_moving.txt.
constexpr uint32_t kMovingTilesMax = 40000u;
struct HoldLimits { float voxels, cos_hold, cos_refresh; };
static HoldLimits hold_limits(const vrt_cam_data &cam, uint32_t width, uint32_t height, uint32_t radius) {
    const float ax = fabsf(cam.inv_proj_mat[0]), ay = fabsf(cam.inv_proj_mat[5]);   // the tangents of half the field of view
    float per_rad = fminf(0.5f * (float)width / ax, 0.5f * (float)height / ay);    // pixels per radian at the image's centre
    float angle = 0.75f * 32.0f * (float)radius / per_rad;
    if (!(angle >= 0.0f) || !(per_rad > 0.0f)) angle = 0.0f;                         // (a projection that is not one: only the same camera is near)
    angle = fminf(angle, 0.5f);
    return HoldLimits{1.3f * (float)radius, cosf(angle), cosf(0.75f * angle)};
}

// Ahead of the frame's launch: which order it launches in (P.tile_order), whether it notes its trips (P.tile_cost), and what
// tile_order_after_frame is to do with them.
int tile_order_before_frame(vrt_ctx *c, vrt::FrameParams &P, hipStream_t st, const vrt_render_opts &o, uint32_t variant, bool kstats, bool edit_in_front,
                            TileOrderPlan &plan) {
    // longest tiles first: the one-launch primary + shadow kernel over the derived tables, plain frames, one frame at a time
    // on the context's own stream (a frame, the sort behind it and the next frame are then ordered by the stream alone)
    const bool lpt = c->tile_lpt && c->in_flight == 1u && st == c->stream && (o.mode == VRT_MODE_PRIMARY_SHADOW || o.mode == VRT_MODE_PRIMARY) && variant == 0u && !kstats &&
                     o.stats == 0u && P.grid && c->tiles_local >= 128u;
    bool tile_sort = false, dilate = false;
    // (a tile's trips depend on the mode too — a primary-only frame has no shadow march: an order made from the other
    // mode's frame is a stale order, and the frame before a sort must be of the same kind)
    if (c->frame_mode != o.mode) c->view_gen++;
    // (VRT_TILE_ORDER_MOVING=2) the order made beside the last frame is for the frame after it: this one may use the one before
    c->mov_cur = c->mov_pend;
    c->mov_pend = c->mov_new;
    c->mov_new.valid = false;
    const bool mov2 = c->tile_lpt_moving == 2u;
    uint32_t mov_wb = 0;   // which of the two trips / order buffers this frame's trips go to
    if (lpt) {
        if (c->tile_buf_tiles != c->tiles_local) {
            const uint32_t chunks = (c->tiles_local + 63u) / 64u;
            HIP_TRY(c, hipMalloc(&c->d_tile_cost, (size_t)c->tiles_local * sizeof(uint32_t)));
            HIP_TRY(c, hipMalloc(&c->d_tile_order, (size_t)c->tiles_local * sizeof(uint32_t)));
            HIP_TRY(c, hipMalloc(&c->d_tile_scratch, (size_t)64u * (chunks + 1u) * sizeof(uint32_t)));
            c->tile_buf_tiles = c->tiles_local;
            c->tile_order_valid = false;
        }
        // an order is used by the very view it was made from, or — a dilated one — by a view whose camera the dilation still covers
        const bool exact = c->tile_order_valid && !c->order_dilated && c->order_view_gen == c->view_gen;
        // (a 4K frame's order is 8 160 blocks to sort — ~ 50 us —, holds for half as many camera steps, and shortens a 354-us frame
        // by the same ~ 5 us: 364 us per frame against 354 in screen order.  Frames of up to kMovingTilesMax tiles — 1080p: 32 400 — ask.)
        const bool moving_ok = c->tile_lpt_moving && c->tiles_local == P.tiles_total && P.tiles_total % P.tiles_x == 0u &&
                               (P.tiles_total <= kMovingTilesMax || c->mov_any_size);
        const bool hold = c->tile_lpt_moving == 1u;   // (6: an order a frame, for the very next one)
        const HoldLimits lim = hold_limits(c->cam, c->width, c->height, c->mov_radius);
        const bool near = !mov2 && c->tile_order_valid && c->order_dilated && moving_ok &&
                          c->view_gen - c->order_view_gen == c->cam_gen - c->order_cam_gen &&
                          (hold ? cameras_within(c->order_cam, c->cam, lim.voxels, lim.cos_hold) : cameras_close(c->order_cam, c->cam));
        if (!exact && !near) c->tile_order_valid = false;   // the order of another view: worse than none
        if (c->tile_order_valid) { P.tile_order = c->d_tile_order; c->ordered_frames++; }
        if (near) c->order_uses++;
        if (mov2 && moving_ok) {
            if (!c->d_mov_cost[0]) {
                for (int k = 0; k < 2; k++) {
                    HIP_TRY(c, hipMalloc(&c->d_mov_cost[k], (size_t)c->tiles_local * sizeof(uint32_t)));
                    HIP_TRY(c, hipMalloc(&c->d_mov_order[k], (size_t)c->tiles_local * sizeof(uint32_t)));
                    if (!c->mov_frame_done[k]) HIP_TRY(c, hipEventCreateWithFlags(&c->mov_frame_done[k], hipEventDisableTiming));
                    if (!c->mov_order_done[k]) HIP_TRY(c, hipEventCreateWithFlags(&c->mov_order_done[k], hipEventDisableTiming));
                }
                // the stream of the second frame in flight — idle while frames go one at a time, and known to run beside the
                // context's own (a stream made for the purpose shared its hardware queue: the order ran between the frames)
                if (!c->extra_stream[0]) HIP_TRY(c, hipStreamCreateWithFlags(&c->extra_stream[0], hipStreamNonBlocking));
                c->mov_stream = c->extra_stream[0];
            }
            mov_wb = c->mov_count & 1u;
            // the order made from the frame before the last one: for a view two camera steps from it at most, nothing but the camera changed
            const bool near2 = !c->tile_order_valid && c->mov_cur.valid && c->view_gen - c->mov_cur.view_gen == c->cam_gen - c->mov_cur.cam_gen &&
                               cameras_close(c->mov_cur.cam, c->cam, 2.0f);
            if (near2) { P.tile_order = c->d_mov_order[c->mov_cur.buf]; c->ordered_frames++; }
        }
        if (!exact) {
            if (c->frame_view_gen == c->view_gen) tile_sort = true;   // the view has come to rest: this frame notes its trips
            else if (moving_ok && !hold) tile_sort = dilate = true;   // it moves: the next frame's order from this frame's trips, dilated
            else if (moving_ok && !(near && cameras_within(c->order_cam, c->cam, 0.75f * lim.voxels, lim.cos_refresh))) {
                // it moves and has no order, or is about to leave the one it has: this frame's trips, dilated, for the frames to come
                if (c->mov_skip) c->mov_skip--;
                else tile_sort = dilate = true;
            }
        }
        // an order made before a chunk was edited: kept for the edit's own frame, made again by the first frame behind it
        // that has no fresh edit in front of it (its launch reads the old order, the sort behind it writes the new one)
        if (exact && c->tile_order_stale && !edit_in_front) tile_sort = true;
        if (tile_sort) P.tile_cost = (dilate && mov2) ? c->d_mov_cost[mov_wb] : c->d_tile_cost;
        // (the side stream's last reader of these trips / writer of this order buffer — two frames ago, or the order this frame
        // launches in — has finished before the frame starts)
        if (mov2 && c->mov_side && ((dilate && c->mov_order_recorded[mov_wb]) || P.tile_order == c->d_mov_order[mov_wb]))
            HIP_TRY(c, hipStreamWaitEvent(st, c->mov_order_done[mov_wb], 0));
        if (mov2 && c->mov_side && c->mov_cur.valid && P.tile_order == c->d_mov_order[c->mov_cur.buf] && c->mov_cur.buf != mov_wb)
            HIP_TRY(c, hipStreamWaitEvent(st, c->mov_order_done[c->mov_cur.buf], 0));
    }
    c->frame_view_gen = c->view_gen;
    c->frame_mode = o.mode;
    plan.sort = tile_sort;
    plan.dilate = dilate;
    plan.beside = mov2;
    plan.wb = mov_wb;
    return VRT_OK;
}

// Behind the frame's launch: the sort of the trips it noted, on the frame's stream (the frame read the old order and is over when
// the sort runs; the next frame starts after it) or, in the experiments build's form 2, beside the next frame.
int tile_order_after_frame(vrt_ctx *c, const vrt::FrameParams &P, hipStream_t st, const TileOrderPlan &plan) {
    const bool tile_sort = plan.sort, dilate = plan.dilate, mov2 = plan.beside;
    const uint32_t mov_wb = plan.wb;
    if (tile_sort) {   // (the frame above read the old order and is over when this runs; the next frame starts after it)
        bool made = true;
        if (dilate && mov2) {
            // beside the next frame: the side stream waits for this frame, sorts its trips, and says when the order is there
            hipStream_t os = c->mov_side ? c->mov_stream : st;
            if (c->mov_side) {
                HIP_TRY(c, hipEventRecord(c->mov_frame_done[mov_wb], st));
                HIP_TRY(c, hipStreamWaitEvent(os, c->mov_frame_done[mov_wb], 0));
            }
            made = vrt::launch_tile_order_blocks(c->d_mov_cost[mov_wb], P.tiles_x, P.tiles_total / P.tiles_x, 1u, c->mov_radius_set ? c->mov_radius : 3u, c->d_mov_order[mov_wb], os,
                                                 c->mov_side ? c->mov_threads : 1024u);
            HIP_TRY(c, hipGetLastError());
            if (c->mov_side) {
                HIP_TRY(c, hipEventRecord(c->mov_order_done[mov_wb], os));
                c->mov_order_recorded[mov_wb] = true;
                c->mov_pending = true;
            }
            c->mov_new.valid = made;
            c->mov_new.view_gen = c->view_gen;
            c->mov_new.cam_gen = c->cam_gen;
            c->mov_new.buf = mov_wb;
            c->mov_new.cam = c->cam;
            c->mov_count++;
        } else {
        if (dilate && c->tile_lpt_moving == 6u) vrt::g_exp.tile_order_moving(c->d_tile_cost, P.tiles_x, P.tiles_total / P.tiles_x, 1u, 2u, c->d_tile_scratch, c->d_tile_order, st);
        else if (dilate) {
            made = vrt::launch_tile_order_blocks(c->d_tile_cost, P.tiles_x, P.tiles_total / P.tiles_x, 1u, c->mov_radius, c->d_tile_order, st, 1024u);
            // orders that are not used — the view moves further per frame than they cover — are asked for less and less often
            c->mov_backoff = c->order_dilated && c->order_uses < 2u ? (c->mov_backoff ? (c->mov_backoff < 64u ? c->mov_backoff * 2u : 64u) : 1u) : 0u;
            c->mov_skip = c->mov_backoff;
            c->order_uses = 0;
        }
        else
        vrt::launch_tile_order(c->d_tile_cost, c->tiles_local, 1u, c->d_tile_scratch, c->d_tile_order, st);   // classes of two trips
        HIP_TRY(c, hipGetLastError());
        c->tile_order_valid = made;   // (a frame of more blocks than the one-launch order holds keeps screen order)
        c->order_view_gen = c->view_gen;
        c->order_dilated = dilate;
        c->order_cam_gen = c->cam_gen;
        c->order_cam = c->cam;
        c->tile_order_stale = false;
        }
    }
    return VRT_OK;
}
