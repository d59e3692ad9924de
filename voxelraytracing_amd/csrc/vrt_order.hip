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
// The kept order of a moving view: a block order dilated over `radius` blocks of 32 pixels each way serves the views whose
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
        const HoldLimits lim = hold_limits(c->cam, c->width, c->height, c->mov_radius);
        const bool near = c->tile_order_valid && c->order_dilated && moving_ok && c->view_gen - c->order_view_gen == c->cam_gen - c->order_cam_gen &&
                          cameras_within(c->order_cam, c->cam, lim.voxels, lim.cos_hold);
        if (!exact && !near) c->tile_order_valid = false;   // the order of another view: worse than none
        if (c->tile_order_valid) { P.tile_order = c->d_tile_order; c->ordered_frames++; }
        if (near) c->order_uses++;
        if (!exact) {
            if (c->frame_view_gen == c->view_gen) tile_sort = true;   // the view has come to rest: this frame notes its trips
            else if (moving_ok && !(near && cameras_within(c->order_cam, c->cam, 0.75f * lim.voxels, lim.cos_refresh))) {
                // it moves and has no order, or is about to leave the one it has: this frame's trips, dilated, for the frames to come
                if (c->mov_skip) c->mov_skip--;
                else tile_sort = dilate = true;
            }
        }
        // an order made before a chunk was edited: kept for the edit's own frame, made again by the first frame behind it
        // that has no fresh edit in front of it (its launch reads the old order, the sort behind it writes the new one)
        if (exact && c->tile_order_stale && !edit_in_front) tile_sort = true;
        // ... and a kept dilated order likewise (the edit may have moved a silhouette further than the dilation covers): this frame's
        // trips whatever the back-off says
        if (near && c->tile_order_stale && !edit_in_front && !tile_sort) tile_sort = dilate = true;
        if (tile_sort) P.tile_cost = c->d_tile_cost;
    }
    c->frame_view_gen = c->view_gen;
    c->frame_mode = o.mode;
    plan.sort = tile_sort;
    plan.dilate = dilate;
    return VRT_OK;
}

// Behind the frame's launch: the sort of the trips it noted, on the frame's stream (the frame read the old order and is over when
// the sort runs; the next frame starts after it).
int tile_order_after_frame(vrt_ctx *c, const vrt::FrameParams &P, hipStream_t st, const TileOrderPlan &plan) {
    if (!plan.sort) return VRT_OK;
    bool made = true;
    if (plan.dilate) {
        made = vrt::launch_tile_order_blocks(c->d_tile_cost, P.tiles_x, P.tiles_total / P.tiles_x, 1u, c->mov_radius, c->d_tile_order, st, 1024u);
        // orders that are not used — the view moves further per frame than they cover — are asked for less and less often
        c->mov_backoff = c->order_dilated && c->order_uses < 2u ? (c->mov_backoff ? (c->mov_backoff < 64u ? c->mov_backoff * 2u : 64u) : 1u) : 0u;
        c->mov_skip = c->mov_backoff;
        c->order_uses = 0;
    } else {
        vrt::launch_tile_order(c->d_tile_cost, c->tiles_local, 1u, c->d_tile_scratch, c->d_tile_order, st);   // classes of two trips
    }
    // (a launch that could not be made — more blocks than the one-launch order holds, an LDS size the device refuses — leaves the
    // frames in screen order; it is not the frame's error)
    if (hipGetLastError() != hipSuccess) made = false;
    c->tile_order_valid = made;
    c->order_view_gen = c->view_gen;
    c->order_dilated = plan.dilate;
    c->order_cam_gen = c->cam_gen;
    c->order_cam = c->cam;
    c->tile_order_stale = false;
    return VRT_OK;
}
