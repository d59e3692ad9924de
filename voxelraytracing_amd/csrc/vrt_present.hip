// vrt_present.hip — ScreenShader::encode_pass (shader.rs:273-293, main.rs:454) and the gather root's assembly of
// tile-major messages into the row-major frame.
#include "vrt_ctx.h"

// ScreenShader::encode_pass into one of the context's screen buffers on the device, asynchronous: the blit is enqueued BEHIND
// THE FRAME it presents, on that frame's stream, into the screen buffer of the frame's set — a host that draws and presents
// frame after frame (main.rs:452-454) keeps its frames in flight; nothing here waits for the device.  *screen: the buffer.
static_assert(vrt_ctx::kMaxInFlight == 4, "one screen buffer per frame set");
// present_kernel's sample position for every column and row of a window of the texture's size, in the kernel's own binary32
// arithmetic (both sides divide correctly rounded; -ffp-contract=off): true if every pixel's sample is — to within 1e-4 of a texel
// — the centre of ITS OWN texel.  (Exactly the centre it is not for every size: at 1920 columns, 51 land 2e-6 to 8e-6 beside it —
// (s + 0.5) / 1920 * 1920 is not always s + 0.5 — and take 0.999998 of texel s and 0.000002 of its neighbour.)  Then
// present_plain_kernel's shortcut is present_kernel's byte: the bilinear sample of decoded unorm8 values v_i = q_i / 255 with at
// least (1 - 1e-4)^2 of the weight on v_s is within 2.1e-4 of v_s, 255 times that within 0.06 of the integer q_s, and the
// quantisation rounds it to q_s; likewise the alpha (0 or 1 per tap).  Inside the crosshair's box every pixel takes present_pixel.
static bool present_is_one_to_one(uint32_t w, uint32_t h) {
    for (int axis = 0; axis < 2; axis++) {
        const uint32_t n = axis ? h : w;
        const volatile float ss = (float)n;
        for (uint32_t s = 0; s < n; s++) {
            const volatile float u = ((float)s + 0.5f) / ss;
            const volatile float ut = u * ss - 0.5f;
            const float fu = floorf(ut);
            const volatile float a = ut - fu;
            int x0 = (int)fu, x1 = (int)fu + 1;
            x0 = x0 < 0 ? 0 : (x0 > (int)n - 1 ? (int)n - 1 : x0);
            x1 = x1 < 0 ? 0 : (x1 > (int)n - 1 ? (int)n - 1 : x1);
            const bool on_x0 = a <= 1.0e-4f && (uint32_t)x0 == s, on_x1 = a >= 1.0f - 1.0e-4f && (uint32_t)x1 == s;
            if (!(on_x0 || on_x1)) return false;
        }
    }
    return true;
}

// The pixels the crosshair's mask can reach: those within `size` of the centre, and two more for the rounding of px, py.
static void crosshair_box(const vrt_crosshair *crosshair, uint32_t screen_w, uint32_t screen_h, uint32_t box[4]) {
    box[0] = box[1] = box[2] = box[3] = 0u;
    if (crosshair->style == 0u) return;   // (for every window: the general kernel asks its pixels the same question)
    const float reach = crosshair->size + 2.0f;
    const float cx = (float)screen_w * 0.5f, cy = (float)screen_h * 0.5f;
    // (a colour or an alpha that is not a number reaches every pixel: 0 * NaN is not 0)
    bool finite = true;
    for (int k = 0; k < 4; k++) finite = finite && crosshair->color[k] - crosshair->color[k] == 0.0f;
    if (finite && reach == reach && reach > 0.0f && reach < 1.0e6f) {
        const float x0 = floorf(cx - reach - 0.5f), x1 = ceilf(cx + reach + 0.5f), y0 = floorf(cy - reach - 0.5f), y1 = ceilf(cy + reach + 0.5f);
        box[0] = x0 > 0.0f ? (uint32_t)x0 : 0u; box[1] = x1 < (float)screen_w ? (uint32_t)(x1 > 0.0f ? x1 : 0.0f) : screen_w;
        box[2] = y0 > 0.0f ? (uint32_t)y0 : 0u; box[3] = y1 < (float)screen_h ? (uint32_t)(y1 > 0.0f ? y1 : 0.0f) : screen_h;
    } else {   // a size that is not a number, negative or huge; a colour that is not finite: every pixel decides for itself
        box[1] = screen_w; box[3] = screen_h;
    }
}

// store_screen (vrt_tile.h) fetches a box pixel's taps from the lanes of the pixel's own 8x8 tile: true if, along this axis,
// every pixel s of [lo, hi) samples — in present_sample's own binary32 arithmetic — either its own texel's centre exactly or two
// taps that both lie in s's tile.  (A tap of weight zero may be any finite texel: x * 0 is 0 whatever x.)
static bool box_taps_in_tile(uint32_t n, uint32_t lo, uint32_t hi) {
    const volatile float ss = (float)n;
    for (uint32_t s = lo; s < hi && s < n; s++) {
        const volatile float u = ((float)s + 0.5f) / ss;
        const volatile float ut = u * ss - 0.5f;
        const float fu = floorf(ut);
        const volatile float a = ut - fu;
        int x0 = (int)fu, x1 = (int)fu + 1;
        x0 = x0 < 0 ? 0 : (x0 > (int)n - 1 ? (int)n - 1 : x0);
        x1 = x1 < 0 ? 0 : (x1 > (int)n - 1 ? (int)n - 1 : x1);
        const bool own = a == 0.0f && (uint32_t)x0 == s;
        const bool both = ((uint32_t)x0 >> 3) == (s >> 3) && ((uint32_t)x1 >> 3) == (s >> 3);
        if (!(own || both)) return false;
    }
    return true;
}

// Whether frames of the context's current size can store their own window pixels (include/vrt.h: vrt_set_presentation); found
// once per size and declaration.
bool presentation_fusable(vrt_ctx *c) {
    if (!c->pres_on) return false;
    if (c->pres_for_w != c->width || c->pres_for_h != c->height) {
        c->pres_for_w = c->width; c->pres_for_h = c->height;
        c->pres_fusable = false;
        if (c->pres_w == c->width && c->pres_h == c->height && !ragged_output(c) && !c->tile_major && !c->compact && c->shard_count == 1u && !c->whole_frame_owner) {
            if (c->one_w != c->width || c->one_h != c->height) {
                c->one_to_one = present_is_one_to_one(c->width, c->height);
                c->one_w = c->width; c->one_h = c->height;
            }
            crosshair_box(&c->pres_ch, c->width, c->height, c->pres_box);
            c->pres_fusable = c->one_to_one && box_taps_in_tile(c->width, c->pres_box[0], c->pres_box[1]) && box_taps_in_tile(c->height, c->pres_box[2], c->pres_box[3]);
        }
    }
    return c->pres_fusable;
}

// Frame set k's screen buffer, large enough for a screen_w x screen_h image and free of its previous blit before anything enqueued
// on `st` from here on writes it.
int screen_buffer_for_frame(vrt_ctx *c, uint32_t k, hipStream_t st, uint32_t screen_w, uint32_t screen_h) {
    const size_t bytes = (size_t)screen_w * screen_h * 4u;
    // the buffer's previous blit ran on the frame set's stream of that time: almost always this one
    if (c->screen_stream[k] && c->screen_stream[k] != st) HIP_TRY(c, hipStreamSynchronize(c->screen_stream[k]));
    if (bytes > c->screen_cap[k]) {
        HIP_TRY(c, hipStreamSynchronize(st));   // an earlier present may still be writing the old buffer
        (void)hipFree(c->d_screen[k]);
        c->d_screen[k] = nullptr; c->screen_cap[k] = 0;
        HIP_TRY(c, hipMalloc(&c->d_screen[k], bytes));
        c->screen_cap[k] = bytes;
    }
    c->screen_stream[k] = st;
    return VRT_OK;
}

static int present_on_device(vrt_ctx *c, const vrt_crosshair *crosshair, uint32_t screen_w, uint32_t screen_h, const char *who,
                             uint8_t **screen, hipStream_t *stream) {
    if (!c->rendered) return fail(c, VRT_ERR_STATE, "%s: nothing rendered yet", who);
    if (c->tile_major || (c->shard_count > 1u && !c->whole_frame_owner))
        return fail(c, VRT_ERR_STATE, "%s: needs the whole row-major frame", who);
    if (screen_w == 0u || screen_h == 0u || (uint64_t)screen_w * screen_h > (1ull << 28))
        return fail(c, VRT_ERR_INVALID_ARG, "%s: screen %ux%u out of range", who, screen_w, screen_h);
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t st = c->last_stream ? c->last_stream : c->stream;
    const uint32_t k = c->last_slot < vrt_ctx::kMaxInFlight ? c->last_slot : 0u;
    // the frame stored the window's image itself (vrt_set_presentation): it is there — or on its way, on the frame's stream
    if (c->last_fused && screen_w == c->pres_w && screen_h == c->pres_h && memcmp(crosshair, &c->pres_ch, sizeof *crosshair) == 0) {
        *screen = c->d_screen[k];
        *stream = st;
        return VRT_OK;
    }
    if (!c->last_has_texels)
        return fail(c, VRT_ERR_STATE, "%s: the last frame stored its window pixels only (VRT_PRESENT_SKIP_TEXELS): it can be presented with the declared "
                    "crosshair and size, nothing else", who);
    int rc = screen_buffer_for_frame(c, k, st, screen_w, screen_h);
    if (rc) return rc;
    c->last_fused = false;   // (the buffer now holds this blit; asked again with the declared crosshair, the frame is blitted again)
    // a window of the texture's size: is every pixel's sample its own texel's centre?  (found once per size, below)
    if (screen_w == c->width && screen_h == c->height && (c->one_w != screen_w || c->one_h != screen_h)) {
        c->one_to_one = present_is_one_to_one(screen_w, screen_h);
        c->one_w = screen_w; c->one_h = screen_h;
    }
    const bool one = screen_w == c->width && screen_h == c->height && c->one_to_one;
    uint32_t box[4];
    crosshair_box(crosshair, screen_w, screen_h, box);
    vrt::launch_present(c->last_out, c->width, c->height, screen_w, screen_h, *crosshair, c->d_screen[k], one, box, st);
    HIP_TRY(c, hipGetLastError());
    *screen = c->d_screen[k];
    *stream = st;
    return VRT_OK;
}

extern "C" {

int vrt_present(vrt_ctx *c, const vrt_crosshair *crosshair, uint32_t screen_w, uint32_t screen_h, uint8_t *rgba8) {
    if (c && c->grp) { const int rc_ = grp_synchronize(c); if (rc_) return rc_; }
    GRP_ROOT(c, vrt_present(d, crosshair, screen_w, screen_h, rgba8));
    if (!c || !crosshair || !rgba8) return fail(c, VRT_ERR_INVALID_ARG, "vrt_present: null argument");
    uint8_t *screen = nullptr;
    hipStream_t st = nullptr;
    const int rc = present_on_device(c, crosshair, screen_w, screen_h, "vrt_present", &screen, &st);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(rgba8, screen, (size_t)screen_w * screen_h * 4u, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    return VRT_OK;
}

int vrt_present_device(vrt_ctx *c, const vrt_crosshair *crosshair, uint32_t screen_w, uint32_t screen_h, void **rgba8_device, uint64_t *bytes) {
    if (c && c->grp) { const int rc_ = grp_synchronize(c); if (rc_) return rc_; }
    GRP_ROOT(c, vrt_present_device(d, crosshair, screen_w, screen_h, rgba8_device, bytes));
    if (!c || !crosshair || !rgba8_device) return fail(c, VRT_ERR_INVALID_ARG, "vrt_present_device: null argument");
    uint8_t *screen = nullptr;
    hipStream_t st = nullptr;
    const int rc = present_on_device(c, crosshair, screen_w, screen_h, "vrt_present_device", &screen, &st);
    if (rc) return rc;
    *rgba8_device = screen;
    if (bytes) *bytes = (uint64_t)screen_w * screen_h * 4u;
    return VRT_OK;
}

int vrt_set_presentation(vrt_ctx *c, const vrt_crosshair *crosshair, uint32_t screen_w, uint32_t screen_h, uint32_t flags) {
    GRP_ROOT(c, vrt_set_presentation(d, crosshair, screen_w, screen_h, flags));   // (a multi-device frame is assembled from messages: never stored by its march)
    if (!c) return VRT_ERR_INVALID_ARG;
    if (flags & ~VRT_PRESENT_SKIP_TEXELS) return fail(c, VRT_ERR_INVALID_ARG, "vrt_set_presentation: unknown flags 0x%x", flags);
    if (crosshair && (screen_w == 0u || screen_h == 0u || (uint64_t)screen_w * screen_h > (1ull << 28)))
        return fail(c, VRT_ERR_INVALID_ARG, "vrt_set_presentation: screen %ux%u out of range", screen_w, screen_h);
    // (the reference writes its blit's uniforms every frame, main.rs:429-432: an unchanged declaration is a compare)
    if (crosshair && c->pres_on && c->pres_w == screen_w && c->pres_h == screen_h && c->pres_flags == flags && memcmp(crosshair, &c->pres_ch, sizeof *crosshair) == 0)
        return VRT_OK;
    c->pres_on = crosshair != nullptr;
    if (crosshair) c->pres_ch = *crosshair;
    c->pres_w = screen_w; c->pres_h = screen_h; c->pres_flags = flags;
    c->pres_for_w = c->pres_for_h = 0u;   // (found again at the next frame)
    c->pres_fusable = false;
    return VRT_OK;
}

int vrt_assemble(vrt_ctx *c, const void *gathered, uint64_t rank_stride_bytes, void *dst) {
    GRP_REFUSE(c, "vrt_assemble");
    if (!c) return VRT_ERR_INVALID_ARG;
    if (!gathered || !dst) return fail(c, VRT_ERR_INVALID_ARG, "vrt_assemble: null argument");
    if (rank_stride_bytes % 16u) return fail(c, VRT_ERR_INVALID_ARG, "vrt_assemble: rank stride must be a multiple of 16 bytes");
    HIP_TRY(c, hipSetDevice(c->device));
    const uint64_t stride = rank_stride_bytes ? rank_stride_bytes / 16u : (uint64_t)c->tiles_padded * 64u;
    const bool in_place = c->shard_count > 1u && !c->tile_major;  // VRT_FLAG_ROW_MAJOR root: its tiles are already in dst
    vrt::launch_assemble((const vrt::Texel *)gathered, (vrt::Texel *)dst, c->width, c->tiles_x, c->tiles_total, c->shard_w0,
                         c->shard_period, in_place, stride, c->stream);
    HIP_TRY(c, hipGetLastError());
    return VRT_OK;
}

int vrt_assemble_compact(vrt_ctx *c, const void *gathered, uint64_t rank_stride_bytes, void *dst) {
    GRP_REFUSE(c, "vrt_assemble_compact");
    if (!c) return VRT_ERR_INVALID_ARG;
    if (!gathered || !dst) return fail(c, VRT_ERR_INVALID_ARG, "vrt_assemble_compact: null argument");
    if (rank_stride_bytes % 8u) return fail(c, VRT_ERR_INVALID_ARG, "vrt_assemble_compact: rank stride must be a multiple of 8 bytes");
    if (!(c->shard_count > 1u && !c->tile_major))
        return fail(c, VRT_ERR_STATE, "vrt_assemble_compact: the gather root must be a VRT_FLAG_ROW_MAJOR shard context (it shades the "
                    "other ranks' records with its own uniforms and has its own tiles in the frame already)");
    int rc = validate_frame(c);
    if (rc) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    rc = ensure_ndc(c);
    if (rc) return rc;
    vrt::FrameParams P;
    memset(&P, 0, sizeof P);
    P.mats = c->d_mats;
    fill_uniforms(c, P);
    const uint64_t stride = rank_stride_bytes ? rank_stride_bytes / 8u : (uint64_t)c->tiles_padded * 64u;
    vrt::launch_assemble_shade(P, gathered, (vrt::Texel *)dst, c->shard_w0, c->shard_period, stride, c->stream);
    HIP_TRY(c, hipGetLastError());
    return VRT_OK;
}

}  // extern "C"
