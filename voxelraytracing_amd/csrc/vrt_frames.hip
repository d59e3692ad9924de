// vrt_frames.hip — GpuResources::new / PixelShader::encode_pass (clientdesktop/src/graphics/mod.rs:155-195,
// shader.rs:371-379; callers main.rs:211-223, 426-454): contexts, the per-frame uniforms, frames in flight, vrt_render,
// read-backs and statistics.  Device memory layout: DESIGN.md section 4.
#include "vrt_ctx.h"

thread_local std::string g_create_err;

int fail(vrt_ctx *ctx, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    else g_create_err = buf;
    return code;
}

// Wait for the frame that may still be running on the second stream.
vrt_host_prof g_host_prof;
namespace vrt { ExpHooks g_exp; }   // all null here: experiments/vrt_exp_register.hip fills them in tools/ab/libvrt_exp.so

int quiesce(vrt_ctx *c) {
    if (c->alt_pending) {
        for (hipStream_t st : c->extra_stream)
            if (st) HIP_TRY(c, hipStreamSynchronize(st));
        c->alt_pending = false;
    }
    if (c->own_pending) {
        HIP_TRY(c, hipStreamSynchronize(c->own_stream));
        c->own_pending = false;
    }
    c->shared_readers_in_flight = false;
    return VRT_OK;
}

// The reference dispatches tex_size / 8 workgroups per axis (main.rs:452) over a result texture of any size
// (main.rs:257-262: 1080 rows, the window's aspect): the columns and rows beyond the last whole 8x8 tile are never stored to
// and keep the fresh texture's zeros.  Buffers that are read as whole frames start out zero for such a size.
bool ragged_output(const vrt_ctx *c) { return ((c->width | c->height) & 7u) != 0u; }
hipError_t zero_now(vrt_ctx *c, void *p, size_t bytes) {
    const hipError_t e = hipMemsetAsync(p, 0, bytes, c->stream);
    return e != hipSuccess ? e : hipStreamSynchronize(c->stream);   // (done before a launch on any other stream can follow)
}

static void layout_tiles(vrt_ctx *c) {
    c->tiles_x = c->width / 8u;
    c->tiles_total = c->tiles_x * (c->height / 8u);
    // tiles are dealt out in periods of P = w0 + N - 1: w0 to rank 0, then one to each of ranks 1..N-1
    const uint32_t P = c->shard_w0 + c->shard_count - 1u;
    c->shard_period = P;
    c->shard_run = c->shard_rank == 0 ? c->shard_w0 : 1u;
    c->shard_first = c->shard_rank == 0 ? 0u : c->shard_w0 + c->shard_rank - 1u;
    const uint32_t full = c->tiles_total / P, rem = c->tiles_total % P;
    c->tiles_padded = (c->tiles_total + P - 1u) / P;
    c->tiles_local = full * c->shard_run + (rem > c->shard_first ? (rem - c->shard_first < c->shard_run ? rem - c->shard_first : c->shard_run) : 0u);
    const uint32_t tm_tiles = c->tiles_local > c->tiles_padded ? c->tiles_local : c->tiles_padded;
    c->slots = c->tile_major ? tm_tiles * 64u : c->width * c->height;
}

static int alloc_output(vrt_ctx *c) {
    (void)hipFree(c->own_out); (void)hipFree(c->d_hits); (void)hipFree(c->d_steps); (void)hipFree(c->d_rgba8); (void)hipFree(c->d_path);
    (void)hipFree(c->d_blk_counts);
    c->own_out = nullptr; c->d_hits = nullptr; c->d_steps = nullptr; c->d_rgba8 = nullptr; c->d_path = nullptr; c->d_blk_counts = nullptr;
    for (auto &p : c->extra_out) { (void)hipFree(p); p = nullptr; }
    for (auto &p : c->extra_blk) { (void)hipFree(p); p = nullptr; }
    for (auto &p : c->extra_path) { (void)hipFree(p); p = nullptr; }
    for (auto &p : c->path_acc) { (void)hipFree(p); p = nullptr; }
    for (auto &p : c->path_grp_counts) { (void)hipFree(p); p = nullptr; }
    for (auto &n : c->path_grp_regions) n = 0;
    (void)hipFree(c->d_tile_cost); c->d_tile_cost = nullptr;
    (void)hipFree(c->d_tile_order); c->d_tile_order = nullptr;
    (void)hipFree(c->d_tile_scratch); c->d_tile_scratch = nullptr;
    c->tile_buf_tiles = 0;
    c->tile_order_valid = false;
    for (auto &n : c->path_acc_texels) n = 0;
    for (auto &n : c->path_buf_records) n = 0;
    layout_tiles(c);
    const size_t n = c->slots ? c->slots : 1;
    HIP_TRY(c, hipMalloc(&c->own_out, n * sizeof(vrt::Texel)));
    // hit buffer: kHitSegments segments, each able to hold every record its workgroups can produce
    const uint32_t nblocks = (c->tiles_local + 3u) / 4u;
    c->hit_seg_cap = ((nblocks + vrt::kHitSegments - 1u) / vrt::kHitSegments) * 256u;
    if (c->hit_seg_cap == 0) c->hit_seg_cap = 256u;
    HIP_TRY(c, hipMalloc(&c->d_hits, (size_t)vrt::kHitSegments * c->hit_seg_cap * sizeof(uint4)));  // >= nblocks * 256
    c->n_blocks = nblocks;
    // launched-ray counts: one per primary workgroup (two-launch variants) or one per tile (the one-launch kernel)
    const size_t ncnt = c->tiles_local ? c->tiles_local : 1;
    HIP_TRY(c, hipMalloc(&c->d_blk_counts, ncnt * sizeof(uint32_t)));
    HIP_TRY(c, hipMemsetAsync(c->d_blk_counts, 0, ncnt * sizeof(uint32_t), c->stream));
    HIP_TRY(c, hipMemsetAsync(c->own_out, 0, n * sizeof(vrt::Texel), c->stream));
    // (c->stream may be the caller's: a VRT_RENDER_OWN_STREAMS frame on own_stream is not ordered behind these memsets, and
    // result sizes that are not whole tiles rely on the zeros)
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    // the extra (stream, output, counts) sets of frames in flight are created when first used (vrt_render)
    c->d_out = c->own_out;  // a resize drops any caller-bound output (its size no longer matches)
    c->last_out = c->own_out;
    c->last_blk = c->d_blk_counts;
    c->rendered = false;
    c->last_fused = false;
    c->last_has_texels = true;
    return VRT_OK;
}

int validate_frame(vrt_ctx *c) {
    const uint32_t S = c->world.size_in_chunks;
    if (S == 0 || (uint64_t)S * S * S > c->n_roots)
        return fail(c, VRT_ERR_STATE, "world.size_in_chunks %u does not fit the %u-entry chunk_roots buffer "
                    "(call vrt_resize_world first, main.rs:441-445)", S, c->n_roots);
    if (c->world.size != S * 32u)
        return fail(c, VRT_ERR_STATE, "world.size %u != size_in_chunks*32 (%u)", c->world.size, S * 32u);
    return VRT_OK;
}

// Frame-uniform pieces of create_ray_from_screen (ray_tracer.wgsl:160-161): one IEEE divide per column and row
// instead of two per pixel.  Same operations in the same order as the shader text, in binary32.
int ensure_ndc(vrt_ctx *c) {
    if (c->d_ndc && c->ndc_w == c->width && c->ndc_h == c->height &&
        memcmp(c->ndc_proj, c->cam.proj_size, sizeof c->ndc_proj) == 0)
        return VRT_OK;
    if (!c->d_ndc || c->ndc_w + c->ndc_h < c->width + c->height) {
        QUIESCE(c);
        (void)hipFree(c->d_ndc);
        c->d_ndc = nullptr;
        HIP_TRY(c, hipMalloc(&c->d_ndc, (size_t)(c->width + c->height) * sizeof(float)));
    }
    std::vector<float> t((size_t)c->width + c->height);
    volatile float px = c->cam.proj_size[0], py = c->cam.proj_size[1];
    for (uint32_t i = 0; i < c->width; i++) { volatile float q = ((float)(int)i * 2.0f) / px; t[i] = q - 1.0f; }
    for (uint32_t i = 0; i < c->height; i++) { volatile float q = ((float)(int)i * 2.0f) / py; t[c->width + i] = q - 1.0f; }
    const int rc = stage_upload(c, c->d_ndc, t.data(), t.size() * sizeof(float));
    if (rc) return rc;
    c->ndc_w = c->width;
    c->ndc_h = c->height;
    memcpy(c->ndc_proj, c->cam.proj_size, sizeof c->ndc_proj);
    return VRT_OK;
}

// A primary ray's origin (:169) and what ray_world asks of it before the first step — finite? on a voxel plane (the nudge, :188-190)? outside
// the world (:285)? — in the operations of create_ray / march_grid (vrt_march.h), once per frame instead of once per lane.
static void cam_origin_facts(const vrt_ctx *c, float origin[3], uint32_t *facts) {
    volatile float o[3];
    for (int k = 0; k < 3; k++) {
        volatile float wm = (float)c->world.min[k];
        o[k] = c->cam.pos[k] - wm;
        origin[k] = o[k];
    }
    uint32_t f = 0u;
    if (!(std::isfinite(o[0]) && std::isfinite(o[1]) && std::isfinite(o[2]))) f |= vrt::kCamNotFinite;
    bool on_plane = false, outside = false;
    volatile float world_max = 0.0f + (float)c->world.size;
    for (int k = 0; k < 3; k++) {
        volatile float fl = floorf(o[k]);
        volatile float frac = o[k] - fl;
        if (frac < 0.001f) on_plane = true;
        if (o[k] <= 0.0f || o[k] >= world_max) outside = true;
    }
    if (on_plane) f |= vrt::kCamOnPlane;
    if (outside) f |= vrt::kCamOutside;
    *facts = f;
}

// ray_sky's sun_dir for a ray starting at the camera (ray_tracer.wgsl:149, origin = cam.pos - world.min :169).
static void cam_sun_dir(const vrt_ctx *c, float out[3]) {
    volatile float d[3];
    for (int k = 0; k < 3; k++) {
        volatile float wm = (float)c->world.min[k];
        volatile float origin = c->cam.pos[k] - wm;
        volatile float a = c->settings.sun_pos[k] - wm;
        d[k] = a - origin;
    }
    volatile float xx = d[0] * d[0], yy = d[1] * d[1], zz = d[2] * d[2];
    volatile float s = xx + yy;
    volatile float dot = s + zz;
    volatile float len = sqrtf(dot);
    for (int k = 0; k < 3; k++) { volatile float q = d[k] / len; out[k] = q; }
}

// Fold the event triples of all frames rendered since the last call into acc_ms (synchronises).
static int fold_events(vrt_ctx *c, float last[3]) {
    if (c->ev_used == 0) return VRT_OK;
    QUIESCE(c);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < c->ev_used; i++) {
        auto &t = c->ev_pool[i];
        float a = 0, b = 0, tot = 0;
        switch (c->ev_kind[i]) {
            case kEvOneKernel:
                HIP_TRY(c, hipEventElapsedTime(&a, t[0], t[1]));
                tot = a;
                break;
            case kEvTwoKernels:
                HIP_TRY(c, hipEventElapsedTime(&a, t[0], t[1]));
                HIP_TRY(c, hipEventElapsedTime(&b, t[2], t[3]));
                HIP_TRY(c, hipEventElapsedTime(&tot, t[0], t[3]));
                break;
            case kEvRecorded:
                HIP_TRY(c, hipEventElapsedTime(&a, t[0], t[1]));
                HIP_TRY(c, hipEventElapsedTime(&b, t[1], t[3]));
                HIP_TRY(c, hipEventElapsedTime(&tot, t[0], t[3]));
                break;
            default:
                continue;  // an empty shard: nothing was launched
        }
        c->acc_ms[0] += a; c->acc_ms[1] += b; c->acc_ms[2] += tot;
        c->acc_frames += 1;
        if (last) { last[0] = a; last[1] = b; last[2] = tot; }
    }
    c->ev_used = 0;
    return VRT_OK;
}

// The frame's uniforms and scene pointers (everything of FrameParams that does not depend on where a frame is written).
void fill_uniforms(const vrt_ctx *c, vrt::FrameParams &P) {
    P.n_nodes = c->max_nodes;
    // only the S^3 entries the frame's WorldData describes are addressable (find_node :120-123)
    P.n_roots = c->world.size_in_chunks * c->world.size_in_chunks * c->world.size_in_chunks;
    P.width = c->width;
    P.height = c->height;
    P.tiles_x = c->tiles_x;
    // tile / tiles_x as a multiply-high: exact while tile * tiles_x < 2^32 (error of the rounded-up reciprocal x tile < 1)
    P.tiles_x_magic = (c->tiles_x > 1u && (uint64_t)c->tiles_total * c->tiles_x < (1ull << 32)) ? (uint32_t)(((1ull << 32) + c->tiles_x - 1u) / c->tiles_x) : 0u;
    P.tiles_total = c->tiles_total;
    P.shard_first = c->shard_first;
    P.shard_run = c->shard_run;
    P.shard_period = c->shard_period;
    P.tiles_local = c->tiles_local;
    P.tile_major = c->tile_major ? 1u : 0u;
    P.compact = c->compact ? 1u : 0u;
    P.cam = c->cam;
    P.settings = c->settings;
    P.world = c->world;
    const vrt_settings &s = c->settings;
    P.finite_settings = std::isfinite(s.sun_intensity) && std::isfinite(s.sky_color[0]) && std::isfinite(s.sky_color[1]) &&
                        std::isfinite(s.sky_color[2]) && std::isfinite(s.sun_pos[0]) && std::isfinite(s.sun_pos[1]) &&
                        std::isfinite(s.sun_pos[2]);
    memcpy(P.liquid, c->liquid_mask, sizeof P.liquid);
    P.liquid_is_range = c->liquid_is_range ? 1u : 0u;
    P.liquid_lo = c->liquid_lo;
    P.liquid_span = c->liquid_span;

    P.ndc_x = c->d_ndc;
    P.ndc_y = c->d_ndc + c->width;
    cam_sun_dir(c, P.cam_sun_dir);
    cam_origin_facts(c, P.cam_origin, &P.cam_origin_facts);
    for (int k = 0; k < 3; k++) {
        volatile float wm = (float)c->world.min[k];
        volatile float a = c->settings.sun_pos[k] - wm;
        P.sun_local[k] = a;
    }
    {
        volatile float wmax = 0.0f + (float)c->world.size;
        P.world_max = wmax;
    }

}

// Where one frame runs and what it writes: the caller's stream and the current output, or — frames in flight — one of
// the context's own (stream, output, launched-ray counts, path buffers, cursors) sets.
struct FrameSet {
    uint32_t slot = 0;   // 0: the context's own set, k: extra set k - 1
    hipStream_t st;
    vrt::Texel *out;
    uint32_t *blk;
    uint4 **path_buf;
    unsigned long long *counters;
};

// Two (or more) frames in flight: plain frames — one launch, or the path trace's chain of launches — alternate between
// the context's sets; anything else (stats, the two-launch variants, a caller's stream or bound buffer without
// VRT_RENDER_OWN_STREAMS) waits for them and runs alone on c->stream.
static int pick_frame_set(vrt_ctx *c, const vrt_render_opts &o, uint32_t variant, bool kstats, FrameSet &f) {
    const bool chain = o.mode == VRT_MODE_PRIMARY || (o.mode == VRT_MODE_PRIMARY_SHADOW && (variant == 0u || (variant == 2u && c->compact))) ||
                       o.mode == VRT_MODE_PATH;
    // VRT_RENDER_OWN_STREAMS: the caller set a stream and / or bound an output but lets this frame run on the context's own
    // streams (nothing on the caller's stream consumes it before a synchronise; frames in flight are bound to different
    // buffers) — the gather root's own tiles in bench.py
    const bool own_streams = (o.flags & VRT_RENDER_OWN_STREAMS) != 0u;
    const bool pipelined = c->in_flight > 1u && chain && !kstats && (own_streams || (c->stream == c->own_stream && c->d_out == c->own_out));
    const bool bound = c->d_out != c->own_out;
    f = FrameSet{0u, c->stream, c->d_out, c->d_blk_counts, &c->d_path, c->d_counters};
    if (!pipelined) {
        QUIESCE(c);
        return VRT_OK;
    }
    if (c->flip) {
        const uint32_t k = c->flip - 1u;
        f.slot = k + 1u;
        if (o.mode == VRT_MODE_PATH) {
            if (!c->extra_counters[k]) HIP_TRY(c, hipMalloc(&c->extra_counters[k], kCounterBytes));
            f.counters = c->extra_counters[k];
            f.path_buf = &c->extra_path[k];
        }
        if (!c->extra_stream[k]) HIP_TRY(c, hipStreamCreateWithFlags(&c->extra_stream[k], hipStreamNonBlocking));
        if (!bound && !c->extra_out[k]) {
            const size_t bytes = (size_t)(c->slots ? c->slots : 1) * sizeof(vrt::Texel);
            HIP_TRY(c, hipMalloc(&c->extra_out[k], bytes));
            if (ragged_output(c)) HIP_TRY(c, zero_now(c, c->extra_out[k], bytes));   // texels no workgroup covers stay zero (main.rs:452)
        }
        if (!c->extra_blk[k]) HIP_TRY(c, hipMalloc(&c->extra_blk[k], (size_t)(c->tiles_local ? c->tiles_local : 1) * sizeof(uint32_t)));
        f.st = c->extra_stream[k];
        f.blk = c->extra_blk[k];
        if (!bound) f.out = c->extra_out[k];  // a bound output is the caller's buffer for this very frame
        c->alt_pending = true;
        const int rc = frame_waits_for_uploads(c, f.st, k + 1u);
        if (rc) return rc;
    } else if (c->stream != c->own_stream) {
        f.st = c->own_stream;
        c->own_pending = true;
        const int rc = frame_waits_for_uploads(c, f.st, 0u);
        if (rc) return rc;
    }
    c->flip = (c->flip + 1u) % c->in_flight;
    return VRT_OK;
}

// The next free event quadruple of the pool (folding the pool into the accumulated times when it is full).
static int next_events(vrt_ctx *c, std::array<hipEvent_t, 4> **ev, uint8_t **kind) {
    if (c->ev_used == c->ev_pool.size()) {
        // (512 frames of events: creating one costs the host a few microseconds, so a context is at full speed once it has
        // rendered that many frames between two vrt_get_stats calls; folding costs one drain per 512 frames)
        if (c->ev_pool.size() >= 512) {
            const int rc = fold_events(c, nullptr);
            if (rc) return rc;
        } else {
            std::array<hipEvent_t, 4> t{nullptr, nullptr, nullptr, nullptr};
            for (auto &e : t) HIP_TRY(c, hipEventCreate(&e));
            c->ev_pool.push_back(t);
            c->ev_kind.push_back(kEvNone);
        }
    }
    *kind = &c->ev_kind[c->ev_used];
    **kind = kEvNone;
    *ev = &c->ev_pool[c->ev_used++];
    return VRT_OK;
}

// Wavefront path trace: per sample one launch per bounce over the compacted live-path buffer.
static int launch_path_frame(vrt_ctx *c, vrt::FrameParams &P, const FrameSet &f, const vrt_render_opts &o, bool kstats, bool literal,
                             std::array<hipEvent_t, 4> &ev, uint8_t &ev_kind) {
    const uint32_t spp = o.spp ? o.spp : 1u, bounces = c->settings.max_ray_bounces;
    // Several samples per launch chain (plain frames, spp > 1): every launch of the chain carries `samples` times the rays —
    // 2.7 rays per lane are not enough to cover a bounce launch's tail — and a frame of 16 spp is 4 x 4
    // launches instead of 16 x 4.  Each sample accumulates into its own plane; the chain's finishing pass adds the planes
    // to the frame in sample order, which is the order one sample per chain adds them in.
    const uint32_t samples = (spp > 1u && !kstats && !literal && P.grid && bounces > 0) ? (spp < c->path_samples ? spp : c->path_samples) : 1u;
    const bool planes = samples > 1u;
    const uint32_t seg_cap = c->hit_seg_cap * samples;
    const size_t cap = (size_t)vrt::kHitSegments * seg_cap;
    if (c->path_buf_records[f.slot] < cap) {   // (grows only; hipFree waits for whatever still uses the old one)
        (void)hipFree(*f.path_buf);
        *f.path_buf = nullptr; c->path_buf_records[f.slot] = 0;
        // two sets of three record planes (+ two of per-ray state for experiments/vrt_path_window.hip's launch)
        HIP_TRY(c, hipMalloc(f.path_buf, (2 * 3 + (vrt::g_exp.path_bounce_window ? 2 : 0)) * cap * sizeof(uint4)));
        c->path_buf_records[f.slot] = cap;
    }
    if (planes && c->path_acc_texels[f.slot] < (size_t)samples * c->slots) {
        (void)hipFree(c->path_acc[f.slot]);
        c->path_acc[f.slot] = nullptr; c->path_acc_texels[f.slot] = 0;
        HIP_TRY(c, hipMalloc(&c->path_acc[f.slot], (size_t)samples * c->slots * sizeof(vrt::Texel)));
        if (ragged_output(c)) HIP_TRY(c, zero_now(c, c->path_acc[f.slot], (size_t)samples * c->slots * sizeof(vrt::Texel)));
        c->path_acc_texels[f.slot] = (size_t)samples * c->slots;
    }
    vrt::Texel *const frame_out = P.out;
    P.hit_seg_cap = seg_cap;
    P.acc = planes ? c->path_acc[f.slot] : nullptr;
    P.acc_slots = c->slots;
    P.chain = 1u;
    if (planes) P.out = c->path_acc[f.slot];   // what the bounce launches accumulate into, through slots that carry the plane
    constexpr uint32_t kSegWords = vrt::kHitSegments * vrt::kSegStride;
    uint32_t *seg[3] = {P.seg_counts, P.seg_counts + kSegWords, P.seg_counts + 2 * kSegWords};
    uint4 *buf[2] = {*f.path_buf, *f.path_buf + 3 * cap};
    P.path_cap = (uint32_t)cap;
    P.in_cap = (uint32_t)cap;
    P.in_seg_cap = seg_cap;
    P.spp = spp;
    P.seed = o.seed;
    // Bounce launches over the derived tables with march cells: every later segment of the frame in ONE launch of the pool
    // kernel (vrt_path.hip); worlds without march cells, stats frames and the literal march: one lane = path launch per bounce.
    const bool pool = !kstats && !literal && P.grid && bounces > 1 && c->path_pool;
    const bool cells = pool && P.mblk && c->path_cells;
    // ... and among those the window launch (vrt_path_window.hip): the primary launch compacts each workgroup's survivors into
    // the workgroup's own region (256 records per sample of the chain), the bounce launch stages the march cells around a
    // group of four regions in LDS
    const bool window = cells && c->path_window && vrt::g_exp.path_bounce_window;   // (the window launch: the experiments build)
    const uint32_t n_regions = (c->tiles_local + 3u) / 4u;
    if (window) {
        if (c->path_grp_regions[f.slot] < n_regions) {
            (void)hipFree(c->path_grp_counts[f.slot]);
            c->path_grp_counts[f.slot] = nullptr; c->path_grp_regions[f.slot] = 0;
            HIP_TRY(c, hipMalloc(&c->path_grp_counts[f.slot], (size_t)n_regions * sizeof(uint32_t)));
            c->path_grp_regions[f.slot] = n_regions;
        }
    }
    P.grp_counts = window ? c->path_grp_counts[f.slot] : nullptr;
    P.grp_cap = 256u * samples;   // (n_regions * grp_cap <= cap: a segment holds what its workgroups can produce)
    P.blk_w = P.blk_h = 4u;
    if (window) {   // a bounce workgroup's regions are one block of tiles: 4 x 4 (4 regions), 8 x 4 (8), 8 x 8 (16)
        const uint32_t nw = vrt::g_exp.window_group_regions(c->path_window_shape);
        P.blk_w = nw == 4u ? 4u : 8u;
        P.blk_h = nw == 16u ? 8u : 4u;
    }
    const bool timed = ev[0] != nullptr;
    if (timed) HIP_TRY(c, hipEventRecord(ev[0], f.st));
    if (bounces == 0) HIP_TRY(c, hipMemsetAsync(f.out, 0, (size_t)c->slots * sizeof(vrt::Texel), f.st));
    bool first = true;
    uint32_t g = 0;   // launch number within the frame (all three cursor sets are zero when it starts: vrt_render cleared them)
    for (uint32_t smp = 0; smp < spp && bounces > 0; smp += samples) {
        P.sample = smp;
        P.chain = spp - smp < samples ? spp - smp : samples;
        for (uint32_t b = 0; b < bounces; b++, g++) {
            P.seg_counts = seg[g % 3u];
            P.seg_in = seg[(g + 2u) % 3u];
            P.seg_clear = seg[(g + 1u) % 3u];
            P.path_out = buf[g & 1u];
            P.path_in = buf[(g + 1u) & 1u];
            P.last_bounce = b + 1 == bounces;
            // one sample per pixel: the lane that ends a path has the pixel's final value (x / 1 = x) — no finishing pass
            if (b == 0) {
                if (window) vrt::g_exp.path_primary_grouped(P, f.st);
                else vrt::launch_path_primary(P, kstats, literal, f.st);
            } else if (cells) {
                // every bounce segment that is left in ONE launch: the waves carry their own survivors from one to the next
                // (one cursor set, one swap of the path buffers per LAUNCH: g counts launches)
                const uint32_t segments = bounces - b;
                P.last_bounce = 1u;
                if (window) vrt::g_exp.path_bounce_window(P, segments, n_regions, samples, c->path_window_shape, c->path_window_lift, f.st);
                else vrt::launch_path_bounce_cells(P, c->path_refill, segments, c->path_pool_batches ? c->path_pool_batches : (c->in_flight > 1u ? 5u : 4u), f.st);
                b += segments - 1u;
            } else {
                vrt::launch_path_bounce(P, kstats, literal, f.st);   // (no march cells, a stats frame, the literal march: lane = path)
            }
            HIP_TRY(c, hipGetLastError());
            if (first) { if (timed) HIP_TRY(c, hipEventRecord(ev[1], f.st)); first = false; }
        }
        if (planes) {
            vrt::launch_path_chain_finish(frame_out, c->path_acc[f.slot], c->slots, P.chain, smp == 0u, smp + P.chain >= spp, spp, f.st);
            HIP_TRY(c, hipGetLastError());
        }
    }
    P.out = frame_out;
    if (first && timed) HIP_TRY(c, hipEventRecord(ev[1], f.st));
    if (bounces > 0 && spp > 1u && !planes) {
        vrt::launch_path_finish(f.out, c->slots, spp, f.st);
        HIP_TRY(c, hipGetLastError());
    }
    if (timed) {
        HIP_TRY(c, hipEventRecord(ev[3], f.st));
        ev_kind = kEvRecorded;
    }
    c->last_spp = spp;
    return VRT_OK;
}

// Primary (+ shadow) rays: one launch (variant 0, 4; primary only) or two (variants 1-3).
static int launch_march_frame(vrt_ctx *c, const vrt::FrameParams &P, const FrameSet &f, bool shadow, uint32_t variant, bool kstats,
                              std::array<hipEvent_t, 4> &ev, uint8_t &ev_kind) {
    if (!c->tiles_local) return VRT_OK;  // an empty shard
    const uint32_t march = variant == 3u ? 0u : variant;  // variant 3 = the grid march in two launches
    // primary + shadow in one launch: the default march, and — on a context whose pixel slots are 8-byte records — the
    // octree walk it falls back to when the world is too large for the derived tables (the two-launch kernels store and
    // re-read 16-byte texels, which such a buffer has no room for)
    const bool fused = shadow && (variant == 0u || (variant == 2u && c->compact));
    c->n_counts = fused ? c->tiles_local : c->n_blocks;
    if (fused) vrt::launch_primary_shadow_fused(P, march, kstats, f.st, ev[0], ev[1]);
    else vrt::launch_primary(P, march, kstats, shadow, f.st, ev[0], ev[1]);
    HIP_TRY(c, hipGetLastError());
    if (ev[0]) ev_kind = kEvOneKernel;
    if (shadow && !fused) {
        vrt::launch_shadow(P, march, kstats, f.st, ev[2], ev[3]);
        HIP_TRY(c, hipGetLastError());
        if (ev[0]) ev_kind = kEvTwoKernels;
    }
    return VRT_OK;
}


extern "C" {

int vrt_create(const vrt_config *cfg, vrt_ctx **out) {
    if (!cfg || !out) return fail(nullptr, VRT_ERR_INVALID_ARG, "vrt_create: null argument");
    *out = nullptr;
    if (cfg->n_devices > 1u) return grp_create(cfg, out);
    // (2^17 nodes of headroom: a chunk rebuild stages up to a chunk's 32 767 + 24 nodes behind a root in one run of 16-byte loads,
    // whose byte offsets must not wrap — vrt_accel.hip)
    if (cfg->max_nodes < 2 || cfg->max_nodes > 0x7FFE0000u)
        return fail(nullptr, VRT_ERR_INVALID_ARG, "max_nodes must be in [2, 2^31 - 2^17] (the pool is addressed through a 32-bit byte offset)");
    if (cfg->width == 0 || cfg->height == 0)
        return fail(nullptr, VRT_ERR_INVALID_ARG, "output %ux%u: dimensions must be non-zero", cfg->width, cfg->height);
    if ((uint64_t)cfg->width * cfg->height > (1ull << 28))
        return fail(nullptr, VRT_ERR_INVALID_ARG, "output too large");
    const uint32_t sc = cfg->shard_count ? cfg->shard_count : 1u;
    if (cfg->shard_rank >= sc) return fail(nullptr, VRT_ERR_INVALID_ARG, "shard_rank %u >= shard_count %u", cfg->shard_rank, sc);
    if (cfg->shard_root_weight > 4096u) return fail(nullptr, VRT_ERR_INVALID_ARG, "shard_root_weight %u out of range", cfg->shard_root_weight);
    if ((cfg->flags & VRT_FLAG_ROW_MAJOR) && (cfg->flags & VRT_FLAG_TILE_MAJOR))
        return fail(nullptr, VRT_ERR_INVALID_ARG, "VRT_FLAG_ROW_MAJOR and VRT_FLAG_TILE_MAJOR exclude each other");
    if ((cfg->flags & VRT_FLAG_COMPACT) && ((cfg->flags & VRT_FLAG_ROW_MAJOR) || (sc == 1u && !(cfg->flags & VRT_FLAG_TILE_MAJOR))))
        return fail(nullptr, VRT_ERR_INVALID_ARG, "VRT_FLAG_COMPACT is for tile-major shard buffers");

    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
        return fail(nullptr, VRT_ERR_DEVICE, "no HIP device available (%s)", hipGetErrorString(e));
    int dev = cfg->device;
    if (dev < 0) {
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    }
    if (dev >= ndev) return fail(nullptr, VRT_ERR_INVALID_ARG, "device %d out of range (%d devices)", dev, ndev);

    vrt_ctx *c = new (std::nothrow) vrt_ctx();
    if (!c) return fail(nullptr, VRT_ERR_OOM, "host allocation failed");
    c->device = dev;
    c->shard_rank = cfg->shard_rank;
    c->shard_count = sc;
    c->shard_w0 = cfg->shard_root_weight ? cfg->shard_root_weight : 1u;
    c->tile_major = (sc > 1u && !(cfg->flags & VRT_FLAG_ROW_MAJOR)) || (cfg->flags & VRT_FLAG_TILE_MAJOR);
    c->compact = (cfg->flags & VRT_FLAG_COMPACT) != 0;
    c->width = cfg->width;
    c->height = cfg->height;
    c->max_nodes = cfg->max_nodes & ~1u;  // NodeBuffer::new forces an even size (shader.rs:10-12)
    c->accel_max_s = kAccelMaxS;
    if (const char *e = getenv("VRT_ACCEL_MAX_S")) {
        const long v = strtol(e, nullptr, 10);
        if (v >= 0 && v < (long)kAccelMaxS) c->accel_max_s = (uint32_t)v;
    }
    c->march_direct_max_s = kMarchDirectMaxS;
    if (const char *e = getenv("VRT_MARCH_DIRECT_MAX_S")) {
        const long v = strtol(e, nullptr, 10);
        if (v >= 0 && v <= (long)kMarchDirectMaxS) c->march_direct_max_s = (uint32_t)v;
    }
    if (const char *e = getenv("VRT_TILE_ORDER")) c->tile_lpt = e[0] != '0';
    // (0: screen order while the view moves — profiles/r05_tile_order_moving.txt)
    if (const char *e = getenv("VRT_TILE_ORDER_MOVING")) c->tile_lpt_moving = e[0] != '0' ? 1u : 0u;
    c->mov_any_size = getenv("VRT_TILE_ORDER_MOVING") != nullptr;   // (asked for by name: also for frames larger than kMovingTilesMax)
    if (const char *e = getenv("VRT_TILE_ORDER_RADIUS")) { const int v = atoi(e); if (v >= 1 && v <= 6) c->mov_radius = (uint32_t)v; }
    if (const char *e = getenv("VRT_PATH_POOL")) c->path_pool = e[0] != '0';
    if (const char *e = getenv("VRT_PATH_CELLS")) c->path_cells = e[0] != '0';
    // (the experiments build, tools/ab/libvrt_exp.so: without its hooks — vrt_exp.h — the next three select nothing)
    if (const char *e = getenv("VRT_PATH_WINDOW")) c->path_window = e[0] == '1';
    if (const char *e = getenv("VRT_PATH_WINDOW_SHAPE")) { const int v = atoi(e); if (v >= 0 && v <= 4) c->path_window_shape = (uint32_t)v; }
    if (const char *e = getenv("VRT_PATH_WINDOW_LIFT")) { const int v = atoi(e); if (v >= -64 && v <= 64) c->path_window_lift = v; }
    if (const char *e = getenv("VRT_PATH_POOL_REFILL")) c->path_refill = (uint32_t)atoi(e);
    if (const char *e = getenv("VRT_PATH_POOL_K")) { const int v = atoi(e); if (v == 4 || v == 5) c->path_pool_batches = (uint32_t)v; }
    if (const char *e = getenv("VRT_PATH_SAMPLES_PER_CHAIN")) { const int v = atoi(e); if (v >= 1 && v <= 16) c->path_samples = (uint32_t)v; }
    if (const char *e = getenv("VRT_TIMING_EVERY")) { const long v = strtol(e, nullptr, 10); if (v >= 1 && v <= 1000000) c->timing_every = (uint32_t)v; }
    memset(c->h_mats, 0, sizeof c->h_mats);
    memset(&c->cam, 0, sizeof c->cam);
    memset(&c->settings, 0, sizeof c->settings);
    memset(&c->world, 0, sizeof c->world);
    memset(&c->stats, 0, sizeof c->stats);

    auto body = [&]() -> int {
        HIP_TRY(c, hipSetDevice(dev));
        HIP_TRY(c, hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
        c->stream = c->own_stream;
        HIP_TRY(c, hipMalloc(&c->d_nodes, (size_t)c->max_nodes * sizeof(uint16_t)));
        // fresh buffer = zeros = every node an air leaf (client/src/world.rs:273-274)
        HIP_TRY(c, hipMemsetAsync(c->d_nodes, 0, (size_t)c->max_nodes * sizeof(uint16_t), c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));   // (uploads run on their own stream)
        // the upload path's fixtures now, not at the first edit (a mapped pinned allocation is tens of milliseconds)
        HIP_TRY(c, hipStreamCreateWithFlags(&c->up_stream, hipStreamNonBlocking));
        HIP_TRY(c, hipEventCreateWithFlags(&c->ev_pool_upload, hipEventDisableTiming));
        HIP_TRY(c, hipHostMalloc((void **)&c->h_ring, vrt_ctx::kRingSegBytes * vrt_ctx::kRingSegs, hipHostMallocMapped));
        HIP_TRY(c, hipHostGetDevicePointer((void **)&c->d_ring, c->h_ring, 0));
        HIP_TRY(c, hipMalloc(&c->d_mats, sizeof c->h_mats));
        HIP_TRY(c, hipMemsetAsync(c->d_mats, 0, sizeof c->h_mats, c->stream));
        HIP_TRY(c, hipMalloc(&c->d_counters, kCounterBytes));
        HIP_TRY(c, hipMemsetAsync(c->d_counters, 0, kCounterBytes, c->stream));
        int r = alloc_roots(c, cfg->world_size_chunks);
        if (r) return r;
        r = alloc_output(c);
        if (r) return r;
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        return VRT_OK;
    };
    const int rc = body();
    if (rc != VRT_OK) {
        g_create_err = c->err;
        vrt_destroy(c);
        return rc;
    }
    *out = c;
    return VRT_OK;
}

void vrt_destroy(vrt_ctx *c) {
    if (!c) return;
    if (c->grp) { grp_destroy(c); return; }
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
    if (c->up_stream) (void)hipStreamSynchronize(c->up_stream);
    for (hipStream_t st : c->extra_stream)
        if (st) (void)hipStreamSynchronize(st);
    for (auto p : c->extra_out) (void)hipFree(p);
    for (auto p : c->extra_blk) (void)hipFree(p);
    for (auto p : c->extra_path) (void)hipFree(p);
    for (auto p : c->extra_counters) (void)hipFree(p);
    for (auto p : c->path_acc) (void)hipFree(p);
    for (auto p : c->path_grp_counts) (void)hipFree(p);
    (void)hipFree(c->d_tile_cost); (void)hipFree(c->d_tile_order); (void)hipFree(c->d_tile_scratch);
    (void)hipFree(c->d_nodes); (void)hipFree(c->d_roots); (void)hipFree(c->d_mats); (void)hipFree(c->own_out);
    (void)hipFree(c->d_hits); (void)hipFree(c->d_counters); (void)hipFree(c->d_steps); (void)hipFree(c->d_rgba8); (void)hipFree(c->d_path);
    (void)hipFree(c->d_blk_counts); (void)hipFree(c->d_clock);
    for (auto &T : c->tabs) {
        (void)free_tables(c, T);
        if (T.ev_updated) (void)hipEventDestroy(T.ev_updated);
    }
    (void)hipFree(c->d_brick_total); (void)hipFree(c->d_chunk_needs);
    if (c->up_stream) { (void)hipStreamSynchronize(c->up_stream); (void)hipStreamDestroy(c->up_stream); }
    if (c->ev_pool_upload) (void)hipEventDestroy(c->ev_pool_upload);
    if (c->ev_walkers) (void)hipEventDestroy(c->ev_walkers);
    if (c->h_ring) (void)hipHostFree(c->h_ring);
    for (auto &evs : c->ring_ev)
        for (auto ev : evs)
            if (ev) (void)hipEventDestroy(ev);
    if (c->ev_frames) (void)hipEventDestroy(c->ev_frames);
    if (c->ev_upload) (void)hipEventDestroy(c->ev_upload); (void)hipFree(c->d_ndc); for (uint8_t *p : c->d_screen) (void)hipFree(p);
    for (auto &t : c->ev_pool)
        for (auto &ev : t)
            if (ev) (void)hipEventDestroy(ev);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    for (hipStream_t st : c->extra_stream)
        if (st) (void)hipStreamDestroy(st);
    delete c;
}

const char *vrt_last_error(const vrt_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int vrt_set_camera(vrt_ctx *c, const vrt_cam_data *cam) {
    GRP_EACH(c, vrt_set_camera(d, cam));
    if (!c || !cam) return fail(c, VRT_ERR_INVALID_ARG, "vrt_set_camera: null argument");
    if (memcmp(&c->cam, cam, sizeof *cam) != 0) { c->view_gen++; c->cam_gen++; }
    c->cam = *cam;
    return VRT_OK;
}

int vrt_set_settings(vrt_ctx *c, const vrt_settings *s) {
    GRP_EACH(c, vrt_set_settings(d, s));
    if (!c || !s) return fail(c, VRT_ERR_INVALID_ARG, "vrt_set_settings: null argument");
    if (memcmp(&c->settings, s, sizeof *s) != 0) c->view_gen++;
    c->settings = *s;
    return VRT_OK;
}

int vrt_set_world(vrt_ctx *c, const vrt_world_data *w) {
    GRP_EACH(c, vrt_set_world(d, w));
    if (!c || !w) return fail(c, VRT_ERR_INVALID_ARG, "vrt_set_world: null argument");
    if (memcmp(&c->world, w, sizeof *w) != 0) c->view_gen++;
    c->world = *w;
    return VRT_OK;
}

int vrt_resize_output(vrt_ctx *c, uint32_t width, uint32_t height) {
    if (c && c->grp) return grp_resize_output(c, width, height);
    if (!c) return VRT_ERR_INVALID_ARG;
    if (width == 0 || height == 0 || (uint64_t)width * height > (1ull << 28))
        return fail(c, VRT_ERR_INVALID_ARG, "output %ux%u: dimensions must be non-zero, at most 2^28 pixels", width, height);
    HIP_TRY(c, hipSetDevice(c->device));
    QUIESCE(c);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->width = width;
    c->height = height;
    return alloc_output(c);
}

int vrt_render(vrt_ctx *c, const vrt_render_opts *opts) {
    if (c && c->grp) return grp_render(c, opts);
    if (!c) return VRT_ERR_INVALID_ARG;
    VRT_PROF(3, "vrt_render (one context)");
    struct IssueClock {   // vrt_get_issue_profile: the calling thread's time in here, whichever way the call leaves
        vrt_ctx *c;
        std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
        ~IssueClock() {
            c->prof_render_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            c->prof_frames += 1u;
        }
    } issue_clock{c};
    vrt_render_opts o;
    memset(&o, 0, sizeof o);
    if (opts) o = *opts;
    if (o.mode > VRT_MODE_PATH) return fail(c, VRT_ERR_INVALID_ARG, "vrt_render: mode %u not supported", o.mode);
    if (o.mode == VRT_MODE_PATH && o.variant != 0) return fail(c, VRT_ERR_INVALID_ARG, "vrt_render: the path trace has one kernel variant");
    if (o.stats > 2u) return fail(c, VRT_ERR_INVALID_ARG, "vrt_render: stats %u (0, 1 = count steps, 2 = clock probe)", o.stats);
    if (o.stats == 2u && (o.mode != VRT_MODE_PRIMARY_SHADOW || (o.variant != 0u)))
        return fail(c, VRT_ERR_INVALID_ARG, "vrt_render: the clock probe (stats = 2) is a build of the default primary + shadow kernel");
    if (!vrt::variant_supported(o.variant)) return fail(c, VRT_ERR_INVALID_ARG, "vrt_render: unknown kernel variant %u", o.variant);
    int rc;
    {
        VRT_PROF(7, "  validate + hipSetDevice");
        rc = validate_frame(c);
        if (rc) return rc;
        HIP_TRY(c, hipSetDevice(c->device));
    }

    if (c->compact && (o.mode == VRT_MODE_PATH || (o.variant != 0u && o.variant != 2u) || c->settings.show_step_count == 1u))
        return fail(c, VRT_ERR_STATE, "vrt_render: a VRT_FLAG_COMPACT context renders primary(+shadow) frames with the default march "
                    "only (no path trace, step-count view, literal or two-launch variants)");
    if (o.stats == 1u && !c->d_steps) {
        HIP_TRY(c, hipMalloc(&c->d_steps, (size_t)(c->slots ? c->slots : 1) * sizeof(uint32_t)));
        if (ragged_output(c)) HIP_TRY(c, zero_now(c, c->d_steps, (size_t)(c->slots ? c->slots : 1) * sizeof(uint32_t)));
    }
    if (o.stats == 2u && !c->d_clock) {
        HIP_TRY(c, hipMalloc(&c->d_clock, 2 * sizeof(unsigned long long)));
        HIP_TRY(c, hipMemsetAsync(c->d_clock, 0, 2 * sizeof(unsigned long long), c->stream));
        const int rc2 = publish_upload(c);
        if (rc2) return rc2;
    }
    {
        VRT_PROF(8, "  ensure_ndc");
        rc = ensure_ndc(c);
        if (rc) return rc;
    }
    uint32_t variant = o.variant;
    // The fast marches never ask whether *air* is liquid (ray_tracer.wgsl:226 asks for every voxel, voxel 0 included): a
    // material table that flags voxel 0 as liquid — nothing the reference's data packs do — is traced by the literal march.
    const bool air_liquid = c->h_mats[0].is_liquid == 1u;
    if (air_liquid) {
        if (c->compact)
            return fail(c, VRT_ERR_STATE, "vrt_render: materials[0].is_liquid == 1 (air flagged liquid) is traced by the literal march only");
        variant = 1u;
    }
    if (variant == 0u || variant == 3u || (o.mode == VRT_MODE_PATH && !air_liquid)) {
        VRT_PROF(9, "  ensure_accel_world");
        rc = ensure_accel_world(c);
        if (rc) return rc;
        if (!c->accel_ok && (variant == 0u || variant == 3u)) variant = 2u;  // world too large for the tables: walk the octree
    }
    // per-lane iteration counts exist in the STATS kernels only; the step-count debug view (F2 in the reference,
    // main.rs:368-370) needs them, so it runs those kernels too
    const bool kstats = o.stats == 1u || c->settings.show_step_count == 1u;

    FrameSet f;
    {
        VRT_PROF(10, "  pick_frame_set (+ its upload waits)");
        rc = pick_frame_set(c, o, variant, kstats, f);
        if (rc) return rc;
    }
    c->last_out = f.out;
    c->last_blk = f.blk;
    c->last_stream = f.st;
    c->last_slot = f.slot;
    if (c->wait_before_frame) {
        HIP_TRY(c, hipStreamWaitEvent(f.st, c->wait_before_frame, 0));
        c->wait_before_frame = nullptr;
    }

    // this frame's table set, brought up to date on its own stream (which waits for the uploads so far first)
    const bool wants_tables = variant == 0u || variant == 3u || (o.mode == VRT_MODE_PATH && !air_liquid);
    constexpr uint32_t kQuietFrames = 64;
    if (c->tables_split && ++c->quiet_frames > kQuietFrames && c->tabs[0].dirty_chunks.empty()) {
        c->tables_split = false;   // no edit for a while: everybody reads tabs[0] again; the other sets go stale
        for (uint32_t k = 1; k < vrt_ctx::kMaxInFlight; k++) c->tabs[k].live = false;
    }
    const uint32_t tab = c->tables_split ? f.slot : 0u;
    const vrt_ctx::Tables &T = c->tabs[tab];
    const bool edit_in_front = !T.dirty_chunks.empty();   // this frame brings an edited chunk's tables up to date first
    c->last_tab = tab;
    // a frame of another frame set that reads the shared set: an edit's update of tabs[0] must wait for it (update_tables)
    const bool shares = wants_tables && c->accel_ok && tab == 0u && f.slot != 0u;
    if (wants_tables && c->accel_ok) {
        VRT_PROF(11, "  frame_waits_for_uploads + update_tables");
        rc = frame_waits_for_uploads(c, f.st, f.slot);
        if (rc) return rc;
        rc = update_tables(c, tab, f.st);
        if (rc) return rc;
        if (tab == 0u && f.slot != 0u) {   // the shared set from another frame set's stream: behind its last update
            if (c->tabs[0].update_pending && f.st != c->stream) HIP_TRY(c, hipStreamWaitEvent(f.st, c->tabs[0].ev_updated, 0));
            c->shared_readers_in_flight = true;
        }
    } else {
        rc = frame_waits_for_uploads(c, f.st, f.slot);   // (a frame on c->stream too: the node pool's uploads have their own stream)
        if (rc) return rc;
    }

    vrt::FrameParams P;
    memset(&P, 0, sizeof P);
    P.nodes = c->d_nodes;
    P.roots = c->d_roots;
    P.mats = c->d_mats;
    if (wants_tables && c->accel_ok && c->accel_S == c->world.size_in_chunks && !c->accel_dirty && T.live && T.dirty_chunks.empty() && !air_liquid) {
        P.grid = T.d_grid;
        P.bricks = T.d_bricks;
        P.grid_dim = c->accel_S * 8u;
        const size_t G = (size_t)c->accel_S * 8u;
        P.grid_bytes = (uint32_t)(G * (G + 1u) * (G + 1u) * sizeof(uint32_t));  // [8S][8S+1][8S+1]: the zero border
        P.brick_bytes = (uint32_t)((size_t)T.brick_cap * 64u * sizeof(uint16_t));
        if (T.d_mblk) {
            P.cdir = T.d_cdir;
            P.cdir_bytes = (uint32_t)(chunk_dir_entries(c->accel_S) * sizeof(uint32_t));
            P.mblk = T.d_mblk;
            // (a direct world: exactly its lines — a position beyond the last slab must be out of range, it reads as zeros)
            P.mblk_bytes = (uint32_t)(c->march_direct ? direct_cell_entries(c->accel_S) * sizeof(uint4) : (size_t)T.mblk_cap * 512u * sizeof(uint4));
            P.march_direct = c->march_direct ? 1u : 0u;
        }
    }
    // a march that walks the octree reads the node pool and chunk_roots: uploads then wait for the frames in flight
    if (!P.grid || variant == 1u || variant == 2u) c->walkers_in_flight = true;
    P.out = f.out;
    P.hits = c->d_hits;
    P.blk_counts = f.blk;
    P.counters = f.counters;
    P.seg_counts = reinterpret_cast<uint32_t *>(f.counters + vrt::kCtrCount);
    P.hit_seg_cap = c->hit_seg_cap;
    P.steps = o.stats == 1u ? c->d_steps : nullptr;
    P.clock = o.stats == 2u ? c->d_clock : nullptr;
    fill_uniforms(c, P);
    // vrt_set_presentation: a frame whose kernel finishes its pixels through store_pixel (every primary-only frame; primary + shadow
    // in one launch) and whose window samples texel for texel also stores the window's image, into its frame set's screen buffer
    const bool one_launch_shadow = o.mode == VRT_MODE_PRIMARY_SHADOW && (variant == 0u || (variant == 2u && c->compact));
    const bool fuse_present = (o.mode == VRT_MODE_PRIMARY || one_launch_shadow) && c->tiles_local && presentation_fusable(c);
    if (fuse_present) {
        rc = screen_buffer_for_frame(c, f.slot, f.st, c->width, c->height);
        if (rc) return rc;
        P.screen = reinterpret_cast<uint32_t *>(c->d_screen[f.slot]);
        P.screen_only = (c->pres_flags & VRT_PRESENT_SKIP_TEXELS) ? 1u : 0u;
        memcpy(P.present_box, c->pres_box, sizeof P.present_box);
        P.crosshair = c->pres_ch;
    }
    c->last_fused = fuse_present;
    c->last_has_texels = !(fuse_present && P.screen_only);

    std::array<hipEvent_t, 4> *ev = nullptr;
    uint8_t *ev_kind = nullptr;
    static std::array<hipEvent_t, 4> no_events{nullptr, nullptr, nullptr, nullptr};
    static uint8_t no_kind = 0;
    if (c->timing_every > 1u && (c->frame_no++ % c->timing_every) != 0u && !kstats && !(o.flags & VRT_RENDER_TIMED)) {
        ev = &no_events;   // an untimed frame: the launches carry no events (vrt_stats' kernel times average the timed ones)
        ev_kind = &no_kind;
    } else {
        VRT_PROF(12, "  next_events");
        rc = next_events(c, &ev, &ev_kind);
        if (rc) return rc;
    }
    // The frame is about to be enqueued on its stream: say so again.  pick_frame_set announced it, but what ran since may
    // have waited for the frames in flight and cleared the announcement with them — next_events folds the event pool every
    // 512 timed frames and drains for it.  Without this the frame was in flight unannounced: vrt_read_output right behind it
    // copied the slot's previous frame (tools/soak_edits.py seed 7, check 1161: once in 180 000 frames), and an upload
    // would not have waited for it.
    if (f.st != c->stream) {
        if (f.st == c->own_stream) c->own_pending = true;
        else c->alt_pending = true;
    }
    // (the same drain clears shared_readers_in_flight: without this an edit right behind the fold frame let the next slot-0
    // frame rebuild chunks of tabs[0] in place while this frame still read them)
    if (shares) c->shared_readers_in_flight = true;
    // the order this frame's tiles are launched in, and whether it notes its trips for an order to come (vrt_order.hip)
    TileOrderPlan order_plan;
    rc = tile_order_before_frame(c, P, f.st, o, variant, kstats, edit_in_front, order_plan);
    if (rc) return rc;
    // the counters feed stats frames and the path trace's segment cursors; a plain primary(+shadow) frame reads none
    if (kstats || o.mode == VRT_MODE_PATH) HIP_TRY(c, hipMemsetAsync(f.counters, 0, kCounterBytes, f.st));
    {
        VRT_PROF(13, "  the launch(es)");
        if (o.mode == VRT_MODE_PATH) rc = launch_path_frame(c, P, f, o, kstats, air_liquid, *ev, *ev_kind);
        else rc = launch_march_frame(c, P, f, o.mode == VRT_MODE_PRIMARY_SHADOW, variant, kstats, *ev, *ev_kind);
        if (rc) return rc;
    }
    rc = tile_order_after_frame(c, P, f.st, order_plan);   // (the sort behind the frame that noted its trips)
    if (rc) return rc;
    c->rendered = true;
    c->flushed_at_call = false;   // (the next frame's first staged range may go out at its call again: vrt_uploads.hip)
    c->last_stats = o.stats == 1u;
    c->last_mode = o.mode;
    c->timing_pending = true;
    return VRT_OK;
}

int vrt_synchronize(vrt_ctx *c) {
    if (c && c->grp) return grp_synchronize(c);
    if (!c) return VRT_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    QUIESCE(c);
    {
        const int rc = flush_staged(c);
        if (rc) return rc;
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->up_stream) HIP_TRY(c, hipStreamSynchronize(c->up_stream));
    c->walkers_in_flight = false;   // nothing is in flight any more
    for (auto &T : c->tabs) T.update_pending = false;
    return VRT_OK;
}

int vrt_read_output(vrt_ctx *c, float *rgb, uint32_t *ids, uint8_t *rgba8) {
    if (c && c->grp) { const int rc_ = grp_synchronize(c); if (rc_) return rc_; }
    GRP_ROOT(c, vrt_read_output(d, rgb, ids, rgba8));
    if (!c) return VRT_ERR_INVALID_ARG;
    if (!c->rendered) return fail(c, VRT_ERR_STATE, "vrt_read_output: nothing rendered yet");
    if (c->compact) return fail(c, VRT_ERR_STATE, "vrt_read_output: a VRT_FLAG_COMPACT context holds 8-byte records, not texels (vrt_assemble_compact shades them)");
    if (!c->last_has_texels) return fail(c, VRT_ERR_STATE, "vrt_read_output: the last frame stored its window pixels only (vrt_set_presentation with VRT_PRESENT_SKIP_TEXELS)");
    HIP_TRY(c, hipSetDevice(c->device));
    QUIESCE(c);
    const size_t npix = (size_t)c->width * c->height;
    if (rgba8) {
        if (c->tile_major) return fail(c, VRT_ERR_STATE, "vrt_read_output: rgba8 readback needs the row-major (unsharded) layout");
        if (!c->d_rgba8) HIP_TRY(c, hipMalloc(&c->d_rgba8, npix * 4));
        vrt::launch_quantize(c->last_out, c->d_rgba8, c->width, c->height, c->stream);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipMemcpyAsync(rgba8, c->d_rgba8, npix * 4, hipMemcpyDeviceToHost, c->stream));
    }
    std::vector<vrt::Texel> t;
    if (rgb || ids) {
        t.resize(c->slots);
        HIP_TRY(c, hipMemcpyAsync(t.data(), c->last_out, t.size() * sizeof(vrt::Texel), hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (!rgb && !ids) return VRT_OK;
    auto put = [&](size_t dst, const vrt::Texel &x) {
        if (rgb) { memcpy(rgb + dst * 3, &x, 12); }
        if (ids) ids[dst] = x.w;
    };
    if (!c->tile_major) {
        for (size_t i = 0; i < npix; i++) put(i, t[i]);
        return VRT_OK;
    }
    // sharded: de-interleave this context's tiles; foreign tiles read as zero
    if (rgb) memset(rgb, 0, npix * 3 * sizeof(float));
    if (ids) memset(ids, 0, npix * sizeof(uint32_t));
    for (uint32_t tl = 0; tl < c->tiles_local; tl++) {
        const uint32_t tile = vrt::shard_tile(tl, c->shard_first, c->shard_run, c->shard_period);
        const uint32_t tx = (tile % c->tiles_x) * 8u, ty = (tile / c->tiles_x) * 8u;
        for (uint32_t p = 0; p < 64; p++) put((size_t)(ty + (p >> 3)) * c->width + tx + (p & 7u), t[(size_t)tl * 64 + p]);
    }
    return VRT_OK;
}

int vrt_read_steps(vrt_ctx *c, uint32_t *steps) {
    GRP_REFUSE(c, "vrt_read_steps");
    if (!c || !steps) return fail(c, VRT_ERR_INVALID_ARG, "vrt_read_steps: null argument");
    if (!c->rendered || !c->last_stats || !c->d_steps)
        return fail(c, VRT_ERR_STATE, "vrt_read_steps: the last frame was not rendered with opts.stats = 1");
    if (c->tile_major) return fail(c, VRT_ERR_STATE, "vrt_read_steps: needs the row-major (unsharded) layout");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(steps, c->d_steps, (size_t)c->width * c->height * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return VRT_OK;
}

int vrt_get_stats(vrt_ctx *c, vrt_stats *out) {
    if (c && c->grp) return grp_get_stats(c, out);
    if (!c || !out) return fail(c, VRT_ERR_INVALID_ARG, "vrt_get_stats: null argument");
    if (!c->rendered) return fail(c, VRT_ERR_STATE, "vrt_get_stats: nothing rendered yet");
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->timing_pending) {
        float last[3] = {0, 0, 0};
        int rc = fold_events(c, last);
        if (rc) return rc;
        std::vector<unsigned long long> hbuf(kCounterBytes / sizeof(unsigned long long));
        HIP_TRY(c, hipMemcpyAsync(hbuf.data(), c->d_counters, kCounterBytes, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        unsigned long long *h = hbuf.data();
        const uint32_t *seg = reinterpret_cast<const uint32_t *>(h + vrt::kCtrCount);
        unsigned long long launched = 0;
        if (c->last_mode == VRT_MODE_PRIMARY_SHADOW && c->n_counts) {
            std::vector<uint32_t> bc(c->n_counts);
            HIP_TRY(c, hipMemcpyAsync(bc.data(), c->last_blk, bc.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            for (uint32_t v : bc) launched += v;
        }
        (void)seg;
        vrt_stats s;
        memset(&s, 0, sizeof s);
        s.primary_rays = (uint64_t)c->tiles_local * 64u * (c->last_mode == VRT_MODE_PATH ? c->last_spp : 1u);
        s.secondary_rays = c->last_mode == VRT_MODE_PRIMARY_SHADOW ? launched : 0;
        if (c->last_mode == VRT_MODE_PATH && c->last_stats) s.secondary_rays = h[vrt::kCtrSecondary];
        if (c->last_stats) {
            s.hits = h[vrt::kCtrHits];
            s.steps = h[vrt::kCtrSteps];
            s.node_visits = h[vrt::kCtrVisits];
            s.primary_steps = h[vrt::kCtrPrimarySteps];
            s.primary_node_visits = h[vrt::kCtrPrimaryVisits];
        }
        if (c->d_clock) {  // clock-probe frames since the last call
            unsigned long long clk[2] = {0, 0};
            HIP_TRY(c, hipMemcpyAsync(clk, c->d_clock, sizeof clk, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipMemsetAsync(c->d_clock, 0, sizeof clk, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            s.clock_shader_ticks = clk[0];
            s.clock_ref_ticks = clk[1];
        }
        s.ms_primary = last[0]; s.ms_secondary = last[1]; s.ms_total = last[2];
        s.frames = c->acc_frames;
        s.sum_ms_primary = c->acc_ms[0]; s.sum_ms_secondary = c->acc_ms[1]; s.sum_ms_total = c->acc_ms[2];
        c->acc_frames = 0;
        c->acc_ms[0] = c->acc_ms[1] = c->acc_ms[2] = 0;
        c->stats = s;
        c->timing_pending = false;
    }
    *out = c->stats;
    return VRT_OK;
}

int vrt_get_issue_profile(vrt_ctx *c, vrt_issue_profile *out) {
    if (!c || !out) return fail(c, VRT_ERR_INVALID_ARG, "vrt_get_issue_profile: null argument");
    if (c->grp) return grp_get_issue_profile(c, out);
    memset(out, 0, sizeof *out);
    out->devices = 1u;
    out->frames = c->prof_frames;
    if (c->prof_frames) out->render_us = c->prof_render_us / (double)c->prof_frames;
    c->prof_render_us = 0.0;
    c->prof_frames = 0u;
    return VRT_OK;
}

int vrt_set_frames_in_flight(vrt_ctx *c, uint32_t n) {
    if (c && c->grp) return grp_set_frames_in_flight(c, n);
    if (!c) return VRT_ERR_INVALID_ARG;
    if (n < 1u || n > vrt_ctx::kMaxInFlight) return fail(c, VRT_ERR_INVALID_ARG, "vrt_set_frames_in_flight: 1..%u", vrt_ctx::kMaxInFlight);
    HIP_TRY(c, hipSetDevice(c->device));
    QUIESCE(c);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->in_flight = n;
    c->flip = 0;
    // the table sets of frame slots that no longer render go stale: nobody brings them up to date, so their dirty lists
    // would only grow (and, full, force whole-world builds on a context whose active sets are fine)
    for (uint32_t k = n; k < vrt_ctx::kMaxInFlight; k++) {
        auto &T = c->tabs[k];
        for (uint32_t ch : T.dirty_chunks) T.chunk_is_dirty[ch] = 0;
        T.dirty_chunks.clear();
        T.live = false;
    }
    return VRT_OK;
}

int vrt_set_stream(vrt_ctx *c, void *hip_stream) {
    GRP_REFUSE(c, "vrt_set_stream");
    if (!c) return VRT_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    QUIESCE(c);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return VRT_OK;
}

int vrt_bind_output(vrt_ctx *c, void *texels) {
    GRP_REFUSE(c, "vrt_bind_output");
    if (!c) return VRT_ERR_INVALID_ARG;
    if (texels && ((uintptr_t)texels % 16u)) return fail(c, VRT_ERR_INVALID_ARG, "vrt_bind_output: texels must be 16-byte aligned");
    // stream-ordered: launches capture the pointer, so frames already enqueued keep writing where they were
    // told to and the next vrt_render uses the new buffer (lets a host ping-pong two gather messages)
    c->d_out = texels ? (vrt::Texel *)texels : c->own_out;
    c->last_out = c->d_out;
    c->rendered = false;
    return VRT_OK;
}

int vrt_device_output(vrt_ctx *c, void **texels, uint64_t *bytes) {
    GRP_ROOT(c, vrt_device_output(d, texels, bytes));
    if (!c) return VRT_ERR_INVALID_ARG;
    if (texels) *texels = c->d_out == c->own_out ? c->last_out : c->d_out;  // own buffers: the one holding the last frame
    if (bytes) *bytes = (uint64_t)c->slots * (c->compact ? 8u : sizeof(vrt::Texel));
    return VRT_OK;
}

int vrt_shard_info(vrt_ctx *c, uint32_t *tiles_local, uint32_t *tiles_padded, uint32_t *tiles_total) {
    GRP_ROOT(c, vrt_shard_info(d, tiles_local, tiles_padded, tiles_total));
    if (!c) return VRT_ERR_INVALID_ARG;
    if (tiles_local) *tiles_local = c->tiles_local;
    if (tiles_padded) *tiles_padded = c->tiles_padded;
    if (tiles_total) *tiles_total = c->tiles_total;
    return VRT_OK;
}

}  // extern "C"
