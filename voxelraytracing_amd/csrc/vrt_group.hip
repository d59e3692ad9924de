// vrt_group.hip — one context over several devices (vrt_config.n_devices > 1); GrpWorker / vrt_group: vrt_ctx.h.
#include "vrt_ctx.h"

static int grp_alloc_messages(vrt_ctx *c) {
    vrt_group *g = c->grp;
    vrt_ctx *root = g->dev[0];
    HIP_TRY(c, hipSetDevice(root->device));
    for (auto &p : g->recv) { (void)hipFree(p); p = nullptr; }
    g->rank_stride = (size_t)root->tiles_padded * 64u * (g->texels ? 16u : 8u);
    for (auto &p : g->recv) {
        HIP_TRY(c, hipMalloc(&p, g->rank_stride * g->dev.size()));
        HIP_TRY(c, hipMemset(p, 0, g->rank_stride * g->dev.size()));
    }
    for (size_t r = 1; r < g->dev.size(); r++) {
        if (!g->staged[r]) continue;
        HIP_TRY(c, hipSetDevice(g->dev[r]->device));
        for (auto &p : g->stage[r]) {
            (void)hipFree(p); p = nullptr;
            HIP_TRY(c, hipMalloc(&p, g->rank_stride));
            HIP_TRY(c, hipMemset(p, 0, g->rank_stride));
        }
    }
    HIP_TRY(c, hipSetDevice(root->device));
    g->slot = 0;
    g->consumed_used[0] = g->consumed_used[1] = false;
    return VRT_OK;
}

int grp_create(const vrt_config *cfg, vrt_ctx **out) {
    const uint32_t n = cfg->n_devices;
    if (n > VRT_MAX_DEVICES) return fail(nullptr, VRT_ERR_INVALID_ARG, "n_devices %u > VRT_MAX_DEVICES", n);
    if (cfg->shard_rank != 0u || cfg->shard_count > 1u)
        return fail(nullptr, VRT_ERR_INVALID_ARG, "a multi-device context shards by itself: shard_rank / shard_count must be 0");
    if (cfg->flags & ~(VRT_FLAG_TEXEL_MESSAGES | VRT_FLAG_STAGED_MESSAGES | VRT_FLAG_POISON_MESSAGES))
        return fail(nullptr, VRT_ERR_INVALID_ARG, "a multi-device context takes VRT_FLAG_TEXEL_MESSAGES, _STAGED_MESSAGES and _POISON_MESSAGES only");
    vrt_ctx *c = new (std::nothrow) vrt_ctx();
    vrt_group *g = new (std::nothrow) vrt_group();
    if (!c || !g) { delete c; delete g; return fail(nullptr, VRT_ERR_OOM, "host allocation failed"); }
    c->grp = g;
    g->texels = (cfg->flags & VRT_FLAG_TEXEL_MESSAGES) != 0u;
    g->poison = (cfg->flags & VRT_FLAG_POISON_MESSAGES) != 0u;
    g->staged.assign(n, (cfg->flags & VRT_FLAG_STAGED_MESSAGES) ? 1 : 0);
    g->stage.assign(n, {nullptr, nullptr});
    // the root's own tiles never cross a link, so it takes more of the frame; measured defaults
    const uint32_t w0 = cfg->shard_root_weight ? cfg->shard_root_weight : (n == 2u ? 4u : n <= 4u ? 3u : 2u);
    DeviceRestore restore;
    auto body = [&]() -> int {
        for (uint32_t r = 0; r < n; r++) {
            vrt_config sub = *cfg;
            sub.n_devices = 0;
            sub.device = cfg->device_ids[r];
            sub.shard_rank = r;
            sub.shard_count = n;
            sub.shard_root_weight = w0;
            sub.flags = r == 0u ? VRT_FLAG_ROW_MAJOR : (g->texels ? 0u : VRT_FLAG_COMPACT);   // (the group's own flags stay here)
            vrt_ctx *d = nullptr;
            const int rc = vrt_create(&sub, &d);
            if (rc) { c->err = g_create_err; return rc; }
            g->dev.push_back(d);
        }
        g->dev[0]->whole_frame_owner = true;
        g->done.resize(n);
        for (uint32_t r = 1; r < n; r++) {
            HIP_TRY(c, hipSetDevice(g->dev[r]->device));
            if (g->dev[r]->device != g->dev[0]->device && !g->staged[r]) {
                // no peer access: the message is rendered at home and copied over (a copy between two devices needs none)
                const hipError_t e = hipDeviceEnablePeerAccess(g->dev[0]->device, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) g->staged[r] = 1;
                (void)hipGetLastError();
            }
            for (auto &ev : g->done[r]) HIP_TRY(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        }
        HIP_TRY(c, hipSetDevice(g->dev[0]->device));
        for (auto &ev : g->consumed) HIP_TRY(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        // issuing threads pay off when the devices are different ones: launches to one device serialise inside the
        // runtime whichever thread makes them (measured with device_ids = {0, ...}: 168 us of host time per frame for 8
        // contexts with workers, 145 without).  VRT_GROUP_THREADS=1 / 0 forces them on / off.
        bool distinct = true;
        for (uint32_t a = 0; a < n; a++)
            for (uint32_t b = a + 1; b < n; b++)
                if (cfg->device_ids[a] == cfg->device_ids[b]) distinct = false;
        const char *e = getenv("VRT_GROUP_THREADS");
        if (e ? e[0] == '1' : distinct)
            for (uint32_t r = 1; r < n; r++) {
                g->workers.emplace_back(new GrpWorker());
                GrpWorker *w = g->workers.back().get();
                w->th = std::thread([w] { w->run(); });
            }
        return grp_alloc_messages(c);
    };
    const int rc = body();
    if (rc) {
        g_create_err = c->err;
        grp_destroy(c);
        return rc;
    }
    *out = c;
    return VRT_OK;
}

void grp_destroy(vrt_ctx *c) {
    vrt_group *g = c->grp;
    DeviceRestore restore;
    for (auto &w : g->workers) w->stop();
    for (vrt_ctx *d : g->dev) {
        (void)hipSetDevice(d->device);
        (void)vrt_synchronize(d);
    }
    for (size_t r = 1; r < g->dev.size() && r < g->stage.size(); r++) {
        (void)hipSetDevice(g->dev[r]->device);
        for (auto p : g->stage[r]) (void)hipFree(p);
    }
    if (!g->dev.empty()) (void)hipSetDevice(g->dev[0]->device);
    for (auto p : g->recv) (void)hipFree(p);
    for (auto ev : g->consumed)
        if (ev) (void)hipEventDestroy(ev);
    for (auto &evs : g->done)
        for (auto ev : evs)
            if (ev) (void)hipEventDestroy(ev);
    for (vrt_ctx *d : g->dev) vrt_destroy(d);
    delete g;
    delete c;
}

int grp_synchronize(vrt_ctx *c) { return grp_each(c, [](vrt_ctx *d) { return vrt_synchronize(d); }); }

int grp_set_frames_in_flight(vrt_ctx *c, uint32_t n) {
    if (n < 1u || n > vrt_group::kSlots) return fail(c, VRT_ERR_INVALID_ARG, "vrt_set_frames_in_flight: 1..%u on a multi-device context", vrt_group::kSlots);
    int rc = grp_synchronize(c);
    if (rc) return rc;
    c->grp->in_flight = n;
    return grp_each(c, [&](vrt_ctx *d) { return vrt_set_frames_in_flight(d, n); });
}

int grp_resize_output(vrt_ctx *c, uint32_t w, uint32_t h) {
    DeviceRestore restore;
    int rc = grp_synchronize(c);
    if (rc) return rc;
    rc = grp_each(c, [&](vrt_ctx *d) { return vrt_resize_output(d, w, h); });
    if (rc) return rc;
    return grp_alloc_messages(c);
}

static int grp_render_frame(vrt_ctx *c, const vrt_render_opts *opts, bool &issued);

// A frame that fails half-way has been enqueued on some devices and not on others, its message slot is taken and nobody will
// record that it was consumed: the group is drained and starts over at slot 0 with no slot owing a wait, so that the frames
// after the error find a consistent group (the error itself goes to the caller as it is).
int grp_render(vrt_ctx *c, const vrt_render_opts *opts) {
    bool issued = false;
    const int rc = grp_render_frame(c, opts, issued);
    if (rc && issued) {
        vrt_group *g = c->grp;
        const std::string keep = c->err;
        (void)grp_synchronize(c);
        c->err = keep;
        g->slot = 0;
        for (bool &u : g->consumed_used) u = false;
    }
    return rc;
}

static inline double us_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
}

static int grp_render_frame(vrt_ctx *c, const vrt_render_opts *opts, bool &issued) {
    VRT_PROF(0, "grp_render (a frame of the group)");
    const auto t_frame = std::chrono::steady_clock::now();
    vrt_group *g = c->grp;
    vrt_ctx *root = g->dev[0];
    vrt_render_opts o;
    memset(&o, 0, sizeof o);
    if (opts) o = *opts;
    if (!g->texels && (o.mode == VRT_MODE_PATH || (o.variant != 0u && o.variant != 2u)))
        return fail(c, VRT_ERR_STATE, "vrt_render: this multi-device context exchanges 8-byte records (primary(+shadow) frames of the default "
                    "march); create it with VRT_FLAG_TEXEL_MESSAGES for the path trace and the other marches");
    if (o.stats == 2u) return fail(c, VRT_ERR_INVALID_ARG, "vrt_render: no clock probe on a multi-device context");
    const uint32_t n = (uint32_t)g->dev.size();
    const bool plain = o.stats == 0u && root->settings.show_step_count != 1u;
    if (!plain || g->in_flight == 1u || g->last_was_stats) {   // a stats frame (counters are read back) stands alone
        const int rc = grp_synchronize(c);
        if (rc) return rc;
    }
    g->last_was_stats = !plain;
    const uint32_t k = g->slot;
    g->slot = (g->slot + 1u) % (g->in_flight > 1u ? vrt_group::kSlots : 1u);
    issued = true;
    o.flags |= VRT_RENDER_OWN_STREAMS;   // every device's frame runs on that context's in-flight streams, into the buffer bound here
    // what is issued to device r >= 1 for this frame — by its worker thread, or here
    const bool threaded = !g->workers.empty();
    g->frame_stream.store(nullptr, std::memory_order_release);
    auto issue = [g, k, o, root, threaded](uint32_t r) -> int {
        VRT_PROF(1, " issue to a shard device");
        vrt_ctx *d = g->dev[r];
        if (hipSetDevice(d->device) != hipSuccess) return fail(d, VRT_ERR_DEVICE, "hipSetDevice(%d) failed", d->device);
        void *slot = (uint8_t *)g->recv[k] + (size_t)r * g->rank_stride;
        int rc = vrt_bind_output(d, g->staged[r] ? g->stage[r][k] : slot);
        // the slot's previous message must have been consumed by device 0 before this frame overwrites it
        d->wait_before_frame = g->consumed_used[k] ? g->consumed[k] : nullptr;
        if (!rc) rc = vrt_render(d, &o);
        d->wait_before_frame = nullptr;
        if (rc) return rc;
        // (a staged message: behind the frame on its stream — which waited for the slot to be consumed — over to device 0)
        if (d->tiles_local && g->staged[r] &&
            hipMemcpyPeerAsync(slot, root->device, g->stage[r][k], d->device, g->rank_stride, d->last_stream) != hipSuccess)
            return fail(d, VRT_ERR_DEVICE, "hipMemcpyPeerAsync from device %d to device %d failed", d->device, root->device);
        VRT_PROF(2, "  record done");
        if (d->tiles_local && hipEventRecord(g->done[r][k], d->last_stream) != hipSuccess)
            return fail(d, VRT_ERR_DEVICE, "hipEventRecord failed on device %d", d->device);
        if (threaded) {
            // this thread, not the caller, tells device 0's frame stream to wait for this message — once the caller has enqueued
            // its own share there (the wait must come behind it: device 0's tiles do not wait for anybody's message)
            hipStream_t x;
            while ((x = g->frame_stream.load(std::memory_order_acquire)) == nullptr) __builtin_ia32_pause();
            if (d->tiles_local && x != vrt_group::no_frame_stream() && hipStreamWaitEvent(x, g->done[r][k], 0) != hipSuccess)
                return fail(d, VRT_ERR_DEVICE, "hipStreamWaitEvent failed for device %d's message", d->device);
        }
        return VRT_OK;
    };
    DeviceRestore restore;
    double shard_sum = 0.0, shard_max = 0.0;   // (vrt_get_issue_profile)
    if (!g->workers.empty()) {
        for (uint32_t r = 1; r < n; r++) g->workers[r - 1]->post([issue, r] { return issue(r); });
    } else {
        for (uint32_t r = 1; r < n; r++) {
            const auto t_r = std::chrono::steady_clock::now();
            const int rc = issue(r);
            if (rc) { c->err = g->dev[r]->err; return rc; }
            const double us = us_since(t_r);
            shard_sum += us;
            shard_max = us > shard_max ? us : shard_max;
        }
    }
    const auto t_root = std::chrono::steady_clock::now();
    int rc = hipSetDevice(root->device) == hipSuccess ? VRT_OK : fail(root, VRT_ERR_DEVICE, "hipSetDevice(%d) failed", root->device);
    if (!rc) rc = vrt_render(root, &o);
    hipStream_t X = root->last_stream ? root->last_stream : root->stream;
    g->frame_stream.store(rc ? vrt_group::no_frame_stream() : X, std::memory_order_release);   // (the issuing threads go on from here)
    const double root_us = us_since(t_root);
    const auto t_join = std::chrono::steady_clock::now();
    // the workers have *enqueued* their frames (their done events are recorded) before the root's stream is told to wait —
    // and they are joined on every path out of here: nothing of a context is ever touched by two threads
    int wrc = VRT_OK;
    for (uint32_t r = 1; r < n && !g->workers.empty(); r++) {
        const int one = g->workers[r - 1]->join();
        if (one && !wrc) { wrc = one; c->err = g->dev[r]->err; }
        const double us = g->workers[r - 1]->last_job_us;
        shard_sum += us;
        shard_max = us > shard_max ? us : shard_max;
    }
    const double join_us = g->workers.empty() ? 0.0 : us_since(t_join);
    if (rc) { c->err = root->err; return rc; }
    if (wrc) return wrc;
    const auto t_tail = std::chrono::steady_clock::now();
    VRT_PROF(4, " root: waits + assemble + record");
    if (!threaded)   // (the issuing threads have enqueued their own waits)
        for (uint32_t r = 1; r < n; r++)
            if (g->dev[r]->tiles_local) HIP_TRY(c, hipStreamWaitEvent(X, g->done[r][k], 0));
    const double waits_us = threaded ? 0.0 : us_since(t_tail);
    // shade / scatter the other devices' messages into the frame the root has just rendered its own tiles into
    vrt::Texel *frame = root->last_out;
    if (g->texels) {
        vrt::launch_assemble((const vrt::Texel *)g->recv[k], frame, root->width, root->tiles_x, root->tiles_total, root->shard_w0,
                             root->shard_period, true, g->rank_stride / 16u, X);
    } else {
        vrt::FrameParams P;
        memset(&P, 0, sizeof P);
        P.mats = root->d_mats;
        fill_uniforms(root, P);
        vrt::launch_assemble_shade(P, g->recv[k], frame, root->shard_w0, root->shard_period, g->rank_stride / 8u, X);
    }
    HIP_TRY(c, hipGetLastError());
    // (testing: a consumed slot holds nothing a later frame could pass for its own)
    if (g->poison) HIP_TRY(c, hipMemsetAsync((uint8_t *)g->recv[k] + g->rank_stride, 0xFF, g->rank_stride * (n - 1u), X));
    HIP_TRY(c, hipEventRecord(g->consumed[k], X));
    g->consumed_used[k] = true;
    g->prof.frames += 1u;
    g->prof.render += us_since(t_frame);
    g->prof.root += root_us;
    g->prof.shard_sum += shard_sum;
    g->prof.shard_max += shard_max;
    g->prof.join += join_us;
    g->prof.tail += us_since(t_tail);
    g->prof.waits += waits_us;
    return VRT_OK;
}

int grp_get_issue_profile(vrt_ctx *c, vrt_issue_profile *out) {
    vrt_group *g = c->grp;
    memset(out, 0, sizeof *out);
    out->devices = (uint32_t)g->dev.size();
    out->issuing_threads = (uint32_t)g->workers.size();
    out->frames = g->prof.frames;
    if (g->prof.frames) {
        const double f = (double)g->prof.frames, others = (double)(g->dev.size() > 1 ? g->dev.size() - 1 : 1);
        out->render_us = g->prof.render / f;
        out->root_issue_us = g->prof.root / f;
        out->shard_issue_us_mean = g->prof.shard_sum / f / others;
        out->shard_issue_us_max = g->prof.shard_max / f;
        out->join_wait_us = g->prof.join / f;
        out->tail_us = g->prof.tail / f;
        out->message_waits_us = g->prof.waits / f;
    }
    g->prof = {};
    for (vrt_ctx *d : g->dev) { d->prof_render_us = 0.0; d->prof_frames = 0; }
    return VRT_OK;
}

int grp_get_stats(vrt_ctx *c, vrt_stats *out) {
    if (!out) return fail(c, VRT_ERR_INVALID_ARG, "vrt_get_stats: null argument");
    DeviceRestore restore;
    int rc = grp_synchronize(c);
    if (rc) return rc;
    vrt_stats acc;
    memset(&acc, 0, sizeof acc);
    bool first = true;
    for (vrt_ctx *d : c->grp->dev) {
        vrt_stats s;
        (void)hipSetDevice(d->device);
        rc = vrt_get_stats(d, &s);
        if (rc) { c->err = d->err; return rc; }
        acc.primary_rays += s.primary_rays; acc.secondary_rays += s.secondary_rays; acc.hits += s.hits;
        acc.steps += s.steps; acc.node_visits += s.node_visits; acc.primary_steps += s.primary_steps;
        acc.primary_node_visits += s.primary_node_visits;
        if (first) {   // kernel times: the root's own launches
            acc.ms_total = s.ms_total; acc.ms_primary = s.ms_primary; acc.ms_secondary = s.ms_secondary; acc.frames = s.frames;
            acc.sum_ms_primary = s.sum_ms_primary; acc.sum_ms_secondary = s.sum_ms_secondary; acc.sum_ms_total = s.sum_ms_total;
            first = false;
        }
    }
    *out = acc;
    return VRT_OK;
}
