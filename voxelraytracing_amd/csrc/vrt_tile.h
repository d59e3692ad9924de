// vrt_tile.h — one finished pixel and one tile = one wave (primary march, shading, the shadow march of the lanes that hit, one
// store per lane), and the presentation arithmetic the blit kernels share with that store (vrt_kernels.hip).
#pragma once

#include "vrt_march.h"

namespace vrt {

// ------------------------------------------------------------------------------------------------
// Presentation arithmetic: fs_main of screen_shader.wgsl:43-65 over the rgba8unorm result texture (ray_tracer.wgsl:179),
// shared by the blit kernels (vrt_kernels.hip: present_kernel, present_plain_kernel) and by the march kernels' own store
// of the window's pixel (store_pixel below, vrt_set_presentation): the same operations on the same values, so the same bytes.
// ------------------------------------------------------------------------------------------------
// textureStore to rgba8unorm / the colour target's unorm8: clamp to [0, 1], scale by 255, round to nearest
__device__ __forceinline__ uint32_t unorm8(float x) { return (uint32_t)rintf(vclamp(x, 0.0f, 1.0f) * 255.0f) & 0xFFu; }

// Where screen pixel (sx, sy) of a screen_w x screen_h window samples the w x h texture through the reference's sampler
// (texture.rs:31-44: lod clamped to [1, 1] => the minification filter, Linear, at every window size; ClampToEdge): the four
// taps' coordinates and the weights; and fs_main's crosshair mask (in_box: the pixel is within the host's box around the
// crosshair, a pixel wider than the mask can reach — outside it the mask is zero without being computed).
struct PresentSample { int x0, x1, y0, y1; float a, b, mask; };
__device__ __forceinline__ PresentSample present_sample(uint32_t w, uint32_t h, uint32_t screen_w, uint32_t screen_h, const vrt_crosshair &ch, uint32_t sx, uint32_t sy,
                                                        bool in_box) {
    const float ssx = (float)screen_w, ssy = (float)screen_h;
    const float cx = ssx * 0.5f, cy = ssy * 0.5f;
    const float u = ((float)sx + 0.5f) / ssx, v = ((float)sy + 0.5f) / ssy;
    const float px = u * ssx, py = v * ssy;
    PresentSample S;
    S.mask = 0.0f;
    if (in_box && ch.style == 1u) {
        const float dx = cx - px, dy = cy - py;
        S.mask = (sqrtf(dx * dx + dy * dy) < ch.size ? 1.0f : 0.0f) * ch.color[3];
    }
    if (in_box && ch.style == 2u) {
        const float dx = fabsf(cx - px), dy = fabsf(cy - py);
        const float wd = ch.size * 0.25f;
        S.mask = (((dx < ch.size && dy < wd) || (dy < ch.size && dx < wd)) ? 1.0f : 0.0f) * ch.color[3];
    }
    const float ut = u * (float)w - 0.5f, vt = v * (float)h - 0.5f;
    const float fu = floorf(ut), fv = floorf(vt);
    S.a = ut - fu;
    S.b = vt - fv;
    S.x0 = min(max((int)fu, 0), (int)w - 1); S.x1 = min(max((int)fu + 1, 0), (int)w - 1);
    S.y0 = min(max((int)fv, 0), (int)h - 1); S.y1 = min(max((int)fv + 1, 0), (int)h - 1);
    return S;
}
// The sample from its taps (decoded unorm8 texels {r, g, b, alpha}), blended with the crosshair, as unorm8 RGBA.  When the
// sample is at tap 00's centre (a == 0 and b == 0) the other three taps have weight zero, and x * 1 + y * 0 is x for the
// finite x and y a decoded unorm8 is: they are not looked at.
__device__ __forceinline__ uint32_t present_blend(const PresentSample &S, const vrt_crosshair &ch, const float v00[4], const float v10[4], const float v01[4],
                                                  const float v11[4]) {
    float texel[4];
    if (S.a == 0.0f && S.b == 0.0f) {
#pragma unroll
        for (int k = 0; k < 4; k++) texel[k] = v00[k];
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float top = v00[k] * (1.0f - S.a) + v10[k] * S.a, bot = v01[k] * (1.0f - S.a) + v11[k] * S.a;
            texel[k] = top * (1.0f - S.b) + bot * S.b;
        }
    }
    // (uniform) outside the crosshair the blend is x * 1 + c * 0 = x — for a FINITE colour c (x >= 0: adding -0 changes nothing
    // either); a crosshair whose colour is not a number takes the blend everywhere, as the shader would
    const bool colour_finite = ((__float_as_uint(ch.color[0]) & 0x7F800000u) != 0x7F800000u) && ((__float_as_uint(ch.color[1]) & 0x7F800000u) != 0x7F800000u) &&
                               ((__float_as_uint(ch.color[2]) & 0x7F800000u) != 0x7F800000u);
    uint32_t q = 0u;
    if (S.mask == 0.0f && colour_finite) {   // unorm8(x * 1 + c * 0) = unorm8(x)
#pragma unroll
        for (int k = 0; k < 4; k++) q |= unorm8(texel[k]) << (8 * k);
        return q;
    }
    const float cc[4] = {ch.color[0], ch.color[1], ch.color[2], 1.0f};
#pragma unroll
    for (int k = 0; k < 4; k++) q |= unorm8(texel[k] * (1.0f - S.mask) + cc[k] * S.mask) << (8 * k);
    return q;
}

// The window's pixel out of the march kernel itself (vrt_set_presentation; FrameParams.screen): the reference's compute pass
// stores rgba8unorm (ray_tracer.wgsl:179) and its blit follows in the same submission (main.rs:452-454) — here a window of the
// texture's size whose every pixel samples its own texel's centre (the host's finding, vrt_present.hip) gets its pixel from
// the lane that traced it: the colour quantised, alpha 1.  A tile that reaches into the box around the crosshair blends it in
// with fs_main's own arithmetic; the taps a sample beside its texel's centre needs (1920 columns: 51 of them sit 2e-6 of a
// texel beside it) are other lanes' pixels of the same tile — the host has checked that for every pixel of the box — and come
// over the wave's cross-lane network.  All 64 lanes of the wave are here (a tile is traced whole).
__device__ __forceinline__ void store_screen(const FrameParams &P, uint32_t px, uint32_t py, V3 color) {
    const uint32_t qr = unorm8(color.x), qg = unorm8(color.y), qb = unorm8(color.z);
    uint32_t q = 0xFF000000u | qr | (qg << 8) | (qb << 16);
    const uint32_t tx0 = __builtin_amdgcn_readfirstlane(px & ~7u), ty0 = __builtin_amdgcn_readfirstlane(py & ~7u);
    if (P.present_box[0] < P.present_box[1] && tx0 < P.present_box[1] && tx0 + 8u > P.present_box[0] && ty0 < P.present_box[3] && ty0 + 8u > P.present_box[2]) {   // (wave-uniform)
        const bool in_box = px >= P.present_box[0] && px < P.present_box[1] && py >= P.present_box[2] && py < P.present_box[3];
        const PresentSample S = present_sample(P.width, P.height, P.width, P.height, P.crosshair, px, py, in_box);
        // this lane's texel as the sampler decodes it; every tap is a pixel of this tile
        const float own[3] = {(float)qr / 255.0f, (float)qg / 255.0f, (float)qb / 255.0f};
        const int l00 = ((S.y0 & 7) << 3) | (S.x0 & 7), l10 = ((S.y0 & 7) << 3) | (S.x1 & 7), l01 = ((S.y1 & 7) << 3) | (S.x0 & 7), l11 = ((S.y1 & 7) << 3) | (S.x1 & 7);
        float v00[4], v10[4], v01[4], v11[4];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            v00[k] = __shfl(own[k], l00, 64); v10[k] = __shfl(own[k], l10, 64);
            v01[k] = __shfl(own[k], l01, 64); v11[k] = __shfl(own[k], l11, 64);
        }
        v00[3] = v10[3] = v01[3] = v11[3] = 1.0f;   // (every texel of a traced tile is covered)
        if (in_box) q = present_blend(S, P.crosshair, v00, v10, v01, v11);
    }
    __builtin_nontemporal_store(q, &P.screen[py * P.width + px]);   // (read by the window system, not by this device: see store_streaming)
}

// One finished pixel: the 16-byte texel, or (VRT_FLAG_COMPACT, a shard whose tiles go over a link) the 8 bytes the
// gather root needs to shade it itself: the id word (+ the sign of norm.y) and water_dist.  Everything else the colour
// depends on — material, face factors, shadow factor, the sky of a miss — is a function of those and of the frame's
// uniforms, which the root holds too (assemble_shade_kernel).  A frame that is presented 1:1 (FrameParams.screen) also — or, with
// VRT_PRESENT_SKIP_TEXELS, only — stores the window's pixel.
__device__ __forceinline__ void store_pixel(const FrameParams &P, uint32_t slot, uint32_t px, uint32_t py, V3 color, uint32_t id, const MarchResult &R) {
    if (P.compact)
        reinterpret_cast<uint2 *>(P.out)[slot] = make_uint2(id | (R.norm.y < 0.0f ? kIdNormYNeg : 0u), __float_as_uint(R.water_dist));
    else if (!P.screen_only)
        store_streaming(&P.out[slot], make_uint4(__float_as_uint(color.x), __float_as_uint(color.y), __float_as_uint(color.z), id));
    if (P.screen) store_screen(P, px, py, color);
}

// One tile = one wave: primary march, shading, the shadow march of the lanes that hit, one store per lane.
template <int MARCH, bool LDS_ROOTS, bool STATS>
__device__ __forceinline__ void trace_tile(const FrameParams &P, const uint32_t *s_roots, const uint32_t *s_liquid, uint32_t t_local,
                                           uint32_t lane, MarchResult &R, MarchResult &S) {
    const uint32_t tile = shard_tile(t_local, P.shard_first, P.shard_run, P.shard_period);
    uint32_t px, py;
    tile_pixel(P, tile, lane, px, py);
    const uint32_t slot = P.tile_major ? t_local * 64u + lane : py * P.width + px;

    V3 origin, dir;
    create_ray(P, (int)px, (int)py, origin, dir);
    R = march<MARCH, LDS_ROOTS, STATS, true>(P, s_roots, s_liquid, origin, dir);
    V3 color;
    uint32_t id = shade<MARCH == 1>(P, R, origin, dir, color);

    const bool launch = R.hit && R.voxel != 0u && !is_liquid(s_liquid, R.voxel);
    if (launch) {
        id |= VRT_ID_SHADOW_RAY;
        const V3 so{R.pos.x + R.norm.x * kShadowBias, R.pos.y + R.norm.y * kShadowBias, R.pos.z + R.norm.z * kShadowBias};
        const V3 sd = normalize_wave(V3{P.sun_local[0] - so.x, P.sun_local[1] - so.y, P.sun_local[2] - so.z});   // (sun_pos - f32(world.min), the host's)
        S = march<MARCH, LDS_ROOTS, STATS>(P, s_roots, s_liquid, so, sd);
        if (S.hit) {
            color.x *= kShadowFactor;
            color.y *= kShadowFactor;
            color.z *= kShadowFactor;
            id |= VRT_ID_SHADOWED;
        }
    }
    store_pixel(P, slot, px, py, color, id, R);
    if (STATS && P.steps) P.steps[slot] = R.iters | (S.iters << 16);
    const unsigned long long ballot = __ballot(launch);
    if (lane == 0) P.blk_counts[t_local] = (uint32_t)__popcll(ballot);  // per tile: the launched-ray count of vrt_get_stats
}

}  // namespace vrt
