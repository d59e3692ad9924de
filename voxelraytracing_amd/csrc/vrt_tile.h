// vrt_tile.h — one finished pixel and one tile = one wave (primary march, shading, the shadow march of the lanes that hit, one
// store per lane): shared by vrt_kernels.hip and the persistent grid of experiments/vrt_kernels_experiments.hip.
#pragma once

#include "vrt_march.h"

namespace vrt {

// One finished pixel: the 16-byte texel, or (VRT_FLAG_COMPACT, a shard whose tiles go over a link) the 8 bytes the
// gather root needs to shade it itself: the id word (+ the sign of norm.y) and water_dist.  Everything else the colour
// depends on — material, face factors, shadow factor, the sky of a miss — is a function of those and of the frame's
// uniforms, which the root holds too (assemble_shade_kernel).
__device__ __forceinline__ void store_pixel(const FrameParams &P, uint32_t slot, V3 color, uint32_t id, const MarchResult &R) {
    if (P.compact)
        reinterpret_cast<uint2 *>(P.out)[slot] = make_uint2(id | (R.norm.y < 0.0f ? kIdNormYNeg : 0u), __float_as_uint(R.water_dist));
    else
        P.out[slot] = make_uint4(__float_as_uint(color.x), __float_as_uint(color.y), __float_as_uint(color.z), id);
}

// One tile = one wave: primary march, shading, the shadow march of the lanes that hit, one store per lane.
template <int MARCH, bool LDS_ROOTS, bool STATS>
__device__ __forceinline__ void trace_tile(const FrameParams &P, const uint32_t *s_roots, const uint32_t *s_liquid, uint32_t t_local,
                                           uint32_t lane, MarchResult &R, MarchResult &S) {
    const uint32_t tile = shard_tile(t_local, P.shard_first, P.shard_run, P.shard_period);
    const uint32_t px = (tile % P.tiles_x) * 8u + (lane & 7u);
    const uint32_t py = (tile / P.tiles_x) * 8u + (lane >> 3);
    const uint32_t slot = P.tile_major ? t_local * 64u + lane : py * P.width + px;

    V3 origin, dir;
    create_ray(P, (int)px, (int)py, origin, dir);
    R = march<MARCH, LDS_ROOTS, STATS>(P, s_roots, s_liquid, origin, dir);
    V3 color;
    uint32_t id = shade<MARCH == 1>(P, R, origin, dir, color);

    const bool launch = R.hit && R.voxel != 0u && !is_liquid(s_liquid, R.voxel);
    if (launch) {
        id |= VRT_ID_SHADOW_RAY;
        const V3 so{R.pos.x + R.norm.x * kShadowBias, R.pos.y + R.norm.y * kShadowBias, R.pos.z + R.norm.z * kShadowBias};
        const V3 sd = normalize_wave(V3{P.settings.sun_pos[0] - (float)P.world.min[0] - so.x,
                                    P.settings.sun_pos[1] - (float)P.world.min[1] - so.y,
                                    P.settings.sun_pos[2] - (float)P.world.min[2] - so.z});
        S = march<MARCH, LDS_ROOTS, STATS>(P, s_roots, s_liquid, so, sd);
        if (S.hit) {
            color.x *= kShadowFactor;
            color.y *= kShadowFactor;
            color.z *= kShadowFactor;
            id |= VRT_ID_SHADOWED;
        }
    }
    store_pixel(P, slot, color, id, R);
    if (STATS && P.steps) P.steps[slot] = R.iters | (S.iters << 16);
    const unsigned long long ballot = __ballot(launch);
    if (lane == 0) P.blk_counts[t_local] = (uint32_t)__popcll(ballot);  // per tile: the launched-ray count of vrt_get_stats
}

}  // namespace vrt
