// vrt_kernels.hip — gfx950 kernels of the per-pixel SVO ray-march.
//
// What the kernels compute is fixed by the reference shader
// (clientdesktop/src/graphics/ray_tracer.wgsl: update :173-180, create_ray_from_screen :159-171,
// ray_world :182-316, find_node/find_chunk_node :76-125, ray_color :131-142, ray_sky :144-157).
// How they compute it is not: one wave64 per 8x8 tile, the in-chunk descent done on the integer
// voxel coordinate (exactly equivalent to the shader's f32 `pos >= center` compares because every
// centre compared against is an integer), chunk roots and the liquid mask staged in LDS, secondary
// rays launched from a wave-compacted hit buffer.  See DESIGN.md §Kernels.
#include "vrt_device.h"

namespace vrt {

// ------------------------------------------------------------------------------------------------
// Node lookup
// ------------------------------------------------------------------------------------------------

struct Leaf {
    uint32_t node;   // the leaf's 16-bit word
    uint32_t depth;  // 0..5
    int bx, by, bz;  // world-local integer min corner of the leaf
};

// find_node (ray_tracer.wgsl:116-125) + find_chunk_node (:76-114), restated on integers.
// vx = floor(pos) per axis (f2i: NaN -> 0).  chunk = v >> 5, local = v & 31; at depth d the shader's
// `pos >= center` is bit (4-d) of the local coordinate.
template <bool LDS_ROOTS>
__device__ __forceinline__ Leaf find_leaf(const FrameParams &P, const uint32_t *s_roots, int vx, int vy, int vz) {
    const uint32_t S = P.world.size_in_chunks;
    uint32_t cidx = (uint32_t)(vx >> 5) + (uint32_t)(vy >> 5) * S + (uint32_t)(vz >> 5) * S * S;
    cidx = min(cidx, P.n_roots - 1u);
    const uint32_t root = LDS_ROOTS ? s_roots[cidx] : P.roots[cidx];
    const uint32_t last = P.n_nodes - 1u;
    uint32_t idx = 0, depth = 0;
    uint32_t node = P.nodes[min(root, last)];
    while ((node & 0x8000u) && depth < 5u) {
        const uint32_t sh = 4u - depth;
        const uint32_t child = ((uint32_t)(vx >> sh) & 1u) | (((uint32_t)(vy >> sh) & 1u) << 1) |
                               (((uint32_t)(vz >> sh) & 1u) << 2);
        idx = (node & 0x7FFFu) + child;
        node = P.nodes[min(root + idx, last)];
        depth += 1u;
    }
    const int m = ~((32 >> depth) - 1);
    Leaf L;
    L.node = node;
    L.depth = depth;
    L.bx = vx & m;  // chunk bits pass through the mask unchanged
    L.by = vy & m;
    L.bz = vz & m;
    return L;
}

__device__ __forceinline__ bool is_liquid(const uint32_t *s_liquid, uint32_t voxel) {
    // voxel_mats[voxel].is_liquid == 1 (ray_tracer.wgsl:226); ids >= 256 clamp to material 255.
    const uint32_t v = min(voxel, 255u);
    return (s_liquid[v >> 5] >> (v & 31u)) & 1u;
}

// ------------------------------------------------------------------------------------------------
// ray_world, ray_tracer.wgsl:182-316
// ------------------------------------------------------------------------------------------------
template <bool LDS_ROOTS>
__device__ __forceinline__ MarchResult march(const FrameParams &P, const uint32_t *s_roots,
                                             const uint32_t *s_liquid, V3 origin, V3 dir) {
    MarchResult R;
    R.hit = false;
    R.pos = V3{0.f, 0.f, 0.f};
    R.norm = V3{0.f, 0.f, 0.f};
    R.water_dist = 0.0f;
    R.voxel = 0u;
    R.iters = 0u;
    R.visits = 0u;

    const V3 mask{dir.x >= 0.0f ? 1.0f : 0.0f, dir.y >= 0.0f ? 1.0f : 0.0f, dir.z >= 0.0f ? 1.0f : 0.0f};
    const V3 imask{1.0f - mask.x, 1.0f - mask.y, 1.0f - mask.z};

    V3 pos = origin;
    if (pos.x - floorf(pos.x) < 0.001f || pos.y - floorf(pos.y) < 0.001f || pos.z - floorf(pos.z) < 0.001f) {
        pos.x += 0.001f * dir.x;
        pos.y += 0.001f * dir.y;
        pos.z += 0.001f * dir.z;
    }
    const float world_min = 0.0f;
    const float world_max = world_min + (float)P.world.size;
    if ((pos.x <= world_min || pos.y <= world_min || pos.z <= world_min) ||
        (pos.x >= world_max || pos.y >= world_max || pos.z >= world_max)) {
        return R;
    }

    const V3 unit{
        sqrtf(1.0f + (dir.y / dir.x) * (dir.y / dir.x) + (dir.z / dir.x) * (dir.z / dir.x)),
        sqrtf(1.0f + (dir.x / dir.y) * (dir.x / dir.y) + (dir.z / dir.y) * (dir.z / dir.y)),
        sqrtf(1.0f + (dir.x / dir.z) * (dir.x / dir.z) + (dir.y / dir.z) * (dir.y / dir.z))};

    uint32_t voxel = 0u;
    float ex = 0.f, ey = 0.f, ez = 0.f;  // exit-axis flags of the last step (norm before the sign)
    float dist_entered_water = -1.0f;
    float total_len = 0.0f;
    uint32_t iter = 0u;
    bool left_world = false;

    while (iter < kMaxSteps) {
        iter += 1u;
        const Leaf L = find_leaf<LDS_ROOTS>(P, s_roots, f2i(floorf(pos.x)), f2i(floorf(pos.y)), f2i(floorf(pos.z)));
        voxel = L.node & 0x7FFFu;
        R.visits += L.depth + 1u;

        const bool liquid = is_liquid(s_liquid, voxel);
        if (voxel != 0u && !liquid) break;
        if (!liquid) {
            if (dist_entered_water != -1.0f) {
                R.water_dist += total_len - dist_entered_water;
                dist_entered_water = -1.0f;
            }
        } else {
            if (dist_entered_water == -1.0f) dist_entered_water = total_len;
        }

        const float size = (float)(32 >> L.depth);
        const V3 nmin{(float)L.bx, (float)L.by, (float)L.bz};
        const V3 nmax{nmin.x + size, nmin.y + size, nmin.z + size};
        const V3 ad{((pos.x - nmin.x) * imask.x + (nmax.x - pos.x) * mask.x) * unit.x,
                    ((pos.y - nmin.y) * imask.y + (nmax.y - pos.y) * mask.y) * unit.y,
                    ((pos.z - nmin.z) * imask.z + (nmax.z - pos.z) * mask.z) * unit.z};

        float step;
        if (ad.x == 0.0f) {
            if (ad.y == 0.0f) step = ad.z;
            else if (ad.z == 0.0f) step = ad.y;
            else step = vmin(ad.y, ad.z);
        } else {
            if (ad.y == 0.0f) {
                if (ad.z == 0.0f) step = ad.x;
                else step = vmin(ad.x, ad.z);
            } else {
                if (ad.z == 0.0f) step = vmin(ad.y, ad.x);
                else step = vmin(ad.x, vmin(ad.y, ad.z));
            }
        }
        total_len += step;
        ex = step == ad.x ? 1.0f : 0.0f;
        ey = step == ad.y ? 1.0f : 0.0f;
        ez = step == ad.z ? 1.0f : 0.0f;
        const float nx = step != ad.x ? 1.0f : 0.0f;
        const float ny = step != ad.y ? 1.0f : 0.0f;
        const float nz = step != ad.z ? 1.0f : 0.0f;
        pos.x += dir.x * (step + 0.001f) * ex + dir.x * step * nx;
        pos.y += dir.y * (step + 0.001f) * ey + dir.y * step * ny;
        pos.z += dir.z * (step + 0.001f) * ez + dir.z * step * nz;

        if ((pos.x < world_min || pos.y < world_min || pos.z < world_min) ||
            (pos.x >= world_max || pos.y >= world_max || pos.z >= world_max)) {
            if (dist_entered_water != -1.0f) R.water_dist += total_len - dist_entered_water;
            left_world = true;
            break;
        }
    }
    R.iters = iter;
    if (left_world) return R;  // hit = false, voxel = 0

    R.hit = true;
    R.pos = pos;
    R.norm = V3{ex * -vsign(dir.x), ey * -vsign(dir.y), ez * -vsign(dir.z)};
    R.voxel = voxel;
    if (dist_entered_water != -1.0f) R.water_dist += total_len - dist_entered_water;
    return R;
}

// ray_sky, ray_tracer.wgsl:144-157
__device__ __forceinline__ V3 ray_sky(const FrameParams &P, V3 origin, V3 dir) {
    const float ground_to_sky_t = vsmoothstep(-0.01f, 0.0f, dir.y);
    const float sky_gradient_t = powf(vsmoothstep(0.0f, 0.4f, dir.y), 0.35f);
    const V3 grad{vmix(1.0f, P.settings.sky_color[0], sky_gradient_t), vmix(0.3f, P.settings.sky_color[1], sky_gradient_t),
                  vmix(0.0f, P.settings.sky_color[2], sky_gradient_t)};
    const V3 sun_dir = vnormalize(V3{P.settings.sun_pos[0] - (float)P.world.min[0] - origin.x,
                                     P.settings.sun_pos[1] - (float)P.world.min[1] - origin.y,
                                     P.settings.sun_pos[2] - (float)P.world.min[2] - origin.z});
    const float sun = (vdot(dir, sun_dir) > (1.0f - 0.01f) && ground_to_sky_t >= 1.0f) ? 1.0f : 0.0f;
    const float add = sun * P.settings.sun_intensity;
    return V3{vmix(0.03f, grad.x, ground_to_sky_t) + add, vmix(0.03f, grad.y, ground_to_sky_t) + add,
              vmix(0.03f, grad.z, ground_to_sky_t) + add};
}

// create_ray_from_screen, ray_tracer.wgsl:159-171 (WGSL v*M = dot with the columns of M)
__device__ __forceinline__ void create_ray(const FrameParams &P, int sx, int sy, V3 &origin, V3 &dir) {
    const float x = ((float)sx * 2.0f) / P.cam.proj_size[0] - 1.0f;
    const float y = ((float)sy * 2.0f) / P.cam.proj_size[1] - 1.0f;
    const float c0 = x, c1 = -y, c2 = -1.0f, c3 = 1.0f;
    const float *ip = P.cam.inv_proj_mat;
    const float e0 = c0 * ip[0] + c1 * ip[1] + c2 * ip[2] + c3 * ip[3];
    const float e1 = c0 * ip[4] + c1 * ip[5] + c2 * ip[6] + c3 * ip[7];
    const float *iv = P.cam.inv_view_mat;
    const float e2 = -1.0f, e3 = 0.0f;
    const V3 w{e0 * iv[0] + e1 * iv[1] + e2 * iv[2] + e3 * iv[3], e0 * iv[4] + e1 * iv[5] + e2 * iv[6] + e3 * iv[7],
               e0 * iv[8] + e1 * iv[9] + e2 * iv[10] + e3 * iv[11]};
    dir = vnormalize(w);
    origin = V3{P.cam.pos[0] - (float)P.world.min[0], P.cam.pos[1] - (float)P.world.min[1],
                P.cam.pos[2] - (float)P.world.min[2]};
}

// Face shading + ray_color + overlay, ray_tracer.wgsl:127-142, 296-314. Returns the id word.
__device__ __forceinline__ uint32_t shade(const FrameParams &P, const MarchResult &R, V3 origin, V3 dir, V3 &color) {
    V3 mc{0.f, 0.f, 0.f};
    if (R.hit) {
        const vrt_material *m = &P.mats[min(R.voxel, 255u)];
        mc = V3{m->color[0], m->color[1], m->color[2]};
        if (R.norm.x != 0.0f) { mc.x *= 0.5f; mc.y *= 0.5f; mc.z *= 0.5f; }
        if (R.norm.z != 0.0f) { mc.x *= 0.7f; mc.y *= 0.7f; mc.z *= 0.7f; }
        if (R.norm.y == -1.0f) { mc.x *= 0.2f; mc.y *= 0.2f; mc.z *= 0.2f; }
        if (P.settings.show_step_count == 1u) {
            const float f = vclamp((float)R.iters / 500.0f, 0.0f, 1.0f);
            mc = V3{f, f, f};
        }
    }
    const V3 sky = ray_sky(P, origin, dir);
    const float fh = R.hit ? 1.0f : 0.0f, fm = R.hit ? 0.0f : 1.0f;
    color = V3{mc.x * fh + sky.x * fm, mc.y * fh + sky.y * fm, mc.z * fh + sky.z * fm};
    if (R.water_dist != 0.0f) {
        const float factor = vclamp(R.water_dist / 14.0f, 0.8f, 1.0f);
        color.x = color.x * (1.0f - factor) + 0.2f * factor;
        color.y = color.y * (1.0f - factor) + 0.5f * factor;
        color.z = color.z * (1.0f - factor) + 1.0f * factor;
    }
    uint32_t id = R.voxel & VRT_ID_VOXEL_MASK;
    if (R.hit) id |= VRT_ID_HIT;
    if (R.norm.x != 0.0f) id |= VRT_ID_NX;
    if (R.norm.y != 0.0f) id |= VRT_ID_NY;
    if (R.norm.z != 0.0f) id |= VRT_ID_NZ;
    if (R.water_dist != 0.0f) id |= VRT_ID_WATER;
    return id;
}

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ void stage_lds(const FrameParams &P, uint32_t *s_roots, uint32_t *s_liquid, bool lds_roots) {
    if (lds_roots)
        for (uint32_t i = threadIdx.x; i < P.n_roots; i += blockDim.x) s_roots[i] = P.roots[i];
    if (threadIdx.x < 8) s_liquid[threadIdx.x] = P.liquid[threadIdx.x];
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// Variant 0: one wave per 8x8 tile (the reference's @workgroup_size(8,8,1) = one CDNA wave),
// 4 tiles per 256-thread workgroup.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kLdsRootsMax = 8192;  // entries (32 KiB): S <= 20

template <bool LDS_ROOTS, bool STATS, bool SHADOW>
__global__ void __launch_bounds__(256) primary_tile_kernel(FrameParams P) {
    extern __shared__ uint32_t smem[];  // [0,8) liquid mask, [8, 8+n_roots) chunk roots
    uint32_t *s_liquid = smem, *s_roots = smem + 8;
    stage_lds(P, s_roots, s_liquid, LDS_ROOTS);

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t t_local = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (t_local >= P.tiles_local) return;
    const uint32_t tile = P.shard_rank + t_local * P.shard_count;
    const uint32_t px = (tile % P.tiles_x) * 8u + (lane & 7u);
    const uint32_t py = (tile / P.tiles_x) * 8u + (lane >> 3);
    const uint32_t slot = P.shard_count > 1u ? t_local * 64u + lane : py * P.width + px;

    V3 origin, dir;
    create_ray(P, (int)px, (int)py, origin, dir);
    const MarchResult R = march<LDS_ROOTS>(P, s_roots, s_liquid, origin, dir);
    V3 color;
    uint32_t id = shade(P, R, origin, dir, color);

    bool launch = false;
    if (SHADOW) {
        launch = R.hit && R.voxel != 0u && !is_liquid(s_liquid, R.voxel);
        if (launch) id |= VRT_ID_SHADOW_RAY;
    }
    P.rgb[slot * 3u + 0u] = color.x;
    P.rgb[slot * 3u + 1u] = color.y;
    P.rgb[slot * 3u + 2u] = color.z;
    P.ids[slot] = id;

    if (SHADOW) {
        // wave-aggregated compaction into the hit buffer: one atomic per wave
        const unsigned long long ballot = __ballot(launch);
        const uint32_t n = (uint32_t)__popcll(ballot);
        uint32_t base = 0;
        if (n) {
            if (lane == (uint32_t)__ffsll((long long)ballot) - 1u)
                base = (uint32_t)atomicAdd(&P.counters[kCtrHitCount], (unsigned long long)n);
            base = __shfl(base, __ffsll((long long)ballot) - 1, 64);
        }
        if (launch) {
            const uint32_t rank = (uint32_t)__popcll(ballot & ((1ull << lane) - 1ull));
            const V3 so{R.pos.x + R.norm.x * kShadowBias, R.pos.y + R.norm.y * kShadowBias, R.pos.z + R.norm.z * kShadowBias};
            P.hits[base + rank] = make_uint4(slot, __float_as_uint(so.x), __float_as_uint(so.y), __float_as_uint(so.z));
        }
    }
    if (STATS) {
        if (P.steps) P.steps[slot] = R.iters;
        const unsigned long long s = wave_sum(R.iters), v = wave_sum(R.visits), h = wave_sum(R.hit ? 1ull : 0ull);
        if (lane == 0) {
            atomicAdd(&P.counters[kCtrSteps], s);
            atomicAdd(&P.counters[kCtrVisits], v);
            atomicAdd(&P.counters[kCtrPrimarySteps], s);
            atomicAdd(&P.counters[kCtrPrimaryVisits], v);
            atomicAdd(&P.counters[kCtrHits], h);
        }
    }
}

// Shadow rays from the compacted hit buffer: lane i of the grid takes record i.
template <bool LDS_ROOTS, bool STATS>
__global__ void __launch_bounds__(256) shadow_kernel(FrameParams P) {
    extern __shared__ uint32_t smem[];  // [0,8) liquid mask, [8, 8+n_roots) chunk roots
    uint32_t *s_liquid = smem, *s_roots = smem + 8;
    stage_lds(P, s_roots, s_liquid, LDS_ROOTS);

    const uint32_t count = (uint32_t)P.counters[kCtrHitCount];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = i < count;
    MarchResult R;
    R.iters = 0; R.visits = 0; R.hit = false;
    uint32_t slot = 0;
    if (active) {
        const uint4 rec = P.hits[i];
        slot = rec.x;
        const V3 so{__uint_as_float(rec.y), __uint_as_float(rec.z), __uint_as_float(rec.w)};
        const V3 sd = vnormalize(V3{P.settings.sun_pos[0] - (float)P.world.min[0] - so.x,
                                    P.settings.sun_pos[1] - (float)P.world.min[1] - so.y,
                                    P.settings.sun_pos[2] - (float)P.world.min[2] - so.z});
        R = march<LDS_ROOTS>(P, s_roots, s_liquid, so, sd);
        if (R.hit) {
            P.rgb[slot * 3u + 0u] *= kShadowFactor;
            P.rgb[slot * 3u + 1u] *= kShadowFactor;
            P.rgb[slot * 3u + 2u] *= kShadowFactor;
            P.ids[slot] |= VRT_ID_SHADOWED;
        }
    }
    if (STATS) {
        if (active && P.steps) P.steps[slot] |= R.iters << 16;
        const unsigned long long s = wave_sum(active ? R.iters : 0u), v = wave_sum(active ? R.visits : 0u);
        if ((threadIdx.x & 63u) == 0 && s) {
            atomicAdd(&P.counters[kCtrSteps], s);
            atomicAdd(&P.counters[kCtrVisits], v);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Output helpers
// ------------------------------------------------------------------------------------------------

// textureStore to rgba8unorm (ray_tracer.wgsl:179): clamp to [0,1], scale by 255, round to nearest.
__global__ void quantize_rgba8_kernel(const float *rgb, uint8_t *rgba8, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t out = 0xFF000000u;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float v = vclamp(rgb[i * 3u + c], 0.0f, 1.0f);
        out |= ((uint32_t)rintf(v * 255.0f) & 0xFFu) << (8 * c);
    }
    reinterpret_cast<uint32_t *>(rgba8)[i] = out;
}

// Gather root: tile-major [rank][tiles_padded][64] -> row-major frame.
__global__ void assemble_kernel(const float *g_rgb, const uint32_t *g_ids, float *dst_rgb, uint32_t *dst_ids,
                                uint32_t width, uint32_t tiles_x, uint32_t tiles_total, uint32_t shard_count,
                                uint64_t stride_rgb, uint64_t stride_ids) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t tile = gid >> 6, p = gid & 63u;
    if (tile >= tiles_total) return;
    const uint32_t rank = tile % shard_count, t_local = tile / shard_count;
    const uint64_t local = (uint64_t)t_local * 64u + p;  // slot inside rank's buffer
    const float *s_rgb = g_rgb + rank * stride_rgb + local * 3u;
    const uint32_t px = (tile % tiles_x) * 8u + (p & 7u), py = (tile / tiles_x) * 8u + (p >> 3);
    const uint32_t dst = py * width + px;
    if (dst_rgb) {
        dst_rgb[dst * 3u + 0u] = s_rgb[0];
        dst_rgb[dst * 3u + 1u] = s_rgb[1];
        dst_rgb[dst * 3u + 2u] = s_rgb[2];
    }
    if (dst_ids) dst_ids[dst] = g_ids[rank * stride_ids + local];
}

// ------------------------------------------------------------------------------------------------
// Launchers (called from vrt_backend.hip)
// ------------------------------------------------------------------------------------------------

static size_t lds_bytes(const FrameParams &P, bool lds_roots) { return (8u + (lds_roots ? P.n_roots : 0u)) * 4u; }

template <bool LDS_ROOTS>
static void launch_primary_v0(const FrameParams &P, bool stats, bool shadow, hipStream_t st) {
    const dim3 grid((P.tiles_local + 3u) / 4u), block(256);
    const size_t lds = lds_bytes(P, LDS_ROOTS);
    if (stats) {
        if (shadow) hipLaunchKernelGGL((primary_tile_kernel<LDS_ROOTS, true, true>), grid, block, lds, st, P);
        else hipLaunchKernelGGL((primary_tile_kernel<LDS_ROOTS, true, false>), grid, block, lds, st, P);
    } else {
        if (shadow) hipLaunchKernelGGL((primary_tile_kernel<LDS_ROOTS, false, true>), grid, block, lds, st, P);
        else hipLaunchKernelGGL((primary_tile_kernel<LDS_ROOTS, false, false>), grid, block, lds, st, P);
    }
}

void launch_primary(const FrameParams &P, uint32_t variant, bool stats, bool shadow, hipStream_t st) {
    (void)variant;
    if (P.tiles_local == 0) return;
    if (P.n_roots <= kLdsRootsMax) launch_primary_v0<true>(P, stats, shadow, st);
    else launch_primary_v0<false>(P, stats, shadow, st);
}

void launch_shadow(const FrameParams &P, uint32_t variant, bool stats, hipStream_t st) {
    (void)variant;
    if (P.tiles_local == 0) return;
    const dim3 grid((P.tiles_local * 64u + 255u) / 256u), block(256);
    if (P.n_roots <= kLdsRootsMax) {
        if (stats) hipLaunchKernelGGL((shadow_kernel<true, true>), grid, block, lds_bytes(P, true), st, P);
        else hipLaunchKernelGGL((shadow_kernel<true, false>), grid, block, lds_bytes(P, true), st, P);
    } else {
        if (stats) hipLaunchKernelGGL((shadow_kernel<false, true>), grid, block, lds_bytes(P, false), st, P);
        else hipLaunchKernelGGL((shadow_kernel<false, false>), grid, block, lds_bytes(P, false), st, P);
    }
}

void launch_quantize(const float *rgb, uint8_t *rgba8, uint32_t n, hipStream_t st) {
    if (!n) return;
    hipLaunchKernelGGL(quantize_rgba8_kernel, dim3((n + 255u) / 256u), dim3(256), 0, st, rgb, rgba8, n);
}

void launch_assemble(const float *g_rgb, const uint32_t *g_ids, float *dst_rgb, uint32_t *dst_ids, uint32_t width,
                     uint32_t tiles_x, uint32_t tiles_total, uint32_t shard_count, uint64_t stride_rgb,
                     uint64_t stride_ids, hipStream_t st) {
    if (!tiles_total) return;
    hipLaunchKernelGGL(assemble_kernel, dim3((tiles_total * 64u + 255u) / 256u), dim3(256), 0, st, g_rgb, g_ids,
                       dst_rgb, dst_ids, width, tiles_x, tiles_total, shard_count, stride_rgb, stride_ids);
}

}  // namespace vrt
