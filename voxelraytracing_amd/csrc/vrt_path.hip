// vrt_path.hip — wavefront path trace (VRT_MODE_PATH) for gfx950.
//
// Structure after the reference's stale, never-dispatched path tracer
// (clientdesktop/src/graphics/path_tracer.wgsl: rng_next* :56-76, ray_color :149-194, seed :328) on top of the
// live march of ray_tracer.wgsl; the deliberate differences (bounce origin outside the hit voxel, clamped
// log argument, no emission, water neither stops nor tints a segment) are DESIGN.md §Path trace and are the
// same in oracle/vrt_oracle.c:trace_path.  log and cos are spelled out in + - * / so that host and device
// agree to the bit: a one-ulp different bounce direction eventually hits a different voxel.
//
// One launch per bounce: bounce 0 traces the primary rays; every later bounce reads the compacted buffer of
// paths that are still alive (same per-segment ballot compaction as the shadow hit buffer), marches them and
// appends the survivors to the other buffer.  A pixel's path has exactly one owner lane per bounce, so
// radiance accumulates into the pixel's texel with plain read-modify-writes.
#include "vrt_march.h"

namespace vrt {

// rng_next, path_tracer.wgsl:56-61
__device__ __forceinline__ float rng_next(uint32_t &state) {
    state = state * 747796405u + 2891336453u;
    uint32_t r = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    r = (r >> 22u) ^ r;
    return (float)r / 4294967295.0f;
}

// ln(x), x normal > 0: x = m * 2^e with m in (sqrt(1/2), sqrt(2)], ln m = 2 atanh((m-1)/(m+1))
__device__ __forceinline__ float vlog(float x) {
    const uint32_t b = __float_as_uint(x);
    int e = (int)(b >> 23) - 127;
    float m = __uint_as_float((b & 0x007FFFFFu) | 0x3F800000u);
    if (m > 1.41421354f) { m = m * 0.5f; e += 1; }
    const float s = (m - 1.0f) / (m + 1.0f);
    const float z = s * s;
    const float p = z * (0.333333343f + z * (0.2f + z * (0.142857149f + z * (0.111111112f + z * 0.0909090936f))));
    return (float)e * 0.693147182f + (s + s * p) * 2.0f;
}

// cos(2*pi*u), u in [0,1]
__device__ __forceinline__ float vcos2pi(float u) {
    const float t = u * 4.0f;
    const float q = floorf(t);
    const float a = (t - q) * 1.57079637f;
    const float a2 = a * a;
    const float sn = a * (1.0f + a2 * (-0.166666672f + a2 * (0.00833333377f + a2 * (-0.000198412701f + a2 * (2.75573188e-06f + a2 * -2.50521079e-08f)))));
    const float cs = 1.0f + a2 * (-0.5f + a2 * (0.0416666679f + a2 * (-0.00138888892f + a2 * (2.48015876e-05f + a2 * (-2.75573199e-07f + a2 * 2.08767559e-09f)))));
    const int qi = (int)q & 3;
    return qi == 0 ? cs : (qi == 1 ? -sn : (qi == 2 ? -cs : sn));
}

// rng_next_norm / rng_next_dir, path_tracer.wgsl:62-72
__device__ __forceinline__ float rng_next_norm(uint32_t &state) {
    const float u1 = rng_next(state);
    float u2 = rng_next(state);
    if (u2 < 1.0e-10f) u2 = 1.0e-10f;
    const float rho = sqrtf(-2.0f * vlog(u2));
    return rho * vcos2pi(u1);
}
__device__ __forceinline__ V3 rng_next_dir(uint32_t &state) {
    const float x = rng_next_norm(state);
    const float y = rng_next_norm(state);
    const float z = rng_next_norm(state);
    return vnormalize(V3{x, y, z});
}

// The material colour of a hit after face shading (ray_tracer.wgsl:296-314) — shade()'s first half.
__device__ __forceinline__ V3 hit_color(const FrameParams &P, const MarchResult &R) {
    const vrt_material *m = &P.mats[min(R.voxel, 255u)];
    V3 mc{m->color[0], m->color[1], m->color[2]};
    if (R.norm.x != 0.0f) { mc.x *= 0.5f; mc.y *= 0.5f; mc.z *= 0.5f; }
    if (R.norm.z != 0.0f) { mc.x *= 0.7f; mc.y *= 0.7f; mc.z *= 0.7f; }
    if (R.norm.y == -1.0f) { mc.x *= 0.2f; mc.y *= 0.2f; mc.z *= 0.2f; }
    if (P.settings.show_step_count == 1u) {
        const float f = vclamp((float)R.iters / 500.0f, 0.0f, 1.0f);
        mc = V3{f, f, f};
    }
    return mc;
}

struct PathState {
    uint32_t slot;
    V3 origin, dir, thr;
    uint32_t rng;
};

// One segment of a path (the body of ray_color's loop, path_tracer.wgsl:155-192). Returns true if the path
// goes on (st updated to the next segment); adds a miss's sky light to `light`.
template <int MARCH, bool LDS_ROOTS, bool STATS>
__device__ __forceinline__ bool path_segment(const FrameParams &P, const uint32_t *s_roots, const uint32_t *s_liquid,
                                             PathState &st, MarchResult &R, V3 &light, bool &missed) {
    R = march<MARCH, LDS_ROOTS, STATS>(P, s_roots, s_liquid, st.origin, st.dir);
    missed = !R.hit;
    if (!R.hit) {
        const V3 sky = ray_sky(P, st.origin, st.dir);
        light = V3{sky.x * st.thr.x, sky.y * st.thr.y, sky.z * st.thr.z};
        return false;
    }
    const V3 mc = hit_color(P, R);
    const float d = vdot(R.norm, st.dir);
    const V3 spec{st.dir.x - 2.0f * R.norm.x * d, st.dir.y - 2.0f * R.norm.y * d, st.dir.z - 2.0f * R.norm.z * d};
    const V3 rd = rng_next_dir(st.rng);
    const V3 sc = vnormalize(V3{R.norm.x + rd.x, R.norm.y + rd.y, R.norm.z + rd.z});
    const float scatter = P.mats[min(R.voxel, 255u)].scatter;
    const V3 nd = vnormalize(V3{vmix(spec.x, sc.x, scatter), vmix(spec.y, sc.y, scatter), vmix(spec.z, sc.z, scatter)});
    st.thr = V3{st.thr.x * mc.x, st.thr.y * mc.y, st.thr.z * mc.z};
    st.origin = V3{R.pos.x + R.norm.x * kShadowBias, R.pos.y + R.norm.y * kShadowBias, R.pos.z + R.norm.z * kShadowBias};
    st.dir = nd;
    return true;
}

// Append the wave's surviving paths to this workgroup's segment of the out buffer (one atomic per wave).
__device__ __forceinline__ void append_paths(const FrameParams &P, bool alive, const PathState &st, uint32_t lane) {
    const unsigned long long ballot = __ballot(alive);
    const uint32_t n = (uint32_t)__popcll(ballot);
    if (!n) return;
    const uint32_t seg = blockIdx.x % kHitSegments;
    const int leader = __ffsll((long long)ballot) - 1;
    uint32_t base = 0;
    if ((int)lane == leader) base = atomicAdd(&P.seg_counts[seg * kSegStride], n);
    base = __shfl(base, leader, 64) + seg * P.hit_seg_cap;
    if (alive) {
        const uint32_t i = base + (uint32_t)__popcll(ballot & ((1ull << lane) - 1ull));
        P.path_out[i] = make_uint4(st.slot, __float_as_uint(st.origin.x), __float_as_uint(st.origin.y), __float_as_uint(st.origin.z));
        P.path_out[P.path_cap + i] = make_uint4(__float_as_uint(st.dir.x), __float_as_uint(st.dir.y), __float_as_uint(st.dir.z), st.rng);
        P.path_out[2u * P.path_cap + i] = make_uint4(__float_as_uint(st.thr.x), __float_as_uint(st.thr.y), __float_as_uint(st.thr.z), 0u);
    }
}

// Bounce 0: primary rays of sample P.sample. Sample 0 initialises the texel {light, id}; later samples add.
template <int MARCH, bool LDS_ROOTS, bool STATS>
__global__ void __launch_bounds__(256) path_primary_kernel(FrameParams P) {
    extern __shared__ uint32_t smem[];
    uint32_t *s_liquid = smem, *s_roots = smem + 24;
    unsigned long long *s_acc = reinterpret_cast<unsigned long long *>(smem + 8);
    if (STATS && threadIdx.x < 8) s_acc[threadIdx.x] = 0ull;
    stage_lds(P, s_roots, s_liquid, LDS_ROOTS);

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t t_local = blockIdx.x * 4u + (threadIdx.x >> 6);
    const bool live = t_local < P.tiles_local;
    if (!STATS && !live) return;
    MarchResult R;
    R.iters = 0; R.visits = 0; R.hit = false;
    if (live) {
        const uint32_t tile = shard_tile(t_local, P.shard_first, P.shard_run, P.shard_period);
        const uint32_t px = (tile % P.tiles_x) * 8u + (lane & 7u);
        const uint32_t py = (tile / P.tiles_x) * 8u + (lane >> 3);
        PathState st;
        st.slot = P.tile_major ? t_local * 64u + lane : py * P.width + px;
        create_ray(P, (int)px, (int)py, st.origin, st.dir);
        st.thr = V3{1.0f, 1.0f, 1.0f};
        // seed: path_tracer.wgsl:328 (y*W + x) + the per-sample stride and frame seed of SURVEY §8d
        st.rng = py * P.width + px + P.sample * (P.width * P.height) + P.seed * 0x9E3779B9u;
        const V3 o0 = st.origin, d0 = st.dir;
        V3 light{0.f, 0.f, 0.f};
        bool missed;
        const bool alive = path_segment<MARCH, LDS_ROOTS, STATS>(P, s_roots, s_liquid, st, R, light, missed) && !P.last_bounce;
        if (P.sample == 0u) {
            // the id word of the primary segment, composed as shade() does
            uint32_t id = R.voxel & VRT_ID_VOXEL_MASK;
            if (R.hit) id |= VRT_ID_HIT;
            if (R.norm.x != 0.0f) id |= VRT_ID_NX;
            if (R.norm.y != 0.0f) id |= VRT_ID_NY;
            if (R.norm.z != 0.0f) id |= VRT_ID_NZ;
            if (R.water_dist != 0.0f) id |= VRT_ID_WATER;
            P.out[st.slot] = make_uint4(__float_as_uint(light.x), __float_as_uint(light.y), __float_as_uint(light.z), id);
        } else if (missed) {
            uint4 t = P.out[st.slot];
            t.x = __float_as_uint(__uint_as_float(t.x) + light.x);
            t.y = __float_as_uint(__uint_as_float(t.y) + light.y);
            t.z = __float_as_uint(__uint_as_float(t.z) + light.z);
            P.out[st.slot] = t;
        }
        (void)o0; (void)d0;
        append_paths(P, alive, st, lane);
        if (STATS && P.steps && P.sample == 0u) P.steps[st.slot] = R.iters;
    }
    if (STATS) {
        block_add(s_acc, 0, R.iters);
        block_add(s_acc, 1, R.visits);
        block_add(s_acc, 2, (R.hit && P.sample == 0u) ? 1ull : 0ull);
        __syncthreads();
        if (threadIdx.x == 0) {
            atomicAdd(&P.counters[kCtrSteps], s_acc[0]);
            atomicAdd(&P.counters[kCtrVisits], s_acc[1]);
            atomicAdd(&P.counters[kCtrPrimarySteps], s_acc[0]);
            atomicAdd(&P.counters[kCtrPrimaryVisits], s_acc[1]);
            atomicAdd(&P.counters[kCtrHits], s_acc[2]);
        }
    }
}

// Bounce b >= 1: lane = one live path of the in buffer.
template <int MARCH, bool LDS_ROOTS, bool STATS>
__global__ void __launch_bounds__(256) path_bounce_kernel(FrameParams P) {
    extern __shared__ uint32_t smem[];
    uint32_t *s_liquid = smem, *s_roots = smem + 24;
    unsigned long long *s_acc = reinterpret_cast<unsigned long long *>(smem + 8);
    if (STATS && threadIdx.x < 8) s_acc[threadIdx.x] = 0ull;
    stage_lds(P, s_roots, s_liquid, LDS_ROOTS);

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t seg = blockIdx.x % kHitSegments, part = blockIdx.x / kHitSegments;
    const uint32_t count = P.seg_in[seg * kSegStride];
    const uint32_t j = part * blockDim.x + threadIdx.x;
    const bool active = j < count;
    if (!STATS && part * blockDim.x >= count) return;
    MarchResult R;
    R.iters = 0; R.visits = 0; R.hit = false;
    bool alive = false;
    PathState st;
    st.slot = 0; st.rng = 0;
    st.origin = st.dir = st.thr = V3{0.f, 0.f, 0.f};
    if (active) {
        const uint32_t i = seg * P.hit_seg_cap + j;
        const uint4 a = P.path_in[i], b = P.path_in[P.path_cap + i], c = P.path_in[2u * P.path_cap + i];
        st.slot = a.x;
        st.origin = V3{__uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w)};
        st.dir = V3{__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z)};
        st.rng = b.w;
        st.thr = V3{__uint_as_float(c.x), __uint_as_float(c.y), __uint_as_float(c.z)};
        V3 light{0.f, 0.f, 0.f};
        bool missed;
        alive = path_segment<MARCH, LDS_ROOTS, STATS>(P, s_roots, s_liquid, st, R, light, missed) && !P.last_bounce;
        if (missed) {
            uint4 t = P.out[st.slot];
            t.x = __float_as_uint(__uint_as_float(t.x) + light.x);
            t.y = __float_as_uint(__uint_as_float(t.y) + light.y);
            t.z = __float_as_uint(__uint_as_float(t.z) + light.z);
            P.out[st.slot] = t;
        }
        if (STATS && P.steps && P.sample == 0u) P.steps[st.slot] += R.iters << 16;
    }
    append_paths(P, alive, st, lane);
    if (STATS) {
        block_add(s_acc, 0, active ? R.iters : 0u);
        block_add(s_acc, 1, active ? R.visits : 0u);
        block_add(s_acc, 2, active ? 1ull : 0ull);
        __syncthreads();
        if (threadIdx.x == 0 && s_acc[2]) {
            atomicAdd(&P.counters[kCtrSteps], s_acc[0]);
            atomicAdd(&P.counters[kCtrVisits], s_acc[1]);
            atomicAdd(&P.counters[kCtrSecondary], s_acc[2]);
        }
    }
}

// rgb /= spp after the last sample
__global__ void path_finish_kernel(Texel *out, uint32_t n, float spp) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint4 t = out[i];
    t.x = __float_as_uint(__uint_as_float(t.x) / spp);
    t.y = __float_as_uint(__uint_as_float(t.y) / spp);
    t.z = __float_as_uint(__uint_as_float(t.z) / spp);
    out[i] = t;
}

static size_t lds_bytes_path(const FrameParams &P, bool lds_roots) { return (24u + (lds_roots ? P.n_roots : 0u)) * 4u; }

// The path trace marches with the grid march when the derived tables exist (P.grid), else with the ancestor-cache walk;
// `literal` (air flagged liquid, vrt_backend.hip) with the shader's text.
#define VRT_PATH_LAUNCH(kernel)                                                                                           \
    do {                                                                                                                  \
        const bool lds = (!P.grid || literal) && P.n_roots <= kLdsRootsMax;                                               \
        const size_t sh = lds_bytes_path(P, lds);                                                                         \
        if (literal) {                                                                                                    \
            if (lds) { if (stats) hipLaunchKernelGGL((kernel<1, true, true>), grid, block, sh, st, P); else hipLaunchKernelGGL((kernel<1, true, false>), grid, block, sh, st, P); } \
            else { if (stats) hipLaunchKernelGGL((kernel<1, false, true>), grid, block, sh, st, P); else hipLaunchKernelGGL((kernel<1, false, false>), grid, block, sh, st, P); } \
        } else if (P.grid) {                                                                                              \
            if (stats) hipLaunchKernelGGL((kernel<0, false, true>), grid, block, sh, st, P);                              \
            else hipLaunchKernelGGL((kernel<0, false, false>), grid, block, sh, st, P);                                   \
        } else if (lds) {                                                                                                 \
            if (stats) hipLaunchKernelGGL((kernel<2, true, true>), grid, block, sh, st, P);                               \
            else hipLaunchKernelGGL((kernel<2, true, false>), grid, block, sh, st, P);                                    \
        } else {                                                                                                          \
            if (stats) hipLaunchKernelGGL((kernel<2, false, true>), grid, block, sh, st, P);                              \
            else hipLaunchKernelGGL((kernel<2, false, false>), grid, block, sh, st, P);                                   \
        }                                                                                                                 \
    } while (0)

void launch_path_primary(const FrameParams &P, bool stats, bool literal, hipStream_t st) {
    if (P.tiles_local == 0) return;
    const dim3 grid((P.tiles_local + 3u) / 4u), block(256);
    VRT_PATH_LAUNCH(path_primary_kernel);
}

void launch_path_bounce(const FrameParams &P, bool stats, bool literal, hipStream_t st) {
    if (P.tiles_local == 0) return;
    const dim3 grid(kHitSegments * (P.hit_seg_cap / 256u)), block(256);
    VRT_PATH_LAUNCH(path_bounce_kernel);
}
#undef VRT_PATH_LAUNCH

void launch_path_finish(Texel *out, uint32_t n, uint32_t spp, hipStream_t st) {
    if (!n) return;
    hipLaunchKernelGGL(path_finish_kernel, dim3((n + 255u) / 256u), dim3(256), 0, st, out, n, (float)spp);
}

}  // namespace vrt
