// vrt_path_common.h — what the path-trace translation units share (vrt_path.hip: the primary launch and the lane = path
// bounce kernels; vrt_path_window.hip: the fused bounce launch over LDS-staged windows of march cells): the RNG of
// path_tracer.wgsl:56-72, log and cos spelled out in + - * / (host and device agree to the bit), and what follows a
// segment's march (path_tracer.wgsl:155-192).  DESIGN.md, path trace.
#pragma once

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
// BANDED: the one division without the general sequence's scaffolding (vrt_march.h: rcp_refined / div_refined).  Its operands are
// always inside what that needs: m + 1 in [1.70, 2.42]; m - 1 is zero — +0 / d is +0 from either form — or of magnitude >= 2^-24
// (m is a binary32 in (0.707, 1.4143]).  So there is no choice to make per wave: 8 instructions for 11, 25 issue cycles for 40.
// vrt_selftest_exact_math runs both forms over every mantissa.
template <bool BANDED>
__device__ __forceinline__ float vlog_t(float x) {
    const uint32_t b = __float_as_uint(x);
    int e = (int)(b >> 23) - 127;
    float m = __uint_as_float((b & 0x007FFFFFu) | 0x3F800000u);
    if (m > 1.41421354f) { m = m * 0.5f; e += 1; }
    const float num = m - 1.0f, den = m + 1.0f;
    const float s = BANDED ? div_refined(num, den, rcp_refined(den)) : num / den;
    const float z = s * s;
    const float p = z * (0.333333343f + z * (0.2f + z * (0.142857149f + z * (0.111111112f + z * 0.0909090936f))));
    return (float)e * 0.693147182f + (s + s * p) * 2.0f;
}
__device__ __forceinline__ float vlog(float x) { return vlog_t<true>(x); }

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

// rng_next_norm / rng_next_dir, path_tracer.wgsl:62-72: three normal deviates rho * cos(2 pi u1), rho = sqrt(-2 ln u2), drawn in
// the shader's order (u1, u2 of x, then of y, then of z), and their direction.  The three square roots take vrt_march.h's banded
// form — v_sqrt_f32 and one ulp either way — when every lane's three arguments are in its range (one ballot for the three): an
// argument is -0 when u2 rounds to 1 (a draw in 2^25), else >= 1.19e-7.  Same operations on the same values as the texts below.
__device__ __forceinline__ float rng_next_u2(uint32_t &state) {
    const float u2 = rng_next(state);
    return u2 < 1.0e-10f ? 1.0e-10f : u2;
}
__device__ __forceinline__ V3 rng_next_dir(uint32_t &state) {
    const float u1x = rng_next(state), tx = -2.0f * vlog(rng_next_u2(state));
    const float u1y = rng_next(state), ty = -2.0f * vlog(rng_next_u2(state));
    const float u1z = rng_next(state), tz = -2.0f * vlog(rng_next_u2(state));
    constexpr float kSqrtBandLo = 1.0e-28f;   // (sqrt_banded's range is [2^-96, inf); the arguments are <= 46.1)
    float rx, ry, rz;
    if (__ballot(!(min3_f32(tx, ty, tz) >= kSqrtBandLo)) == 0ull) {
        rx = sqrt_banded(tx); ry = sqrt_banded(ty); rz = sqrt_banded(tz);
    } else {
        rx = sqrtf(tx); ry = sqrtf(ty); rz = sqrtf(tz);
    }
    return normalize_wave(V3{rx * vcos2pi(u1x), ry * vcos2pi(u1y), rz * vcos2pi(u1z)});
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

// What follows a segment's march (the rest of the body of ray_color's loop, path_tracer.wgsl:155-192).  Returns true if the
// path goes on (st updated to the next segment); a miss puts the sky's light, weighted, into `light`.
__device__ __forceinline__ bool path_after_march(const FrameParams &P, PathState &st, const MarchResult &R, V3 &light, bool &missed) {
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
    const V3 sc = normalize_wave(V3{R.norm.x + rd.x, R.norm.y + rd.y, R.norm.z + rd.z});
    const float scatter = P.mats[min(R.voxel, 255u)].scatter;
    const V3 nd = normalize_wave(V3{vmix(spec.x, sc.x, scatter), vmix(spec.y, sc.y, scatter), vmix(spec.z, sc.z, scatter)});
    st.thr = V3{st.thr.x * mc.x, st.thr.y * mc.y, st.thr.z * mc.z};
    st.origin = V3{R.pos.x + R.norm.x * kShadowBias, R.pos.y + R.norm.y * kShadowBias, R.pos.z + R.norm.z * kShadowBias};
    st.dir = nd;
    return true;
}

// the nudge off a voxel face at the start of a march (ray_tracer.wgsl:204-207)
__device__ __forceinline__ V3 nudged(V3 pos, V3 dir) {
    if (pos.x - floorf(pos.x) < 0.001f || pos.y - floorf(pos.y) < 0.001f || pos.z - floorf(pos.z) < 0.001f) {
        pos.x += 0.001f * dir.x;
        pos.y += 0.001f * dir.y;
        pos.z += 0.001f * dir.z;
    }
    return pos;
}

// how many lanes of the mask are below this one (two instructions over the mask's halves; no per-lane mask to keep)
__device__ __forceinline__ uint32_t lanes_below(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
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

// A wave's pool of rays (the bounce launches below): K batches of 64 paths, 4 words each in the wave's own LDS — unit[3]
// (phase A -> the hand-out), then pos[3] + the packed end state (the march -> phase C): 4 KiB per wave
#ifndef VRT_POOL_K
#define VRT_POOL_K 4
#endif
constexpr uint32_t kPoolBatches = VRT_POOL_K;            // K
constexpr uint32_t kPoolEntries = kPoolBatches * 64u;
constexpr uint32_t kPoolWords = 4u * kPoolEntries;
constexpr uint32_t kPoolRefillAt = 16u;                  // idle lanes (of 64) that send the wave back to the pool

constexpr uint32_t kCellsPoolBytesPerWave = kPoolWords * 4u + kPoolEntries * 2u;   // the pool + a u16 order per entry

// what a launch of path_bounce_cells_kernel is given (one argument: the kernel reads it again for every segment)
struct CellsLaunch {
    FrameParams P;
    uint32_t refill_at;   // a wave takes rays from its pool when this many of its lanes are idle
    uint32_t segments;    // bounce segments in this launch: all that the frame's paths have left
};

}  // namespace vrt
