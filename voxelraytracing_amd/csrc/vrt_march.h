// vrt_march.h — device functions of the SVO march shared by the kernel translation units
// (vrt_kernels.hip: primary + shadow; vrt_path.hip: wavefront path trace).  See vrt_kernels.hip's header
// comment and DESIGN.md §Exact reductions for what is done and why it is exact.
#pragma once

#include "vrt_device.h"


namespace vrt {

// f32 -> i32 with the hardware's own NaN -> 0 (v_cvt_i32_f32; what WGSL's i32(f32) specifies), truncating.
// For x >= 0 this is floor(x); the march never uses it for a negative coordinate (it has left the world).
// Measured on MI355X: NaN -> 0, +-inf and |x| >= 2^31 saturate.  (v_cvt_flr_i32_f32 maps NaN to INT_MAX.)
__device__ __forceinline__ int trunc2i(float x) {
    int r;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

// ------------------------------------------------------------------------------------------------
// Correctly rounded f32 division and square root, without the general case's scaffolding.
//
// With -fhip-fp32-correctly-rounded-divide-sqrt `n / d` is 11 instructions: two v_div_scale (pre-scale operands whose
// exponents are extreme), v_rcp, two FMAs that refine the reciprocal, a multiply and three FMAs that refine the quotient,
// v_div_fmas (the last FMA, undoing the scaling) and v_div_fixup (zeros, infinities, NaNs); sqrtf is 17: scale a tiny
// argument by 2^32, v_sqrt, try one ulp up and down with two FMA residuals, scale back, pass 0 / inf / NaN through.
// ~38 and ~60 issue cycles — and a ray's set-up is 9 divides and 4 square roots (normalise, the three unit steps), twice
// per pixel with the shadow ray: two thirds of a frame's non-march work.
// When every operand's magnitude is in [2^-30, 2^30] none of the scaffolding does anything: v_div_scale returns its
// operand and clears VCC, v_div_fmas is a plain FMA, v_div_fixup returns its first operand, the square root's argument is
// neither tiny nor special.  What is left is below — the *same* instructions on the same values, so the same bits — and
// the reciprocal's refinement is shared by the divides that share a denominator.  The choice is made per wave (one
// ballot): a wave with a lane outside the band takes the compiler's general sequence for all its lanes.
// tests/test_gpu_exact_math.py compares both forms bit for bit over the band's whole exponent range.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kBandLo = 0x30800000u;              // 2^-30
constexpr uint32_t kBandSpan = 0x4E800000u - kBandLo;  // .. 2^30 (both ends' binades included: ample margin either side)
__device__ __forceinline__ bool in_band(float x) { return (__float_as_uint(x) & 0x7FFFFFFFu) - kBandLo <= kBandSpan; }   // NaN, inf, 0: no
__device__ __forceinline__ bool in_band3(V3 v) {
    const uint32_t a = (__float_as_uint(v.x) & 0x7FFFFFFFu) - kBandLo, b = (__float_as_uint(v.y) & 0x7FFFFFFFu) - kBandLo,
                   c = (__float_as_uint(v.z) & 0x7FFFFFFFu) - kBandLo;
    return max(max(a, b), c) <= kBandSpan;
}
// the reciprocal of d as the division's expansion refines it (v_rcp_f32, two FMAs)
__device__ __forceinline__ float rcp_refined(float d) {
    const float r = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
// n / d given r = rcp_refined(d): the expansion's multiply and four FMAs
__device__ __forceinline__ float div_refined(float n, float d, float r) {
    const float q0 = n * r;
    const float e0 = __builtin_fmaf(-d, q0, n);
    const float q1 = __builtin_fmaf(e0, r, q0);
    const float e1 = __builtin_fmaf(-d, q1, n);
    return __builtin_fmaf(e1, r, q1);
}
// sqrt(x) for 2^-96 <= x < inf: v_sqrt_f32, then one ulp down / up if the residual says so
__device__ __forceinline__ float sqrt_banded(float x) {
    const float s = __builtin_amdgcn_sqrtf(x);
    const float dn = __uint_as_float(__float_as_uint(s) - 1u), up = __uint_as_float(__float_as_uint(s) + 1u);
    const float rd = __builtin_fmaf(-dn, s, x), ru = __builtin_fmaf(-up, s, x);
    float r = (0.0f >= rd) ? dn : s;
    r = (0.0f < ru) ? up : r;
    return r;
}
// |unit step| per axis (ray_tracer.wgsl:209-213): sqrt(1 + (dir.b / dir.a)^2 + (dir.c / dir.a)^2)
__device__ __forceinline__ V3 unit_steps(V3 dir) {
    if (__ballot(!in_band3(dir)) == 0ull) {
        // (ratios within 2^+-60, their squares within 2^+-120: the sums are finite and >= 1)
        const float rx = rcp_refined(dir.x), ry = rcp_refined(dir.y), rz = rcp_refined(dir.z);
        const float yx = div_refined(dir.y, dir.x, rx), zx = div_refined(dir.z, dir.x, rx);
        const float xy = div_refined(dir.x, dir.y, ry), zy = div_refined(dir.z, dir.y, ry);
        const float xz = div_refined(dir.x, dir.z, rz), yz = div_refined(dir.y, dir.z, rz);
        return V3{fabsf(sqrt_banded(1.0f + yx * yx + zx * zx)), fabsf(sqrt_banded(1.0f + xy * xy + zy * zy)),
                  fabsf(sqrt_banded(1.0f + xz * xz + yz * yz))};
    }
    return V3{fabsf(sqrtf(1.0f + (dir.y / dir.x) * (dir.y / dir.x) + (dir.z / dir.x) * (dir.z / dir.x))),
              fabsf(sqrtf(1.0f + (dir.x / dir.y) * (dir.x / dir.y) + (dir.z / dir.y) * (dir.z / dir.y))),
              fabsf(sqrtf(1.0f + (dir.x / dir.z) * (dir.x / dir.z) + (dir.y / dir.z) * (dir.y / dir.z)))};
}
// vnormalize for a whole wave (same choice)
__device__ __forceinline__ V3 normalize_wave(V3 v) {
    if (__ballot(!in_band3(v)) == 0ull) {
        const float len = sqrt_banded(vdot(v, v));   // in [2^-30, 2^31]
        const float r = rcp_refined(len);
        return V3{div_refined(v.x, len, r), div_refined(v.y, len, r), div_refined(v.z, len, r)};
    }
    return vnormalize(v);
}

// Node words are read through a raw buffer descriptor over the pool: the hardware range check makes a read
// past the end return 0 (an air leaf) instead of faulting, for free — no per-load clamp, 32-bit offsets.
// (What a storage read past the end yields is implementation-defined in WGSL; the oracle reads 0 too.)
using NodeBuf = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ NodeBuf node_buffer(const FrameParams &P) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(P.nodes), 0, P.n_nodes * 2u, 0x00020000);
}
__device__ __forceinline__ uint32_t load_node(NodeBuf nb, uint32_t idx) {
    return (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(nb, idx * 2u, 0, 0);
}

__device__ __forceinline__ bool is_liquid(const uint32_t *s_liquid, uint32_t voxel) {
    // voxel_mats[voxel].is_liquid == 1 (ray_tracer.wgsl:226); ids >= 256 clamp to material 255.
    const uint32_t v = min(voxel, 255u);
    return (s_liquid[v >> 5] >> (v & 31u)) & 1u;
}

// The same question for the grid march: the liquid ids as a range when they form one (wave-uniform choice).
__device__ __forceinline__ bool is_liquid_ranged(const FrameParams &P, const uint32_t *s_liquid, uint32_t voxel) {
    if (P.liquid_is_range) return voxel - P.liquid_lo <= P.liquid_span;
    return is_liquid(s_liquid, voxel);
}

// ------------------------------------------------------------------------------------------------
// MARCH = 1: literal restatement of ray_world (ray_tracer.wgsl:182-316) with the descent restarted from
// the chunk root every step.  Kept as the A/B baseline for DESIGN.md's evidence table and as an
// in-backend cross-check of the fast march.
// ------------------------------------------------------------------------------------------------
struct Leaf {
    uint32_t node;   // the leaf's 16-bit word
    uint32_t depth;  // 0..5
    int bx, by, bz;  // world-local integer min corner of the leaf
};

template <bool LDS_ROOTS>
__device__ __forceinline__ Leaf find_leaf(const FrameParams &P, const uint32_t *s_roots, int vx, int vy, int vz) {
    const uint32_t S = P.world.size_in_chunks;
    uint32_t cidx = (uint32_t)(vx >> 5) + (uint32_t)(vy >> 5) * S + (uint32_t)(vz >> 5) * S * S;
    cidx = min(cidx, P.n_roots - 1u);
    const uint32_t root = LDS_ROOTS ? s_roots[cidx] : P.roots[cidx];
    const NodeBuf nb = node_buffer(P);
    uint32_t idx = 0, depth = 0;
    uint32_t node = load_node(nb, root);
    while ((node & 0x8000u) && depth < 5u) {
        const uint32_t sh = 4u - depth;
        const uint32_t child = ((uint32_t)(vx >> sh) & 1u) | (((uint32_t)(vy >> sh) & 1u) << 1) |
                               (((uint32_t)(vz >> sh) & 1u) << 2);
        idx = (node & 0x7FFFu) + child;
        node = load_node(nb, root + idx);
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

template <bool LDS_ROOTS>
__device__ __forceinline__ MarchResult march_literal(const FrameParams &P, const uint32_t *s_roots,
                                                     const uint32_t *s_liquid, V3 origin, V3 dir) {
    MarchResult R;
    R.hit = false;
    R.pos = V3{0.f, 0.f, 0.f};
    R.norm = V3{0.f, 0.f, 0.f};
    R.water_dist = 0.0f;
    R.voxel = 0u;
    R.iters = 0u;
    R.visits = 0u;
    R.trips = 0u;

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
    const bool stepped = iter > 1u || voxel == 0u || is_liquid(s_liquid, voxel);  // a step ran before the exit

    R.hit = true;
    R.pos = pos;
    // `norm` is only assigned inside the loop (:272): a hit on the very first lookup leaves it zero, which
    // differs from 0 * -sign(dir) when dir is NaN
    if (stepped) R.norm = V3{ex * -vsign(dir.x), ey * -vsign(dir.y), ez * -vsign(dir.z)};
    R.voxel = voxel;
    if (dist_entered_water != -1.0f) R.water_dist += total_len - dist_entered_water;
    return R;
}

// ------------------------------------------------------------------------------------------------
// MARCH = 2: the ancestor-cache march (reads the octree itself).  Same positions, same leaves, same results, bit for bit; the reductions
// used (each argued in DESIGN.md §Exact reductions):
//   (a) `pos >= center` at depth d  ==  bit (4-d) of floor(pos) & 31            (centres are integers)
//   (b) (pos-min)*imask + (max-pos)*mask  ==  mask ? max-pos : -(min-pos)         (x*0 adds a signed zero)
//   (c) the :247-270 branch tree  ==  minNum over the non-zero axis distances, ad.z if all are zero
//   (d) dir*(step+.001)*e + dir*step*(1-e)  ==  dir * (e ? step+.001 : step)       (the dropped term is a
//       zero with dir's sign)
//   (e) pos<0 || pos>=size  ==  (unsigned)floor(pos) >= size                      (size is an integer)
//   (f) the leaf of the new position is found by resuming the descent below the deepest split ancestor
//       it shares with the previous position instead of from the chunk root.
// ------------------------------------------------------------------------------------------------
template <bool LDS_ROOTS>
__device__ __forceinline__ MarchResult march_fast(const FrameParams &P, const uint32_t *s_roots,
                                                  const uint32_t *s_liquid, V3 origin, V3 dir) {
    MarchResult R;
    R.hit = false;
    R.pos = V3{0.f, 0.f, 0.f};
    R.norm = V3{0.f, 0.f, 0.f};
    R.water_dist = 0.0f;
    R.voxel = 0u;
    R.iters = 0u;
    R.visits = 0u;
    R.trips = 0u;

    const bool mx = dir.x >= 0.0f, my = dir.y >= 0.0f, mz = dir.z >= 0.0f;

    V3 pos = origin;
    if (pos.x - floorf(pos.x) < 0.001f || pos.y - floorf(pos.y) < 0.001f || pos.z - floorf(pos.z) < 0.001f) {
        pos.x += 0.001f * dir.x;
        pos.y += 0.001f * dir.y;
        pos.z += 0.001f * dir.z;
    }
    const float world_max = P.world_max;   // 0.0 + f32(world.size), the host's
    if ((pos.x <= 0.0f || pos.y <= 0.0f || pos.z <= 0.0f) || (pos.x >= world_max || pos.y >= world_max || pos.z >= world_max))
        return R;

    const V3 unit = unit_steps(dir);   // (:209-213)

    const uint32_t S = P.world.size_in_chunks;
    const uint32_t wsize = P.world.size;
    const NodeBuf nb = node_buffer(P);
    const float qnan = __builtin_nanf("");

    int vx = trunc2i(pos.x), vy = trunc2i(pos.y), vz = trunc2i(pos.z);  // pos > 0 here (or NaN -> 0)
    int pvx = ~vx, pvy = vy, pvz = vz;  // ~vx: the first lookup sees "another chunk"
    uint32_t root = 0u, node = 0u, depth = 0u;
    uint32_t b0 = 0u, b1 = 0u, b2 = 0u, b3 = 0u, b4 = 0u;  // child-block base of the split ancestor at depth 0..4
    const int mxm = mx ? -1 : 0, mym = my ? -1 : 0, mzm = mz ? -1 : 0;

    uint32_t voxel = 0u;
    bool ex = false, ey = false, ez = false;
    float dew = -1.0f;  // dist_entered_water
    float total_len = 0.0f;
    uint32_t iter = 0u;
    bool left_world = false;

    for (;;) {
        iter += 1u;
        // ---- find_node ----
        const uint32_t diff = (uint32_t)((vx ^ pvx) | (vy ^ pvy) | (vz ^ pvz));
        pvx = vx; pvy = vy; pvz = vz;
        bool resume = false;
        if (diff >= 32u) {  // another chunk (or the first lookup): start at that chunk's root
            uint32_t cidx = __umul24(__umul24((uint32_t)(vz >> 5), S) + (uint32_t)(vy >> 5), S) + (uint32_t)(vx >> 5);
            cidx = min(cidx, P.n_roots - 1u);
            root = LDS_ROOTS ? s_roots[cidx] : P.roots[cidx];
            node = load_node(nb, root);
            depth = 0u;
        } else {
            // k = number of leading local-coordinate bits shared with the previous position (5 if none differ)
            const uint32_t k = (uint32_t)__clz((int)diff) - 27u;
            if (k < depth) { depth = k; resume = true; }
        }
#define VRT_LEVEL(D, BD)                                                                               \
        if (depth == D && (resume || (node & 0x8000u))) {                                              \
            if (!resume) BD = node & 0x7FFFu;                                                          \
            resume = false;                                                                            \
            const uint32_t sel = (((uint32_t)vx >> (4 - D)) & 1u) | ((((uint32_t)vy >> (4 - D)) & 1u) << 1) | \
                                 ((((uint32_t)vz >> (4 - D)) & 1u) << 2);                               \
            node = load_node(nb, root + BD + sel);                                                     \
            depth = D + 1u;                                                                            \
        }
        VRT_LEVEL(0, b0)
        VRT_LEVEL(1, b1)
        VRT_LEVEL(2, b2)
        VRT_LEVEL(3, b3)
        VRT_LEVEL(4, b4)
#undef VRT_LEVEL
        voxel = node & 0x7FFFu;
        R.visits += depth + 1u;

        bool liquid = false;
        if (voxel != 0u) {
            liquid = is_liquid(s_liquid, voxel);
            if (!liquid) break;  // solid: the hit
        }
        if (liquid) {
            if (dew == -1.0f) dew = total_len;
        } else if (dew != -1.0f) {
            R.water_dist += total_len - dew;
            dew = -1.0f;
        }

        // ---- step to the leaf's exit face ----
        const int sz = 32 >> depth;
        const int m = ~(sz - 1);
        const float tx = (float)((vx & m) + (sz & mxm)) - pos.x;
        const float ty = (float)((vy & m) + (sz & mym)) - pos.y;
        const float tz = (float)((vz & m) + (sz & mzm)) - pos.z;
        const float adx = (mx ? tx : -tx) * unit.x;
        const float ady = (my ? ty : -ty) * unit.y;
        const float adz = (mz ? tz : -tz) * unit.z;
        const bool zx = adx == 0.0f, zy = ady == 0.0f, zz = adz == 0.0f;
        float step = __builtin_fminf(__builtin_fminf(zx ? qnan : adx, zy ? qnan : ady), zz ? qnan : adz);
        if (zx && zy && zz) step = adz;
        total_len += step;
        ex = step == adx;
        ey = step == ady;
        ez = step == adz;
        const float sp = step + 0.001f;
        pos.x += dir.x * (ex ? sp : step);
        pos.y += dir.y * (ey ? sp : step);
        pos.z += dir.z * (ez ? sp : step);

        // (e): pos < 0 on some axis (NaN-ignoring min: a NaN is not < 0), or floor(pos) >= size.  trunc == floor
        // for the non-negative coordinates that survive the first test.
        vx = trunc2i(pos.x);
        vy = trunc2i(pos.y);
        vz = trunc2i(pos.z);
        if (__builtin_fminf(__builtin_fminf(pos.x, pos.y), pos.z) < 0.0f ||
            max(max((uint32_t)vx, (uint32_t)vy), (uint32_t)vz) >= wsize) {
            if (dew != -1.0f) R.water_dist += total_len - dew;
            left_world = true;
            break;
        }
        if (iter >= kMaxSteps) break;
    }
    R.iters = iter;
    if (left_world) return R;
    const bool stepped = iter > 1u || voxel == 0u || is_liquid(s_liquid, voxel);  // a step ran before the exit

    R.hit = true;
    R.pos = pos;
    if (stepped)  // see march_literal: norm stays zero when no step was taken
        R.norm = V3{(ex ? 1.0f : 0.0f) * -vsign(dir.x), (ey ? 1.0f : 0.0f) * -vsign(dir.y), (ez ? 1.0f : 0.0f) * -vsign(dir.z)};
    R.voxel = voxel;
    if (dew != -1.0f) R.water_dist += total_len - dew;
    return R;
}

// ------------------------------------------------------------------------------------------------
// MARCH = 0 (default): the grid march.  Same positions, same leaves, same results as the two above, bit for
// bit; the leaf under a position comes from the derived cell grid / brick pool of vrt_accel.hip (at most two
// loads, no loop, nothing carried between steps) instead of a walk of the octree.  Reductions (b)-(d) as in
// march_fast, plus:
//   (b'') |t| * |unit|  ==  (mask ? t : -t) * unit up to the sign of a zero          (abs is a free operand modifier)
//   (c')  the :247-270 branch tree on bit patterns: min3_u32(bits - 1) + 1
//   (h)   leaf bounds from lo = leaf size - 1 (what the table stores): low = v & ~lo, high = (v | lo) + 1, chosen per
//         axis by one bit-field insert with the direction mask
//   (i)   the exit-axis flags and "left the world" are not carried through the loop: they are functions of the
//         last step's operands, which each lane still holds when it leaves
//   (l)   the table entry is the step's decision: an air leaf of the cell grid is its own lo (1..31), everything else
//         (other leaf, split cell, beyond the world) is one unsigned compare away — and "beyond the world" is an
//         entry like any other: the grid has a zero border and a load past its end returns 0, so the march has no
//         bounds test of its own (e).  That holds for rays whose origin and direction are finite — every
//         coordinate then stays within one voxel of the world, where the border is; a wave with a non-finite ray
//         (`careful`, wave-uniform) runs the shader's own test (e) on every step instead.
// ------------------------------------------------------------------------------------------------
using TableBuf = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ TableBuf table_buffer(const void *p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float min3_nan_ignoring(float a, float b, float c) {
    float r;  // IEEE mode: a quiet-NaN operand is ignored; every operand here is the result of an arithmetic op
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

__device__ __forceinline__ uint32_t mad_i24(int a, uint32_t uniform_b, uint32_t c) {
    uint32_t r;  // a * b + c on signed 24-bit factors, b wave-uniform
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(uniform_b), "v"(c));
    return r;
}

__device__ __forceinline__ uint32_t bfi(uint32_t mask, uint32_t a, uint32_t b) {
    // (mask & a) | (~mask & b) as gfx950's three-input boolean op, truth table 0xCA (index = mask << 2 | a << 1 | b):
    // v_bitop3_b32 issues at full rate (2.35 cycles per wave-instruction per SIMD, profiles/r04_valu_issue_rates.txt) where
    // v_bfi_b32 — like every other three-operand integer instruction — takes 4.25
    uint32_t r;
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xca" : "=v"(r) : "v"(mask), "v"(a), "v"(b));
    return r;
}

// ... the same with a small constant mask (an inline operand: no register, no move): the low bits of a over the rest of b.  Chains
// of these merge bit fields of several registers in one full-rate instruction each, where shift / and / or sequences end in the
// half-rate three-operand forms (v_and_or_b32, v_or3_b32, v_lshl_or_b32)
template <uint32_t MASK>
__device__ __forceinline__ uint32_t low_bits_of(uint32_t a, uint32_t b) {
    static_assert(MASK <= 64u, "an inline constant");
    uint32_t r;
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xca" : "=v"(r) : "n"(MASK), "v"(a), "v"(b));
    return r;
}
// a | (b & c) in one instruction (truth table 0xF8: index = a << 2 | b << 1 | c)
__device__ __forceinline__ uint32_t or_and(uint32_t a, uint32_t b, uint32_t c_mask_reg) {
    uint32_t r;
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xf8" : "=v"(r) : "v"(a), "v"(b), "v"(c_mask_reg));
    return r;
}

// |a| * b: the absolute value is an operand modifier, not an instruction
__device__ __forceinline__ float abs_mul(float a, float b) {
    float r;
    asm("v_mul_f32_e64 %0, |%1|, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// f32 -> i32 rounding towards -inf (v_cvt_flr_i32_f32): the integer part of every coordinate inside the world, -1 for
// the positions just beyond a low face (a step overshoots a face by 0.001 |dir|).
__device__ __forceinline__ int flr2i(float x) {
    int r;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
// v_min3_f32 as the instruction (fminf would add a canonicalising v_max per operand); a quiet NaN operand is ignored
__device__ __forceinline__ float min3_f32(float a, float b, float c) {
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t min3_u32(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

__device__ __forceinline__ bool finite3(V3 v) {
    // exponent field all ones <=> inf or NaN
    return ((__float_as_uint(v.x) & 0x7F800000u) != 0x7F800000u) && ((__float_as_uint(v.y) & 0x7F800000u) != 0x7F800000u) &&
           ((__float_as_uint(v.z) & 0x7F800000u) != 0x7F800000u);
}

// CAMERA: `origin` is the camera's (a primary ray): what the march asks of the origin alone is FrameParams.cam_origin_facts, the host's.
template <bool STATS, bool CAMERA = false>
__device__ __forceinline__ MarchResult march_grid(const FrameParams &P, const uint32_t *s_liquid, V3 origin, V3 dir) {
    MarchResult R;
    R.hit = false;
    R.pos = V3{0.f, 0.f, 0.f};
    R.norm = V3{0.f, 0.f, 0.f};
    R.water_dist = 0.0f;
    R.voxel = 0u;
    R.iters = 0u;
    R.visits = 0u;
    R.trips = 0u;

    const bool mx = dir.x >= 0.0f, my = dir.y >= 0.0f, mz = dir.z >= 0.0f;
    // (l): is any ray of this wave not finite?  (Wave-uniform; evaluated before lanes leave.)
    const bool careful = CAMERA ? ((P.cam_origin_facts & kCamNotFinite) != 0u || __ballot(!finite3(dir)) != 0ull)
                                : __ballot(!(finite3(origin) && finite3(dir))) != 0ull;

    V3 pos = origin;
    const bool nudge = CAMERA ? (P.cam_origin_facts & kCamOnPlane) != 0u
                              : (pos.x - floorf(pos.x) < 0.001f || pos.y - floorf(pos.y) < 0.001f || pos.z - floorf(pos.z) < 0.001f);
    if (nudge) {
        pos.x += 0.001f * dir.x;
        pos.y += 0.001f * dir.y;
        pos.z += 0.001f * dir.z;
    }
    const float world_max = P.world_max;   // 0.0 + f32(world.size), the host's
    if (CAMERA && !(P.cam_origin_facts & kCamOnPlane)) {   // (wave-uniform: the origin as it is)
        if (P.cam_origin_facts & kCamOutside) return R;
    } else if ((pos.x <= 0.0f || pos.y <= 0.0f || pos.z <= 0.0f) || (pos.x >= world_max || pos.y >= world_max || pos.z >= world_max))
        return R;

    const V3 unit = unit_steps(dir);   // (:209-213)
    float ux = fabsf(unit.x), uy = fabsf(unit.y), uz = fabsf(unit.z);
    asm("" : "+v"(ux), "+v"(uy), "+v"(uz));   // (held in registers: the compiler would otherwise redo the three |.| on every step)
    // (q) the exit plane as a float without a conversion and without an instruction for the "+ 1": the bit selector of the insert
    // below has its top nine bits set (an air leaf's entry of the cell grid has them as it is: vrt_device.h kAirLeaf), so the
    // insert takes those bits from the direction mask — which carries the exponent of 2^23 there — and its result is the bit
    // pattern of 2^23 + plane for 0 <= plane < 2^23; adding -2^23, or 1 - 2^23 when the exit plane is the high one ((v | lo) + 1),
    // is exact (integers below 2^24).  Two full-rate instructions for v_cvt_f32_i32's half-rate one and the integer increment.
    // (Until round 6 the "+ 1" and the exponent were an integer subtraction of their own: three instructions per step.)
    constexpr uint32_t kTwo23 = 0x4B000000u;
    uint32_t mxm = kTwo23 | (mx ? 0x007FFFFFu : 0u), mym = kTwo23 | (my ? 0x007FFFFFu : 0u), mzm = kTwo23 | (mz ? 0x007FFFFFu : 0u);
    float cx = mx ? -8388607.0f : -8388608.0f, cy = my ? -8388607.0f : -8388608.0f, cz = mz ? -8388607.0f : -8388608.0f;
    asm("" : "+v"(mxm), "+v"(mym), "+v"(mzm));   // (held in registers)

    const uint32_t wsize = P.world.size;
    const TableBuf gb = table_buffer(P.grid, P.grid_bytes), bb = table_buffer(P.bricks, P.brick_bytes);
    // (the bricks as an array of 16-bit entries — stride 2, indexed — for (s): the index is a merge, not an addition and a shift)
    const TableBuf bb16 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(P.bricks), 2, P.brick_bytes / 2u, 0x00020000);
    // (s): the liquids as the loop asks for them — voxel - liq_lo <= liq_span, never true of air (id 0).  A material table whose liquids
    // are not one range makes every voxel a candidate: the general step asks the table
    uint32_t liq_lo = P.liquid_lo, liq_span = P.liquid_span;
    if (!P.liquid_is_range) { liq_lo = 1u; liq_span = 0xFFFFFFFEu; }
    else if (liq_lo == 0u) { if (liq_span) { liq_lo = 1u; liq_span -= 1u; } else liq_lo = 0x80000000u; }
    // rows and slabs carry one border entry / row: [8S][8S + 1][8S + 1]; both strides < 2^23 (grid_dim <= 800)
    const uint32_t row_bytes = (P.grid_dim + 1u) * 4u, slab_bytes = (P.grid_dim + 1u) * row_bytes;

    int vx = trunc2i(pos.x), vy = trunc2i(pos.y), vz = trunc2i(pos.z);  // pos > 0 here (or NaN -> 0)
    // the voxel of the last lookup.  Only the slow path writes it; a lane on the fast path (a plain air leaf, not in water)
    // always finds it 0: whatever sent it through the slow path before — an air or liquid brick entry, leaving water —
    // left it 0 or left the lane in water, and a lane in water takes the slow path
    uint32_t voxel = 0u;
    // operands of the last step taken; a step is never negative, so -1 says "none yet" (the normal then stays zero, :272)
    float step = -1.0f, adx = 0.0f, ady = 0.0f, adz = 0.0f;
    float dew = -1.0f;  // dist_entered_water
    // (l) lanes that have more to do than step through an air leaf carry a bit that pushes every entry out of the air-leaf
    // range, so the fast path asks one question only: lanes inside water (dew != -1: bookkeeping even in air), and every
    // lane of a `careful` wave (the shader's own bounds test, the lookup repeated at its i32(f32) coordinates)
    // (as a threshold: the air leaves are the highest entries there are — kAirLeaf | lo — so "not an air leaf" is e < kAirLeaf,
    // and a lane that has more to do compares against a number above every entry: one compare asks the question)
    constexpr uint32_t kPlain = kAirLeaf, kSlow = 0xFFFFFFFFu;
    uint32_t slow_below = careful ? kSlow : kPlain;
    float total_len = 0.0f;
    uint32_t iter = 0u;      // wave-uniform trip count (loop control)
    uint32_t looked_up = 0u; // STATS: node lookups of this lane (:221) — one less than its trips if it left through the border

    // ---- the step to the leaf's exit face (:243-283), for a leaf of size lo + 1; `sel` = kAirLeaf | lo ----
    auto take_step = [&](uint32_t sel) __attribute__((always_inline)) {
        // (h) the exit plane: low = v & ~lo, high = (v | lo) + 1 — the insert gives v & ~lo or v | lo under the bits of 2^23
        const float tx = (__uint_as_float(bfi(sel, mxm, (uint32_t)vx)) + cx) - pos.x;
        const float ty = (__uint_as_float(bfi(sel, mym, (uint32_t)vy)) + cy) - pos.y;
        const float tz = (__uint_as_float(bfi(sel, mzm, (uint32_t)vz)) + cz) - pos.z;
        // (b'') |t| * |unit| has the bits of (mask ? t : -t) * unit, except that a zero is always +0 (never observed)
        adx = abs_mul(tx, ux);
        ady = abs_mul(ty, uy);
        adz = abs_mul(tz, uz);
        // (c') the minimum over the non-zero distances, on bit patterns: non-negative floats order like unsigned integers,
        // NaNs above every number, and bits - 1 sends +0 to the very top, so one unsigned min3 replaces the shader's
        // branch tree (:247-270): the smallest non-zero number; a NaN only if nothing else is non-zero; +0 if all are zero
        // (p) ... and when every lane's three distances are above zero (or NaN beside a number above zero) the plain float
        // minimum is that same number: one v_min3_f32 and one compare instead of three decrements, the unsigned minimum
        // and an increment.  A wave in which some lane's minimum is zero or NaN takes the bit-pattern form.
        step = min3_f32(adx, ady, adz);
        if (__ballot(!(step > 0.0f)) != 0ull)
            step = __uint_as_float(min3_u32(__float_as_uint(adx) - 1u, __float_as_uint(ady) - 1u, __float_as_uint(adz) - 1u) + 1u);
        total_len += step;
        const float sp = step + 0.001f;
        pos.x += dir.x * (step == adx ? sp : step);
        pos.y += dir.y * (step == ady ? sp : step);
        pos.z += dir.z * (step == adz ? sp : step);
        // the next lookup's coordinates: floor — just beyond a face it is -1 or `size`, which the grid's border answers
        // with 0 (a `careful` wave looks its coordinates up again, as the shader has them)
        vx = flr2i(pos.x);
        vy = flr2i(pos.y);
        vz = flr2i(pos.z);
    };
    auto lookup = [&]() __attribute__((always_inline)) {
        return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(
            gb, mad_i24(vz >> 2, slab_bytes, mad_i24(vy >> 2, row_bytes, (uint32_t)vx & ~3u)), 0, 0);
    };

    // Two loops (m): the inner one runs the steps in which NO lane of the wave has anything to decide — every lane in a
    // plain air leaf — and has only wave-uniform branches: no exec-mask bookkeeping, four scalar instructions per step.
    // A step in which some lane is in the slow path (brick, hit, water, leaving the world; 28 % of the wave's steps)
    // leaves it, runs once through the general body below — where a lane that is done breaks out, divergently — and
    // returns.  Same steps in the same order for every lane; `iter` counts them for the whole wave.
    for (;;) {
        uint32_t e;
        if constexpr (!STATS) {
            // (r) The inner loop as the instructions themselves.  What the compiler makes of the C++ below is the same vector
            // instructions, but it wraps every step in seven scalar instructions, four branches (three taken) and six s_nop /
            // s_waitcnt: its loop exits become 64-bit flag registers that are set, selected, and-ed with exec and copied to vcc, and
            // around the helpers above — one-instruction asm statements — it pads for hazards they do not have.  The scalar unit
            // is shared by a CU's four SIMDs, and the step's scalar work showed as its price (profiles/r06_step_asm.txt).  Here:
            // one scalar instruction (the trip count, as iter - kMaxSteps: its carry is the exit), three branches of which the
            // loop's own is the only one taken, one s_waitcnt; 34 vector instructions.
            // Same operations on the same operands in the same order of roundings as take_step() — which the general step below
            // still uses; tests/test_gpu_parity.py holds both against the oracle bit for bit.
            // Leaves with `e` = the entry some lane has to decide about (nothing of that step done), or after kMaxSteps lookups.
            uint32_t t0, t1, t2;
            unsigned long long sa, sb, sx, sd, sn, sw;
            uint32_t parked;
            uint32_t trips = __builtin_amdgcn_readfirstlane(iter) - kMaxSteps;   // (wave-uniform already: tells the compiler)
            asm volatile(
                "s_mov_b64 %[sd], 0\n\t"
                "s_setprio 1\n\t"                                           // (the waves in the loop before the ones that set up or shade: + 1 %)
                "v_cmp_ne_u32_e64 %[sw], %[kair], %[below]\n"                // lanes in water, a careful wave: the general step's
                ".Lvrt_step_%=:\n\t"
                "v_ashrrev_i32_e32 %[t1], 2, %[vy]\n\t"
                "v_and_b32_e32 %[t2], -4, %[vx]\n\t"
                "v_mad_i32_i24 %[t1], %[t1], %[row], %[t2]\n\t"
                "v_ashrrev_i32_e32 %[t0], 2, %[vz]\n\t"
                "v_mad_i32_i24 %[t0], %[t0], %[slab], %[t1]\n\t"
                "buffer_load_dword %[e], %[t0], %[desc], 0 offen\n\t"
                "s_waitcnt vmcnt(0)\n\t"
                "v_cmp_lt_u32_e32 vcc, %[e], %[below]\n\t"
                "s_cbranch_vccnz .Lvrt_split_%=\n"                         // some lane is not in a plain air leaf
                ".Lvrt_planes_%=:\n\t"
                "v_bitop3_b32 %[ax], %[e], %[mx], %[vx] bitop3:0xca\n\t"     // (h), (q): the exit planes under the bits of 2^23
                "v_bitop3_b32 %[ay], %[e], %[my], %[vy] bitop3:0xca\n\t"
                "v_bitop3_b32 %[az], %[e], %[mz], %[vz] bitop3:0xca\n\t"
                "v_add_f32_e32 %[ax], %[cx], %[ax]\n\t"
                "v_add_f32_e32 %[ay], %[cy], %[ay]\n\t"
                "v_add_f32_e32 %[az], %[cz], %[az]\n\t"
                "v_sub_f32_e32 %[ax], %[ax], %[px]\n\t"
                "v_sub_f32_e32 %[ay], %[ay], %[py]\n\t"
                "v_sub_f32_e32 %[az], %[az], %[pz]\n\t"
                "v_mul_f32_e64 %[ax], |%[ax]|, %[ux]\n\t"                    // (b'')
                "v_mul_f32_e64 %[ay], |%[ay]|, %[uy]\n\t"
                "v_mul_f32_e64 %[az], |%[az]|, %[uz]\n\t"
                "v_min3_f32 %[st], %[ax], %[ay], %[az]\n\t"                  // (p)
                "v_cmp_nlt_f32_e32 vcc, 0, %[st]\n\t"
                "s_cbranch_vccnz .Lvrt_zero_%=\n"                            // some lane's smallest distance is zero or NaN: (c')
                ".Lvrt_move_%=:\n\t"
                // (the half-rate instructions — compares, selects, conversions — each between two plain ones: next to a plain instruction
                // a half-rate one issues in a plain one's time, next to another half-rate one it takes its own: tools/valu_rates.hip k_mix_*)
                "v_add_f32_e32 %[t0], 0x3a83126f, %[st]\n\t"                // step + 0.001
                "v_cmp_eq_f32_e32 vcc, %[st], %[ax]\n\t"
                "v_add_f32_e32 %[tl], %[tl], %[st]\n\t"
                "v_cndmask_b32_e32 %[t1], %[st], %[t0], vcc\n\t"
                "v_mul_f32_e32 %[t1], %[dx], %[t1]\n\t"
                "v_cmp_eq_f32_e32 vcc, %[st], %[ay]\n\t"
                "v_add_f32_e32 %[px], %[px], %[t1]\n\t"
                "v_cndmask_b32_e32 %[t2], %[st], %[t0], vcc\n\t"
                "v_mul_f32_e32 %[t2], %[dy], %[t2]\n\t"
                "v_cmp_eq_f32_e32 vcc, %[st], %[az]\n\t"
                "v_add_f32_e32 %[py], %[py], %[t2]\n\t"
                "v_cndmask_b32_e32 %[t0], %[st], %[t0], vcc\n\t"
                "v_mul_f32_e32 %[t0], %[dz], %[t0]\n\t"
                "v_cvt_flr_i32_f32_e32 %[vx], %[px]\n\t"
                "v_add_f32_e32 %[pz], %[pz], %[t0]\n\t"
                "v_cvt_flr_i32_f32_e32 %[vy], %[py]\n\t"
                "v_cvt_flr_i32_f32_e32 %[vz], %[pz]\n\t"
                "s_add_u32 %[it], %[it], 1\n\t"                              // (carries when the count reaches kMaxSteps, :220)
                "s_cbranch_scc0 .Lvrt_step_%=\n\t"
                "s_branch .Lvrt_out_%=\n"
                ".Lvrt_zero_%=:\n\t"
                "v_add_u32_e32 %[t0], -1, %[ax]\n\t"
                "v_add_u32_e32 %[t1], -1, %[ay]\n\t"
                "v_add_u32_e32 %[t2], -1, %[az]\n\t"
                "v_min3_u32 %[t0], %[t0], %[t1], %[t2]\n\t"
                "v_add_u32_e32 %[st], 1, %[t0]\n\t"
                "s_branch .Lvrt_move_%=\n"
                // (s) Some lane is not in a plain air leaf (vcc).  Two steps in five of C2's are that for one reason only: plain lanes in
                // split cells, on voxels of their bricks that are air (a ray close to a surface walks leaves of one and two voxels) —
                // and for those the step is the one above with the brick's entry as the selector.  One step in eleven has a lane that
                // STOPS — at the border (entry 0), on a leaf or a brick's voxel that is solid: such a lane is parked (off the exec
                // mask, its registers as they are, the voxel it stopped on recorded) and the others march on; `parked` tells the
                // caller, which breaks out of its loop for it.  What is left for the general step: a lane in water or a careful wave
                // (the threshold is not kAirLeaf) and a voxel that is a liquid — the loop leaves with `e` as it was loaded.
                ".Lvrt_split_%=:\n\t"
                "s_cmp_lg_u64 %[sw], 0\n\t"                                 // (a lane in water, a careful wave: asked once, at the loop's entry)
                "s_cbranch_scc1 .Lvrt_out_%=\n\t"
                "s_and_saveexec_b64 %[sx], vcc\n\t"                         // the lanes with something to decide: border, leaf, split cell
                "v_cmp_gt_i32_e32 vcc, 0, %[e]\n\t"
                "s_and_saveexec_b64 %[sa], vcc\n\t"                         // the lanes in split cells
                "s_andn2_b64 %[sn], %[sa], exec\n\t"                        // the others: at the border, in a leaf that is not air
                "s_cbranch_execz .Lvrt_decide_%=\n\t"
                "v_lshlrev_b32_e32 %[t0], 2, %[vy]\n\t"                     // u = (x&3) | (y&3) << 2 | (z&3) << 4
                "v_lshlrev_b32_e32 %[t1], 4, %[vz]\n\t"
                "v_bitop3_b32 %[t0], 3, %[vx], %[t0] bitop3:0xca\n\t"
                "v_bitop3_b32 %[t0], 15, %[t0], %[t1] bitop3:0xca\n\t"      // (z's upper bits on top)
                "v_bitop3_b32 %[t0], %[kbrick], %[e], %[t0] bitop3:0xca\n\t"  // brick * 64 from the entry (bits 6..30), u below: the 16-bit entry's index
                "buffer_load_ushort %[t1], %[t0], %[bdesc], 0 idxen\n\t"
                "s_waitcnt vmcnt(0)\n\t"
                "v_cmp_lt_u32_e32 vcc, 1, %[t1]\n\t"                        // voxel << 1 | lo: a voxel that is not air
                "s_or_b64 %[sb], %[sn], vcc\n\t"
                "s_cbranch_scc1 .Lvrt_decide_%=\n\t"                        // someone stops, or meets a liquid
                "v_and_or_b32 %[e], %[t1], 1, %[kair]\n\t"                  // all air: the selector of the voxel's leaf (one voxel or two)
                "s_mov_b64 exec, %[sx]\n\t"
                "s_branch .Lvrt_planes_%=\n"
                ".Lvrt_decide_%=:\n\t"
                "v_lshrrev_b32_e32 %[t2], 1, %[t1]\n\t"                     // the voxel of a brick's entry ...
                "s_mov_b64 exec, %[sn]\n\t"
                "v_lshrrev_b32_e32 %[t2], 16, %[e]\n\t"                     // ... of a leaf (the border: 0)
                "s_mov_b64 exec, %[sa]\n\t"
                "v_subrev_u32_e32 %[t0], %[liqlo], %[t2]\n\t"
                "v_cmp_ge_u32_e32 vcc, %[liqspan], %[t0]\n\t"               // a liquid (air is none: liqlo >= 1): the general step's
                "s_cbranch_vccnz .Lvrt_leave_%=\n\t"
                "v_cmp_ne_u32_e32 vcc, 0, %[t2]\n\t"                        // solid
                "v_cmp_eq_u32_e64 %[sb], 0, %[e]\n\t"                       // the border
                "s_or_b64 %[sb], %[sb], vcc\n\t"                            // the lanes that stop here
                "s_or_b64 %[sd], %[sd], %[sb]\n\t"
                "s_andn2_b64 exec, %[sa], %[sb]\n\t"                        // split cells, air voxels:
                "v_and_or_b32 %[e], %[t1], 1, %[kair]\n\t"                  // the selector of the voxel's leaf
                "s_mov_b64 exec, %[sb]\n\t"
                "v_mov_b32_e32 %[vox], %[t2]\n\t"                           // what a parked lane stopped on
                "s_andn2_b64 exec, %[sx], %[sb]\n\t"                        // the lanes that march on
                "s_cbranch_scc1 .Lvrt_planes_%=\n\t"
                "s_branch .Lvrt_out_%=\n"                                   // none: every lane is parked
                ".Lvrt_leave_%=:\n\t"
                "s_mov_b64 exec, %[sx]\n"
                ".Lvrt_out_%=:\n\t"
                "s_setprio 0\n\t"
                "s_or_b64 exec, exec, %[sd]\n\t"
                "v_cndmask_b32_e64 %[parked], 0, 1, %[sd]"
                : [px] "+v"(pos.x), [py] "+v"(pos.y), [pz] "+v"(pos.z), [tl] "+v"(total_len), [vx] "+v"(vx), [vy] "+v"(vy), [vz] "+v"(vz),
                  [st] "+v"(step), [ax] "+v"(adx), [ay] "+v"(ady), [az] "+v"(adz), [e] "=&v"(e), [it] "+s"(trips),
                  [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [sa] "=&s"(sa), [sb] "=&s"(sb), [sx] "=&s"(sx), [sd] "=&s"(sd), [sn] "=&s"(sn), [sw] "=&s"(sw),
                  [vox] "+v"(voxel), [parked] "=&v"(parked)
                : [mx] "v"(mxm), [my] "v"(mym), [mz] "v"(mzm), [cx] "v"(cx), [cy] "v"(cy), [cz] "v"(cz), [ux] "v"(ux), [uy] "v"(uy), [uz] "v"(uz),
                  [dx] "v"(dir.x), [dy] "v"(dir.y), [dz] "v"(dir.z), [below] "v"(slow_below), [desc] "s"(gb), [row] "s"(row_bytes), [slab] "s"(slab_bytes),
                  [bdesc] "s"(bb16), [kair] "s"(kAirLeaf), [kbrick] "s"(0x7FFFFFC0u), [liqlo] "s"(liq_lo), [liqspan] "s"(liq_span)
                : "vcc", "scc", "memory");
            iter = trips + kMaxSteps;
            if (parked) break;              // stopped in the loop: border, or the solid voxel now in `voxel`
            if (iter >= kMaxSteps) break;   // (wave-uniform) at most kMaxSteps lookups (:220)
        } else {   // (the kernels that count: the same loop as the compiler lays it out, with the per-lane counters)
            bool exhausted = false;
            for (;;) {
                e = lookup();
                if (__ballot(e < slow_below) != 0ull) break;
                iter += 1u;
                if (STATS) { looked_up += 1u; R.visits += (uint32_t)__clz((int)((e & 31u) + 1u)) - 25u; }
                take_step(e);   // an air leaf of the cell grid: the entry is the selector
                if (iter >= kMaxSteps) { exhausted = true; break; }
            }
            if (exhausted) break;   // (wave-uniform) at most kMaxSteps lookups (:220)
        }
        // ---- the general step: find_node's answer may be a brick, a non-air leaf, the border ----
        iter += 1u;
        uint32_t lo = e;
        bool stop = false;
        if (e < slow_below) {
            if (careful) {  // wave-uniform: the shader's test (:285) on the shader's coordinates (i32(NaN) = 0), then its lookup
                vx = trunc2i(pos.x);
                vy = trunc2i(pos.y);
                vz = trunc2i(pos.z);
                e = 0u;
                if (!(min3_nan_ignoring(pos.x, pos.y, pos.z) < 0.0f || max(max((uint32_t)vx, (uint32_t)vy), (uint32_t)vz) >= wsize))
                    e = lookup();
                lo = e;
            }
            stop = e == 0u;  // border, or past either end of the grid: the position is outside the world
            if (!stop) {
                voxel = 0u;
                if (is_split_entry(e)) {
                    const uint32_t u = ((uint32_t)vx & 3u) | (((uint32_t)vy & 3u) << 2) | (((uint32_t)vz & 3u) << 4);
                    const uint32_t b = (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(bb, (e + u) << 1, 0, 0);  // the shift drops bit 31
                    lo = kAirLeaf | (b & 1u);
                    voxel = b >> 1;
                } else if (e < kAirLeaf) {   // a leaf that is not air (an air leaf is its own selector)
                    lo = kAirLeaf | (e & 31u);
                    voxel = e >> 16;
                }
                if (STATS) { looked_up += 1u; R.visits += (uint32_t)__clz((int)((lo & 31u) + 1u)) - 25u; }
                if (voxel != 0u) {
                    if (!is_liquid_ranged(P, s_liquid, voxel)) stop = true;                // solid: the hit
                    else if (dew == -1.0f) { dew = total_len; slow_below = kSlow; }    // liquid: water bookkeeping (:231-242)
                } else if (dew != -1.0f) {
                    R.water_dist += total_len - dew;
                    dew = -1.0f;
                    if (!careful) slow_below = kPlain;
                }
            }
        } else if (STATS) {
            looked_up += 1u;
            R.visits += (uint32_t)__clz((int)((lo & 31u) + 1u)) - 25u;  // depth + 1 node words on the reference's walk
        }
        if (stop) break;
        take_step(lo);
        if (iter >= kMaxSteps) break;  // every lane still here has looked up exactly `iter` nodes
    }
    // per-lane lookup counts are kept by the STATS kernels only (counters, step-count debug view)
    R.iters = STATS ? looked_up : 0u;
    R.trips = iter;
    if (dew != -1.0f) R.water_dist += total_len - dew;
    // (i): "left the world" (:285-290) from the position itself; a lane that left through a solid leaf or by exhaustion
    // holds a position that passes this test
    if (min3_nan_ignoring(pos.x, pos.y, pos.z) < 0.0f ||
        max(max((uint32_t)trunc2i(pos.x), (uint32_t)trunc2i(pos.y)), (uint32_t)trunc2i(pos.z)) >= wsize)
        return R;
    const bool stepped = step != -1.0f;

    R.hit = true;
    R.pos = pos;
    if (stepped)  // see march_literal: norm stays zero when no step was taken
        R.norm = V3{(step == adx ? 1.0f : 0.0f) * -vsign(dir.x), (step == ady ? 1.0f : 0.0f) * -vsign(dir.y),
                    (step == adz ? 1.0f : 0.0f) * -vsign(dir.z)};
    R.voxel = voxel;
    return R;
}

template <int MARCH, bool LDS_ROOTS, bool STATS = false, bool CAMERA = false>
__device__ __forceinline__ MarchResult march(const FrameParams &P, const uint32_t *s_roots, const uint32_t *s_liquid,
                                             V3 origin, V3 dir) {
    if (MARCH == 1) return march_literal<LDS_ROOTS>(P, s_roots, s_liquid, origin, dir);
    if (MARCH == 2) return march_fast<LDS_ROOTS>(P, s_roots, s_liquid, origin, dir);
    return march_grid<STATS, CAMERA>(P, s_liquid, origin, dir);
}

// ray_sky, ray_tracer.wgsl:144-157
// CAM_ORIGIN: `origin` is the camera (a primary ray), whose sun direction the host evaluated once (P.cam_sun_dir).
template <bool CAM_ORIGIN = false>
__device__ __forceinline__ V3 ray_sky(const FrameParams &P, V3 origin, V3 dir) {
    const float ground_to_sky_t = vsmoothstep(-0.01f, 0.0f, dir.y);
    // pow(t, 0.35), t in [0, 1], as exp2(0.35 * log2 t) on the hardware transcendentals (what a GPU's WGSL pow is);
    // nothing branches on it and it stays within 1e-6 of libm's (bar: 1e-4). ocml's powf is ~190 instructions.
    const float sky_gradient_t = __builtin_amdgcn_exp2f(0.35f * __builtin_amdgcn_logf(vsmoothstep(0.0f, 0.4f, dir.y)));
    const V3 grad{vmix(1.0f, P.settings.sky_color[0], sky_gradient_t), vmix(0.3f, P.settings.sky_color[1], sky_gradient_t),
                  vmix(0.0f, P.settings.sky_color[2], sky_gradient_t)};
    const V3 sun_dir = CAM_ORIGIN ? V3{P.cam_sun_dir[0], P.cam_sun_dir[1], P.cam_sun_dir[2]}
                                  : normalize_wave(V3{P.sun_local[0] - origin.x, P.sun_local[1] - origin.y, P.sun_local[2] - origin.z});
    const float sun = (vdot(dir, sun_dir) > (1.0f - 0.01f) && ground_to_sky_t >= 1.0f) ? 1.0f : 0.0f;
    const float add = sun * P.settings.sun_intensity;
    return V3{vmix(0.03f, grad.x, ground_to_sky_t) + add, vmix(0.03f, grad.y, ground_to_sky_t) + add,
              vmix(0.03f, grad.z, ground_to_sky_t) + add};
}

// create_ray_from_screen, ray_tracer.wgsl:159-171 (WGSL v*M = dot with the columns of M)
__device__ __forceinline__ void create_ray(const FrameParams &P, int sx, int sy, V3 &origin, V3 &dir) {
    const float x = P.ndc_x[sx];  // ((float)sx * 2.0f) / P.cam.proj_size[0] - 1.0f, tabulated per frame
    const float y = P.ndc_y[sy];  // ((float)sy * 2.0f) / P.cam.proj_size[1] - 1.0f
    const float c0 = x, c1 = -y, c2 = -1.0f, c3 = 1.0f;
    const float *ip = P.cam.inv_proj_mat;
    const float e0 = c0 * ip[0] + c1 * ip[1] + c2 * ip[2] + c3 * ip[3];
    const float e1 = c0 * ip[4] + c1 * ip[5] + c2 * ip[6] + c3 * ip[7];
    const float *iv = P.cam.inv_view_mat;
    const float e2 = -1.0f, e3 = 0.0f;
    const V3 w{e0 * iv[0] + e1 * iv[1] + e2 * iv[2] + e3 * iv[3], e0 * iv[4] + e1 * iv[5] + e2 * iv[6] + e3 * iv[7],
               e0 * iv[8] + e1 * iv[9] + e2 * iv[10] + e3 * iv[11]};
    dir = normalize_wave(w);
    origin = V3{P.cam_origin[0], P.cam_origin[1], P.cam_origin[2]};   // cam.pos - f32(world.min), the host's (vrt_frames.hip: cam_origin_facts)
}

// Face shading + ray_color + overlay, ray_tracer.wgsl:127-142, 296-314. Returns the id word.
// `color = vox*f32(hit) + sky*f32(!hit)` (:135): with finite settings the sky term of a hit is a zero and
// the material term of a miss is a zero, so only one side is evaluated (exact up to the sign of a zero).
// Primary rays only (the sky's origin is the camera).
template <bool EXACT_SKY>
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
    if (EXACT_SKY || !P.finite_settings) {
        const V3 sky = ray_sky<true>(P, origin, dir);
        const float fh = R.hit ? 1.0f : 0.0f, fm = R.hit ? 0.0f : 1.0f;
        color = V3{mc.x * fh + sky.x * fm, mc.y * fh + sky.y * fm, mc.z * fh + sky.z * fm};
    } else if (R.hit) {
        color = mc;
    } else {
        color = ray_sky<true>(P, origin, dir);
    }
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

// stats frames: per-block reduction in LDS, then one atomic per counter per block
__device__ __forceinline__ void block_add(unsigned long long *s_acc, int slot, unsigned long long v) {
    const unsigned long long s = wave_sum(v);
    if ((threadIdx.x & 63u) == 0 && s) atomicAdd(&s_acc[slot], s);
}

constexpr uint32_t kLdsRootsMax = 8192;  // entries (32 KiB): S <= 20

}  // namespace vrt
