// experiments/vrt_path_experiments.hip — path-trace structures that were built, measured and NOT chosen (profiles/DECISIONS.md): the
// round-2 LDS pool kernel over cell grid + bricks with its straggler chain, and the path trace as one launch of persistent waves.
// They compile into tools/ab/libvrt_exp.so only (make -C voxelraytracing_amd/csrc experiments) and reach the backend through the hooks
// of vrt_exp.h; the tests that hold their frames to the shipping kernels' run against that build.
#include <cstdlib>

#include "../vrt_path_common.h"
#include "../vrt_exp.h"

namespace vrt {

// ------------------------------------------------------------------------------------------------
// The path trace as ONE launch (VRT_PATH_PERSISTENT=1; built and measured, not the default): persistent waves, lanes
// refilled in batches.
//
// The wavefront kernels above run one launch per bounce; a bounce launch marches rays whose directions were just
// randomised, so a wave's lanes finish after very different numbers of steps and the wave runs as long as its slowest
// lane (measured: 36-39 % lane utilisation).  Here a lane owns a *pixel* — all its samples, all their segments, in the
// order the oracle traces them, the radiance summed in a register — and a wave keeps marching with the lanes it has:
//   * the march loop is resumable (its state lives in the lane's registers across the phases below);
//   * when kRefillAt lanes have finished their segment, the wave leaves the march loop once, and those lanes do what
//     comes next together: shade, draw the bounce direction, or end the sample / the pixel and take the next pixel of
//     the wave's tile queue — then all lanes re-enter the march loop;
//   * tiles come from eight per-XCD ticket counters (one atomic per 64 pixels, any wave steals from any queue).
// Every path executes exactly the instructions the wavefront kernels execute for it, so the frame is bit-identical to
// theirs (tests) — scheduling is the only difference.  No path buffers in HBM, one launch per frame whatever spp is.
// Measured on C4 (1080p, 4 bounces): 10.8 Grays/s at the best batch size (40 waiting lanes; 6.2 at 8, 8.1 at 64) against
// 13.0 for the launch-per-bounce kernels: what a lane does between two segments — shading, six RNG draws with three
// logarithms and cosines, two normalisations, the nine divides and four square roots of a ray's set-up, ~800 VALU
// instructions — costs as much as marching the segment, and here it is issued for a batch of 16-40 lanes where the
// launch-per-bounce kernels issue it for 64.  What the batches win in the march they lose between the segments.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kRefillAt = 40;   // lanes that must be waiting before the wave leaves the march loop for them (measured optimum)

struct Segment {      // one ray being marched (march_grid's loop state, vrt_march.h)
    V3 pos, dir;
    float ux, uy, uz;
    uint32_t mxm, mym, mzm;
    int vx, vy, vz;
    float step, adx, ady, adz, dew, total_len, water_dist;
    uint32_t slow_bit, voxel, iter;
    bool careful;
};

// march_grid's prologue.  false: the ray starts outside the world (a miss before any lookup).
__device__ __forceinline__ bool segment_begin(const FrameParams &P, V3 origin, V3 dir, Segment &m) {
    m.dir = dir;
    m.careful = !(finite3(origin) && finite3(dir));
    V3 pos = origin;
    if (pos.x - floorf(pos.x) < 0.001f || pos.y - floorf(pos.y) < 0.001f || pos.z - floorf(pos.z) < 0.001f) {
        pos.x += 0.001f * dir.x;
        pos.y += 0.001f * dir.y;
        pos.z += 0.001f * dir.z;
    }
    m.pos = pos;
    m.water_dist = 0.0f;
    m.voxel = 0u;
    m.step = -1.0f;
    m.adx = m.ady = m.adz = 0.0f;
    m.dew = -1.0f;
    m.total_len = 0.0f;
    m.iter = 0u;
    m.slow_bit = m.careful ? 0x80000000u : 0u;
    const float world_max = 0.0f + (float)P.world.size;
    if ((pos.x <= 0.0f || pos.y <= 0.0f || pos.z <= 0.0f) || (pos.x >= world_max || pos.y >= world_max || pos.z >= world_max)) return false;
    const V3 unit = unit_steps(dir);
    m.ux = fabsf(unit.x); m.uy = fabsf(unit.y); m.uz = fabsf(unit.z);
    m.mxm = dir.x >= 0.0f ? ~0u : 0u; m.mym = dir.y >= 0.0f ? ~0u : 0u; m.mzm = dir.z >= 0.0f ? ~0u : 0u;
    m.vx = trunc2i(pos.x); m.vy = trunc2i(pos.y); m.vz = trunc2i(pos.z);
    return true;
}

// One trip of march_grid's loop.  true: the segment is over (solid hit, left the world, or kMaxSteps lookups).
__device__ __forceinline__ bool segment_trip(const FrameParams &P, const uint32_t *s_liquid, TableBuf gb, TableBuf bb, uint32_t row_bytes,
                                             uint32_t slab_bytes, Segment &m) {
    m.iter += 1u;
    uint32_t e = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(
        gb, mad_i24(m.vz >> 2, slab_bytes, mad_i24(m.vy >> 2, row_bytes, (uint32_t)m.vx & ~3u)), 0, 0);
    uint32_t lo = e;
    if ((e | m.slow_bit) - 1u >= 31u) {
        if (m.careful) {
            m.vx = trunc2i(m.pos.x);
            m.vy = trunc2i(m.pos.y);
            m.vz = trunc2i(m.pos.z);
            e = 0u;
            if (!(min3_nan_ignoring(m.pos.x, m.pos.y, m.pos.z) < 0.0f ||
                  max(max((uint32_t)m.vx, (uint32_t)m.vy), (uint32_t)m.vz) >= P.world.size))
                e = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(
                    gb, mad_i24(m.vz >> 2, slab_bytes, mad_i24(m.vy >> 2, row_bytes, (uint32_t)m.vx & ~3u)), 0, 0);
            lo = e;
        }
        if (e == 0u) return true;   // outside the world
        m.voxel = 0u;
        if ((int)e < 0) {
            const uint32_t u = ((uint32_t)m.vx & 3u) | (((uint32_t)m.vy & 3u) << 2) | (((uint32_t)m.vz & 3u) << 4);
            const uint32_t b = (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(bb, (e + u) << 1, 0, 0);
            lo = b & 1u;
            m.voxel = b >> 1;
        } else if (e > 31u) {
            lo = e & 31u;
            m.voxel = e >> 16;
        }
        if (m.voxel != 0u) {
            if (!is_liquid_ranged(P, s_liquid, m.voxel)) return true;   // solid: the hit
            if (m.dew == -1.0f) { m.dew = m.total_len; m.slow_bit = 0x80000000u; }
        } else if (m.dew != -1.0f) {
            m.water_dist += m.total_len - m.dew;
            m.dew = -1.0f;
            if (!m.careful) m.slow_bit = 0u;
        }
    }
    const float tx = (float)(int)(bfi(lo, m.mxm, (uint32_t)m.vx) - m.mxm) - m.pos.x;
    const float ty = (float)(int)(bfi(lo, m.mym, (uint32_t)m.vy) - m.mym) - m.pos.y;
    const float tz = (float)(int)(bfi(lo, m.mzm, (uint32_t)m.vz) - m.mzm) - m.pos.z;
    m.adx = abs_mul(tx, m.ux);
    m.ady = abs_mul(ty, m.uy);
    m.adz = abs_mul(tz, m.uz);
    m.step = min3_f32(m.adx, m.ady, m.adz);   // (p) of vrt_march.h: the float minimum when no lane's is zero or NaN
    if (__ballot(!(m.step > 0.0f)) != 0ull)
        m.step = __uint_as_float(min3_u32(__float_as_uint(m.adx) - 1u, __float_as_uint(m.ady) - 1u, __float_as_uint(m.adz) - 1u) + 1u);
    m.total_len += m.step;
    const float sp = m.step + 0.001f;
    m.pos.x += m.dir.x * (m.step == m.adx ? sp : m.step);
    m.pos.y += m.dir.y * (m.step == m.ady ? sp : m.step);
    m.pos.z += m.dir.z * (m.step == m.adz ? sp : m.step);
    m.vx = flr2i(m.pos.x);
    m.vy = flr2i(m.pos.y);
    m.vz = flr2i(m.pos.z);
    return m.iter >= kMaxSteps;
}

// march_grid's epilogue: the MarchResult of a finished segment (`started` false: segment_begin said miss).
__device__ __forceinline__ MarchResult segment_end(const FrameParams &P, const Segment &m, bool started) {
    MarchResult R;
    R.hit = false;
    R.pos = V3{0.f, 0.f, 0.f};
    R.norm = V3{0.f, 0.f, 0.f};
    R.water_dist = 0.0f;
    R.voxel = 0u;
    R.iters = 0u;
    R.visits = 0u;
    if (!started) return R;
    R.water_dist = m.water_dist;
    if (m.dew != -1.0f) R.water_dist += m.total_len - m.dew;
    if (min3_nan_ignoring(m.pos.x, m.pos.y, m.pos.z) < 0.0f ||
        max(max((uint32_t)trunc2i(m.pos.x), (uint32_t)trunc2i(m.pos.y)), (uint32_t)trunc2i(m.pos.z)) >= P.world.size)
        return R;
    R.hit = true;
    R.pos = m.pos;
    if (m.step != -1.0f)
        R.norm = V3{(m.step == m.adx ? 1.0f : 0.0f) * -vsign(m.dir.x), (m.step == m.ady ? 1.0f : 0.0f) * -vsign(m.dir.y),
                    (m.step == m.adz ? 1.0f : 0.0f) * -vsign(m.dir.z)};
    R.voxel = m.voxel;
    return R;
}

// ------------------------------------------------------------------------------------------------
// Bounce b >= 1 with a wave-local ray pool (the default for plain frames over the derived tables).
//
// A bounce launch marches rays whose directions were just drawn at random: most end within a few steps on the terrain
// next to their origin, a few graze it for a hundred.  With lane = path for the whole kernel a wave runs as long as its
// longest ray: 65 wave-steps for a mean of 12 per ray — 19 % lane utilisation inside the march loop, which is three
// quarters of the kernel's instructions (profiles/r02_path_pmc_summary.txt).  Here a wave owns K x 64 paths and works
// in three phases, the two arithmetic-heavy ones at full width:
//   A  K batches, lane = path: load the record, do the march's prologue (the nudge off a voxel face, the nine divides
//      and three square roots of the unit steps), park {pos, dir, unit} in the wave's own LDS pool;
//   B  march: a lane takes the next ray of the pool when it has none; when `refill_at` lanes have finished theirs (each
//      parks its end state in the pool entry it came from) the wave leaves the loop once and those lanes take the
//      next ones — a dozen LDS reads, not the ~800 instructions of shading and set-up that made the persistent kernel
//      below lose what it won;
//   C  K batches, lane = path again: the record once more, the march's end state from the pool, then exactly what
//      path_bounce_kernel does after its march (shade, draw the bounce, accumulate a miss, append the survivor).
// Every ray executes the instructions the other kernels execute for it: bit-identical frames (tests).  The pool is
// wave-local: no barriers between the phases, a wave that has nothing left leaves.
// ------------------------------------------------------------------------------------------------
#ifndef VRT_POOL_RAYS
#define VRT_POOL_RAYS 1
#endif
// rays a lane marches at once.  2 and 3 were measured (their loads in flight together, their arithmetic interleaved): 117 and
// more registers instead of 60, half the waves per SIMD, 12.4 and 10.4 Grays/s on C4 against 14.5 — profiles/r02_path_pool_sweeps.txt
constexpr uint32_t kPoolRays = VRT_POOL_RAYS;
constexpr uint32_t kPoolEjectAt = 16;
constexpr uint32_t kPoolEjected = 0xFFFFFFFFu;   // a pool entry's packed end state: the ray went on to the continuation launch


#ifdef VRT_EXP_POOLDBG
__device__ unsigned long long g_pool_dbg[16384 * 8];   // experiment: per wave {n, A, B, C cycles, wave-steps, refills, start, end (100 MHz)}
#define POOLDBG_T(x) const unsigned long long x = __builtin_amdgcn_s_memtime()
#else
#define POOLDBG_T(x)
#endif

template <bool CONT>
__global__ void __launch_bounds__(256) path_bounce_pool_kernel(FrameParams P, uint32_t refill_at, uint32_t eject_at) {
    extern __shared__ uint32_t smem[];
    uint32_t *s_liquid = smem;
    if (threadIdx.x < 8) s_liquid[threadIdx.x] = P.liquid[threadIdx.x];
    if (blockIdx.x == 0 && P.seg_clear) P.seg_clear[threadIdx.x * kSegStride] = 0u;   // kHitSegments == blockDim.x cursors
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    float *pool = reinterpret_cast<float *>(smem + 8) + wave * kPoolWords;
    constexpr uint32_t E = kPoolEntries;

    // this wave's paths: the workgroup takes up to 4 E records of its segment, split evenly over its waves
    const uint32_t seg = blockIdx.x % kHitSegments, part = blockIdx.x / kHitSegments;
    const uint32_t count = P.seg_in[seg * kSegStride];
    const uint32_t wg_begin = part * 4u * E;
    if (wg_begin >= count) return;
    const uint32_t n_wg = min(4u * E, count - wg_begin), per = (n_wg + 3u) / 4u;
    if (wave * per >= n_wg) return;
    const uint32_t n = min(per, n_wg - wave * per);   // <= E
    const uint32_t base = seg * P.in_seg_cap + wg_begin + wave * per;
    const float world_max = 0.0f + (float)P.world.size;
#ifdef VRT_EXP_POOLDBG
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    uint32_t dbg_steps = 0, dbg_refills = 0, dbg_bricks = 0, dbg_dry_steps = 0, dbg_dry_lanes = 0, dbg_wet_lanes = 0;
    unsigned long long dbg_lat_grid = 0, dbg_lat_brick = 0;
#endif
    POOLDBG_T(t0);

    // ---- A: the unit steps of every ray (nine divides, three square roots), full width ----
    for (uint32_t k = 0; k * 64u < n; k++) {
        const uint32_t i = k * 64u + lane;
        if (i < n) {
            const uint4 b = P.path_in[P.in_cap + base + i];
            const V3 dir{__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z)};
            const V3 unit = unit_steps(dir);
            pool[0u * E + i] = unit.x; pool[1u * E + i] = unit.y; pool[2u * E + i] = unit.z;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    POOLDBG_T(t1);
    // ---- B: the marches, lanes refilled from the pool.  march_grid's loop (vrt_march.h) made resumable: a ray that has
    // stopped keeps its end state in its registers until the wave's next refill parks it; water is not tracked (no
    // output of a path segment depends on it), so a liquid voxel is simply not a hit.
    // (Written for kPoolRays rays per lane — independent rays, their loads in flight together; one is what is built.) ----
    {
        const TableBuf gb = table_buffer(P.grid, P.grid_bytes), bb = table_buffer(P.bricks, P.brick_bytes);
        const uint32_t row_bytes = (P.grid_dim + 1u) * 4u, slab_bytes = (P.grid_dim + 1u) * row_bytes;
        const uint32_t wsize = P.world.size;
        struct Ray {
            V3 pos, dir;
            float ux, uy, uz, step, adx, ady, adz;
            uint32_t mxm, mym, mzm, voxel, iter, idx;
            int vx, vy, vz;
            bool marching, parked, not_finite;
        };
        Ray ray[kPoolRays];
#pragma unroll
        for (uint32_t r = 0; r < kPoolRays; r++) {
            Ray &q = ray[r];
            q.pos = q.dir = V3{0.f, 0.f, 0.f};
            q.ux = q.uy = q.uz = 0.f; q.step = -1.f; q.adx = q.ady = q.adz = 0.f;
            q.mxm = q.mym = q.mzm = 0u; q.voxel = 0u; q.iter = 0u; q.idx = 0u;
            q.vx = q.vy = q.vz = 0;
            q.marching = false; q.parked = true; q.not_finite = false;
        }
        uint32_t next = 0u;   // wave-uniform: the pool's first ray not handed out yet
        const unsigned long long below = (1ull << lane) - 1ull;
        auto cell_offset = [&](const Ray &q) __attribute__((always_inline)) {
            return mad_i24(q.vz >> 2, slab_bytes, mad_i24(q.vy >> 2, row_bytes, (uint32_t)q.vx & ~3u));
        };
        auto park = [&](Ray &q) __attribute__((always_inline)) {   // the end state segment_end needs: where, through which faces, on what
            uint32_t packed = q.voxel << 8;
            if (q.step != -1.0f) packed |= (q.step == q.adx ? 1u : 0u) | (q.step == q.ady ? 2u : 0u) | (q.step == q.adz ? 4u : 0u);
            pool[0u * E + q.idx] = q.pos.x; pool[1u * E + q.idx] = q.pos.y; pool[2u * E + q.idx] = q.pos.z;
            pool[3u * E + q.idx] = __uint_as_float(packed);
            q.parked = true;
        };
        auto take = [&](Ray &q, uint32_t idx) __attribute__((always_inline)) {
            // the rest of segment_begin: consecutive records for the lanes that refill, so the loads coalesce
            q.idx = idx;
            const uint32_t rec = base + idx;
            const uint4 a = P.path_in[rec], b = P.path_in[P.in_cap + rec];
            const V3 origin{__uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w)};
            q.dir = V3{__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z)};
            q.not_finite = !(finite3(origin) && finite3(q.dir));
            q.ux = pool[0u * E + idx]; q.uy = pool[1u * E + idx]; q.uz = pool[2u * E + idx];
            q.mxm = q.dir.x >= 0.0f ? ~0u : 0u; q.mym = q.dir.y >= 0.0f ? ~0u : 0u; q.mzm = q.dir.z >= 0.0f ? ~0u : 0u;
            q.voxel = 0u;
            q.marching = true;
            q.parked = false;
            uint4 d = make_uint4(0u, 0u, 0u, kContFresh);
            if (CONT) d = P.path_in[3u * P.in_cap + rec];
            if (CONT && !(d.w & kContFresh)) {
                // a ray the bounce launch handed on: where it stood, how many lookups it has had, and through which
                // faces its last step left (as a step / distances triple that compares the same way)
                q.pos = V3{__uint_as_float(d.x), __uint_as_float(d.y), __uint_as_float(d.z)};
                q.iter = d.w & 0xFFFFu;
                q.step = 1.0f;
                q.adx = (d.w & 0x10000u) ? 1.0f : 2.0f; q.ady = (d.w & 0x20000u) ? 1.0f : 2.0f; q.adz = (d.w & 0x40000u) ? 1.0f : 2.0f;
                q.vx = flr2i(q.pos.x); q.vy = flr2i(q.pos.y); q.vz = flr2i(q.pos.z);   // as take_step left them
            } else {
                q.pos = nudged(origin, q.dir);
                q.step = -1.0f; q.adx = q.ady = q.adz = 0.0f;
                q.iter = 0u;
                if ((q.pos.x <= 0.0f || q.pos.y <= 0.0f || q.pos.z <= 0.0f) || (q.pos.x >= world_max || q.pos.y >= world_max || q.pos.z >= world_max)) {
                    // starts outside the world: a miss before any lookup.  Its end state says so (a position outside)
                    q.marching = false;
                    q.pos = V3{-1.0f, -1.0f, -1.0f};
                }
                q.vx = trunc2i(q.pos.x); q.vy = trunc2i(q.pos.y); q.vz = trunc2i(q.pos.z);
            }
        };
        // the step to the leaf's exit face for a leaf of size lo + 1 (take_step of march_grid), then the lookup limit
        auto step_or_stop = [&](Ray &q, uint32_t lo, bool stop) __attribute__((always_inline)) {
#ifdef VRT_EXP_POOL_VALU   // tools/ab experiments only: the marginal cost of extra instructions per step
#pragma unroll
            for (int k_ = 0; k_ < VRT_EXP_POOL_VALU; k_++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(q.idx) : "v"(0u));
#endif
#ifdef VRT_EXP_POOL_SALU
#pragma unroll
            for (int k_ = 0; k_ < VRT_EXP_POOL_SALU; k_++) asm volatile("s_mov_b32 vcc_lo, 0" ::: "vcc");
#endif
            if (!stop) {
                const float tx = (float)(int)(bfi(lo, q.mxm, (uint32_t)q.vx) - q.mxm) - q.pos.x;
                const float ty = (float)(int)(bfi(lo, q.mym, (uint32_t)q.vy) - q.mym) - q.pos.y;
                const float tz = (float)(int)(bfi(lo, q.mzm, (uint32_t)q.vz) - q.mzm) - q.pos.z;
                q.adx = abs_mul(tx, q.ux);
                q.ady = abs_mul(ty, q.uy);
                q.adz = abs_mul(tz, q.uz);
                q.step = min3_f32(q.adx, q.ady, q.adz);   // (p) of vrt_march.h
                if (__ballot(!(q.step > 0.0f)) != 0ull)
                    q.step = __uint_as_float(min3_u32(__float_as_uint(q.adx) - 1u, __float_as_uint(q.ady) - 1u, __float_as_uint(q.adz) - 1u) + 1u);
                const float sp = q.step + 0.001f;
                q.pos.x += q.dir.x * (q.step == q.adx ? sp : q.step);
                q.pos.y += q.dir.y * (q.step == q.ady ? sp : q.step);
                q.pos.z += q.dir.z * (q.step == q.adz ? sp : q.step);
                q.vx = flr2i(q.pos.x);
                q.vy = flr2i(q.pos.y);
                q.vz = flr2i(q.pos.z);
                stop = q.iter >= kMaxSteps;
            }
            q.marching = !stop;
        };
        // (l) of vrt_march.h: the general step as march_grid has it for a wave with a ray that is not finite — the shader's
        // own bounds test, its lookup at i32(f32) coordinates
        auto careful_step = [&](Ray &q) __attribute__((always_inline)) {
            q.iter += 1u;
            q.vx = trunc2i(q.pos.x);
            q.vy = trunc2i(q.pos.y);
            q.vz = trunc2i(q.pos.z);
            uint32_t e = 0u;
            if (!(min3_nan_ignoring(q.pos.x, q.pos.y, q.pos.z) < 0.0f || max(max((uint32_t)q.vx, (uint32_t)q.vy), (uint32_t)q.vz) >= wsize))
                e = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(gb, cell_offset(q), 0, 0);
            uint32_t lo = e;
            bool stop = e == 0u;   // border, or past either end of the grid: the position is outside the world
            if (!stop) {
                q.voxel = 0u;
                if ((int)e < 0) {
                    const uint32_t u = ((uint32_t)q.vx & 3u) | (((uint32_t)q.vy & 3u) << 2) | (((uint32_t)q.vz & 3u) << 4);
                    const uint32_t b = (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(bb, (e + u) << 1, 0, 0);
                    lo = b & 1u;
                    q.voxel = b >> 1;
                } else if (e > 31u) {
                    lo = e & 31u;
                    q.voxel = e >> 16;
                }
                stop = q.voxel != 0u && !is_liquid_ranged(P, s_liquid, q.voxel);   // solid: the hit
            }
            step_or_stop(q, lo, stop);
        };
        for (;;) {
            // ---- refill: park what has stopped, hand out the pool's next rays (ray 0 of every lane first) ----
            uint32_t handed = 0u;
#pragma unroll
            for (uint32_t r = 0; r < kPoolRays; r++) {
                Ray &q = ray[r];
                if (!q.marching && !q.parked) park(q);
                const unsigned long long idle = __ballot(!q.marching);
                const uint32_t at = next + handed + (uint32_t)__popcll(idle & below);
                if (!q.marching && at < n) take(q, at);
                handed += (uint32_t)__popcll(idle);
            }
            next = min(n, next + handed);
#ifdef VRT_EXP_POOLDBG
            dbg_refills++;
#endif
            bool any = false, nf = false;
#pragma unroll
            for (uint32_t r = 0; r < kPoolRays; r++) { any |= ray[r].marching; nf |= ray[r].marching && ray[r].not_finite; }
            if (__ballot(any) == 0ull) {
                if (next >= n) break;   // the pool is empty and nobody marches
                continue;               // (every ray handed out started outside the world)
            }
            const bool careful = __ballot(nf) != 0ull;   // per refill round
            for (;;) {
                if (careful) {   // wave-uniform, rare
#pragma unroll
                    for (uint32_t r = 0; r < kPoolRays; r++)
                        if (ray[r].marching) careful_step(ray[r]);
                } else if (kPoolRays == 1u) {
                    // one ray per lane (what is built): everything inside one region of marching lanes — the scalar unit's
                    // time shows in this kernel (61 % of it, against 42 % of the VALU's: profiles/r02_path_pool_sweeps.txt),
                    // and every region of lanes is four or five scalar instructions
                    Ray &q = ray[0];
                    if (q.marching) {
                        const uint32_t e = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(gb, cell_offset(q), 0, 0);
#ifdef VRT_EXP_POOL_LOAD   // tools/ab experiments only: one more load per step, 1 = the line just read, 2 = a line nobody shares
                        {
                            const uint32_t off_ = VRT_EXP_POOL_LOAD == 1 ? cell_offset(q) : ((q.iter * 0x9E3779B9u + q.idx * 0x85EBCA6Bu + lane * 0xC2B2AE35u) % (P.grid_bytes / 4u)) * 4u;
                            const uint32_t x_ = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(gb, off_, 0, 0);
                            asm volatile("" :: "v"(x_));
                        }
#endif
                        q.iter += 1u;
                        const bool brick = (int)e < 0;
                        uint32_t b = 0u;
                        if (brick) {
                            const uint32_t u = ((uint32_t)q.vx & 3u) | (((uint32_t)q.vy & 3u) << 2) | (((uint32_t)q.vz & 3u) << 4);
                            b = (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(bb, (e + u) << 1, 0, 0);  // the shift drops bit 31
                        }
                        const uint32_t lo = brick ? (b & 1u) : (e & 31u);
                        q.voxel = brick ? (b >> 1) : (e >> 16);
                        const bool liquid = P.liquid_is_range ? (q.voxel - P.liquid_lo <= P.liquid_span) : is_liquid(s_liquid, q.voxel);
                        step_or_stop(q, lo, (e == 0u) | ((q.voxel != 0u) & !liquid));
                    }
                } else {
                    // the same decisions without a branch per case — a bounce wave has a ray in every case on nearly every
                    // step, and each divergent branch is half a dozen scalar instructions of exec-mask bookkeeping: an air
                    // leaf is the e <= 31 instance of "leaf" (lo = e & 31, voxel = e >> 16 = 0), the border (e = 0) is a
                    // leaf of nothing.  First all the grid loads, then all the brick loads, then the arithmetic
                    uint32_t e[kPoolRays], b[kPoolRays];
                    bool brick[kPoolRays], any_brick = false;
#pragma unroll
                    for (uint32_t r = 0; r < kPoolRays; r++) {
                        e[r] = 0u;
                        if (ray[r].marching) e[r] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(gb, cell_offset(ray[r]), 0, 0);
                    }
#pragma unroll
                    for (uint32_t r = 0; r < kPoolRays; r++) {
                        brick[r] = ray[r].marching && (int)e[r] < 0;
                        any_brick |= brick[r];
                        b[r] = 0u;
                    }
                    if (__ballot(any_brick) != 0ull) {   // wave-uniform: the second, dependent load only if some ray needs it
#pragma unroll
                        for (uint32_t r = 0; r < kPoolRays; r++) {
                            const Ray &q = ray[r];
                            const uint32_t u = ((uint32_t)q.vx & 3u) | (((uint32_t)q.vy & 3u) << 2) | (((uint32_t)q.vz & 3u) << 4);
                            if (brick[r]) b[r] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(bb, (e[r] + u) << 1, 0, 0);  // the shift drops bit 31
                        }
                    }
#pragma unroll
                    for (uint32_t r = 0; r < kPoolRays; r++) {
                        Ray &q = ray[r];
                        if (q.marching) {
                            q.iter += 1u;
                            const uint32_t lo = brick[r] ? (b[r] & 1u) : (e[r] & 31u);
                            q.voxel = brick[r] ? (b[r] >> 1) : (e[r] >> 16);
                            const bool liquid = P.liquid_is_range ? (q.voxel - P.liquid_lo <= P.liquid_span) : is_liquid(s_liquid, q.voxel);
                            step_or_stop(q, lo, (e[r] == 0u) | ((q.voxel != 0u) & !liquid));
                        }
                    }
                }
                uint32_t n_march = 0u;
#pragma unroll
                for (uint32_t r = 0; r < kPoolRays; r++) n_march += (uint32_t)__popcll(__ballot(ray[r].marching));
#ifdef VRT_EXP_POOLDBG
                dbg_steps++;
                if (next >= n) { dbg_dry_steps++; dbg_dry_lanes += n_march; } else dbg_wet_lanes += n_march;
#endif
                if (n_march == 0u || (next < n && 64u * kPoolRays - n_march >= refill_at)) break;
                if (!CONT && next >= n && n_march <= eject_at) break;   // the pool is dry and few rays are left: hand them on
            }
            if (!CONT && next >= n && P.cont_out) {
                // ---- the stragglers go to the straggler chain: their path record and where they stand ----
                uint32_t n_left = 0u;
#pragma unroll
                for (uint32_t r = 0; r < kPoolRays; r++) n_left += (uint32_t)__popcll(__ballot(ray[r].marching));
                if (n_left != 0u && n_left <= eject_at) {
                    uint32_t at0 = 0;
                    if (lane == 0) at0 = atomicAdd(&P.cont_counts[seg * kSegStride], n_left);
                    at0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)at0);
#pragma unroll
                    for (uint32_t r = 0; r < kPoolRays; r++) {
                        Ray &q = ray[r];
                        const unsigned long long ballot = __ballot(q.marching);
                        if (q.marching) {   // (a segment of the straggler records is as large as a segment of the paths)
                            const uint32_t rec = base + q.idx, o = seg * P.hit_seg_cap + at0 + (uint32_t)__popcll(ballot & below);
                            uint32_t w = q.iter;
                            if (q.step != -1.0f) w |= (q.step == q.adx ? 0x10000u : 0u) | (q.step == q.ady ? 0x20000u : 0u) | (q.step == q.adz ? 0x40000u : 0u);
                            P.cont_out[o] = P.path_in[rec];
                            P.cont_out[P.path_cap + o] = P.path_in[P.in_cap + rec];
                            P.cont_out[2u * P.path_cap + o] = P.path_in[2u * P.in_cap + rec];
                            P.cont_out[3u * P.path_cap + o] = make_uint4(__float_as_uint(q.pos.x), __float_as_uint(q.pos.y), __float_as_uint(q.pos.z), w);
                            pool[3u * E + q.idx] = __uint_as_float(kPoolEjected);
                            q.marching = false;
                            q.parked = true;
                        }
                        at0 += (uint32_t)__popcll(ballot);
                    }
                }
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    POOLDBG_T(t2);

    // ---- C: what follows the march, full width ----
    for (uint32_t k = 0; k * 64u < n; k++) {
        const uint32_t i = k * 64u + lane;
        bool alive = false;
        PathState st;
        st.slot = 0; st.rng = 0;
        st.origin = st.dir = st.thr = V3{0.f, 0.f, 0.f};
        if (i < n) {
            const uint32_t rec = base + i;
            const uint4 a = P.path_in[rec], b = P.path_in[P.in_cap + rec], c = P.path_in[2u * P.in_cap + rec];
            st.slot = a.x;
            st.origin = V3{__uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w)};
            st.dir = V3{__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z)};
            st.rng = b.w;
            st.thr = V3{__uint_as_float(c.x), __uint_as_float(c.y), __uint_as_float(c.z)};
            // segment_end on the parked end state (a path segment's water is nobody's business: DESIGN.md, path trace)
            const V3 pos{pool[0u * E + i], pool[1u * E + i], pool[2u * E + i]};
            const uint32_t packed = __float_as_uint(pool[3u * E + i]);
            if (packed != kPoolEjected) {
            MarchResult R;
            R.hit = false;
            R.pos = V3{0.f, 0.f, 0.f};
            R.norm = V3{0.f, 0.f, 0.f};
            R.water_dist = 0.0f;
            R.voxel = 0u;
            R.iters = 0u;
            R.visits = 0u;
            if (!(min3_nan_ignoring(pos.x, pos.y, pos.z) < 0.0f ||
                  max(max((uint32_t)trunc2i(pos.x), (uint32_t)trunc2i(pos.y)), (uint32_t)trunc2i(pos.z)) >= P.world.size)) {
                R.hit = true;
                R.pos = pos;
                R.norm = V3{((packed & 1u) ? 1.0f : 0.0f) * -vsign(st.dir.x), ((packed & 2u) ? 1.0f : 0.0f) * -vsign(st.dir.y),
                            ((packed & 4u) ? 1.0f : 0.0f) * -vsign(st.dir.z)};
                R.voxel = packed >> 8;
            }
            V3 light{0.f, 0.f, 0.f};
            bool missed;
            alive = path_after_march(P, st, R, light, missed) && !P.last_bounce;
            if (missed) {
                uint4 t = P.out[st.slot];
                t.x = __float_as_uint(__uint_as_float(t.x) + light.x);
                t.y = __float_as_uint(__uint_as_float(t.y) + light.y);
                t.z = __float_as_uint(__uint_as_float(t.z) + light.z);
                P.out[st.slot] = t;
            }
            }
        }
        if (!CONT) {
            append_paths(P, alive, st, lane);
        } else {
            // a straggler's next segment stays with the stragglers: the bounce launch that marches its generation is
            // already running (or done)
            const unsigned long long ballot = __ballot(alive);
            const uint32_t n_alive = (uint32_t)__popcll(ballot);
            if (n_alive != 0u) {   // (alive implies !P.last_bounce, and then the host gave a cont_out)
                const int leader = __ffsll((long long)ballot) - 1;
                uint32_t at = 0;
                if ((int)lane == leader) at = atomicAdd(&P.cont_counts[seg * kSegStride], n_alive);
                at = (uint32_t)__shfl((int)at, leader, 64) + (uint32_t)__popcll(ballot & ((1ull << lane) - 1ull));
                if (alive) {
                    const uint32_t o = seg * P.hit_seg_cap + at;
                    P.cont_out[o] = make_uint4(st.slot, __float_as_uint(st.origin.x), __float_as_uint(st.origin.y), __float_as_uint(st.origin.z));
                    P.cont_out[P.path_cap + o] = make_uint4(__float_as_uint(st.dir.x), __float_as_uint(st.dir.y), __float_as_uint(st.dir.z), st.rng);
                    P.cont_out[2u * P.path_cap + o] = make_uint4(__float_as_uint(st.thr.x), __float_as_uint(st.thr.y), __float_as_uint(st.thr.z), 0u);
                    P.cont_out[3u * P.path_cap + o] = make_uint4(0u, 0u, 0u, kContFresh);
                }
            }
        }
    }
#ifdef VRT_EXP_POOLDBG
    {
        POOLDBG_T(t3);
        const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0 && blockIdx.x * 4u + wave < 16384u) {
            unsigned long long *d = &g_pool_dbg[(blockIdx.x * 4u + wave) * 8u];
            d[0] = n | (CONT ? (1ull << 32) : 0ull); d[1] = t1 - t0; d[2] = t2 - t1; d[3] = t3 - t2; d[4] = dbg_steps | ((unsigned long long)dbg_dry_steps << 32); d[5] = dbg_wet_lanes | ((unsigned long long)dbg_dry_lanes << 32); d[6] = r0; d[7] = r1;
        }
    }
#endif
}

#ifdef VRT_EXP_POOLDBG
extern "C" void vrt_exp_pool_dbg(unsigned long long *out) {   // read (16384 x 8 words) and reset
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pool_dbg), sizeof(unsigned long long) * 16384 * 8);
    void *p = nullptr;
    (void)hipGetSymbolAddress(&p, HIP_SYMBOL(g_pool_dbg));
    (void)hipMemset(p, 0, sizeof(unsigned long long) * 16384 * 8);
}
#endif

__device__ __forceinline__ uint32_t path_xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u; }

// Next tile of the frame for this wave: its own XCD's queue first, then the others (tile i of queue x = x + 8 i).
// ~0u: the frame has no tiles left.  Wave-uniform.
__device__ __forceinline__ uint32_t next_tile(uint32_t *heads, uint32_t tiles, uint32_t &queue_round, uint32_t xcc, uint32_t lane) {
    while (queue_round < 8u) {
        const uint32_t q = (xcc + queue_round) & 7u;
        uint32_t i = 0;   // (called by all 64 lanes: lane 0 takes the ticket, everybody reads it)
        if (lane == 0) i = __hip_atomic_fetch_add(&heads[q * 16u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t t = q + 8u * (uint32_t)__builtin_amdgcn_readfirstlane((int)i);
        if (t < tiles) return t;
        queue_round += 1u;   // that queue is empty for good
    }
    return ~0u;
}

__global__ void __launch_bounds__(256) path_persistent_kernel(FrameParams P, uint32_t *heads, uint32_t refill_at) {
    extern __shared__ uint32_t smem[];
    uint32_t *s_liquid = smem;
    if (threadIdx.x < 8) s_liquid[threadIdx.x] = P.liquid[threadIdx.x];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t xcc = path_xcc_id();
    const TableBuf gb = table_buffer(P.grid, P.grid_bytes), bb = table_buffer(P.bricks, P.brick_bytes);
    const uint32_t row_bytes = (P.grid_dim + 1u) * 4u, slab_bytes = (P.grid_dim + 1u) * row_bytes;
    const uint32_t bounces = P.settings.max_ray_bounces;
    const float fspp = (float)P.spp;

    // the wave's queue: pixels q_pos .. 63 of tile q_tile are still to be handed out
    uint32_t queue_round = 0u;
    uint32_t q_tile = next_tile(heads, P.tiles_local, queue_round, xcc, lane), q_pos = 0u;

    enum : uint32_t { kMarching = 0u, kWaiting = 1u, kRetired = 2u };
    uint32_t state = kWaiting;   // kWaiting: the segment is over (or there is none yet) — something has to be decided
    bool have_pixel = false, started = false;
    Segment m;
    m.pos = m.dir = V3{0.f, 0.f, 0.f};
    m.ux = m.uy = m.uz = 0.f; m.mxm = m.mym = m.mzm = 0u; m.vx = m.vy = m.vz = 0;
    m.step = -1.f; m.adx = m.ady = m.adz = 0.f; m.dew = -1.f; m.total_len = 0.f; m.water_dist = 0.f;
    m.slow_bit = 0u; m.voxel = 0u; m.iter = 0u; m.careful = false;
    PathState st;
    st.slot = 0u; st.rng = 0u;
    st.origin = st.dir = st.thr = V3{0.f, 0.f, 0.f};
    V3 sum{0.f, 0.f, 0.f};
    uint32_t px = 0u, py = 0u, sample = 0u, bounce = 0u, id0 = 0u;

    for (;;) {
        // ---- 1. (waiting lanes) what comes after the segment this lane has just finished ----
        bool new_segment = false;
        if (state == kWaiting && have_pixel) {
            const MarchResult R = segment_end(P, m, started);
            if (sample == 0u && bounce == 0u) {   // the id word of the primary segment, composed as shade() does
                id0 = R.voxel & VRT_ID_VOXEL_MASK;
                if (R.hit) id0 |= VRT_ID_HIT;
                if (R.norm.x != 0.0f) id0 |= VRT_ID_NX;
                if (R.norm.y != 0.0f) id0 |= VRT_ID_NY;
                if (R.norm.z != 0.0f) id0 |= VRT_ID_NZ;
                if (R.water_dist != 0.0f) id0 |= VRT_ID_WATER;
            }
            bool path_over;
            if (!R.hit) {   // path_segment(): a miss adds the sky's light and ends the path
                const V3 sky = ray_sky(P, st.origin, st.dir);
                sum.x += sky.x * st.thr.x;
                sum.y += sky.y * st.thr.y;
                sum.z += sky.z * st.thr.z;
                path_over = true;
            } else {
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
                bounce += 1u;
                path_over = bounce >= bounces;
                new_segment = !path_over;
            }
            if (path_over) {
                sample += 1u;
                if (sample < P.spp) {   // the pixel's next sample: its primary ray again, a fresh RNG stream
                    create_ray(P, (int)px, (int)py, st.origin, st.dir);
                    st.thr = V3{1.0f, 1.0f, 1.0f};
                    st.rng = py * P.width + px + sample * (P.width * P.height) + P.seed * 0x9E3779B9u;
                    bounce = 0u;
                    new_segment = true;
                } else {
                    P.out[st.slot] = make_uint4(__float_as_uint(sum.x / fspp), __float_as_uint(sum.y / fspp), __float_as_uint(sum.z / fspp), id0);
                    have_pixel = false;
                }
            }
        }
        // ---- 2. (the whole wave) lanes without a pixel take the next ones of the wave's queue.  The queue's position is
        // wave state: it is advanced here, outside any divergent branch, with every lane of the wave present ----
        for (uint32_t round = 0; round < 4u; round++) {   // (a hand-out spans at most two tiles; the bound is a belt)
            const unsigned long long want = __ballot(state == kWaiting && !have_pixel);
            if (want == 0ull || q_tile == ~0u) break;
            if (q_pos == 64u) {
                q_tile = next_tile(heads, P.tiles_local, queue_round, xcc, lane);
                q_pos = 0u;
                continue;
            }
            const uint32_t rank = (uint32_t)__popcll(want & ((1ull << lane) - 1ull));
            const uint32_t avail = 64u - q_pos;
            if (state == kWaiting && !have_pixel && rank < avail) {
                const uint32_t i = q_pos + rank;
                const uint32_t tile = shard_tile(q_tile, P.shard_first, P.shard_run, P.shard_period);
                px = (tile % P.tiles_x) * 8u + (i & 7u);
                py = (tile / P.tiles_x) * 8u + (i >> 3);
                st.slot = P.tile_major ? q_tile * 64u + i : py * P.width + px;
                create_ray(P, (int)px, (int)py, st.origin, st.dir);
                st.thr = V3{1.0f, 1.0f, 1.0f};
                st.rng = py * P.width + px + P.seed * 0x9E3779B9u;   // sample 0
                sum = V3{0.f, 0.f, 0.f};
                sample = 0u;
                bounce = 0u;
                have_pixel = true;
                new_segment = true;
            }
            const uint32_t n = (uint32_t)__popcll(want);
            q_pos += n < avail ? n : avail;
        }
        // ---- 3. (waiting lanes) the next segment's set-up, or retirement ----
        if (state == kWaiting) {
            if (new_segment) {
                started = segment_begin(P, st.origin, st.dir, m);
                if (started) state = kMarching;   // (a ray that starts outside the world is over at once: stays waiting)
            } else if (!have_pixel && q_tile == ~0u) {
                state = kRetired;   // nothing left to hand out: this lane is done for the frame
            }
        }
        const unsigned long long marching = __ballot(state == kMarching);
        if (marching == 0ull) {
            if (__ballot(state == kWaiting) == 0ull) break;   // every lane retired: the wave is done
            continue;
        }
        // ---- march: all lanes that have a ray, until enough of them are waiting again ----
        for (;;) {
            if (state == kMarching && segment_trip(P, s_liquid, gb, bb, row_bytes, slab_bytes, m)) state = kWaiting;
            const uint32_t n_march = (uint32_t)__popcll(__ballot(state == kMarching));
            const uint32_t n_wait = (uint32_t)__popcll(__ballot(state == kWaiting));
            if (n_march == 0u || n_wait >= refill_at) break;
        }
    }
}

void launch_path_persistent(const FrameParams &P, uint32_t *heads, uint32_t n_cus, hipStream_t st) {
    if (P.tiles_local == 0) return;
    // as many waves as the chip holds at this kernel's register count, but no more than there are tiles
    const uint32_t waves = min(n_cus * 4u * 8u, P.tiles_local);
    static uint32_t refill_at = 0;
    if (!refill_at) {
        const char *e = getenv("VRT_PATH_REFILL");   // experiments: how many waiting lanes end a march phase
        refill_at = e ? (uint32_t)atoi(e) : kRefillAt;
        if (refill_at < 1u || refill_at > 64u) refill_at = kRefillAt;
    }
    hipLaunchKernelGGL(path_persistent_kernel, dim3((waves + 3u) / 4u), dim3(256), 8u * 4u, st, P, heads, refill_at);
}

// `continuations`: a launch of the straggler chain (P.path_in = four-plane records: rays a bounce launch handed on and the
// next segments of the chain's own survivors), which marches every ray to its end.
void launch_path_bounce_pool(const FrameParams &P, bool continuations, uint32_t refill_at, uint32_t eject_at, hipStream_t st) {
    if (P.tiles_local == 0) return;
    // refill_at: idle ray slots that send a wave back to its pool (0: the default); eject_at: marching rays at or below which
    // a wave whose pool is dry hands them to the straggler chain (only with a cont_out)
    const uint32_t refill = refill_at >= 1u && refill_at <= 64u * kPoolRays ? refill_at : kPoolRefillAt;
    const uint32_t eject = eject_at <= 64u ? eject_at : kPoolEjectAt;
    const uint32_t parts = (P.in_seg_cap + 4u * kPoolEntries - 1u) / (4u * kPoolEntries);
    const dim3 grid(kHitSegments * parts), block(256);
    const size_t sh = (8u + 4u * kPoolWords) * 4u;
    if (continuations) hipLaunchKernelGGL(path_bounce_pool_kernel<true>, grid, block, sh, st, P, refill, 0u);
    else hipLaunchKernelGGL(path_bounce_pool_kernel<false>, grid, block, sh, st, P, refill, P.cont_out ? eject : 0u);
}

}  // namespace vrt
