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
#include <cstdlib>

#include "vrt_path_primary.h"

namespace vrt {

// One segment of a path: its march, then path_after_march.
template <int MARCH, bool LDS_ROOTS, bool STATS>
__device__ __forceinline__ bool path_segment(const FrameParams &P, const uint32_t *s_roots, const uint32_t *s_liquid,
                                             PathState &st, MarchResult &R, V3 &light, bool &missed) {
    R = march<MARCH, LDS_ROOTS, STATS>(P, s_roots, s_liquid, st.origin, st.dir);
    return path_after_march(P, st, R, light, missed);
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
    if (blockIdx.x == 0 && P.seg_clear) P.seg_clear[threadIdx.x * kSegStride] = 0u;
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


// ------------------------------------------------------------------------------------------------
// Bounce b >= 1 over the MARCH CELLS (vrt_accel.hip; the default for plain frames of worlds that have them): the pool
// kernel above with a march loop that has ONE load per step and no dependent load at all.
//
// What held the pool kernel at a quarter of its issue rate was the pair of dependent loads of a step in a split cell —
// cell entry, then the voxel's brick entry: 290 + 515 cycles of a 2 200-cycle wave-step, half of them L1 misses
// (profiles/r02_path_pmc_summary.txt) — on the critical path of every ray, and a launch lasts as long as its longest
// chain of steps.  A march cell answers both questions of a step from one 16-byte entry: the leaf's size (a leaf
// cell's lo, or the split cell's size-2 mask) and whether the voxel stops the ray (64 bits; liquids are transparent to
// a path segment, so they count as air — the tables are built with the material table's liquid set).  Which voxel it
// stopped on is only asked after the march, at full width, from the brick (phase C).
// Also new here: phase C takes the rays that hit and the rays that missed in separate batches (a wave executes both sides of that branch otherwise: ~510 + ~130 vector
// instructions per ray), and on the last bounce the rays that hit are not shaded at all (their bounce would be dropped).
// Every ray executes the arithmetic the other kernels execute for it: bit-identical frames (tests).
// ------------------------------------------------------------------------------------------------



// KB: batches of 64 rays in a wave's pool.  4 (256 rays, 32 waves per CU) everywhere but for small worlds with two frames in flight:
// there 5 (320 rays, 28 waves per CU) — fewer dry tails per ray, and the four wave slots a CU keeps free let the next frame's primary
// launch start beside this one: C4 + 2.8 % (one frame at a time - 2.2 %, C5 - 3.8 %: profiles/r05_pool_k5.txt)
template <bool DIRECT, uint32_t KB>
__attribute__((amdgpu_waves_per_eu(8, 8)))
__global__ void __launch_bounds__(256) path_bounce_cells_kernel(CellsLaunch L) {
    const FrameParams &K = L.P;
    extern __shared__ uint32_t smem[];
    uint32_t *s_liquid = smem;
    if (threadIdx.x < 8) s_liquid[threadIdx.x] = K.liquid[threadIdx.x];
    if (blockIdx.x == 0 && K.seg_clear) K.seg_clear[threadIdx.x * kSegStride] = 0u;   // kHitSegments == blockDim.x cursors
    __syncthreads();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr uint32_t E = KB * 64u, kWords = 4u * E;

    // this wave's paths: the workgroup takes up to 4 E records of its segment, split evenly over its waves
    const uint32_t seg = blockIdx.x % kHitSegments, part = blockIdx.x / kHitSegments;
    const uint32_t count = K.seg_in[seg * kSegStride];
    const uint32_t wg_begin = part * 4u * E;
    if (wg_begin >= count) return;
    const uint32_t n_wg = min(4u * E, count - wg_begin), per = (n_wg + 3u) / 4u;
    if (wave * per >= n_wg) return;
    uint32_t n = __builtin_amdgcn_readfirstlane(min(per, n_wg - wave * per));   // <= E
    const uint32_t base = __builtin_amdgcn_readfirstlane(seg * K.in_seg_cap + wg_begin + wave * per);
    // The wave keeps its paths for ALL the segments that are left (`segments` of them): the survivors of one segment are
    // compacted — by the wave alone, no cursor, no atomic — into the same index range of the other path buffer and are the
    // wave's pool for the next.  A launch per bounce ends when its slowest wave does (221 wave-steps against 81 on average),
    // three times per frame; here a wave that is done with one segment starts the next, and the launch waits for the slowest
    // SUM.  (The pools shrink — C4: 243, 198, 146 paths a wave — and phases A and C run their last batch partly empty.)
    // (as offsets from one pointer, so that the address of a record stays "scalar base + lane index")
    uint4 *const recs = K.path_in < K.path_out ? const_cast<uint4 *>(K.path_in) : K.path_out;
    uint32_t in_at = __builtin_amdgcn_readfirstlane((uint32_t)(K.path_in - recs));
    uint32_t out_at = __builtin_amdgcn_readfirstlane((uint32_t)(K.path_out - recs));
    for (uint32_t left = L.segments;; left--) {   // (left: segments still to do, this one included)
    // What does not change from one segment to the next is made anew for every one of them — the launch's parameters read
    // again from the kernel-argument segment (scalar loads), the lane's number and what follows from it computed again —
    // and not kept in registers around the whole loop: that is 15 VGPRs and 30 SGPRs too many for eight waves a SIMD.
    typedef const __attribute__((address_space(4))) CellsLaunch *KernArgs;
    KernArgs kargs = (KernArgs)__builtin_amdgcn_kernarg_segment_ptr();   // (L is the kernel's only argument)
    asm volatile("" : "+s"(kargs));
    const FrameParams &P = ((const CellsLaunch *)kargs)->P;
    const uint32_t refill_at = ((const CellsLaunch *)kargs)->refill_at;
    uint32_t none = 0u;
    asm volatile("" : "+s"(none));
    const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, none));
    float *pool = reinterpret_cast<float *>(smem + 8) + (wave + none) * kWords;
    uint16_t *order = reinterpret_cast<uint16_t *>(smem + 8 + 4u * kWords) + (wave + none) * E;
    const float world_max = P.world_max;   // 0.0 + f32(world.size), the host's
    const bool last_bounce = left == 1u;   // (the launch's last segment is the paths' last)

    // ---- A: the unit steps of every ray (nine divides, three square roots), full width ----
#pragma unroll
    for (uint32_t k = 0; k < KB; k++) {
        const uint32_t i = k * 64u + lane;
        if (i < n) {
            const uint4 b = recs[in_at + P.in_cap + base + i];
            {   // the origin plane is touched too: the hand-outs then find both planes of the record in L2 (the previous launch
                // wrote them, 69 MB ago; + 1 %.  Touching the next hand-outs' records at every refill instead: - 2 %)
                const uint4 a_ = recs[in_at + base + i];
                asm volatile("" :: "v"(a_.x));
            }
            const V3 unit = unit_steps(V3{__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z)});
            pool[0u * E + i] = unit.x; pool[1u * E + i] = unit.y; pool[2u * E + i] = unit.z;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- B: the marches, lanes refilled from the pool; a ray that has stopped keeps its end state in its registers until
    // the wave's next refill parks it.  Water is not tracked (no output of a path segment depends on it). ----
    {
        const TableBuf mb = table_buffer(P.mblk, P.mblk_bytes), db = table_buffer(P.cdir, P.cdir_bytes);
        const TableBuf bb = table_buffer(P.bricks, P.brick_bytes);
        // the chunk directory: [S][S+1][S+1] with a zero border; a direct world: [4S][4S+1][4S+1] lines of 128 bytes
        const uint32_t drow = (P.grid_dim / 8u + 1u) * 4u, dslab = (P.grid_dim / 8u + 1u) * drow;
        const uint32_t row128 = (P.grid_dim / 2u + 1u) * 128u, slab128 = (P.grid_dim / 2u + 1u) * row128;   // < 2^23: S <= 16
        const uint32_t wsize = P.world.size;
        V3 pos{0.f, 0.f, 0.f}, dir{0.f, 0.f, 0.f};
        float ux = 0.f, uy = 0.f, uz = 0.f, step = -1.f, adx = 0.f, ady = 0.f, adz = 0.f;
        // the direction masks and float constants of (q) of vrt_march.h: the bits of 2^23 over all ones or none; -2^23, + 1 towards +
        uint32_t mxm = 0u, mym = 0u, mzm = 0u, ref = 0u, iter = 0u, idx = 0u;
        float cx = 0.f, cy = 0.f, cz = 0.f;
        int vx = 0, vy = 0, vz = 0;
        // the chunk the ray is in — its coordinates as one number — and where that chunk's block of march cells begins
        constexpr uint32_t kNoChunk = 0x7FFFFFFFu;
        uint32_t ckey = kNoChunk, cblock = 0u;
        bool marching = false, parked = true, not_finite = false;
        uint32_t next = 0u;   // wave-uniform: the pool's first ray not handed out yet

        // the end state phase C needs: where, through which faces, on what (bit 3: `what` is the split cell's brick)
        auto park = [&]() __attribute__((always_inline)) {
            uint32_t packed = (int)ref < 0 ? (8u | ((ref & 0x7FFFFFC0u) >> 2)) : ((ref >> 16) << 4);
            if (step != -1.0f) packed |= (step == adx ? 1u : 0u) | (step == ady ? 2u : 0u) | (step == adz ? 4u : 0u);
            pool[0u * E + idx] = pos.x; pool[1u * E + idx] = pos.y; pool[2u * E + idx] = pos.z;
            pool[3u * E + idx] = __uint_as_float(packed);
            parked = true;
        };
        auto take = [&](uint32_t at) __attribute__((always_inline)) {
            idx = at;
            const uint32_t rec = base + idx;
            const uint4 a = recs[in_at + rec], b = recs[in_at + P.in_cap + rec];
            const V3 origin{__uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w)};
            dir = V3{__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z)};
            not_finite = !(finite3(origin) && finite3(dir));
            ux = pool[0u * E + idx]; uy = pool[1u * E + idx]; uz = pool[2u * E + idx];
            constexpr uint32_t kTwo23 = 0x4B000000u;
            mxm = kTwo23 | (dir.x >= 0.0f ? 0x007FFFFFu : 0u); mym = kTwo23 | (dir.y >= 0.0f ? 0x007FFFFFu : 0u); mzm = kTwo23 | (dir.z >= 0.0f ? 0x007FFFFFu : 0u);
            cx = dir.x >= 0.0f ? -8388607.0f : -8388608.0f; cy = dir.y >= 0.0f ? -8388607.0f : -8388608.0f; cz = dir.z >= 0.0f ? -8388607.0f : -8388608.0f;
            ref = 0u;
            ckey = kNoChunk;
            marching = true;
            parked = false;
            pos = nudged(origin, dir);
            step = -1.0f; adx = ady = adz = 0.0f;
            iter = 0u;
            if ((pos.x <= 0.0f || pos.y <= 0.0f || pos.z <= 0.0f) || (pos.x >= world_max || pos.y >= world_max || pos.z >= world_max)) {
                // starts outside the world: a miss before any lookup.  Its end state says so (a position outside), and is
                // parked right here: the wave may find nothing left to march and never come back to the refill
                marching = false;
                pos = V3{-1.0f, -1.0f, -1.0f};
                park();
            }
            vx = trunc2i(pos.x); vy = trunc2i(pos.y); vz = trunc2i(pos.z);
        };
        // the step to the leaf's exit face for a leaf of size lo + 1 (take_step of march_grid)
        auto take_step = [&](uint32_t lo) __attribute__((always_inline)) {
            // (h), (q) of vrt_march.h: the exit plane (v | lo) + 1 or v & ~lo, as a float without a conversion
            const uint32_t sel = kAirLeaf | lo;
            const float tx = (__uint_as_float(bfi(sel, mxm, (uint32_t)vx)) + cx) - pos.x;
            const float ty = (__uint_as_float(bfi(sel, mym, (uint32_t)vy)) + cy) - pos.y;
            const float tz = (__uint_as_float(bfi(sel, mzm, (uint32_t)vz)) + cz) - pos.z;
            adx = abs_mul(tx, ux);
            ady = abs_mul(ty, uy);
            adz = abs_mul(tz, uz);
            step = min3_f32(adx, ady, adz);   // (p) of vrt_march.h
            if (__ballot(!(step > 0.0f)) != 0ull)
                step = __uint_as_float(min3_u32(__float_as_uint(adx) - 1u, __float_as_uint(ady) - 1u, __float_as_uint(adz) - 1u) + 1u);
            const float sp = step + 0.001f;
            pos.x += dir.x * (step == adx ? sp : step);
            pos.y += dir.y * (step == ady ? sp : step);
            pos.z += dir.z * (step == adz ? sp : step);
            vx = flr2i(pos.x);
            vy = flr2i(pos.y);
            vz = flr2i(pos.z);
        };
        // (l) of vrt_march.h: the general step as march_grid has it for a wave with a ray that is not finite — the shader's
        // own bounds test, its lookup at i32(f32) coordinates; over the cell grid and the bricks (rare: NaN cameras)
        auto careful_step = [&]() __attribute__((always_inline)) {
            iter += 1u;
            vx = trunc2i(pos.x);
            vy = trunc2i(pos.y);
            vz = trunc2i(pos.z);
            uint32_t e = 0u;   // the cell's entry of the cell grid: the march cell's first word
            if (!(min3_nan_ignoring(pos.x, pos.y, pos.z) < 0.0f || max(max((uint32_t)vx, (uint32_t)vy), (uint32_t)vz) >= wsize)) {
                const uint32_t sub = ((((uint32_t)vz >> 2) & 1u) << 2) | ((((uint32_t)vy >> 2) & 1u) << 1) | (((uint32_t)vx >> 2) & 1u);
                uint32_t off;
                if (DIRECT) {
                    off = mad_i24(vz >> 3, slab128, mad_i24(vy >> 3, row128, ((uint32_t)(vx >> 3) << 7) + (sub << 4)));
                } else {   // (inside the world: the chunk has an entry in the directory)
                    const uint32_t block = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(db, mad_i24(vz >> 5, dslab, mad_i24(vy >> 5, drow, (uint32_t)(vx >> 5) << 2)), 0, 0) << 13;
                    const uint32_t line = ((((((uint32_t)vz >> 3) & 3u) << 2) | (((uint32_t)vy >> 3) & 3u)) << 2) | (((uint32_t)vx >> 3) & 3u);
                    off = block + (((line << 3) | sub) << 4);
                }
                e = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(mb, off, 0, 0);
            }
            uint32_t lo = e, voxel = 0u;
            bool stop = e == 0u;   // border, or past either end of the grid: the position is outside the world
            if (!stop) {
                if ((int)e < 0) {
                    const uint32_t u = ((uint32_t)vx & 3u) | (((uint32_t)vy & 3u) << 2) | (((uint32_t)vz & 3u) << 4);
                    const uint32_t b = (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(bb, (e + u) << 1, 0, 0);
                    lo = b & 1u;
                    voxel = b >> 1;
                } else if (e > 31u) {
                    lo = e & 31u;
                    voxel = e >> 16;
                }
                stop = voxel != 0u && !is_liquid_ranged(P, s_liquid, voxel);   // solid: the hit
            }
            ref = voxel << 16;   // (the voxel itself, as a leaf cell's entry has it)
            if (!stop) {
                take_step(lo);
                stop = iter >= kMaxSteps;
            }
            marching = !stop;
        };
        for (;;) {
            // ---- refill: park what has stopped, hand out the pool's next rays ----
            if (!marching && !parked) park();
            {
                const unsigned long long idle = __ballot(!marching);
                const uint32_t at = next + lanes_below(idle);
                if (!marching && at < n) take(at);
                next = min(n, next + (uint32_t)__popcll(idle));
            }
            if (__ballot(marching) == 0ull) {
                if (next >= n) break;   // the pool is empty and nobody marches (every ray is parked: take() parks the ones that start outside)
                continue;
            }
            // (two loops, chosen per refill round: one loop with both bodies costs a register move per loop-carried value and
            // body on every trip — 22 of them, a quarter of the step)
            if (__ballot(marching && not_finite) != 0ull) {   // wave-uniform, rare
                for (;;) {
                    if (marching) careful_step();
                    const uint32_t n_march = (uint32_t)__popcll(__ballot(marching));
                    if (n_march == 0u || (next < n && 64u - n_march >= refill_at)) break;
                }
                continue;
            }
            {
                // (r) of vrt_march.h, for this march: the steps of the lanes that march as the instructions themselves.  A lane whose
                // voxel does not let it pass leaves the exec mask — its end state stays in its registers, `ref` = its cell's entry —
                // and the others march on until few enough are left (the refill condition: the loop above has its C++ form) or a lane runs out
                // of lookups (:220; rare: the code behind the loop ends that ray).  v60..v63: the march cell (.x the entry, .y the
                // size-2 bits, .z .w which voxels a ray passes).  What the compiler had made of this loop in C++: 20 scalar instructions and
                // four vector ones of control flow per step, its `marching` flag a register that is compared, counted and selected.
                const uint32_t leave_at = next < n ? 64u - refill_at : 0u;   // leave when no more lanes than this still march
                uint32_t t0, t1, t2, t3, u;
                unsigned long long sx = __ballot(marching), sa;   // sx: the lanes that march, then — inside — the lanes the loop was entered with
// the lanes that march (exec is saved in the register that named them); the cell inside its line of 2 x 2 x 2 (bits 2 of x, y, z: [.. z2 y2 x2] in t0) ...
#define VBM_HEAD \
                    "s_and_saveexec_b64 %[sx], %[sx]\n" \
                    ".Lvbm_step_%=:\n\t" \
                    "v_lshrrev_b32_e32 %[t0], 2, %[vx]\n\t" \
                    "v_lshrrev_b32_e32 %[t1], 1, %[vy]\n\t" \
                    "v_bitop3_b32 %[t0], 1, %[t0], %[t1] bitop3:0xca\n\t" \
                    "v_bitop3_b32 %[t0], 3, %[t0], %[vz] bitop3:0xca\n\t"
// ... a direct world: the line among the lines of the whole world (bits 3 and up of the coordinates)
#define VBM_ADDRESS_DIRECT \
                    "v_ashrrev_i32_e32 %[t2], 3, %[vy]\n\t" \
                    "v_ashrrev_i32_e32 %[t1], 3, %[vz]\n\t" \
                    "v_bitop3_b32 %[t0], 7, %[t0], %[vx] bitop3:0xca\n\t" \
                    "v_lshlrev_b32_e32 %[t0], 4, %[t0]\n\t" \
                    "v_mad_i32_i24 %[t0], %[t2], %[row], %[t0]\n\t" \
                    "v_mad_i32_i24 %[t0], %[t1], %[slab], %[t0]\n\t" \
                    "buffer_load_dwordx4 v[60:63], %[t0], %[mdesc], 0 offen\n\t" \
                    "v_lshlrev_b32_e32 %[u], 2, %[vy]\n\t" \
                    "v_lshlrev_b32_e32 %[t3], 4, %[vz]\n\t"
// ... a world with a chunk directory: the chunk's block of cells — looked up when the ray has entered another chunk (its coordinates
// as one number, base 128: -1 .. S <= 100 stay apart: z's upper bits | y's bits 5..11 | x's bits 5..11) —, the line inside the block (bits 3, 4: [z4 z3 | y4 y3 | x4 x3 | z2 y2 x2])
#define VBM_ADDRESS_DIRECTORY \
                    "v_lshlrev_b32_e32 %[u], 2, %[vy]\n\t" \
                    "v_lshlrev_b32_e32 %[t3], 4, %[vz]\n\t" \
                    "v_ashrrev_i32_e32 %[t1], 5, %[vz]\n\t" \
                    "v_bfe_u32 %[t2], %[vy], 5, 7\n\t" \
                    "v_lshl_or_b32 %[t1], %[t1], 7, %[t2]\n\t" \
                    "v_bfe_u32 %[t2], %[vx], 5, 7\n\t" \
                    "v_lshl_or_b32 %[t1], %[t1], 7, %[t2]\n\t" \
                    "v_ashrrev_i32_e32 %[t2], 5, %[vx]\n\t" \
                    "v_cmp_ne_u32_e32 vcc, %[t1], %[ckey]\n\t" \
                    "s_and_saveexec_b64 %[sa], vcc\n\t" \
                    "s_cbranch_execz .Lvbm_same_%=\n\t" \
                    "v_mov_b32_e32 %[ckey], %[t1]\n\t" \
                    "v_ashrrev_i32_e32 %[t1], 5, %[vy]\n\t" \
                    "v_lshlrev_b32_e32 %[t2], 2, %[t2]\n\t" \
                    "v_mad_i32_i24 %[t1], %[t1], %[drow], %[t2]\n\t" \
                    "v_ashrrev_i32_e32 %[t2], 5, %[vz]\n\t" \
                    "v_mad_i32_i24 %[t1], %[t2], %[dslab], %[t1]\n\t" \
                    "buffer_load_dword %[cblock], %[t1], %[ddesc], 0 offen\n\t" \
                    "s_waitcnt vmcnt(0)\n\t" \
                    "v_lshlrev_b32_e32 %[cblock], 13, %[cblock]\n" \
                    ".Lvbm_same_%=:\n\t" \
                    "s_mov_b64 exec, %[sa]\n\t" \
                    "v_bitop3_b32 %[t1], 31, %[vx], %[u] bitop3:0xca\n\t" \
                    "v_and_b32_e32 %[t2], 0x180, %[t3]\n\t" \
                    "v_and_b32_e32 %[t1], 0x7f, %[t1]\n\t" \
                    "v_bitop3_b32 %[t0], 7, %[t0], %[t1] bitop3:0xca\n\t" \
                    "v_or_b32_e32 %[t0], %[t0], %[t2]\n\t" \
                    "v_lshl_add_u32 %[t0], %[t0], 4, %[cblock]\n\t" \
                    "buffer_load_dwordx4 v[60:63], %[t0], %[mdesc], 0 offen\n\t"
// one 16-byte load answers the step — behind it, while it is in flight: u = (x&3) | (y&3) << 2 | (z&3) << 4 under z's upper bits, the
// shift that brings u's bit of .z .w to the top (does a ray pass the voxel?  the sign says), the position of the size-2 bit.  The lanes
// that stop are off from there; the others: the selector (lo, or the size-2 bit of a split cell's voxel, under nine set bits), the step
#define VBM_BODY \
                    "v_bitop3_b32 %[u], 3, %[vx], %[u] bitop3:0xca\n\t" \
                    "v_bitop3_b32 %[u], 15, %[u], %[t3] bitop3:0xca\n\t" \
                    "v_add_u32_e32 %[it], 1, %[it]\n\t" \
                    "v_sub_u32_e32 %[t3], 63, %[u]\n\t" \
                    "v_bfe_u32 %[t2], %[u], 1, 5\n\t" \
                    "s_waitcnt vmcnt(0)\n\t" \
                    "v_lshlrev_b64 v[62:63], %[t3], v[62:63]\n\t" \
                    "v_mov_b32_e32 %[ref], v60\n\t" \
                    "v_cmp_gt_i32_e32 vcc, 0, v63\n\t" \
                    "s_and_b64 exec, exec, vcc\n\t" \
                    "s_cbranch_scc0 .Lvbm_out_%=\n\t" \
                    "v_bfe_u32 %[t0], v61, %[t2], 1\n\t" \
                    "v_and_or_b32 %[t0], v60, 31, %[t0]\n\t" \
                    "v_or_b32_e32 %[t0], 0xff800000, %[t0]\n\t" \
                    "v_bitop3_b32 %[ax], %[t0], %[mx], %[vx] bitop3:0xca\n\t" \
                    "v_bitop3_b32 %[ay], %[t0], %[my], %[vy] bitop3:0xca\n\t" \
                    "v_bitop3_b32 %[az], %[t0], %[mz], %[vz] bitop3:0xca\n\t" \
                    "v_add_f32_e32 %[ax], %[cx], %[ax]\n\t" \
                    "v_add_f32_e32 %[ay], %[cy], %[ay]\n\t" \
                    "v_add_f32_e32 %[az], %[cz], %[az]\n\t" \
                    "v_sub_f32_e32 %[ax], %[ax], %[px]\n\t" \
                    "v_sub_f32_e32 %[ay], %[ay], %[py]\n\t" \
                    "v_sub_f32_e32 %[az], %[az], %[pz]\n\t" \
                    "v_mul_f32_e64 %[ax], |%[ax]|, %[ux]\n\t" \
                    "v_mul_f32_e64 %[ay], |%[ay]|, %[uy]\n\t" \
                    "v_mul_f32_e64 %[az], |%[az]|, %[uz]\n\t" \
                    "v_min3_f32 %[st], %[ax], %[ay], %[az]\n\t" \
                    "v_cmp_nlt_f32_e32 vcc, 0, %[st]\n\t" \
                    "s_cbranch_vccnz .Lvbm_zero_%=\n" \
                    ".Lvbm_move_%=:\n\t" \
                    "v_add_f32_e32 %[t0], 0x3a83126f, %[st]\n\t" \
                    "v_cmp_eq_f32_e32 vcc, %[st], %[ax]\n\t" \
                    "v_cndmask_b32_e32 %[t1], %[st], %[t0], vcc\n\t" \
                    "v_cmp_eq_f32_e32 vcc, %[st], %[ay]\n\t" \
                    "v_cndmask_b32_e32 %[t2], %[st], %[t0], vcc\n\t" \
                    "v_cmp_eq_f32_e32 vcc, %[st], %[az]\n\t" \
                    "v_cndmask_b32_e32 %[t0], %[st], %[t0], vcc\n\t" \
                    "v_mul_f32_e32 %[t1], %[dx], %[t1]\n\t" \
                    "v_mul_f32_e32 %[t2], %[dy], %[t2]\n\t" \
                    "v_mul_f32_e32 %[t0], %[dz], %[t0]\n\t" \
                    "v_add_f32_e32 %[px], %[px], %[t1]\n\t" \
                    "v_add_f32_e32 %[py], %[py], %[t2]\n\t" \
                    "v_add_f32_e32 %[pz], %[pz], %[t0]\n\t" \
                    "v_cvt_flr_i32_f32_e32 %[vx], %[px]\n\t" \
                    "v_cvt_flr_i32_f32_e32 %[vy], %[py]\n\t" \
                    "v_cvt_flr_i32_f32_e32 %[vz], %[pz]\n\t" \
                    "v_cmp_lt_u32_e32 vcc, 0x1f3, %[it]\n\t" \
                    "s_cbranch_vccnz .Lvbm_out_%=\n\t" \
                    "s_bcnt1_i32_b64 vcc_lo, exec\n\t" \
                    "s_cmp_gt_u32 vcc_lo, %[leave]\n\t" \
                    "s_cbranch_scc1 .Lvbm_step_%=\n\t" \
                    "s_branch .Lvbm_out_%=\n" \
                    ".Lvbm_zero_%=:\n\t" \
                    "v_add_u32_e32 %[t0], -1, %[ax]\n\t" \
                    "v_add_u32_e32 %[t1], -1, %[ay]\n\t" \
                    "v_add_u32_e32 %[t2], -1, %[az]\n\t" \
                    "v_min3_u32 %[t0], %[t0], %[t1], %[t2]\n\t" \
                    "v_add_u32_e32 %[st], 1, %[t0]\n\t" \
                    "s_branch .Lvbm_move_%=\n" \
                    ".Lvbm_out_%=:\n\t" \
                    "s_mov_b64 vcc, exec\n\t" \
                    "s_mov_b64 exec, %[sx]\n\t" \
                    "v_cndmask_b32_e64 %[t3], 0, 1, vcc"
#define VBM_OUTPUTS \
                    [px] "+v"(pos.x), [py] "+v"(pos.y), [pz] "+v"(pos.z), [vx] "+v"(vx), [vy] "+v"(vy), [vz] "+v"(vz), [st] "+v"(step), [ax] "+v"(adx), \
                    [ay] "+v"(ady), [az] "+v"(adz), [ref] "+v"(ref), [it] "+v"(iter), [t0] "=&v"(t0), \
                    [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [u] "=&v"(u), [sx] "+s"(sx)
#define VBM_OUTPUTS_DIRECTORY , [ckey] "+v"(ckey), [cblock] "+v"(cblock), [sa] "=&s"(sa)
#define VBM_INPUTS \
                    [mx] "v"(mxm), [my] "v"(mym), [mz] "v"(mzm), [cx] "v"(cx), [cy] "v"(cy), [cz] "v"(cz), [ux] "v"(ux), [uy] "v"(uy), [uz] "v"(uz), \
                    [dx] "v"(dir.x), [dy] "v"(dir.y), [dz] "v"(dir.z), [mdesc] "s"(mb), [leave] "s"(leave_at)
#define VBM_INPUTS_DIRECT , [row] "s"(row128), [slab] "s"(slab128)
#define VBM_INPUTS_DIRECTORY , [ddesc] "s"(db), [drow] "s"(drow), [dslab] "s"(dslab)
#define VBM_CLOBBERS "vcc", "scc", "memory", "v60", "v61", "v62", "v63"
                if constexpr (DIRECT) asm volatile(VBM_HEAD VBM_ADDRESS_DIRECT VBM_BODY : VBM_OUTPUTS : VBM_INPUTS VBM_INPUTS_DIRECT : VBM_CLOBBERS);
                else asm volatile(VBM_HEAD VBM_ADDRESS_DIRECTORY VBM_BODY : VBM_OUTPUTS VBM_OUTPUTS_DIRECTORY : VBM_INPUTS VBM_INPUTS_DIRECTORY : VBM_CLOBBERS);
                marching = t3 != 0u;   // (the loop's last word: which lanes still march)
                if (marching && iter >= kMaxSteps) {
                    // out of lookups in air or in a liquid (:220, :293): the segment ends as a hit on the voxel of the last lookup — which
                    // for a split cell is in its brick, at the position that was looked up
                    marching = false;
                    ref = (int)ref < 0 ? ((uint32_t)__builtin_amdgcn_raw_buffer_load_b16(bb, ((ref & 0x7FFFFFFFu) + (u & 63u)) << 1, 0, 0) >> 1) << 16 : ref;
                }
                continue;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- between B and C: the rays that hit first, then the rays that missed (their end position is outside the world) ----
    uint32_t n_hit = 0u;
    {
        bool hit[KB];
        uint32_t cnt[KB];
#pragma unroll
        for (uint32_t k = 0; k < KB; k++) {
            const uint32_t i = k * 64u + lane;
            hit[k] = false;
            if (i < n) {
                const V3 pos{pool[0u * E + i], pool[1u * E + i], pool[2u * E + i]};
                hit[k] = !(min3_nan_ignoring(pos.x, pos.y, pos.z) < 0.0f ||
                           max(max((uint32_t)trunc2i(pos.x), (uint32_t)trunc2i(pos.y)), (uint32_t)trunc2i(pos.z)) >= P.world.size);
            }
            cnt[k] = (uint32_t)__popcll(__ballot(hit[k]));
            n_hit += cnt[k];
        }
        uint32_t at_hit = 0u, at_miss = n_hit;
#pragma unroll
        for (uint32_t k = 0; k < KB; k++) {
            const uint32_t i = k * 64u + lane;
            const unsigned long long mh = __ballot(hit[k]), mm = __ballot(i < n && !hit[k]);
            if (hit[k]) order[at_hit + lanes_below(mh)] = (uint16_t)i;
            else if (i < n) order[at_miss + lanes_below(mm)] = (uint16_t)i;
            at_hit += (uint32_t)__popcll(mh);
            at_miss += (uint32_t)__popcll(mm);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- C: what follows the march, full width.  On the last bounce a ray that hit has nothing left to do: its bounce
    // would be dropped and only a miss adds light ----
    const TableBuf bb = table_buffer(P.bricks, P.brick_bytes);
    uint32_t n_out = 0u;   // survivors so far: the next segment's pool
    for (uint32_t j0 = last_bounce ? n_hit & ~63u : 0u; j0 < n; j0 += 64u) {
        const uint32_t j = j0 + lane;
        bool alive = false;
        PathState st;
        st.slot = 0; st.rng = 0;
        st.origin = st.dir = st.thr = V3{0.f, 0.f, 0.f};
        if (j < n && !(last_bounce && j < n_hit)) {
            const uint32_t i = order[j];
            const uint32_t rec = base + i;
            const uint4 a = recs[in_at + rec], b = recs[in_at + P.in_cap + rec], c = recs[in_at + 2u * P.in_cap + rec];
            st.slot = a.x;
            st.origin = V3{__uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w)};
            st.dir = V3{__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z)};
            st.rng = b.w;
            st.thr = V3{__uint_as_float(c.x), __uint_as_float(c.y), __uint_as_float(c.z)};
            // segment_end on the parked end state (a path segment's water is nobody's business: DESIGN.md, path trace)
            const V3 pos{pool[0u * E + i], pool[1u * E + i], pool[2u * E + i]};
            const uint32_t packed = __float_as_uint(pool[3u * E + i]);
            MarchResult R;
            R.hit = false;
            R.pos = V3{0.f, 0.f, 0.f};
            R.norm = V3{0.f, 0.f, 0.f};
            R.water_dist = 0.0f;
            R.voxel = 0u;
            R.iters = 0u;
            R.visits = 0u;
            if (j < n_hit) {
                R.hit = true;
                R.pos = pos;
                R.norm = V3{((packed & 1u) ? 1.0f : 0.0f) * -vsign(st.dir.x), ((packed & 2u) ? 1.0f : 0.0f) * -vsign(st.dir.y),
                            ((packed & 4u) ? 1.0f : 0.0f) * -vsign(st.dir.z)};
                R.voxel = packed >> 4;
                if (packed & 8u) {   // stopped in a split cell: the voxel is in the cell's brick, at the end position
                    const uint32_t u = ((uint32_t)trunc2i(pos.x) & 3u) | (((uint32_t)trunc2i(pos.y) & 3u) << 2) | (((uint32_t)trunc2i(pos.z) & 3u) << 4);
                    R.voxel = (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(bb, (((packed >> 4) << 6) + u) << 1, 0, 0) >> 1;
                }
            }
            V3 light{0.f, 0.f, 0.f};
            bool missed;
            alive = path_after_march(P, st, R, light, missed) && !last_bounce;
            if (missed) {
                uint4 t = P.out[st.slot];
                t.x = __float_as_uint(__uint_as_float(t.x) + light.x);
                t.y = __float_as_uint(__uint_as_float(t.y) + light.y);
                t.z = __float_as_uint(__uint_as_float(t.z) + light.z);
                P.out[st.slot] = t;
            }
        }
        if (left != 1u) {   // the survivors, compacted into this wave's own range of the other buffer
            const unsigned long long m = __ballot(alive);
            if (alive) {
                const uint32_t o = out_at + base + n_out + lanes_below(m);
                recs[o] = make_uint4(st.slot, __float_as_uint(st.origin.x), __float_as_uint(st.origin.y), __float_as_uint(st.origin.z));
                recs[P.path_cap + o] = make_uint4(__float_as_uint(st.dir.x), __float_as_uint(st.dir.y), __float_as_uint(st.dir.z), st.rng);
                recs[2u * P.path_cap + o] = make_uint4(__float_as_uint(st.thr.x), __float_as_uint(st.thr.y), __float_as_uint(st.thr.z), 0u);
            }
            n_out += (uint32_t)__popcll(m);
        }
    }
    if (left == 1u || n_out == 0u) break;
    // the next segment: the records just written are read back by other lanes of this wave (same CU, same L1)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    n = n_out;
    {
        const uint32_t t = in_at;
        in_at = out_at;
        out_at = t;
    }
    }
}


// The end of a launch chain of several samples: the frame's running sum plus the chain's planes, in sample order — the
// order the one-sample-per-chain launches add them in and the oracle's — and the division once the last chain is in.
__global__ void path_chain_finish_kernel(Texel *out, const Texel *acc, uint32_t n, uint32_t chain, uint32_t first, uint32_t last, float spp) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint4 t = acc[i];   // the chain's first sample: the frame's first (light and id word as they are), or one more term
    if (!first) {
        const uint4 o = out[i];
        t.x = __float_as_uint(__uint_as_float(o.x) + __uint_as_float(t.x));
        t.y = __float_as_uint(__uint_as_float(o.y) + __uint_as_float(t.y));
        t.z = __float_as_uint(__uint_as_float(o.z) + __uint_as_float(t.z));
        t.w = o.w;
    }
    for (uint32_t s = 1; s < chain; s++) {
        const uint4 a = acc[(size_t)s * n + i];
        t.x = __float_as_uint(__uint_as_float(t.x) + __uint_as_float(a.x));
        t.y = __float_as_uint(__uint_as_float(t.y) + __uint_as_float(a.y));
        t.z = __float_as_uint(__uint_as_float(t.z) + __uint_as_float(a.z));
    }
    if (last) {
        t.x = __float_as_uint(__uint_as_float(t.x) / spp);
        t.y = __float_as_uint(__uint_as_float(t.y) / spp);
        t.z = __float_as_uint(__uint_as_float(t.z) / spp);
    }
    out[i] = t;
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


// The path trace marches with the grid march when the derived tables exist (P.grid), else with the ancestor-cache walk;
// `literal` (air flagged liquid, vrt_frames.hip) with the shader's text.
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
    if (P.acc) {   // several samples per launch chain: plain frames over the derived tables only (vrt_frames.hip)
        hipLaunchKernelGGL((path_primary_kernel<0, false, false, true>), grid, block, lds_bytes_path(P, false), st, P);
        return;
    }
    VRT_PATH_LAUNCH(path_primary_kernel);
}

// the pool kernel over the march cells (P.mblk): `segments` bounce segments in this one launch (every wave carries its own
// survivors from one to the next; P.path_in / P.path_out are the two buffers it goes back and forth between)
void launch_path_bounce_cells(const FrameParams &P, uint32_t refill_at, uint32_t segments, uint32_t pool_batches, hipStream_t st) {
    if (P.tiles_local == 0 || segments == 0) return;
    const uint32_t refill = refill_at >= 1u && refill_at <= 64u ? refill_at : kPoolRefillAt;
    const uint32_t kb = (pool_batches == 5u && P.march_direct) ? 5u : 4u, entries = kb * 64u;   // (320-ray pools: direct worlds only)
    const uint32_t parts = (P.in_seg_cap + 4u * entries - 1u) / (4u * entries);
    const dim3 grid(kHitSegments * parts), block(256);
    const size_t sh = 8u * 4u + 4u * (entries * 16u + entries * 2u);   // per wave: the pool + a u16 order per entry
    const CellsLaunch L{P, refill, segments};
    if (kb == 5u) hipLaunchKernelGGL((path_bounce_cells_kernel<true, 5u>), grid, block, sh, st, L);
    else if (P.march_direct) hipLaunchKernelGGL((path_bounce_cells_kernel<true, 4u>), grid, block, sh, st, L);
    else hipLaunchKernelGGL((path_bounce_cells_kernel<false, 4u>), grid, block, sh, st, L);
}

void launch_path_bounce(const FrameParams &P, bool stats, bool literal, hipStream_t st) {
    if (P.tiles_local == 0) return;
    const dim3 grid(kHitSegments * (P.hit_seg_cap / 256u)), block(256);
    VRT_PATH_LAUNCH(path_bounce_kernel);
}
#undef VRT_PATH_LAUNCH

void launch_path_chain_finish(Texel *out, const Texel *acc, uint32_t n, uint32_t chain, bool first, bool last, uint32_t spp, hipStream_t st) {
    if (!n) return;
    hipLaunchKernelGGL(path_chain_finish_kernel, dim3((n + 255u) / 256u), dim3(256), 0, st, out, acc, n, chain, first ? 1u : 0u, last ? 1u : 0u,
                       (float)spp);
}

void launch_path_finish(Texel *out, uint32_t n, uint32_t spp, hipStream_t st) {
    if (!n) return;
    hipLaunchKernelGGL(path_finish_kernel, dim3((n + 255u) / 256u), dim3(256), 0, st, out, n, (float)spp);
}

}  // namespace vrt
