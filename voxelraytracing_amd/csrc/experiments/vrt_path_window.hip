// vrt_path_window.hip — the path trace's bounce segments over LDS-STAGED WINDOWS of march cells (gfx950).
//
// What a bounce segment is made of (profiles/r04_path_*): a ray leaves a surface in a random direction
// and two thirds of its lookups fall within 32 voxels of where it started — but every one of them is a 16-byte gather from
// a line nobody else in the wave wants (37 distinct lines per wave-step, 47 % L1 hits), and a wave-step lasts as long as its
// slowest miss: 1.3 us, with the vector pipes a quarter busy.  Removing *some* of a wave-step's misses does nothing (round
// 4's air map); removing *all* of them needs every lane of the wave inside memory that cannot miss.
//
// So the rays are grouped by where they start and the march cells around them are staged in LDS:
//   * the primary launch (path_primary_kernel<GROUPED>, vrt_path.hip) takes the screen's tiles in blocks of 4 x 4 and compacts
//     each workgroup's survivors into the workgroup's own region of the path buffer: four consecutive regions are the
//     paths of 32 x 32 pixels, whose primary hits lie within a few voxels of each other;
//   * a bounce workgroup (4 waves, one region each) averages its rays' origins, and stages the WINDOW of march cells around
//     that point — WCX x WCY x WCZ cells of 4^3 voxels, 16 bytes each, copied line by line (128 bytes: 2 x 2 x 2 cells)
//     from the derived table — in its LDS;
//   * mode 0: every ray marches while it is inside the window, one ds_read_b128 per step, no global load in the loop at all;
//     a ray that stops is parked as before, a ray that steps out of the window is parked where it stands (position, lookups
//     so far, the faces its last step left through) and listed;
//   * mode 1: the listed rays go on from where they stand with the march-cell loads of path_bounce_cells_kernel (they are the
//     long ones: open air, or terrain far from where they started);
//   * then, as before: phase C at full width, the survivors compacted into the wave's own range, the next segment from the
//     same window (a ray that stopped in mode 0 starts its next segment inside it).
// Every ray executes the arithmetic the other kernels execute for it, in the same order: bit-identical frames (tests).
//
// A ray's state between the phases (unit steps; end state) lives in two 16-byte planes in global memory beside its path
// record instead of in a per-wave LDS pool: the LDS belongs to the window.
//
// Spec: clientdesktop/src/graphics/path_tracer.wgsl:149-194 over the march of ray_tracer.wgsl:220-291.
#include <type_traits>

#include "../vrt_path_primary.h"
#include "../vrt_exp.h"

namespace vrt {

constexpr uint32_t kWinEntries = 256u;                // a wave's rays per segment in a window launch
constexpr uint32_t kDeepEntriesMax = 4096u;           // ... in a deep launch (no window: the LDS holds nothing but the rays' order)
constexpr uint32_t kWinRefillAt = 16u;
constexpr uint32_t kEndLeft = 0x80000000u;            // a parked end state's w: the ray left the window in mid-march
constexpr uint32_t kEndNotFinite = 0x40000000u;       // ... and its origin or direction is not finite (the careful march)

struct WindowLaunch {
    FrameParams P;
    uint32_t segments;     // bounce segments in this launch: all that the frame's paths have left
    uint32_t wave_rays;    // rays a wave takes per segment: kWinEntries, or (no window) all a region can hold
    uint32_t parts;        // a region's records are handed to `parts` workgroups, wave_rays each
    uint32_t n_regions;    // regions of the path buffer = workgroups of the primary launch
    int32_t lift;          // the window's centre above the mean origin, in voxels (rays leave a surface upwards)
};
// A ray's state between the phases lives in two more planes of the path buffer, behind its two sets of three record planes:
// plane 6 {|unit step| xyz, -}, plane 7 {position xyz, packed end state}, indexed like the records of the current segment.
constexpr uint32_t kPlaneUnit = 6u, kPlaneEnd = 7u;
// a lane's index into its wave's rays, with the lane's flags riding on it (the scalar registers are all taken)
constexpr uint32_t kIdxMask = 0xFFFu, kParked = 0x1000u, kGone = 0x2000u, kNotFinite = 0x4000u;

#ifdef VRT_EXP_WINDBG
// experiment (tools/ab build, tools/window_probe.py): per wave of the launch {rays, start, end (100 MHz), shader-clock ticks in the
// window's staging | phase A << 32, mode 0 | mode 1 << 32, phase C, wave-steps mode 0 | mode 1 << 32, lane-steps likewise, rays that left}
__device__ unsigned long long g_win_dbg[16384 * 8];
extern "C" void vrt_exp_win_dbg(unsigned long long *out) {   // read and reset
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_win_dbg), sizeof(unsigned long long) * 16384 * 8);
    void *p = nullptr;
    (void)hipGetSymbolAddress(&p, HIP_SYMBOL(g_win_dbg));
    (void)hipMemset(p, 0, sizeof(unsigned long long) * 16384 * 8);
}
#define WINDBG(...) __VA_ARGS__
#else
#define WINDBG(...)
#endif

// NW waves per workgroup (one region each) share a window: 4 x 8 KiB, 8 x 32 KiB and 16 x 64 KiB all leave the CU its 32 waves
template <bool DIRECT, int WCX, int WCY, int WCZ, int NW>
#ifndef VRT_WINDOW_NO_WAVES_ATTR
__attribute__((amdgpu_waves_per_eu(8, 8)))
#endif
__global__ void __launch_bounds__(NW * 64) path_bounce_window_kernel(WindowLaunch L) {
    static_assert(WCX % 2 == 0 && WCY % 2 == 0 && WCZ % 2 == 0, "the window is whole lines of 2 x 2 x 2 cells");
    constexpr bool kWindow = WCX > 0;   // (no window: every ray marches over the table itself — path_bounce_cells_kernel with its rays' state in global memory)
    constexpr uint32_t kCells = WCX * WCY * WCZ;
    constexpr uint32_t LX = kWindow ? WCX / 2 : 1, LY = kWindow ? WCY / 2 : 1;
    const uint32_t E = __builtin_amdgcn_readfirstlane(L.wave_rays);
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint4 *const s_win = reinterpret_cast<uint4 *>(smem);   // [WCZ][WCY][WCX] march cells
    uint32_t *const s_misc = smem + 4u * kCells;             // [0, 8) the liquid mask, [8, 12) the origins' sums and count
    const FrameParams &K = L.P;
    WINDBG(const unsigned long long dbg_r0 = __builtin_amdgcn_s_memrealtime(); const unsigned long long dbg_c0 = __builtin_amdgcn_s_memtime();
           unsigned long long dbg_a = 0, dbg_b0 = 0, dbg_b1 = 0, dbg_c = 0; uint32_t dbg_ws[2] = {0u, 0u}, dbg_ls[2] = {0u, 0u}, dbg_left = 0u, dbg_rays = 0u, dbg_rf[2] = {0u, 0u}, dbg_rn[2] = {0u, 0u};)
    if (threadIdx.x < 8) s_misc[threadIdx.x] = K.liquid[threadIdx.x];
    else if (threadIdx.x < 12) s_misc[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t group = blockIdx.x / L.parts, part = blockIdx.x - group * L.parts;
    const uint32_t region = group * NW + wave;
    uint32_t n = 0u;
    if (region < L.n_regions) {
        const uint32_t count = min(K.grp_counts[region], K.grp_cap);
        if (count > part * E) n = min(E, count - part * E);
    }
    n = __builtin_amdgcn_readfirstlane(n);
    const uint32_t base = __builtin_amdgcn_readfirstlane(region * K.grp_cap + part * E);
    uint4 *const recs = K.path_in < K.path_out ? const_cast<uint4 *>(K.path_in) : K.path_out;
    uint32_t in_at = __builtin_amdgcn_readfirstlane((uint32_t)(K.path_in - recs));
    uint32_t out_at = __builtin_amdgcn_readfirstlane((uint32_t)(K.path_out - recs));

    // ---- the window: around the mean of the workgroup's origins (those inside the world) ----
    if constexpr (kWindow) {
        int wx0, wy0, wz0;
        const uint32_t lane0 = threadIdx.x & 63u;
        const float wmax = (float)K.world.size;
        int sx = 0, sy = 0, sz = 0, sc = 0;
        for (uint32_t k = 0; k * 64u < n; k++) {
            const uint32_t i = k * 64u + lane0;
            if (i < n) {
                const uint4 a = recs[in_at + base + i];
                const float ox = __uint_as_float(a.y), oy = __uint_as_float(a.z), oz = __uint_as_float(a.w);
                if (ox >= 0.0f && oy >= 0.0f && oz >= 0.0f && ox < wmax && oy < wmax && oz < wmax) {
                    sx += (int)ox; sy += (int)oy; sz += (int)oz; sc += 1;
                }
            }
        }
        if (sc) {
            atomicAdd(&s_misc[8], (uint32_t)sx); atomicAdd(&s_misc[9], (uint32_t)sy); atomicAdd(&s_misc[10], (uint32_t)sz);
            atomicAdd(&s_misc[11], (uint32_t)sc);
        }
        __syncthreads();
        const uint32_t cnt = s_misc[11];
        int mx = (int)(K.world.size / 2u), my = mx, mz = mx;
        if (cnt) { mx = (int)(s_misc[8] / cnt); my = (int)(s_misc[9] / cnt); mz = (int)(s_misc[10] / cnt); }
        // (whole lines of 8 voxels; a window may reach beyond the world: those cells are zeros, as the table's border is)
        wx0 = __builtin_amdgcn_readfirstlane((mx - 2 * WCX + 4) & ~7);
        wy0 = __builtin_amdgcn_readfirstlane((my + L.lift - 2 * WCY + 4) & ~7);
        wz0 = __builtin_amdgcn_readfirstlane((mz - 2 * WCZ + 4) & ~7);
        if (threadIdx.x == 0) { s_misc[12] = (uint32_t)wx0; s_misc[13] = (uint32_t)wy0; s_misc[14] = (uint32_t)wz0; }   // (read back per segment)
        const TableBuf mb = table_buffer(K.mblk, K.mblk_bytes);
        const uint32_t gd = K.grid_dim, S1 = gd / 8u + 1u;
        const uint32_t row128 = (gd / 2u + 1u) * 128u, slab128 = (gd / 2u + 1u) * row128;
        const int cx0 = wx0 >> 2, cy0 = wy0 >> 2, cz0 = wz0 >> 2;
        for (uint32_t c0 = threadIdx.x; c0 < kCells; c0 += 4u * NW * 64u) {
            uint4 v[4];
            uint32_t at[4];
#pragma unroll
            for (uint32_t q = 0; q < 4u; q++) {   // (four loads in flight per thread; eight consecutive threads read one line)
                const uint32_t c = c0 + q * NW * 64u;
                const uint32_t sub = c & 7u, line = c >> 3;
                const uint32_t lx = line % LX, ly = (line / LX) % LY, lz = line / (LX * LY);
                const uint32_t ox = 2u * lx + (sub & 1u), oy = 2u * ly + ((sub >> 1) & 1u), oz = 2u * lz + (sub >> 2);
                const int cx = cx0 + (int)ox, cy = cy0 + (int)oy, cz = cz0 + (int)oz;   // the cell in the world
                at[q] = (oz * WCY + oy) * WCX + ox;
                v[q] = make_uint4(0u, 0u, 0u, 0u);
                if (c < kCells && (uint32_t)cx < gd && (uint32_t)cy < gd && (uint32_t)cz < gd) {
                    uint32_t off;
                    if (DIRECT) {
                        off = (uint32_t)(cz >> 1) * slab128 + (uint32_t)(cy >> 1) * row128 + ((uint32_t)(cx >> 1) << 7) + (sub << 4);
                    } else {
                        const uint32_t block = K.cdir[((uint32_t)(cz >> 3) * S1 + (uint32_t)(cy >> 3)) * S1 + (uint32_t)(cx >> 3)] << 13;
                        const uint32_t bl = (((((uint32_t)cz >> 1) & 3u) << 2 | (((uint32_t)cy >> 1) & 3u)) << 2) | (((uint32_t)cx >> 1) & 3u);
                        off = block + (((bl << 3) | sub) << 4);
                    }
                    v[q] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(mb, off, 0, 0));
                }
            }
#pragma unroll
            for (uint32_t q = 0; q < 4u; q++)
                if (c0 + q * NW * 64u < kCells) s_win[at[q]] = v[q];
        }
        __syncthreads();
    }
    if (n == 0u) return;   // (no barrier below: the waves go their own ways from here)
    WINDBG(const unsigned long long dbg_c1 = __builtin_amdgcn_s_memtime();)

    for (uint32_t left = L.segments;; left--) {   // (left: segments still to do, this one included)
    // What does not change from one segment to the next is made anew for every one of them (path_bounce_cells_kernel)
    typedef const __attribute__((address_space(4))) WindowLaunch *KernArgs;
    KernArgs kargs = (KernArgs)__builtin_amdgcn_kernarg_segment_ptr();   // (L is the kernel's only argument)
    asm volatile("" : "+s"(kargs));
    const FrameParams &P = ((const WindowLaunch *)kargs)->P;
    constexpr uint32_t refill_at = kWinRefillAt;
    uint4 *const ray_unit = recs + kPlaneUnit * P.path_cap, *const ray_end = recs + kPlaneEnd * P.path_cap;
    uint32_t none = 0u;
    asm volatile("" : "+s"(none));
    const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, none));
    const uint32_t *const s_liquid = s_misc;
    uint16_t *const order = reinterpret_cast<uint16_t *>(s_misc + 16) + (wave + none) * E;   // phase C's order of the rays
    uint16_t *const lefts = order;   // during the marches: the rays that left the window (dead before the order is made)
    const float world_max = 0.0f + (float)P.world.size;
    const bool last_bounce = left == 1u;   // (the launch's last segment is the paths' last)

    // ---- A: the unit steps of every ray (nine divides, three square roots), full width ----
    WINDBG(const unsigned long long dbg_ta = __builtin_amdgcn_s_memtime(); dbg_rays += n;)
    for (uint32_t k = 0; k * 64u < n; k++) {
        const uint32_t i = k * 64u + lane;
        if (i < n) {
            const uint4 b = recs[in_at + P.in_cap + base + i];
            const V3 unit = unit_steps(V3{__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z)});
            ray_unit[base + i] = make_uint4(__float_as_uint(unit.x), __float_as_uint(unit.y), __float_as_uint(unit.z), 0u);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();

    WINDBG(asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); dbg_a += __builtin_amdgcn_s_memtime() - dbg_ta;)
    // ---- B: the marches, lanes refilled from the wave's list; a ray that has stopped (or left the window) keeps its end state
    // in its registers until the wave's next refill parks it.  Water is not tracked (no output of a path segment depends on it). ----
    {
        V3 pos{0.f, 0.f, 0.f}, dir{0.f, 0.f, 0.f};
        float ux = 0.f, uy = 0.f, uz = 0.f, step = -1.f, adx = 0.f, ady = 0.f, adz = 0.f;
        uint32_t mxm = 0u, mym = 0u, mzm = 0u, ref = 0u, iter = 0u, idx = kParked;   // idx: kIdxMask | kParked | kGone (out of the window, in mid-march) | kNotFinite
        int vx = 0, vy = 0, vz = 0;
        constexpr uint32_t kNoChunk = 0x7FFFFFFFu;
        uint32_t ckey = kNoChunk, cblock = 0u;
        bool marching = false;
        uint32_t n_left = 0u;   // wave-uniform: rays listed in `lefts`

        // the end state: where, through which faces, on what (bit 3: `what` is the split cell's brick) — or, for a ray that left
        // the window, how many lookups it has had
        auto park = [&]() __attribute__((always_inline)) {
            uint32_t packed = (int)ref < 0 ? (8u | ((ref & 0x7FFFFFC0u) >> 2)) : ((ref >> 16) << 4);
            if (idx & kGone) packed = kEndLeft | ((idx & kNotFinite) ? kEndNotFinite : 0u) | (iter << 3);
            if (step != -1.0f) packed |= (step == adx ? 1u : 0u) | (step == ady ? 2u : 0u) | (step == adz ? 4u : 0u);
            ray_end[base + (idx & kIdxMask)] = make_uint4(__float_as_uint(pos.x), __float_as_uint(pos.y), __float_as_uint(pos.z), packed);
            idx |= kParked;
        };
        // the step to the leaf's exit face for a leaf of size lo + 1 (take_step of march_grid; (h), (q) of vrt_march.h)
        auto take_step = [&](uint32_t lo) __attribute__((always_inline)) {
            const float tx = (__uint_as_float(bfi(lo, mxm, (uint32_t)vx) - mxm) - 8388608.0f) - pos.x;
            const float ty = (__uint_as_float(bfi(lo, mym, (uint32_t)vy) - mym) - 8388608.0f) - pos.y;
            const float tz = (__uint_as_float(bfi(lo, mzm, (uint32_t)vz) - mzm) - 8388608.0f) - pos.z;
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
        // mode 0: inside the window (LDS), 1: the rays that left it (the table itself); mode 2 (no window): every ray over the table.
        // Instances of the same text: what only one of them needs (the window's origin; the tables' descriptors) is not kept in
        // registers through the other
        auto run = [&](auto MODE) __attribute__((always_inline)) {
            constexpr uint32_t mode = decltype(MODE)::value;
            constexpr bool fresh = mode != 1u, lds = mode == 0u;   // rays taken from their records / march cells read from the window
            const uint32_t n_list = fresh ? n : n_left;
            uint32_t next = 0u;   // wave-uniform: the list's first ray not handed out yet
            if (n_list == 0u) return;
            const TableBuf mb = table_buffer(P.mblk, P.mblk_bytes), db = table_buffer(P.cdir, P.cdir_bytes);
            const TableBuf bb = table_buffer(P.bricks, P.brick_bytes);
            const uint32_t drow = (P.grid_dim / 8u + 1u) * 4u, dslab = (P.grid_dim / 8u + 1u) * drow;
            const uint32_t row128 = (P.grid_dim / 2u + 1u) * 128u, slab128 = (P.grid_dim / 2u + 1u) * row128;   // < 2^23: S <= 16
            const uint32_t wsize = P.world.size;
            // (l) of vrt_march.h: the general step for a wave with a ray that is not finite (mode 1 only; rare: NaN cameras)
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

            const int wx0 = lds ? __builtin_amdgcn_readfirstlane((int)s_misc[12]) : 0, wy0 = lds ? __builtin_amdgcn_readfirstlane((int)s_misc[13]) : 0,
                      wz0 = lds ? __builtin_amdgcn_readfirstlane((int)s_misc[14]) : 0;
            WINDBG(const unsigned long long dbg_tb = __builtin_amdgcn_s_memtime();)
            for (;;) {
                // ---- refill: park what has stopped or left, hand out the list's next rays ----
                WINDBG(const unsigned long long dbg_tr = __builtin_amdgcn_s_memtime(); const uint32_t dbg_next0 = next;)
                {
                    const unsigned long long went = __ballot(!marching && (idx & (kParked | kGone)) == kGone);
                    if (!marching && !(idx & kParked)) {
                        if (idx & kGone) lefts[n_left + lanes_below(went)] = (uint16_t)(idx & kIdxMask);
                        park();
                    }
                    n_left += (uint32_t)__popcll(went);
                }
                {
                    const unsigned long long idle = __ballot(!marching);
                    const uint32_t at = next + lanes_below(idle);
                    if (!marching && at < n_list) {
                        constexpr uint32_t kTwo23 = 0x4B000000u;
                        idx = fresh ? at : (uint32_t)lefts[at];   // (flags clear)
                        const uint32_t rec = base + idx;
                        const uint4 b = recs[in_at + P.in_cap + rec], un = ray_unit[rec];
                        dir = V3{__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z)};
                        ux = __uint_as_float(un.x); uy = __uint_as_float(un.y); uz = __uint_as_float(un.z);
                        mxm = (dir.x >= 0.0f ? ~0u : 0u) - kTwo23; mym = (dir.y >= 0.0f ? ~0u : 0u) - kTwo23; mzm = (dir.z >= 0.0f ? ~0u : 0u) - kTwo23;
                        ref = 0u;
                        ckey = kNoChunk;
                        marching = true;
                        if (fresh) {
                            const uint4 a = recs[in_at + rec];
                            const V3 origin{__uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w)};
                            const bool not_finite = !(finite3(origin) && finite3(dir));
                            pos = nudged(origin, dir);
                            step = -1.0f; adx = ady = adz = 0.0f;
                            iter = 0u;
                            if ((pos.x <= 0.0f || pos.y <= 0.0f || pos.z <= 0.0f) || (pos.x >= world_max || pos.y >= world_max || pos.z >= world_max)) {
                                // starts outside the world: a miss before any lookup.  Its end state says so (a position outside), and is
                                // parked right here
                                marching = false;
                                pos = V3{-1.0f, -1.0f, -1.0f};
                                park();
                            } else if (not_finite) {
                                if (lds) {   // the careful march is mode 1's
                                    marching = false;
                                    idx |= kGone | kNotFinite;
                                } else {
                                    idx |= kNotFinite;
                                }
                            }
                            vx = trunc2i(pos.x); vy = trunc2i(pos.y); vz = trunc2i(pos.z);
                        } else {
                            // where it stood when it left the window, how many lookups it has had, and through which faces its last step
                            // left (as a step / distances triple that compares the same way)
                            const uint4 e = ray_end[rec];
                            pos = V3{__uint_as_float(e.x), __uint_as_float(e.y), __uint_as_float(e.z)};
                            if (e.w & kEndNotFinite) idx |= kNotFinite;
                            iter = (e.w >> 3) & 0xFFFFu;
                            step = (e.w & 7u) ? 1.0f : -1.0f;
                            adx = (e.w & 1u) ? 1.0f : 2.0f; ady = (e.w & 2u) ? 1.0f : 2.0f; adz = (e.w & 4u) ? 1.0f : 2.0f;
                            vx = flr2i(pos.x); vy = flr2i(pos.y); vz = flr2i(pos.z);   // as take_step left them (a fresh ray is inside the world: the same)
                        }
                    }
                    next = min(n_list, next + (uint32_t)__popcll(idle));
                }
                WINDBG(asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); if (next != dbg_next0) { dbg_rf[lds ? 0 : 1] += (uint32_t)((__builtin_amdgcn_s_memtime() - dbg_tr) >> 4); dbg_rn[lds ? 0 : 1] += 1u; })
                if (__ballot(marching) == 0ull) {
                    if (next >= n_list && __ballot(!(idx & kParked)) == 0ull) break;   // the list is empty, nobody marches, everything is parked
                    continue;
                }
                if (!lds && __ballot(marching && (idx & kNotFinite)) != 0ull) {   // wave-uniform, rare
                    for (;;) {
                        if (marching) careful_step();
                        const uint32_t n_march = (uint32_t)__popcll(__ballot(marching));
                        if (n_march == 0u || (next < n_list && 64u - n_march >= refill_at)) break;
                    }
                    continue;
                }
                for (;;) {
                    WINDBG(dbg_ws[lds ? 0 : 1] += 1u; dbg_ls[lds ? 0 : 1] += (uint32_t)__popcll(__ballot(marching));)
                    if (marching) {
                        uint4 c;
                        bool look = true;
                        if (lds) {
                            const uint32_t lx = (uint32_t)(vx - wx0), ly = (uint32_t)(vy - wy0), lz = (uint32_t)(vz - wz0);
                            look = (lx < 4u * WCX) & (ly < 4u * WCY) & (lz < 4u * WCZ);
                            c = s_win[look ? ((lz >> 2) * WCY + (ly >> 2)) * WCX + (lx >> 2) : 0u];
                        } else {
                            const uint32_t sub = ((((uint32_t)vz >> 2) & 1u) << 2) | ((((uint32_t)vy >> 2) & 1u) << 1) | (((uint32_t)vx >> 2) & 1u);
                            uint32_t off;
                            if (DIRECT) {
                                off = mad_i24(vz >> 3, slab128, mad_i24(vy >> 3, row128, ((uint32_t)(vx >> 3) << 7) + (sub << 4)));
                            } else {
                                const uint32_t key = (uint32_t)((((vz >> 5) << 7) + (vy >> 5)) << 7) + (uint32_t)(vx >> 5);
                                if (key != ckey) {
                                    ckey = key;
                                    cblock = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(db, mad_i24(vz >> 5, dslab, mad_i24(vy >> 5, drow, (uint32_t)(vx >> 5) << 2)), 0, 0) << 13;
                                }
                                const uint32_t line = (((((uint32_t)vz >> 3) & 3u) << 2 | (((uint32_t)vy >> 3) & 3u)) << 2) | (((uint32_t)vx >> 3) & 3u);
                                off = cblock + (((line << 3) | sub) << 4);
                            }
                            c = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(mb, off, 0, 0));
                        }
                        if (!look) {   // out of the window before this lookup: it is mode 1's
                            marching = false;
                            idx |= kGone;
                        } else {
                            iter += 1u;
                            // u = (x&3) | (y&3) << 2 | (z&3) << 4, with z's upper bits left on top: the shifts below use the low bits only
                            const uint32_t u = ((((uint32_t)vz << 2) | ((uint32_t)vy & 3u)) << 2) | ((uint32_t)vx & 3u);
                            const uint32_t passes = (uint32_t)((((unsigned long long)c.w << 32) | c.z) >> (u & 63u)) & 1u;
                            const uint32_t lo = (c.x & 31u) | __builtin_amdgcn_ubfe(c.y, (u >> 1) & 31u, 1u);
                            bool stop = passes == 0u;
                            ref = c.x;
                            if (!stop) {
                                take_step(lo);
                                if (iter >= kMaxSteps) {
                                    // out of lookups in air or in a liquid (:220, :293): the segment ends as a hit on the voxel of the last
                                    // lookup — which for a split cell is in its brick, at the position that was looked up
                                    stop = true;
                                    ref = (int)c.x < 0 ? ((uint32_t)__builtin_amdgcn_raw_buffer_load_b16(bb, ((c.x & 0x7FFFFFFFu) + (u & 63u)) << 1, 0, 0) >> 1) << 16 : c.x;
                                }
                            }
                            marching = !stop;
                        }
                    }
                    const uint32_t n_march = (uint32_t)__popcll(__ballot(marching));
                    if (n_march == 0u || (next < n_list && 64u - n_march >= refill_at)) break;
                }
            }
            // the end states just written are read back by other lanes of this wave (same CU, same L1)
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            WINDBG(if (lds) { dbg_b0 += __builtin_amdgcn_s_memtime() - dbg_tb; dbg_left += n_left; } else dbg_b1 += __builtin_amdgcn_s_memtime() - dbg_tb;)
        };
        if constexpr (kWindow) {
            run(std::integral_constant<uint32_t, 0u>{});
            run(std::integral_constant<uint32_t, 1u>{});
        } else {
            run(std::integral_constant<uint32_t, 2u>{});
        }
    }
    WINDBG(const unsigned long long dbg_tc = __builtin_amdgcn_s_memtime();)

    // ---- between B and C: the rays that hit first, then the rays that missed (their end position is outside the world) ----
    uint32_t n_hit = 0u;
    {
        auto hit_at = [&](uint32_t i) __attribute__((always_inline)) {
            bool h = false;
            if (i < n) {
                const uint4 e = ray_end[base + i];
                const V3 pos{__uint_as_float(e.x), __uint_as_float(e.y), __uint_as_float(e.z)};
                h = !(min3_nan_ignoring(pos.x, pos.y, pos.z) < 0.0f ||
                      max(max((uint32_t)trunc2i(pos.x), (uint32_t)trunc2i(pos.y)), (uint32_t)trunc2i(pos.z)) >= P.world.size);
            }
            return h;
        };
        for (uint32_t k = 0; k * 64u < n; k++) n_hit += (uint32_t)__popcll(__ballot(hit_at(k * 64u + lane)));
        uint32_t at_hit = 0u, at_miss = n_hit;
        for (uint32_t k = 0; k * 64u < n; k++) {
            const uint32_t i = k * 64u + lane;
            const bool h = hit_at(i);
            const unsigned long long mh = __ballot(h), mm = __ballot(i < n && !h);
            if (h) order[at_hit + lanes_below(mh)] = (uint16_t)i;
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
    uint32_t n_out = 0u;   // survivors so far: the next segment's rays
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
            const uint4 e = ray_end[rec];
            st.slot = a.x;
            st.origin = V3{__uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w)};
            st.dir = V3{__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z)};
            st.rng = b.w;
            st.thr = V3{__uint_as_float(c.x), __uint_as_float(c.y), __uint_as_float(c.z)};
            // segment_end on the parked end state (a path segment's water is nobody's business: DESIGN.md, path trace)
            const V3 pos{__uint_as_float(e.x), __uint_as_float(e.y), __uint_as_float(e.z)};
            const uint32_t packed = e.w;
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
    WINDBG(asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); dbg_c += __builtin_amdgcn_s_memtime() - dbg_tc;
           if (left == 1u || n_out == 0u) {
               const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
               if (lane == 0 && blockIdx.x * NW + wave < 16384u) {
                   unsigned long long *d = &g_win_dbg[(blockIdx.x * NW + wave) * 8u];
                   d[0] = dbg_rays | ((unsigned long long)dbg_left << 32); d[1] = dbg_r0; d[2] = r1; d[3] = ((dbg_c1 - dbg_c0) >> 4) | ((dbg_a >> 4) << 32);
                   d[4] = (dbg_b0 >> 4) | ((dbg_b1 >> 4) << 32); d[5] = (dbg_c >> 4) | ((unsigned long long)(dbg_rf[0] & 0xFFFFu) << 32) | ((unsigned long long)(dbg_rf[1] & 0xFFFFu) << 48); d[0] |= 0ull; d[6] = (dbg_ws[0] & 0xFFFFu) | ((unsigned long long)(dbg_rn[0] & 0xFFFFu) << 16) | ((unsigned long long)(dbg_ws[1] & 0xFFFFu) << 32) | ((unsigned long long)(dbg_rn[1] & 0xFFFFu) << 48); d[7] = dbg_ls[0] | ((unsigned long long)dbg_ls[1] << 32);
               }
           })
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

// the primary launch of a frame whose bounce launch is the window kernel (plain frames over the derived tables)
void launch_path_primary_grouped(const FrameParams &P, hipStream_t st) {
    if (P.tiles_local == 0) return;
    const dim3 grid((P.tiles_local + 3u) / 4u), block(256);
    if (P.acc) hipLaunchKernelGGL((path_primary_kernel<0, false, false, true, true>), grid, block, lds_bytes_path(P, false), st, P);
    else hipLaunchKernelGGL((path_primary_kernel<0, false, false, false, true>), grid, block, lds_bytes_path(P, false), st, P);
}

// LDS of a launch: the window, 16 words of liquid mask and sums, a u16 list per wave
static size_t window_lds_bytes(uint32_t cells, uint32_t nw, uint32_t wave_rays) { return (size_t)cells * 16u + 16u * 4u + (size_t)nw * wave_rays * 2u; }

// waves (= regions of the primary launch) per bounce workgroup for a window shape
uint32_t window_group_regions(uint32_t shape) { return shape == 3u ? 16u : (shape == 1u || shape == 2u) ? 8u : 4u; }

// `shape`: 0 = 8 x 8 x 8 cells (32^3 voxels, 8 KiB) for 4 waves, 1 = 12 x 12 x 12 (48^3, 27 KiB) for 8, 2 = 16 x 8 x 16
// (64 x 32 x 64, 32 KiB) for 8, 3 = 16 x 16 x 16 (64^3, 64 KiB) for 16; 4 = no window: 4 waves, each with ALL its region's rays
// (`samples` x 256: the deep launch)
void launch_path_bounce_window(const FrameParams &P, uint32_t segments, uint32_t n_regions, uint32_t samples, uint32_t shape, int32_t lift, hipStream_t st) {
    if (P.tiles_local == 0 || segments == 0 || n_regions == 0 || samples == 0) return;
    const uint32_t nw = window_group_regions(shape);
    const bool deep = shape >= 4u && 256u * samples <= kDeepEntriesMax;
    const uint32_t wave_rays = deep ? 256u * samples : kWinEntries, parts = deep ? 1u : samples;
    const dim3 grid(((n_regions + nw - 1u) / nw) * parts), block(nw * 64u);
    const WindowLaunch L{P, segments, wave_rays, parts, n_regions, lift};
#define VRT_WINDOW_LAUNCH(X, Y, Z, NW)                                                                                   \
    do {                                                                                                                 \
        const size_t sh = window_lds_bytes(X * Y * Z, NW, wave_rays);                                                    \
        if (P.march_direct) {                                                                                            \
            if (sh > 48u * 1024u) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&path_bounce_window_kernel<true, X, Y, Z, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); \
            hipLaunchKernelGGL((path_bounce_window_kernel<true, X, Y, Z, NW>), grid, block, sh, st, L);                  \
        } else {                                                                                                         \
            if (sh > 48u * 1024u) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&path_bounce_window_kernel<false, X, Y, Z, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); \
            hipLaunchKernelGGL((path_bounce_window_kernel<false, X, Y, Z, NW>), grid, block, sh, st, L);                 \
        }                                                                                                                \
    } while (0)
    switch (shape) {
        case 0: VRT_WINDOW_LAUNCH(8, 8, 8, 4); break;
        case 1: VRT_WINDOW_LAUNCH(12, 12, 12, 8); break;
        case 2: VRT_WINDOW_LAUNCH(16, 8, 16, 8); break;
        case 3: VRT_WINDOW_LAUNCH(16, 16, 16, 16); break;
        default: VRT_WINDOW_LAUNCH(0, 0, 0, 4); break;
    }
#undef VRT_WINDOW_LAUNCH
}

}  // namespace vrt
