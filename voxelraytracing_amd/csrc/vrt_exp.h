// vrt_exp.h — where the experiments build plugs in.
//
// The structures that were built, measured and not chosen (profiles/DECISIONS.md: the persistent grid, the persistent path kernel,
// the LDS pool over cell grid + bricks with its straggler chain, the moving-camera tile order, the LDS-staged window bounce launch)
// live in csrc/experiments/*.hip and are linked into tools/ab/libvrt_exp.so only (make experiments).  The product's translation
// units know them through these hooks: all null in libvrt.so — every switch that would select one is then without effect —
// and filled in by experiments/vrt_exp_register.hip's initialiser in the experiments build.  No #ifdef on either side.
#pragma once

#include <hip/hip_runtime.h>

#include "vrt_device.h"

namespace vrt {

struct ExpHooks {
    // primary + shadow as a persistent grid over per-XCD tile queues (vrt_render_opts.variant = 4)
    void (*primary_shadow_persistent)(const FrameParams &P, uint32_t *heads, uint32_t n_cus, hipStream_t st, hipEvent_t e0, hipEvent_t e1) = nullptr;
    // the tile order of a moving view as round 4's six small launches behind every frame (VRT_TILE_ORDER_MOVING=6)
    void (*tile_order_moving)(uint32_t *cost, uint32_t tiles_x, uint32_t tiles_y, uint32_t shift, uint32_t radius, uint32_t *scratch, uint32_t *order, hipStream_t st) = nullptr;
    // ... the shipping one-launch order made beside the next frame on a side stream, every frame (VRT_TILE_ORDER_MOVING=2)
    bool tile_order_beside = false;
    // the path trace as one launch of persistent waves (VRT_PATH_PERSISTENT=1)
    void (*path_persistent)(const FrameParams &P, uint32_t *heads, uint32_t n_cus, hipStream_t st) = nullptr;
    // the round-2 pool kernel over cell grid + bricks, with its straggler chain (VRT_PATH_CELLS=0, VRT_PATH_POOL_CHAIN=1)
    void (*path_bounce_pool)(const FrameParams &P, bool continuations, uint32_t refill_at, uint32_t eject_at, hipStream_t st) = nullptr;
    // the shipping bounce launch with its probes compiled in (experiments/vrt_path_cells_probe.hip: variant builds only)
    void (*path_bounce_cells)(const FrameParams &P, uint32_t refill_at, uint32_t segments, uint32_t lds_pad, hipStream_t st) = nullptr;
    // the bounce launch over LDS-staged windows of march cells (VRT_PATH_WINDOW=1; round 5)
    void (*path_primary_grouped)(const FrameParams &P, hipStream_t st) = nullptr;
    uint32_t (*window_group_regions)(uint32_t shape) = nullptr;
    void (*path_bounce_window)(const FrameParams &P, uint32_t segments, uint32_t n_regions, uint32_t samples, uint32_t shape, int32_t lift, hipStream_t st) = nullptr;
};

extern __attribute__((visibility("hidden"))) ExpHooks g_exp;   // (vrt_frames.hip)

}  // namespace vrt
