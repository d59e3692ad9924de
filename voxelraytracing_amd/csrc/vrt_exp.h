// vrt_exp.h — where the experiments build plugs in.
//
// The one structure of round 5 that was built, measured and not chosen and is kept one more round — the LDS-staged window bounce
// launch (profiles/r05_window_ab.txt) — lives in csrc/experiments/vrt_path_window.hip and is linked into tools/ab/libvrt_exp.so
// only (make experiments).  The product's translation units know it through these hooks: all null in libvrt.so — the switch
// that would select it is then without effect — and filled in by experiments/vrt_exp_register.hip's initialiser in the
// experiments build.  No #ifdef on either side.  (The older closed structures — persistent grid, persistent path kernel, LDS pool
// over cell grid + bricks with its straggler chain, a tile order per moving frame — were deleted in round 6; their evidence stays
// in profiles/DECISIONS.md and their sources in the history before that commit.)
#pragma once

#include <hip/hip_runtime.h>

#include "vrt_device.h"

namespace vrt {

struct ExpHooks {
    // the bounce launch over LDS-staged windows of march cells (VRT_PATH_WINDOW=1; round 5)
    void (*path_primary_grouped)(const FrameParams &P, hipStream_t st) = nullptr;
    uint32_t (*window_group_regions)(uint32_t shape) = nullptr;
    void (*path_bounce_window)(const FrameParams &P, uint32_t segments, uint32_t n_regions, uint32_t samples, uint32_t shape, int32_t lift, hipStream_t st) = nullptr;
};

extern __attribute__((visibility("hidden"))) ExpHooks g_exp;   // (vrt_frames.hip)

}  // namespace vrt
