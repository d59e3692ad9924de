// experiments/vrt_exp_register.hip — fills the hooks of vrt_exp.h when the library is tools/ab/libvrt_exp.so, and marks that
// build for the tests (vrt_experiments_build: tests/test_gpu_parity.py::needs_experiments).
#include "../vrt_exp.h"

namespace vrt {
void launch_primary_shadow_persistent(const FrameParams &P, uint32_t *heads, uint32_t n_cus, hipStream_t st, hipEvent_t e0, hipEvent_t e1);
void launch_tile_order_moving(uint32_t *cost, uint32_t tiles_x, uint32_t tiles_y, uint32_t shift, uint32_t radius, uint32_t *scratch, uint32_t *order, hipStream_t st);
void launch_path_persistent(const FrameParams &P, uint32_t *heads, uint32_t n_cus, hipStream_t st);
void launch_path_bounce_pool(const FrameParams &P, bool continuations, uint32_t refill_at, uint32_t eject_at, hipStream_t st);
void launch_path_primary_grouped(const FrameParams &P, hipStream_t st);
uint32_t window_group_regions(uint32_t shape);
void launch_path_bounce_window(const FrameParams &P, uint32_t segments, uint32_t n_regions, uint32_t samples, uint32_t shape, int32_t lift, hipStream_t st);

namespace {
struct Register {
    Register() {
        g_exp.primary_shadow_persistent = launch_primary_shadow_persistent;
        g_exp.tile_order_moving = launch_tile_order_moving;
        g_exp.tile_order_beside = true;
        g_exp.path_persistent = launch_path_persistent;
        g_exp.path_bounce_pool = launch_path_bounce_pool;
        g_exp.path_primary_grouped = launch_path_primary_grouped;
        g_exp.window_group_regions = window_group_regions;
        g_exp.path_bounce_window = launch_path_bounce_window;
    }
} g_register;
}  // namespace
}  // namespace vrt

extern "C" int vrt_experiments_build(void) { return 1; }
