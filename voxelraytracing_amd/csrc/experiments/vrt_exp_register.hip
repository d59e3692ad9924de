// experiments/vrt_exp_register.hip — fills the hooks of vrt_exp.h when the library is tools/ab/libvrt_exp.so, and marks that
// build for the tests (vrt_experiments_build: tests/test_gpu_parity.py::needs_experiments).
#include "../vrt_exp.h"

namespace vrt {
void launch_path_primary_grouped(const FrameParams &P, hipStream_t st);
uint32_t window_group_regions(uint32_t shape);
void launch_path_bounce_window(const FrameParams &P, uint32_t segments, uint32_t n_regions, uint32_t samples, uint32_t shape, int32_t lift, hipStream_t st);

namespace {
struct Register {
    Register() {
        g_exp.path_primary_grouped = launch_path_primary_grouped;
        g_exp.window_group_regions = window_group_regions;
        g_exp.path_bounce_window = launch_path_bounce_window;
    }
} g_register;
}  // namespace
}  // namespace vrt

extern "C" int vrt_experiments_build(void) { return 1; }
