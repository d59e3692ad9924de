// tools/issue_cost.hip — what a shard device's frame costs its issuing thread, form by form (round 4's review, item 3: <= 5 us per
// shard device asked).  A shard's frame is: wait until the root has consumed the message slot, one kernel launch with this frame's
// uniforms, record `done`.  Host microseconds per frame over 4000 frames on one stream:
//   direct          hipStreamWaitEvent + hipLaunchKernelGGL (560-byte by-value argument) + hipEventRecord          (what ships)
//   stop-event      hipStreamWaitEvent + hipExtLaunchKernelGGL(..., nullptr, done): the dispatch's own completion signal is `done`
//   query + stop    hipEventQuery(consumed) — it completed long ago, so no wait is enqueued — + the launch with its stop event
//   by pointer      the same, the uniforms read from mapped host memory (the launch carries two pointers)
//   launch only     hipLaunchKernelGGL alone (the floor)
// hipcc --offload-arch=gfx950 -O2 tools/issue_cost.hip -o /tmp/issue_cost
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstring>

struct Params { float v[140]; };   // the size of vrt::FrameParams
__global__ void k_by_value(Params p, float *out) { if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = p.v[0] + p.v[139]; }
__global__ void k_by_pointer(const Params *p, float *out) { if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = p->v[0] + p->v[139]; }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    const int N = 4000;
    hipStream_t st, other;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&other, hipStreamNonBlocking));
    hipEvent_t consumed, done, done_timed;
    CK(hipEventCreateWithFlags(&consumed, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&done, hipEventDisableTiming));
    CK(hipEventCreate(&done_timed));
    float *out;
    CK(hipMalloc(&out, 64));
    Params *mapped, *mapped_dev;
    CK(hipHostMalloc((void **)&mapped, sizeof(Params), hipHostMallocMapped));
    CK(hipHostGetDevicePointer((void **)&mapped_dev, mapped, 0));
    Params p;
    memset(&p, 0, sizeof p);
    CK(hipEventRecord(consumed, other));
    CK(hipDeviceSynchronize());
    auto run = [&](const char *name, auto body) -> int {
        for (int warm = 0; warm < 2; warm++) {
            const double t0 = now_us();
            for (int i = 0; i < N; i++) { p.v[0] = (float)i; mapped->v[0] = (float)i; if (body()) return 1; }
            const double t1 = now_us();
            CK(hipStreamSynchronize(st));
            if (warm) printf("%-14s %6.2f us of host time per frame, %6.2f us per frame to completion\n", name, (t1 - t0) / N, (now_us() - t0) / N);
        }
        return 0;
    };
    if (run("direct", [&]() -> int { CK(hipStreamWaitEvent(st, consumed, 0)); hipLaunchKernelGGL(k_by_value, dim3(64), dim3(256), 0, st, p, out); CK(hipEventRecord(done, st)); return 0; })) return 1;
    if (run("stop-event", [&]() -> int { CK(hipStreamWaitEvent(st, consumed, 0)); hipExtLaunchKernelGGL(k_by_value, dim3(64), dim3(256), 0, st, nullptr, done, 0, p, out); return 0; })) return 1;
    if (run("stop-ev timed", [&]() -> int { CK(hipStreamWaitEvent(st, consumed, 0)); hipExtLaunchKernelGGL(k_by_value, dim3(64), dim3(256), 0, st, nullptr, done_timed, 0, p, out); return 0; })) return 1;
    if (run("query + stop", [&]() -> int { if (hipEventQuery(consumed) != hipSuccess) CK(hipStreamWaitEvent(st, consumed, 0)); hipExtLaunchKernelGGL(k_by_value, dim3(64), dim3(256), 0, st, nullptr, done, 0, p, out); return 0; })) return 1;
    if (run("by pointer", [&]() -> int { if (hipEventQuery(consumed) != hipSuccess) CK(hipStreamWaitEvent(st, consumed, 0)); hipExtLaunchKernelGGL(k_by_pointer, dim3(64), dim3(256), 0, st, nullptr, done, 0, (const Params *)mapped_dev, out); return 0; })) return 1;
    if (run("launch only", [&]() -> int { hipLaunchKernelGGL(k_by_value, dim3(64), dim3(256), 0, st, p, out); return 0; })) return 1;
    // is the stop event usable from another stream (the root's stream waits for it)?
    hipExtLaunchKernelGGL(k_by_value, dim3(64), dim3(256), 0, st, nullptr, done, 0, p, out);
    CK(hipStreamWaitEvent(other, done, 0));
    CK(hipStreamSynchronize(other));
    printf("a stop event is waited for by another stream: ok\n");
    return 0;
}
