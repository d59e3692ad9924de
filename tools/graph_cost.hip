// tools/graph_cost.hip — what a captured graph would save the host per device and frame (round 3's review asked for one
// hipGraph per device and frame shape in the multi-device context).  A shard device's frame is: wait for the slot's
// `consumed` event, one kernel launch with this frame's uniforms, record `done`.  Host microseconds per frame, 4000 frames:
//   direct      hipStreamWaitEvent + hipLaunchKernelGGL (560-byte by-value argument) + hipEventRecord
//   graph       hipGraphLaunch of {event-wait node, kernel node, event-record node}, the uniforms read from mapped host memory
//   graph+set   the same with hipGraphExecKernelNodeSetParams per frame (uniforms in the kernel arguments)
// hipcc --offload-arch=gfx950 -O2 tools/graph_cost.hip -o /tmp/graph_cost
#include <hip/hip_runtime.h>
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
    hipEvent_t consumed, done;
    CK(hipEventCreateWithFlags(&consumed, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&done, hipEventDisableTiming));
    float *out;
    CK(hipMalloc(&out, 64));
    Params *mapped, *mapped_dev;
    CK(hipHostMalloc((void **)&mapped, sizeof(Params), hipHostMallocMapped));
    CK(hipHostGetDevicePointer((void **)&mapped_dev, mapped, 0));
    Params p;
    memset(&p, 0, sizeof p);
    CK(hipEventRecord(consumed, other));
    CK(hipDeviceSynchronize());

    // direct
    for (int warm = 0; warm < 2; warm++) {
        const double t0 = now_us();
        for (int i = 0; i < N; i++) {
            p.v[0] = (float)i;
            CK(hipStreamWaitEvent(st, consumed, 0));
            hipLaunchKernelGGL(k_by_value, dim3(64), dim3(256), 0, st, p, out);
            CK(hipEventRecord(done, st));
        }
        const double t1 = now_us();
        CK(hipStreamSynchronize(st));
        if (warm) printf("direct     %6.2f us of host time per frame (wait + launch + record), %6.2f us per frame to completion\n", (t1 - t0) / N, (now_us() - t0) / N);
    }

    // graph, uniforms through mapped memory
    hipGraph_t g;
    CK(hipGraphCreate(&g, 0));
    hipGraphNode_t n_wait, n_kernel, n_rec;
    CK(hipGraphAddEventWaitNode(&n_wait, g, nullptr, 0, consumed));
    void *args_ptr[2] = {(void *)&mapped_dev, (void *)&out};
    hipKernelNodeParams kp;
    memset(&kp, 0, sizeof kp);
    kp.func = (void *)k_by_pointer;
    kp.gridDim = dim3(64);
    kp.blockDim = dim3(256);
    kp.kernelParams = args_ptr;
    CK(hipGraphAddKernelNode(&n_kernel, g, &n_wait, 1, &kp));
    CK(hipGraphAddEventRecordNode(&n_rec, g, &n_kernel, 1, done));
    hipGraphExec_t ge;
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int warm = 0; warm < 2; warm++) {
        const double t0 = now_us();
        for (int i = 0; i < N; i++) {
            mapped->v[0] = (float)i;
            CK(hipGraphLaunch(ge, st));
        }
        const double t1 = now_us();
        CK(hipStreamSynchronize(st));
        if (warm) printf("graph      %6.2f us of host time per frame (hipGraphLaunch of 3 nodes, uniforms in mapped host memory), %6.2f us per frame to completion\n", (t1 - t0) / N, (now_us() - t0) / N);
    }

    // graph, uniforms in the kernel arguments: set per frame
    hipGraph_t g2;
    CK(hipGraphCreate(&g2, 0));
    hipGraphNode_t m_wait, m_kernel, m_rec;
    CK(hipGraphAddEventWaitNode(&m_wait, g2, nullptr, 0, consumed));
    void *args_val[2] = {(void *)&p, (void *)&out};
    hipKernelNodeParams kv;
    memset(&kv, 0, sizeof kv);
    kv.func = (void *)k_by_value;
    kv.gridDim = dim3(64);
    kv.blockDim = dim3(256);
    kv.kernelParams = args_val;
    CK(hipGraphAddKernelNode(&m_kernel, g2, &m_wait, 1, &kv));
    CK(hipGraphAddEventRecordNode(&m_rec, g2, &m_kernel, 1, done));
    hipGraphExec_t ge2;
    CK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
    for (int warm = 0; warm < 2; warm++) {
        const double t0 = now_us();
        for (int i = 0; i < N; i++) {
            p.v[0] = (float)i;
            CK(hipGraphExecKernelNodeSetParams(ge2, m_kernel, &kv));
            CK(hipGraphLaunch(ge2, st));
        }
        const double t1 = now_us();
        CK(hipStreamSynchronize(st));
        if (warm) printf("graph+set  %6.2f us of host time per frame (hipGraphExecKernelNodeSetParams + hipGraphLaunch), %6.2f us per frame to completion\n", (t1 - t0) / N, (now_us() - t0) / N);
    }
    // a lone kernel launch, for scale
    {
        const double t0 = now_us();
        for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_by_value, dim3(64), dim3(256), 0, st, p, out);
        const double t1 = now_us();
        CK(hipStreamSynchronize(st));
        printf("launch     %6.2f us of host time per kernel launch alone, %6.2f us per launch to completion\n", (t1 - t0) / N, (now_us() - t0) / N);
    }
    return 0;
}
