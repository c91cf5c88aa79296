// What do the stream-order primitives cost ON THE DEVICE TIMELINE of the stream that carries them?  Chains of small dependent kernels
// (20 us each, chip-filling) on stream 0, timed with events, with between every two kernels: nothing | hipEventRecord(ev, s0) |
// record + hipStreamWaitEvent(s1, ev) (a fork: s1 idle) | a wait on s0 for an event another stream recorded long ago (satisfied).
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/probe/event_probe.hip -o /tmp/ep && /tmp/ep
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ __launch_bounds__(256) void work(float* p, int iters) {
    float v = p[blockIdx.x * 256 + threadIdx.x];
    for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
    p[blockIdx.x * 256 + threadIdx.x] = v;
}
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err_)); return 1; } } while (0)

int main(int argc, char**) {
    float* p;
    CK(hipMalloc(&p, 2048 * 256 * 4));
    CK(hipMemset(p, 0, 2048 * 256 * 4));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));   // (as torch creates its streams: no implicit order with the null stream)
    hipEvent_t t0, t1, ev[64], old;
    CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    const bool dev_release = argc > 1;          // any argument: events created with hipEventReleaseToDevice
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming | (dev_release ? hipEventReleaseToDevice : 0)));
    printf("events: hipEventDisableTiming%s\n", dev_release ? " | hipEventReleaseToDevice" : "");
    CK(hipEventCreateWithFlags(&old, hipEventDisableTiming));
    hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, s2, p, 10);
    CK(hipEventRecord(old, s2));
    CK(hipDeviceSynchronize());
    const int n = 200;
    for (int iters : {2000, 200}) {
        for (int mode = 0; mode < 5; ++mode) {
            for (int pass = 0; pass < 2; ++pass) {
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(t0, 0));
                for (int i = 0; i < n; ++i) {
                    hipLaunchKernelGGL(work, dim3(2048), dim3(256), 0, 0, p, iters);
                    if (mode == 1) CK(hipEventRecord(ev[i & 63], 0));
                    if (mode == 2) { CK(hipEventRecord(ev[i & 63], 0)); CK(hipStreamWaitEvent(s1, ev[i & 63], 0)); }
                    if (mode == 3) CK(hipStreamWaitEvent(0, old, 0));
                    if (mode == 4) { CK(hipEventRecord(ev[i & 63], 0)); CK(hipStreamWaitEvent(s1, ev[i & 63], 0));
                                     hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, s1, p + 1024 * 256, 50); }
                }
                CK(hipEventRecord(t1, 0));
                CK(hipEventSynchronize(t1));
                float ms = 0;
                CK(hipEventElapsedTime(&ms, t0, t1));
                if (pass) printf("kernel iters %4d  %-52s %7.2f us per kernel\n", iters,
                                 mode == 0 ? "back to back" : mode == 1 ? "+ event record on the stream" :
                                 mode == 2 ? "+ record, another (idle) stream waits" : mode == 3 ? "+ wait for an event that fired long ago" :
                                 "+ record, other stream waits and runs a small kernel", ms / n * 1e3);
            }
        }
    }
    return 0;
}
