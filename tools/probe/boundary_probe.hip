// What does a kernel boundary cost behind a kernel that WRITES a lot?  A(write MB with policy P) -> B(tiny, dependent) x reps on one
// stream, timed with events; P = plain stores | non-temporal stores (nt) | sc1 (agent-scope write-through) | sc0 sc1 nt.
// If the end-of-kernel release (L2 write-back of whatever A left dirty) is a visible part of the 8-9 us between dependent launches
// of the step ("Round 5"), a write-through policy on the big writers shortens every boundary behind them.
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/probe/boundary_probe.hip -o /tmp/bp && /tmp/bp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int P>
__global__ __launch_bounds__(256) void writer(u32x4* dst, const u32x4* src, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        u32x4 v = src[i];
        v.x += 1;
        u32x4* p = dst + i;
        if (P == 0) *p = v;
        else if (P == 1) __builtin_nontemporal_store(v, p);
        else if (P == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
        else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
    }
}
__global__ void tiny(const u32x4* a, unsigned* out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = a[0].x;
}

#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err_)); return 1; } } while (0)

template <int P>
int run(const char* name, u32x4* dst, u32x4* src, size_t n16, unsigned* out, int reps) {
    hipEvent_t s, e;
    CK(hipEventCreate(&s)); CK(hipEventCreate(&e));
    for (int pass = 0; pass < 2; ++pass) {          // pass 0: A only; pass 1: A + dependent tiny B
        for (int w = 0; w < 3; ++w) { hipLaunchKernelGGL(writer<P>, dim3(2048), dim3(256), 0, 0, dst, src, n16); }
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(s, 0));
        for (int r = 0; r < reps; ++r) {
            hipLaunchKernelGGL(writer<P>, dim3(2048), dim3(256), 0, 0, dst, src, n16);
            if (pass) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, 0, dst, out);
        }
        CK(hipEventRecord(e, 0));
        CK(hipEventSynchronize(e));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, s, e));
        printf("%-14s %6.1f MB  %s: %8.2f us per iteration\n", name, n16 * 16 / 1e6, pass ? "A + tiny B" : "A alone   ", ms / reps * 1e3);
    }
    return 0;
}

int main() {
    const size_t sizes[] = {8u << 20, 57u << 20, 256u << 20};
    for (size_t bytes : sizes) {
        const size_t n16 = bytes / 16;
        u32x4 *dst, *src;
        unsigned* out;
        CK(hipMalloc(&dst, bytes)); CK(hipMalloc(&src, bytes)); CK(hipMalloc(&out, 64));
        CK(hipMemset(src, 1, bytes));
        if (run<0>("plain", dst, src, n16, out, 50)) return 1;
        if (run<1>("nontemporal", dst, src, n16, out, 50)) return 1;
        if (run<2>("sc1", dst, src, n16, out, 50)) return 1;
        if (run<3>("sc0 sc1 nt", dst, src, n16, out, 50)) return 1;
        CK(hipFree(dst)); CK(hipFree(src)); CK(hipFree(out));
    }
    return 0;
}
