// Empirical semantics of ds_read_b64_tr_b8 on gfx950 (the ISA manual is not in this image): fill LDS with a byte pattern,
// give every lane the address (row = 8*(l/16) + (l%16)/2, 8-byte column chunk = l%2) of a [32 rows][16 bytes] image and
// print what each lane receives.  hipcc --offload-arch=gfx950 tr8_probe.hip -o tr8_probe && ./tr8_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v2i __attribute__((ext_vector_type(2)));
__global__ void k(unsigned char* out, int variant) {
    __shared__ unsigned char s[32 * 16];
    for (int i = threadIdx.x; i < 32 * 16; i += 64) s[i] = (unsigned char)i;     // s[r][c] = 16 r + c (mod 256)
    __syncthreads();
    const int l = threadIdx.x, g = l >> 4, i = l & 15;
    int row, chunk;
    if (variant == 0) { row = g * 8 + (i >> 1); chunk = i & 1; }       // two lanes per row
    else { row = g * 8 + (i & 7); chunk = i >> 3; }                     // lanes 0-7 left chunk of rows 0-7, 8-15 right chunk
    const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)s + row * 16 + chunk * 8;
    v2i v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) v2i*)(size_t)addr);
    ((v2i*)out)[l] = v;
}
int main() {
    unsigned char* d; unsigned char h[512];
    hipMalloc(&d, 512);
    for (int variant = 0; variant < 2; ++variant) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, variant);
        hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
        printf("variant %d\n", variant);
        for (int l = 0; l < 64; ++l) {
            printf("lane %2d:", l);
            for (int b = 0; b < 8; ++b) printf(" (r%2d,c%2d)", h[l * 8 + b] >> 4, h[l * 8 + b] & 15);
            printf("\n");
        }
    }
    return 0;
}
