// Depthwise Conv1d (groups == channels) forward / data gradient / weight gradient over channels-last
// activations (gfx950).  Jasper's separable blocks (jasper.py:318-341) put a depthwise K = 33..75 conv in
// front of a 1x1 pointwise conv: 2*K FLOP per element against 4 bytes of HBM traffic, i.e. HBM / cache
// bound, so this is deliberately NOT reshaped into a GEMM: one lane owns 8 adjacent channels (16 B) of one
// output frame, taps are re-read through L1/L2 (adjacent lanes and frames share them), weights are tap-major
// [K][C] fp32 so the 8 channels of a tap are one 32-byte load.  All arithmetic is fp32; activations are bf16
// (hi [+ lo] split pairs in the fp32 parity mode).
//
// Replaces nn.Conv1d(groups=in_channels) inside MaskedConv1d (jasper.py:96-105,127,319-330) forward and its
// autograd gradients, including the masked_fill of the NEXT MaskedConv1d on the output (jasper.py:116-119).
#include "common.h"

namespace {

__device__ __forceinline__ void ld8(const bf16_raw* hi, const bf16_raw* lo, int64_t off, float v[8]) {
    const u16x8 a = *reinterpret_cast<const u16x8*>(hi + off);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = bf16_bits_to_f32(a[j]);
    if (lo) {
        const u16x8 b = *reinterpret_cast<const u16x8*>(lo + off);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += bf16_bits_to_f32(b[j]);
    }
}
__device__ __forceinline__ void ldw8(const float* w, int64_t off, float v[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(w + off), b = *reinterpret_cast<const f32x4*>(w + off + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = a[j]; v[4 + j] = b[j]; }
}
__device__ __forceinline__ void st8(bf16_raw* hi, bf16_raw* lo, int64_t off, const float v[8]) {
    u16x8 h, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        bf16_raw a, b;
        f32_split_bf16(v[j], a, b);
        h[j] = a; l[j] = b;
    }
    *reinterpret_cast<u16x8*>(hi + off) = h;
    if (lo) *reinterpret_cast<u16x8*>(lo + off) = l;
}

// y[n][t][c] = sum_k w[k][c] * xp[n][t*s + k*d][c];  rows t >= lens[n] are written as zero
__global__ __launch_bounds__(256) void dw_fwd_kernel(const bf16_raw* x_hi, const bf16_raw* x_lo, int R, const float* w,
                                                      bf16_raw* y_hi, bf16_raw* y_lo, int N, int Tout, int C, int K, int s,
                                                      int d, const int32_t* lens) {
    const int G = C >> 3;
    const unsigned total = (unsigned)N * Tout * G;
    for (unsigned it = blockIdx.x * 256u + threadIdx.x; it < total; it += gridDim.x * 256u) {
        const unsigned row = it / (unsigned)G;
        const int cg = (int)(it - row * G);
        const int n = (int)(row / (unsigned)Tout);
        const int t = (int)(row - (unsigned)n * Tout);
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        if (!lens || t < lens[n]) {
            const int64_t xb = ((int64_t)n * R + (int64_t)t * s) * C + cg * 8;
            for (int k = 0; k < K; ++k) {
                float xv[8], wv[8];
                ld8(x_hi, x_lo, xb + (int64_t)k * d * C, xv);
                ldw8(w, (int64_t)k * C + cg * 8, wv);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += wv[j] * xv[j];
            }
        }
        st8(y_hi, y_lo, (int64_t)row * C + cg * 8, acc);
    }
}

// dxp[n][v][c] = sum_k w[k][c] * dy[n][v - k*d][c]   (stride 1), dy rows outside [0, min(Tout, lens[n])) count as 0
template <bool GF32, bool OF32>
__global__ __launch_bounds__(256) void dw_dgrad_kernel(const void* dy, int dy_rows, const float* w, void* dxp, int N, int Tp,
                                                        int Tout, int C, int K, int d, const int32_t* lens) {
    const int G = C >> 3;
    const unsigned total = (unsigned)N * Tp * G;
    for (unsigned it = blockIdx.x * 256u + threadIdx.x; it < total; it += gridDim.x * 256u) {
        const unsigned row = it / (unsigned)G;
        const int cg = (int)(it - row * G);
        const int n = (int)(row / (unsigned)Tp);
        const int v = (int)(row - (unsigned)n * Tp);
        int lim = Tout;
        if (lens && lens[n] < lim) lim = lens[n];
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        for (int k = 0; k < K; ++k) {
            const int t = v - k * d;
            if (t >= 0 && t < lim) {
                float gv[8], wv[8];
                const int64_t off = ((int64_t)n * dy_rows + t) * C + cg * 8;
                if (GF32) ldw8(reinterpret_cast<const float*>(dy), off, gv);
                else ld8(reinterpret_cast<const bf16_raw*>(dy), nullptr, off, gv);
                ldw8(w, (int64_t)k * C + cg * 8, wv);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += wv[j] * gv[j];
            }
        }
        const int64_t o = (int64_t)row * C + cg * 8;
        if (OF32) {
            float* p = reinterpret_cast<float*>(dxp) + o;
            *reinterpret_cast<f32x4*>(p) = f32x4{acc[0], acc[1], acc[2], acc[3]};
            *reinterpret_cast<f32x4*>(p + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
        } else {
            st8(reinterpret_cast<bf16_raw*>(dxp), nullptr, o, acc);
        }
    }
}

// dw[k][c] += sum_{n, t < min(Tout, lens[n])} dy[n][t][c] * xp[n][t*s + k*d][c]
// grid (K, row chunks); a thread owns one channel group for its rows; LDS reduce, then fp32 atomics.
constexpr int WG_ROWS = 256;
template <bool GF32>
__global__ __launch_bounds__(256) void dw_wgrad_kernel(const void* dy, int dy_rows, const bf16_raw* x_hi,
                                                        const bf16_raw* x_lo, int R, float* dw, int N, int Tout, int C,
                                                        int s, int d, const int32_t* lens) {
    extern __shared__ float red[];                 // [RPB][C]
    const int G = C >> 3;
    const int RPB = 256 / G;
    const int tid = threadIdx.x;
    const int rr = tid / G, cg = tid - rr * G;
    const int k = blockIdx.x;
    const int64_t rows = (int64_t)N * Tout;
    const int64_t row0 = (int64_t)blockIdx.y * WG_ROWS;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    if (rr < RPB) {
        int64_t rend = row0 + WG_ROWS;
        if (rend > rows) rend = rows;
        for (int64_t row = row0 + rr; row < rend; row += RPB) {
            const int n = (int)(row / Tout), t = (int)(row - (int64_t)n * Tout);
            if (lens && t >= lens[n]) continue;
            float gv[8], xv[8];
            const int64_t goff = ((int64_t)n * dy_rows + t) * C + cg * 8;
            if (GF32) ldw8(reinterpret_cast<const float*>(dy), goff, gv);
            else ld8(reinterpret_cast<const bf16_raw*>(dy), nullptr, goff, gv);
            ld8(x_hi, x_lo, ((int64_t)n * R + (int64_t)t * s + (int64_t)k * d) * C + cg * 8, xv);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += gv[j] * xv[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) red[rr * C + cg * 8 + j] = acc[j];
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        float a = 0.f;
        for (int q = 0; q < RPB; ++q) a += red[q * C + c];
        atomicAdd(dw + (int64_t)k * C + c, a);
    }
}

int blocks_for(int64_t items) {
    int64_t b = (items + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace

extern "C" int w2l_dwconv_fwd(const void* x_hi, const void* x_lo, int x_rows, const float* w, void* y_hi, void* y_lo, int N,
                              int Tout, int C, int K, int stride, int dil, const int32_t* lens, void* stream) {
    W2L_CHECK_ARG(x_hi && w && y_hi, "dwconv_fwd: null pointer");
    W2L_CHECK_ARG(N > 0 && Tout > 0 && C > 0 && C % 8 == 0 && K > 0 && stride > 0 && dil > 0, "dwconv_fwd: bad sizes");
    W2L_CHECK_ARG((int64_t)(Tout - 1) * stride + (int64_t)(K - 1) * dil < x_rows, "dwconv_fwd: padded input too short");
    W2L_CHECK_ARG((int64_t)N * Tout * (C / 8) < (1LL << 31), "dwconv_fwd: tensor too large");
    hipLaunchKernelGGL(dw_fwd_kernel, dim3(blocks_for((int64_t)N * Tout * (C / 8))), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_raw*)x_hi, (const bf16_raw*)x_lo, x_rows, w, (bf16_raw*)y_hi, (bf16_raw*)y_lo, N, Tout, C, K,
                       stride, dil, lens);
    W2L_CHECK_LAUNCH();
    return 0;
}

extern "C" int w2l_dwconv_dgrad(const void* dy, int dy_f32, int dy_rows, const float* w, void* dxp, int dxp_f32, int N, int Tp,
                                int Tout, int C, int K, int dil, const int32_t* lens, void* stream) {
    W2L_CHECK_ARG(dy && w && dxp, "dwconv_dgrad: null pointer");
    W2L_CHECK_ARG(N > 0 && Tp > 0 && Tout > 0 && Tout <= dy_rows && C > 0 && C % 8 == 0 && K > 0 && dil > 0,
                  "dwconv_dgrad: bad sizes");
    W2L_CHECK_ARG((int64_t)N * Tp * (C / 8) < (1LL << 31), "dwconv_dgrad: tensor too large");
    const dim3 grid(blocks_for((int64_t)N * Tp * (C / 8))), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (dy_f32 && dxp_f32) hipLaunchKernelGGL((dw_dgrad_kernel<true, true>), grid, block, 0, st, dy, dy_rows, w, dxp, N, Tp, Tout, C, K, dil, lens);
    else if (dy_f32) hipLaunchKernelGGL((dw_dgrad_kernel<true, false>), grid, block, 0, st, dy, dy_rows, w, dxp, N, Tp, Tout, C, K, dil, lens);
    else if (dxp_f32) hipLaunchKernelGGL((dw_dgrad_kernel<false, true>), grid, block, 0, st, dy, dy_rows, w, dxp, N, Tp, Tout, C, K, dil, lens);
    else hipLaunchKernelGGL((dw_dgrad_kernel<false, false>), grid, block, 0, st, dy, dy_rows, w, dxp, N, Tp, Tout, C, K, dil, lens);
    W2L_CHECK_LAUNCH();
    return 0;
}

extern "C" int w2l_dwconv_wgrad(const void* dy, int dy_f32, int dy_rows, const void* x_hi, const void* x_lo, int x_rows,
                                float* dw, int N, int Tout, int C, int K, int stride, int dil, const int32_t* lens,
                                void* stream) {
    W2L_CHECK_ARG(dy && x_hi && dw, "dwconv_wgrad: null pointer");
    W2L_CHECK_ARG(N > 0 && Tout > 0 && Tout <= dy_rows && C > 0 && C % 8 == 0 && C <= 2048 && K > 0 && stride > 0 && dil > 0,
                  "dwconv_wgrad: bad sizes");
    W2L_CHECK_ARG((int64_t)(Tout - 1) * stride + (int64_t)(K - 1) * dil < x_rows, "dwconv_wgrad: padded input too short");
    const int G = C / 8, RPB = 256 / G;
    const dim3 grid(K, (unsigned)(((int64_t)N * Tout + WG_ROWS - 1) / WG_ROWS)), block(256);
    const size_t lds = (size_t)RPB * C * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    if (dy_f32) hipLaunchKernelGGL((dw_wgrad_kernel<true>), grid, block, lds, st, dy, dy_rows, (const bf16_raw*)x_hi, (const bf16_raw*)x_lo, x_rows, dw, N, Tout, C, stride, dil, lens);
    else hipLaunchKernelGGL((dw_wgrad_kernel<false>), grid, block, lds, st, dy, dy_rows, (const bf16_raw*)x_hi, (const bf16_raw*)x_lo, x_rows, dw, N, Tout, C, stride, dil, lens);
    W2L_CHECK_LAUNCH();
    return 0;
}
