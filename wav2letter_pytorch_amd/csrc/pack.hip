// Layout kernels at the edges of the hot path (HBM-bound, tiled through LDS so
// both the read and the write side are coalesced):
//   * fp32 master weights -> bf16 operand layouts for the forward and dgrad GEMMs,
//   * fp32 [N][C][T] spectrograms -> padded channels-last bf16,
//   * fp32 classifier gradient -> zero-haloed bf16 + bias-gradient column sums.
#include "common.h"

namespace {

__device__ __forceinline__ void put_split(bf16_raw* hi, bf16_raw* lo, int64_t off, float v) {
    bf16_raw h, l;
    f32_split_bf16(v, h, l);
    hi[off] = h;
    if (lo) lo[off] = l;
}

__global__ void pack_weights_kernel(const float* w, int64_t s_co, int64_t s_ci, int64_t s_kw, int Cout, int Cin,
                                    int Kw, int CoutP, int CinP, bf16_raw* fwd_hi, bf16_raw* fwd_lo,
                                    bf16_raw* dgr_hi, bf16_raw* dgr_lo) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32, kw = blockIdx.z;
    // gather with the lane index on whichever source axis is denser
    const bool ci_fast = s_ci <= s_co;
    for (int j = ty; j < 32; j += 8) {
        const int co = ci_fast ? co0 + j : co0 + tx;
        const int ci = ci_fast ? ci0 + tx : ci0 + j;
        float v = 0.f;
        if (co < Cout && ci < Cin) v = w[co * s_co + ci * s_ci + kw * s_kw];
        if (ci_fast) tile[j][tx] = v; else tile[tx][j] = v;          // tile[co_l][ci_l]
    }
    __syncthreads();
    if (fwd_hi) {
        for (int j = ty; j < 32; j += 8) {
            const int co = co0 + j, ci = ci0 + tx;
            if (co < CoutP && ci < CinP) put_split(fwd_hi, fwd_lo, ((int64_t)kw * CoutP + co) * CinP + ci, tile[j][tx]);
        }
    }
    if (dgr_hi) {
        for (int j = ty; j < 32; j += 8) {
            const int ci = ci0 + j, co = co0 + tx;
            if (co < CoutP && ci < CinP)
                put_split(dgr_hi, dgr_lo, ((int64_t)(Kw - 1 - kw) * CinP + ci) * CoutP + co, tile[tx][j]);
        }
    }
}

// Fused SGD(momentum, nesterov, weight decay) update of a tap-major conv weight [Kw][Cout][Cin] (fp32 master,
// gradient and momentum buffer in the same dense layout) that also emits the bf16 GEMM operands of the NEXT
// step: w_fwd (same layout) and w_dgr [Kw][Cin][Cout] with taps flipped (transposed through LDS).  One pass:
// read p, g, m; write p, m, 2 x bf16 -- instead of torch's multi-pass foreach update plus a separate pack.
// Update rule = torch.optim.SGD (dampening 0): g += wd*p; m = first ? g : mu*m + g; g = nesterov ? g + mu*m : m;
// p -= lr*g.
// one float -> one OCP e4m3 byte (round to nearest even, saturating at +-448)
__device__ __forceinline__ uint8_t quant1_e4m3(float v) {
    v = fminf(fmaxf(v, -448.f), 448.f);
    return (uint8_t)(__builtin_amdgcn_cvt_pk_fp8_f32(v, 0.f, 0, false) & 0xFF);
}

__global__ void sgd_pack_kernel(float* p, float* g, float* m, int first, float lr, float mu, float wd,
                                int nesterov, int zero_grad, int Cout, int Cin, int Kw, bf16_raw* fwd_hi, bf16_raw* fwd_lo,
                                bf16_raw* dgr_hi, bf16_raw* dgr_lo, uint8_t* fwd_q, uint8_t* dgr_q, float q_scale) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32, kw = blockIdx.z;
    for (int j = ty; j < 32; j += 8) {
        const int co = co0 + j, ci = ci0 + tx;
        float v = 0.f;
        if (co < Cout && ci < Cin) {
            const int64_t off = ((int64_t)kw * Cout + co) * Cin + ci;
            float pv = p[off];
            float gv = g[off] + wd * pv;
            if (zero_grad) g[off] = 0.f;       // the buffer comes back as the next step's (split-K, atomically accumulated) dW
            float mv = first ? gv : mu * m[off] + gv;
            m[off] = mv;
            gv = nesterov ? gv + mu * mv : mv;
            pv -= lr * gv;
            p[off] = pv;
            v = pv;
            if (fwd_hi) put_split(fwd_hi, fwd_lo, off, pv);
            // fp8 mode: the e4m3 operand of the next forward, quantised from the bf16-rounded value like w2l_quantize_e4m3
            if (fwd_q) fwd_q[off] = quant1_e4m3(bf16_bits_to_f32(f32_to_bf16_bits(pv)) * q_scale);
        }
        tile[j][tx] = v;
    }
    __syncthreads();
    if (dgr_hi) {
        for (int j = ty; j < 32; j += 8) {
            const int ci = ci0 + j, co = co0 + tx;
            if (co < Cout && ci < Cin) {
                const int64_t o = ((int64_t)(Kw - 1 - kw) * Cin + ci) * Cout + co;
                put_split(dgr_hi, dgr_lo, o, tile[tx][j]);
                if (dgr_q) dgr_q[o] = quant1_e4m3(bf16_bits_to_f32(f32_to_bf16_bits(tile[tx][j])) * q_scale);
            }
        }
    }
}

// ---- Novograd (novograd.py:86-112): per-TENSOR second moment v = EMA of ||g||^2 --------------------------------
// Three launches per conv weight: deterministic partial sums of g^2, a one-block finalize that updates v (first-step
// select on the device, optional running max) and leaves denom = sqrt(v) + eps in device memory, and the update fused
// with the bf16 operand pack like sgd_pack_kernel:  g' = g/denom + wd*p;  g' *= (1-beta1) if grad_averaging;
// m = beta1*m + g';  p -= lr*m.
__global__ __launch_bounds__(256) void sqnorm_partial_kernel(const float* g, int64_t n, float* partial) {
    __shared__ float red[256];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float v = g[i];
        s += v * v;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void novograd_finalize_kernel(const float* partial, int nblocks, float beta2, float eps,
                                                                float* v, float* vmax, float* denom) {
    __shared__ float red[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < nblocks; i += 256) s += partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float norm = red[0];
        const float old = v[0];
        float nv = old == 0.f ? norm : beta2 * old + (1.f - beta2) * norm;      // novograd.py:92-95
        v[0] = nv;
        if (vmax) {                                                             // amsgrad: novograd.py:97-102
            nv = fmaxf(vmax[0], nv);
            vmax[0] = nv;
        }
        denom[0] = sqrtf(nv) + eps;
    }
}

__global__ void novograd_pack_kernel(float* p, const float* g, float* m, const float* denom, float lr, float beta1, float wd,
                                     int grad_averaging, int Cout, int Cin, int Kw, bf16_raw* fwd_hi, bf16_raw* fwd_lo,
                                     bf16_raw* dgr_hi, bf16_raw* dgr_lo) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32, kw = blockIdx.z;
    const float dn = denom[0];
    for (int j = ty; j < 32; j += 8) {
        const int co = co0 + j, ci = ci0 + tx;
        float v = 0.f;
        if (co < Cout && ci < Cin) {
            const int64_t off = ((int64_t)kw * Cout + co) * Cin + ci;
            float pv = p[off];
            float gv = g[off] / dn;
            if (wd != 0.f) gv += wd * pv;
            if (grad_averaging) gv *= 1.f - beta1;
            const float mv = beta1 * m[off] + gv;
            m[off] = mv;
            pv -= lr * mv;
            p[off] = pv;
            v = pv;
            if (fwd_hi) put_split(fwd_hi, fwd_lo, off, pv);
        }
        tile[j][tx] = v;
    }
    __syncthreads();
    if (dgr_hi) {
        for (int j = ty; j < 32; j += 8) {
            const int ci = ci0 + j, co = co0 + tx;
            if (co < Cout && ci < Cin)
                put_split(dgr_hi, dgr_lo, ((int64_t)(Kw - 1 - kw) * Cin + ci) * Cout + co, tile[tx][j]);
        }
    }
}

// padded row r of an utterance -> source frame t, or -1 for a zero row
__device__ __forceinline__ int pad_src_row(int r, int T, int pad_l, int pad_r, int pad_mode) {
    int t = r - pad_l;
    if (t >= 0 && t < T) return t;
    if (pad_mode != 1 || t < -pad_l || t >= T + pad_r) return -1;
    t = t < 0 ? -t : 2 * (T - 1) - t;               // ReflectionPad1d: mirror without repeating the edge
    return (t >= 0 && t < T) ? t : -1;
}

__global__ void nct_to_ntc_kernel(const float* x, int C, int T, int CP, int R, int pad_l, int pad_r, int pad_mode,
                                  const int32_t* lens, bf16_raw* out_hi, bf16_raw* out_lo) {
    __shared__ float tile[32][33];                   // [c_l][r_l]
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32, n = blockIdx.z;
    const int len = lens ? lens[n] : T;
    const int r = r0 + tx;
    const int t = r < R ? pad_src_row(r, T, pad_l, pad_r, pad_mode) : -1;
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j;
        float v = 0.f;
        if (t >= 0 && t < len && c < C) v = x[((int64_t)n * C + c) * T + t];
        tile[j][tx] = v;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int rr = r0 + j, c = c0 + tx;
        if (rr < R && c < CP) put_split(out_hi, out_lo, ((int64_t)n * R + rr) * CP + c, tile[tx][j]);
    }
}

// shared-halo layout: row = halo + n*(T+halo) + t
__global__ void pad_cast_kernel(const float* g, int T, int C, int CP, int halo, bf16_raw* out_hi, bf16_raw* out_lo,
                                int64_t total) {
    const int P = T + halo;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % CP);
        const int64_t q = i / CP - halo;
        float v = 0.f;
        if (q >= 0) {
            const int64_t n = q / P;
            const int t = (int)(q - n * P);
            if (t < T && c < C) v = g[(n * T + t) * C + c];
        }
        put_split(out_hi, out_lo, i, v);
    }
}

__global__ void colsum_kernel(const float* g, int64_t rows, int C, int CP, float* colsum) {
    const int c = blockIdx.x;
    float s = 0.f;
    if (c < C)
        for (int64_t r = threadIdx.x; r < rows; r += blockDim.x) s += g[r * C + c];
    __shared__ float red[256];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) colsum[c] = red[0];
}

}  // namespace

extern "C" int w2l_pack_weights(const float* w, int64_t s_co, int64_t s_ci, int64_t s_kw, int Cout, int Cin, int Kw,
                                int CoutP, int CinP, void* w_fwd_hi, void* w_fwd_lo, void* w_dgr_hi, void* w_dgr_lo,
                                void* stream) {
    W2L_CHECK_ARG(w && (w_fwd_hi || w_dgr_hi), "pack_weights: null pointer");
    W2L_CHECK_ARG(Cout > 0 && Cin > 0 && Kw > 0 && CoutP >= Cout && CinP >= Cin, "pack_weights: bad sizes");
    W2L_CHECK_ARG(!(w_fwd_lo && !w_fwd_hi) && !(w_dgr_lo && !w_dgr_hi), "pack_weights: lo without hi");
    dim3 grid((CinP + 31) / 32, (CoutP + 31) / 32, Kw), block(32, 8);
    hipLaunchKernelGGL(pack_weights_kernel, grid, block, 0, (hipStream_t)stream, w, s_co, s_ci, s_kw, Cout, Cin, Kw,
                       CoutP, CinP, (bf16_raw*)w_fwd_hi, (bf16_raw*)w_fwd_lo, (bf16_raw*)w_dgr_hi, (bf16_raw*)w_dgr_lo);
    W2L_CHECK_LAUNCH();
    return 0;
}

extern "C" int w2l_sgd_pack(float* p, float* g, float* m, int first_step, float lr, float momentum,
                            float weight_decay, int nesterov, int zero_grad, int Cout, int Cin, int Kw, void* w_fwd_hi,
                            void* w_fwd_lo, void* w_dgr_hi, void* w_dgr_lo, void* w_fwd_q, void* w_dgr_q, float q_scale,
                            void* stream) {
    W2L_CHECK_ARG((!w_fwd_q && !w_dgr_q) || (q_scale > 0.f && w_fwd_hi && w_dgr_hi), "sgd_pack: e4m3 operands need a scale");
    W2L_CHECK_ARG(p && g && m, "sgd_pack: null pointer");
    W2L_CHECK_ARG(Cout > 0 && Cin > 0 && Kw > 0, "sgd_pack: bad sizes");
    W2L_CHECK_ARG(!(w_fwd_lo && !w_fwd_hi) && !(w_dgr_lo && !w_dgr_hi), "sgd_pack: lo without hi");
    dim3 grid((Cin + 31) / 32, (Cout + 31) / 32, Kw), block(32, 8);
    hipLaunchKernelGGL(sgd_pack_kernel, grid, block, 0, (hipStream_t)stream, p, g, m, first_step, lr, momentum,
                       weight_decay, nesterov, zero_grad, Cout, Cin, Kw, (bf16_raw*)w_fwd_hi, (bf16_raw*)w_fwd_lo,
                       (bf16_raw*)w_dgr_hi, (bf16_raw*)w_dgr_lo, (uint8_t*)w_fwd_q, (uint8_t*)w_dgr_q, q_scale);
    W2L_CHECK_LAUNCH();
    return 0;
}

extern "C" int w2l_novograd_pack(float* p, const float* g, float* exp_avg, float* exp_avg_sq, float* max_exp_avg_sq,
                                 float* scratch, int scratch_floats, float lr, float beta1, float beta2, float eps,
                                 float weight_decay, int grad_averaging, int Cout, int Cin, int Kw, void* w_fwd_hi,
                                 void* w_fwd_lo, void* w_dgr_hi, void* w_dgr_lo, void* stream) {
    W2L_CHECK_ARG(p && g && exp_avg && exp_avg_sq && scratch, "novograd_pack: null pointer");
    W2L_CHECK_ARG(Cout > 0 && Cin > 0 && Kw > 0 && scratch_floats >= 2, "novograd_pack: bad sizes");
    W2L_CHECK_ARG(!(w_fwd_lo && !w_fwd_hi) && !(w_dgr_lo && !w_dgr_hi), "novograd_pack: lo without hi");
    const int64_t n = (int64_t)Cout * Cin * Kw;
    int nblocks = (int)((n + 256 * 16 - 1) / (256 * 16));          // ~16 elements per thread
    if (nblocks > scratch_floats - 1) nblocks = scratch_floats - 1;
    if (nblocks > 1024) nblocks = 1024;
    if (nblocks < 1) nblocks = 1;
    hipStream_t st = (hipStream_t)stream;
    float* denom = scratch;                                         // scratch = [denom | partial sums]
    float* partial = scratch + 1;
    hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(nblocks), dim3(256), 0, st, g, n, partial);
    W2L_CHECK_LAUNCH();
    hipLaunchKernelGGL(novograd_finalize_kernel, dim3(1), dim3(256), 0, st, partial, nblocks, beta2, eps, exp_avg_sq,
                       max_exp_avg_sq, denom);
    W2L_CHECK_LAUNCH();
    dim3 grid((Cin + 31) / 32, (Cout + 31) / 32, Kw), block(32, 8);
    hipLaunchKernelGGL(novograd_pack_kernel, grid, block, 0, st, p, g, exp_avg, denom, lr, beta1, weight_decay, grad_averaging,
                       Cout, Cin, Kw, (bf16_raw*)w_fwd_hi, (bf16_raw*)w_fwd_lo, (bf16_raw*)w_dgr_hi, (bf16_raw*)w_dgr_lo);
    W2L_CHECK_LAUNCH();
    return 0;
}

extern "C" int w2l_nct_to_ntc(const float* x, int N, int C, int T, int CP, int pad_l, int pad_r, int pad_mode,
                              const int32_t* lens, void* out_hi, void* out_lo, void* stream) {
    W2L_CHECK_ARG(x && out_hi, "nct_to_ntc: null pointer");
    W2L_CHECK_ARG(N > 0 && C > 0 && T > 0 && CP >= C && pad_l >= 0 && pad_r >= 0, "nct_to_ntc: bad sizes");
    W2L_CHECK_ARG(pad_mode != 1 || (pad_l < T && pad_r < T), "nct_to_ntc: reflect padding (%d,%d) needs pad < T=%d",
                  pad_l, pad_r, T);
    const int R = pad_l + T + pad_r;
    dim3 grid((R + 31) / 32, (CP + 31) / 32, N), block(32, 8);
    hipLaunchKernelGGL(nct_to_ntc_kernel, grid, block, 0, (hipStream_t)stream, x, C, T, CP, R, pad_l, pad_r, pad_mode,
                       lens, (bf16_raw*)out_hi, (bf16_raw*)out_lo);
    W2L_CHECK_LAUNCH();
    return 0;
}

extern "C" int w2l_pad_cast(const float* g, int N, int T, int C, int CP, int halo, void* out_hi, void* out_lo,
                            float* colsum, void* stream) {
    W2L_CHECK_ARG(g && out_hi, "pad_cast: null pointer");
    W2L_CHECK_ARG(N > 0 && T > 0 && C > 0 && CP >= C && halo >= 0, "pad_cast: bad sizes");
    const int64_t total = ((int64_t)halo + (int64_t)N * (T + halo)) * CP;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pad_cast_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g, T, C, CP, halo,
                       (bf16_raw*)out_hi, (bf16_raw*)out_lo, total);
    W2L_CHECK_LAUNCH();
    if (colsum) {
        hipLaunchKernelGGL(colsum_kernel, dim3(CP), dim3(256), 0, (hipStream_t)stream, g, (int64_t)N * T, C, CP, colsum);
        W2L_CHECK_LAUNCH();
    }
    return 0;
}
