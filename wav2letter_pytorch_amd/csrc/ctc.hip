// log_softmax / softmax over the label axis, CTC loss (alpha-beta, log domain) with its
// gradient, and the greedy-decode argmax (gfx950).
//
// Replaces F.log_softmax / F.softmax (wav2letter.py:86-87, jasper.py:469-473),
// nn.CTCLoss(blank=0, reduction='mean', zero_infinity=True) forward AND backward
// (base_asr_models.py:23,81,90; torch native _ctc_loss / _ctc_loss_backward) and
// torch.max(probs, 2) (decoder.py:136).
//
// CTC structure: the alpha and the beta recursions are independent chains over time,
// so they run CONCURRENTLY in two workgroups per utterance (grid = N x 2), one lane per
// extended-label state, previous-step values exchanged through LDS, the next frame's
// log-probs prefetched into registers while the current step's log-sum-exp runs.  A
// third, fully parallel kernel turns alpha+beta into posteriors and the gradient.
#include "common.h"
#include <math.h>
#include <stdlib.h>

namespace {

constexpr float NEG_INF = -INFINITY;

// The recursion is a 500-step serial chain per utterance, so the per-step latency is what matters: the
// hardware v_exp_f32 / v_log_f32 forms (~1 ulp-level relative error) are used instead of libm's expf/logf.
// The log-domain values reach magnitudes of ~1e3 where one fp32 ulp (1.2e-4) already dominates that error.
__device__ __forceinline__ float lse2(float a, float b) {
    const float m = fmaxf(a, b);
    if (m == NEG_INF) return NEG_INF;
    return m + __logf(__expf(a - m) + __expf(b - m));
}
// (one of the three terms is exp(0): with v_max3 / v_med3 / v_min3 -- one instruction each -- only the two smaller ones cost a
// transcendental; the recursion is bound by the quarter-rate v_exp / v_log issue of its waves, see ctc_alpha_beta_kernel)
__device__ __forceinline__ float lse3(float a, float b, float c) {
    const float m = __builtin_fmaxf(__builtin_fmaxf(a, b), c);
    if (m == NEG_INF) return NEG_INF;
    const float lo = __builtin_fminf(__builtin_fminf(a, b), c);
    const float md = __builtin_amdgcn_fmed3f(a, b, c);
    return m + __logf(1.f + __expf(md - m) + __expf(lo - m));
}

// ---------------------------------------------------------------- softmax family
// one wave per row; C <= 64 * PER (label sets here: 29)
__global__ __launch_bounds__(256) void log_softmax_fwd_kernel(const float* logits, int64_t rows, int C, int CP, int mode,
                                                               float* out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* src = logits + row * CP;
    float m = NEG_INF;
    for (int c = lane; c < C; c += 64) m = fmaxf(m, src[c]);
#pragma unroll
    for (int k = 32; k > 0; k >>= 1) m = fmaxf(m, __shfl_xor(m, k, 64));
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += expf(src[c] - m);
#pragma unroll
    for (int k = 32; k > 0; k >>= 1) s += __shfl_xor(s, k, 64);
    const float lse = m + logf(s);
    for (int c = lane; c < C; c += 64) {
        const float lp = src[c] - lse;
        out[row * C + c] = mode == 0 ? lp : expf(lp);
    }
}

// mode 0: g_logit = g - exp(out) * sum(g);  mode 1 (softmax): g_logit = out * (g - sum(g*out))
__global__ __launch_bounds__(256) void log_softmax_bwd_kernel(const float* gout, const float* out, int64_t rows, int C,
                                                               int mode, float* glogits) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* g = gout + row * C;
    const float* o = out + row * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += mode == 0 ? g[c] : g[c] * o[c];
#pragma unroll
    for (int k = 32; k > 0; k >>= 1) s += __shfl_xor(s, k, 64);
    for (int c = lane; c < C; c += 64)
        glogits[row * C + c] = mode == 0 ? g[c] - expf(o[c]) * s : o[c] * (g[c] - s);
}

// ---------------------------------------------------------------- CTC alpha / beta
// workspace layout (floats): log_alpha [N][T][L] | log_beta [N][T][L], L = 2*Smax+1
struct CtcParams {
    const float* lp;           // [N][T][C]
    const int32_t* targets;    // [N][Smax]
    const int32_t* in_len;
    const int32_t* tg_len;
    int N, T, C, Smax, L, blank, zero_inf;
    float* alpha;
    float* beta;
    float* nll;                // [N]
};

// NT = block size (the states rounded up to whole waves, at most 1024), SPT = states per thread (1 up to 1024 states).  LP_LDS: the utterance's whole log-prob matrix (T x C fp32,
// 58 KB at T'=500) is staged in LDS once, so a time step never waits on an L2 round trip; otherwise the
// emissions are prefetched from global memory one step ahead (long utterances, T*C*4 > ~150 KB).
template <int NT, int SPT, bool LP_LDS>
__global__ __launch_bounds__(NT) void ctc_alpha_beta_kernel(CtcParams p) {
    extern __shared__ float sh[];              // [2][Lpad + 4] (two guard cells on each side) | [T*C] log-probs
    const int n = blockIdx.x;
    const bool is_beta = blockIdx.y == 1;
    const int tid = threadIdx.x;
    const int S = min(max(p.tg_len[n], 0), p.Smax);
    const int Tn = min(max(p.in_len[n], 0), p.T);
    const int L = 2 * S + 1;
    const int Lp = SPT * NT + 4;
    float* buf0 = sh + 2;
    float* buf1 = sh + Lp + 2;
    const float* lp = p.lp + (int64_t)n * p.T * p.C;
    if (LP_LDS) {
        float* lds_lp = sh + 2 * Lp;
        const int total = Tn * p.C;            // rows are contiguous: one linear, coalesced copy
        for (int i = tid; i < total; i += NT) lds_lp[i] = lp[i];
        lp = lds_lp;
        __syncthreads();
    }
    float* dst = (is_beta ? p.beta : p.alpha) + (int64_t)n * p.T * p.L;
    const int32_t* tg = p.targets + (int64_t)n * p.Smax;

    // per-state constants: label, whether the skip transition (s-2 / s+2) is allowed
    int lab[SPT];
    bool skip[SPT];
#pragma unroll
    for (int k = 0; k < SPT; ++k) {
        const int s = tid + k * NT;
        lab[k] = p.blank;
        skip[k] = false;
        if (s < L && (s & 1)) {
            lab[k] = tg[s >> 1];
            if (!is_beta) skip[k] = s >= 3 && tg[(s >> 1) - 1] != lab[k];
            else skip[k] = s + 2 < L && tg[(s >> 1) + 1] != lab[k];
        }
    }
    if (tid < 2) { buf0[-2 + tid] = NEG_INF; buf1[-2 + tid] = NEG_INF; }       // guards below state 0
    if (tid < 2) { buf0[Lp - 4 + tid] = NEG_INF; buf1[Lp - 4 + tid] = NEG_INF; }  // guards above

    if (Tn == 0) {
        if (!is_beta && tid == 0) p.nll[n] = S == 0 ? 0.f : (p.zero_inf ? 0.f : INFINITY);
        return;
    }
    // initial column
    const int t_first = is_beta ? Tn - 1 : 0;
    float cur[SPT], nxt_lp[SPT];
#pragma unroll
    for (int k = 0; k < SPT; ++k) {
        const int s = tid + k * NT;
        float v = NEG_INF;
        if (s < L) {
            const bool init = is_beta ? (s >= L - 2) : (s <= 1);
            if (init) v = lp[(int64_t)t_first * p.C + lab[k]];
            dst[(int64_t)t_first * p.L + s] = v;
        }
        cur[k] = v;
        buf0[s] = v;
        nxt_lp[k] = 0.f;
    }
    const int dir = is_beta ? -1 : 1;
    if (Tn > 1) {
#pragma unroll
        for (int k = 0; k < SPT; ++k)
            if (tid + k * NT < L) nxt_lp[k] = lp[(int64_t)(t_first + dir) * p.C + lab[k]];
    }
    __syncthreads();
    float* prev = buf0;
    float* next = buf1;
    for (int i = 1; i < Tn; ++i) {
        const int t = t_first + dir * i;
        float e[SPT];
#pragma unroll
        for (int k = 0; k < SPT; ++k) e[k] = nxt_lp[k];
        if (i + 1 < Tn) {                                   // prefetch the following frame's emissions
#pragma unroll
            for (int k = 0; k < SPT; ++k)
                if (tid + k * NT < L) nxt_lp[k] = lp[(int64_t)(t + dir) * p.C + lab[k]];
        }
#pragma unroll
        for (int k = 0; k < SPT; ++k) {
            const int s = tid + k * NT;
            const float a0 = cur[k];
            const float a1 = prev[s - dir];
            const float a2 = skip[k] ? prev[s - 2 * dir] : NEG_INF;
            float v = lse3(a0, a1, a2) + e[k];
            if (s >= L) v = NEG_INF;
            cur[k] = v;
            next[s] = v;
            if (s < L) dst[(int64_t)t * p.L + s] = v;
        }
        // LDS-only barrier: the alpha/beta stores and the emission prefetch stay in flight across it
        // (__syncthreads() would drain vmcnt: one L2 round trip per time step on a 500-step serial chain)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        float* tmp = prev; prev = next; next = tmp;
    }
    if (!is_beta && tid == 0) {
        const float a = prev[L - 1];
        const float b = L > 1 ? prev[L - 2] : NEG_INF;
        float nll = -lse2(a, b);
        p.nll[n] = nll;                                      // may be +inf; the grad kernel applies zero_infinity
    }
}

// grad[n][t][c] = (exp(lp) - posterior(t,c)) * gscale_n   for t < in_len[n], else 0
// posterior(t,c) = sum_{s: l'_s = c} exp(alpha_t(s) + beta_t(s) - lp[t][c] + nll)
// summed in the linear domain relative to the per-frame maximum of (alpha+beta-lp), whose
// exponent offset (max + nll) lies in [-log L, 0] because the posteriors of a frame sum to 1.
__global__ __launch_bounds__(256) void ctc_grad_kernel(CtcParams p, float* grad, float* nll_out) {
    __shared__ float acc_all[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int n = blockIdx.y;
    const int t = blockIdx.x * 4 + w;
    if (t >= p.T) return;
    float* acc = acc_all[w];
    const int S = min(max(p.tg_len[n], 0), p.Smax);
    const int Tn = min(max(p.in_len[n], 0), p.T);
    const int L = 2 * S + 1;
    const float nll = p.nll[n];
    const bool inf = !(nll < INFINITY);
    float* g = grad + ((int64_t)n * p.T + t) * p.C;
    if (t >= Tn || (inf && p.zero_inf)) {
        for (int c = lane; c < p.C; c += 64) g[c] = 0.f;
        return;
    }
    const float* lp = p.lp + ((int64_t)n * p.T + t) * p.C;
    const float* al = p.alpha + ((int64_t)n * p.T + t) * p.L;
    const float* be = p.beta + ((int64_t)n * p.T + t) * p.L;
    const int32_t* tg = p.targets + (int64_t)n * p.Smax;
    for (int c = lane; c < 64; c += 64) acc[c] = 0.f;
    float m = NEG_INF;
    for (int s = lane; s < L; s += 64) {
        const int c = (s & 1) ? tg[s >> 1] : p.blank;
        const float v = al[s] + be[s];
        if (v > NEG_INF) m = fmaxf(m, v - lp[c]);
    }
#pragma unroll
    for (int k = 32; k > 0; k >>= 1) m = fmaxf(m, __shfl_xor(m, k, 64));
    __builtin_amdgcn_wave_barrier();
    if (m > NEG_INF) {
        for (int s = lane; s < L; s += 64) {
            const int c = (s & 1) ? tg[s >> 1] : p.blank;
            const float v = al[s] + be[s];
            if (v > NEG_INF) atomicAdd(&acc[c], expf(v - lp[c] - m));
        }
    }
    __builtin_amdgcn_wave_barrier();
    const float gs = 1.f / ((float)p.N * (float)max(S, 1));
    const float post_scale = m > NEG_INF ? expf(m + nll) : 0.f;
    for (int c = lane; c < p.C; c += 64) {
        const float post = acc[c] * post_scale;
        g[c] = (expf(lp[c]) - post) * gs;
    }
    (void)nll_out;
}

__global__ void ctc_loss_reduce_kernel(float* nll, const int32_t* tg_len, int N, int Smax, int zero_inf, float* loss) {
    // single block; N is a batch size
    __shared__ float red[256];
    float s = 0.f;
    for (int n = threadIdx.x; n < N; n += 256) {
        float v = nll[n];
        if (zero_inf && !(v < INFINITY)) { v = 0.f; nll[n] = 0.f; }
        const int S = min(max(tg_len[n], 0), Smax);
        s += v / (float)max(S, 1);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = red[0] / (float)N;
}

// ---------------------------------------------------------------- argmax (ties -> lowest index)
__global__ __launch_bounds__(256) void argmax_kernel(const float* probs, int64_t rows, int C, int32_t* idx) {
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    const float* p = probs + row * C;
    float best = p[0];
    int bi = 0;
    bool nan_found = best != best;
    for (int c = 1; c < C && !nan_found; ++c) {
        const float v = p[c];
        if (v != v) { bi = c; nan_found = true; }          // torch.max propagates the first NaN
        else if (v > best) { best = v; bi = c; }
    }
    idx[row] = bi;
}

}  // namespace

extern "C" int w2l_log_softmax_fwd(const float* logits, int N, int T, int C, int CP, int mode, float* out, void* stream) {
    W2L_CHECK_ARG(logits && out && N > 0 && T > 0 && C > 0 && CP >= C, "log_softmax_fwd: bad arguments");
    const int64_t rows = (int64_t)N * T;
    hipLaunchKernelGGL(log_softmax_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       logits, rows, C, CP, mode, out);
    W2L_CHECK_LAUNCH();
    return 0;
}

extern "C" int w2l_log_softmax_bwd(const float* gout, const float* out, int N, int T, int C, int mode, float* glogits,
                                   void* stream) {
    W2L_CHECK_ARG(gout && out && glogits && N > 0 && T > 0 && C > 0, "log_softmax_bwd: bad arguments");
    const int64_t rows = (int64_t)N * T;
    hipLaunchKernelGGL(log_softmax_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, gout,
                       out, rows, C, mode, glogits);
    W2L_CHECK_LAUNCH();
    return 0;
}

extern "C" int64_t w2l_ctc_workspace_bytes(int N, int T, int Smax) {
    if (N <= 0 || T <= 0 || Smax < 0) return 0;
    return 2 * (int64_t)N * T * (2 * Smax + 1) * (int64_t)sizeof(float);
}

extern "C" int w2l_ctc_loss(const float* log_probs, const int32_t* targets, const int32_t* input_lengths,
                            const int32_t* target_lengths, int N, int T, int C, int Smax, int blank, int zero_infinity,
                            float* nll, float* loss, float* grad, void* workspace, void* stream) {
    W2L_CHECK_ARG(log_probs && input_lengths && target_lengths && nll && loss && workspace, "ctc_loss: null pointer");
    W2L_CHECK_ARG(targets || Smax == 0, "ctc_loss: null targets");
    W2L_CHECK_ARG(N > 0 && T > 0 && C > 0 && C <= 64 && Smax >= 0 && blank >= 0 && blank < C, "ctc_loss: bad sizes");
    const int L = 2 * Smax + 1;
    W2L_CHECK_ARG(L <= 8 * 1024, "ctc_loss: target length %d too long (max 4095)", Smax);
    CtcParams p;
    p.lp = log_probs; p.targets = targets; p.in_len = input_lengths; p.tg_len = target_lengths;
    p.N = N; p.T = T; p.C = C; p.Smax = Smax; p.L = L; p.blank = blank; p.zero_inf = zero_infinity;
    p.alpha = (float*)workspace;
    p.beta = p.alpha + (int64_t)N * T * L;
    p.nll = nll;
    // one thread per extended-label state while 4 x 256 covers them (S <= 511: every utterance of <= ~40 s), else 1024
    // threads with up to 8 states each (S <= 4095: the T = 16 000 frame utterances of BASELINE config 5)
    // One extended-label state per thread while a block can hold them (L <= 1024: every utterance of <= ~80 s), the block just
    // wide enough in waves; beyond that 1024 threads with up to 8 states each (S <= 4095: the T = 16 000 frame utterances of
    // BASELINE config 5).  Round 6 measured what bounds a time step: not the barrier or the LDS round trip but the waves'
    // transcendental issue (v_exp / v_log at a quarter of the rate) -- fewer, fatter threads are slower in proportion (L = 321:
    // 64 x 6 states 509 us, 128 x 3 320, 256 x 2 224, 384 x 1 156 for loss + gradient at N = 32, T' = 500), idle state slots
    // cost like busy ones, and two waves per SIMD hide each other's latency.
    static const int kWide[] = {64, 128, 192, 256, 320, 384, 448, 512, 640, 768, 896, 1024};
    int nt = 1024;
    for (int w : kWide)
        if (L <= w) { nt = w; break; }
    int spt = (L + nt - 1) / nt;
    if (spt > 1) spt = spt <= 4 ? spt : (spt <= 6 ? 6 : 8);    // the instantiated variants of the 1024-thread form
    dim3 grid(N, 2), block(nt);
    size_t lds = 2 * (size_t)(spt * nt + 4) * sizeof(float);
    const size_t lp_bytes = (size_t)T * C * sizeof(float);
    const bool lp_lds = lds + lp_bytes <= 150 * 1024;
    if (lp_lds) lds += lp_bytes;
#define W2L_CTC_LAUNCH(NT, SPT)                                                                              \
    do {                                                                                                     \
        if (lp_lds) {                                                                                        \
            W2L_CHECK_HIP(w2l_allow_big_lds((const void*)ctc_alpha_beta_kernel<NT, SPT, true>));             \
            hipLaunchKernelGGL((ctc_alpha_beta_kernel<NT, SPT, true>), grid, block, lds, (hipStream_t)stream, p);  \
        } else {                                                                                             \
            hipLaunchKernelGGL((ctc_alpha_beta_kernel<NT, SPT, false>), grid, block, lds, (hipStream_t)stream, p); \
        }                                                                                                    \
    } while (0)
    if (spt == 1) {
        switch (nt) {
            case 64: W2L_CTC_LAUNCH(64, 1); break;
            case 128: W2L_CTC_LAUNCH(128, 1); break;
            case 192: W2L_CTC_LAUNCH(192, 1); break;
            case 256: W2L_CTC_LAUNCH(256, 1); break;
            case 320: W2L_CTC_LAUNCH(320, 1); break;
            case 384: W2L_CTC_LAUNCH(384, 1); break;
            case 448: W2L_CTC_LAUNCH(448, 1); break;
            case 512: W2L_CTC_LAUNCH(512, 1); break;
            case 640: W2L_CTC_LAUNCH(640, 1); break;
            case 768: W2L_CTC_LAUNCH(768, 1); break;
            case 896: W2L_CTC_LAUNCH(896, 1); break;
            default: W2L_CTC_LAUNCH(1024, 1); break;
        }
    } else {
        switch (spt) {
            case 2: W2L_CTC_LAUNCH(1024, 2); break;
            case 3: W2L_CTC_LAUNCH(1024, 3); break;
            case 4: W2L_CTC_LAUNCH(1024, 4); break;
            case 6: W2L_CTC_LAUNCH(1024, 6); break;
            default: W2L_CTC_LAUNCH(1024, 8); break;
        }
    }
#undef W2L_CTC_LAUNCH
    W2L_CHECK_LAUNCH();
    if (grad) {
        hipLaunchKernelGGL(ctc_grad_kernel, dim3((T + 3) / 4, N), dim3(256), 0, (hipStream_t)stream, p, grad, nll);
        W2L_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(ctc_loss_reduce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, nll, target_lengths, N,
                       p.Smax, zero_infinity, loss);
    W2L_CHECK_LAUNCH();
    return 0;
}

extern "C" int w2l_argmax(const float* probs, int64_t rows, int C, int32_t* idx, void* stream) {
    W2L_CHECK_ARG(probs && idx && rows > 0 && C > 0, "argmax: bad arguments");
    hipLaunchKernelGGL(argmax_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, probs, rows,
                       C, idx);
    W2L_CHECK_LAUNCH();
    return 0;
}
