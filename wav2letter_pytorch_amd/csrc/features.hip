// Log-mel feature front-end and spectrogram augmentation masks on gfx950.
//
// Replaces the per-utterance host pipeline of the reference's SpectrogramExtractor
// (data/data_loader.py:64-88): dither, pre-emphasis, torch.stft(center=True, reflect), |.|^2,
// mel filterbank, log1p(. + 2^-24), per-feature normalisation over time -- batched over N
// utterances, one launch each for (frames -> log-mel), (per-feature statistics) and
// (normalise + transpose into the collate layout [N][n_mels][T_max], data_loader.py:149-158).
// And the masked_fill of SpecAugment / SpecCutout (data/augmentations.py:39-58,79-99).
//
// These kernels are HBM / latency bound (20 MB of audio per 32 ten-second utterances); the FFT is a
// radix-2 DIT in LDS, one wave per frame, deliberately NOT reshaped into a GEMM.
#include "common.h"

namespace {

constexpr int FRAMES_PER_BLOCK = 4;   // one wave per frame

struct LogmelParams {
    const float* audio;
    const int32_t* n_samples;
    const float* noise;
    const float* window;
    const float* fbT;        // [n_bins][n_mels]
    const int32_t* fb_range; // [n_mels][2]: first / one-past-last non-zero bin of each filter (NULL: dense)
    float* logmel;           // [N][Tmax][n_mels]
    int64_t audio_stride;
    float dither, preemph, guard;
    int N, win_length, n_fft, log2n, hop, n_mels, Tmax, take_log;
};

__device__ __forceinline__ int reflect_index(int j, int L) {
    // torch.stft(center=True, pad_mode='reflect'): mirror without repeating the edge sample
    if (j < 0) j = -j;
    if (j >= L) j = 2 * (L - 1) - j;
    j = j < 0 ? 0 : j;                   // only reachable when L <= n_fft/2 (rejected by the host wrapper); stay in bounds
    return j >= L ? L - 1 : j;
}

__global__ __launch_bounds__(64 * FRAMES_PER_BLOCK) void logmel_kernel(LogmelParams p) {
    extern __shared__ float lds[];       // twiddles cos[n/2], sin[n/2] | per wave: re[n], im[n]
    const int nf = p.n_fft, half = nf >> 1;
    float* tw_c = lds;
    float* tw_s = lds + half;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* re = lds + nf + wave * 2 * nf;
    float* im = re + nf;
    const int n = blockIdx.y;
    const int t = blockIdx.x * FRAMES_PER_BLOCK + wave;
    const int L = p.n_samples[n];
    const int T_n = 1 + L / p.hop;
    const bool live = t < T_n && t < p.Tmax;

    for (int k = threadIdx.x; k < half; k += blockDim.x) {
        float s, c;
        sincospif(-2.0f * (float)k / (float)nf, &s, &c);      // exp(-2 pi i k / n)
        tw_c[k] = c;
        tw_s[k] = s;
    }
    const float* x = p.audio + (int64_t)n * p.audio_stride;
    const float* z = p.noise ? p.noise + (int64_t)n * p.audio_stride : nullptr;
    const int woff = (nf - p.win_length) >> 1;               // torch.stft centres a short window inside n_fft
    const int j0 = t * p.hop - half;
    for (int i = lane; i < nf; i += 64) {
        float v = 0.f;
        const int wi = i - woff;
        if (live && wi >= 0 && wi < p.win_length) {
            const int m = reflect_index(j0 + i, L);
            float a = x[m];
            if (z) a = __fadd_rn(a, __fmul_rn(z[m], p.dither));
            if (m > 0) {
                float b = x[m - 1];
                if (z) b = __fadd_rn(b, __fmul_rn(z[m - 1], p.dither));
                a = __fsub_rn(a, __fmul_rn(p.preemph, b));
            }
            v = a * p.window[wi];
        }
        const int r = (int)(__brev((unsigned)i) >> (32 - p.log2n));
        re[r] = v;
        im[r] = 0.f;
    }
    __syncthreads();                     // twiddle table complete (the only block-wide dependency)
    // From here each wave works on its own re/im arrays.  LDS instructions of one wave execute in order, so a stage's
    // stores are visible to the next stage's loads without a block barrier; wave_barrier() only pins the compiler's order.
    for (int s = 1; s <= p.log2n; ++s) {
        const int hm = 1 << (s - 1);
        const int tstep = nf >> s;
        for (int b = lane; b < half; b += 64) {
            const int k = b & (hm - 1);
            const int i0 = ((b >> (s - 1)) << s) + k;
            const int i1 = i0 + hm;
            const float c = tw_c[k * tstep], sn = tw_s[k * tstep];
            const float xr = re[i1], xi = im[i1];
            const float tr = c * xr - sn * xi, ti = c * xi + sn * xr;
            const float ur = re[i0], ui = im[i0];
            re[i0] = ur + tr;
            im[i0] = ui + ti;
            re[i1] = ur - tr;
            im[i1] = ui - ti;
        }
        __builtin_amdgcn_wave_barrier();
    }
    // power spectrum, as the reference forms it: sqrt(re^2 + im^2), then squared (data_loader.py:69-70)
    float pw[ (1024 / 2 + 64) / 64 ];
    const int nbins = half + 1;
    int cnt = 0;
    for (int k = lane; k < nbins; k += 64) {
        const float mag = sqrtf(re[k] * re[k] + im[k] * im[k]);
        pw[cnt++] = mag * mag;
    }
    cnt = 0;
    for (int k = lane; k < nbins; k += 64) re[k] = pw[cnt++];      // lane k overwrites only what lane k read
    __builtin_amdgcn_wave_barrier();
    if (t < p.Tmax) {
        float* out = p.logmel + ((int64_t)n * p.Tmax + t) * p.n_mels;
        for (int m = lane; m < p.n_mels; m += 64) {
            float acc = 0.f;
            if (live) {
                // triangular filters are non-zero on a short run of bins: summing that run only is exact
                const int k0 = p.fb_range ? p.fb_range[2 * m] : 0;
                const int k1 = p.fb_range ? p.fb_range[2 * m + 1] : nbins;
                for (int k = k0; k < k1; ++k) acc += p.fbT[k * p.n_mels + m] * re[k];
                if (p.take_log) acc = log1pf(acc + p.guard);
            }
            out[m] = acc;                 // frames past the utterance's length are zero
        }
    }
}

// ---- n_fft = 512: the FFT lives in registers -------------------------------------------------------------------------
// 512 = 8 x 8 x 8.  One wave per frame, eight complex values per lane, three in-register 8-point DFTs separated by two
// LDS transposes (instead of nine LDS-resident radix-2 stages):
//   n = 64a + b:            lane b      DFT-8 over a -> Y_b[c],      x W512^(b c)
//   b = 8e + f:             lane (c,f)  DFT-8 over e -> U_cf[g],     x W64^(f g)
//                           lane (c,g)  DFT-8 over f -> X[c + 8g + 64h], h = 0..7
// Transpose images use a row stride of 72 words and a (f + g) mod 8 skew: every ds access below is bank-conflict-free.
__device__ __forceinline__ void dft8(float (&re)[8], float (&im)[8]) {
    const float R = 0.70710678118654752f;
    // length-2
    const float e0r = re[0] + re[4], e0i = im[0] + im[4], e1r = re[0] - re[4], e1i = im[0] - im[4];
    const float e2r = re[2] + re[6], e2i = im[2] + im[6], e3r = re[2] - re[6], e3i = im[2] - im[6];
    const float o0r = re[1] + re[5], o0i = im[1] + im[5], o1r = re[1] - re[5], o1i = im[1] - im[5];
    const float o2r = re[3] + re[7], o2i = im[3] + im[7], o3r = re[3] - re[7], o3i = im[3] - im[7];
    // length-4 (W4 = -i: (-i)(a + ib) = b - ia)
    const float E0r = e0r + e2r, E0i = e0i + e2i, E2r = e0r - e2r, E2i = e0i - e2i;
    const float E1r = e1r + e3i, E1i = e1i - e3r, E3r = e1r - e3i, E3i = e1i + e3r;
    const float O0r = o0r + o2r, O0i = o0i + o2i, O2r = o0r - o2r, O2i = o0i - o2i;
    const float O1r = o1r + o3i, O1i = o1i - o3r, O3r = o1r - o3i, O3i = o1i + o3r;
    // length-8: W8^1 = (1 - i)/sqrt2, W8^2 = -i, W8^3 = (-1 - i)/sqrt2
    const float t1r = R * (O1r + O1i), t1i = R * (O1i - O1r);
    const float t2r = O2i, t2i = -O2r;
    const float t3r = R * (O3i - O3r), t3i = -R * (O3r + O3i);
    re[0] = E0r + O0r; im[0] = E0i + O0i; re[4] = E0r - O0r; im[4] = E0i - O0i;
    re[1] = E1r + t1r; im[1] = E1i + t1i; re[5] = E1r - t1r; im[5] = E1i - t1i;
    re[2] = E2r + t2r; im[2] = E2i + t2i; re[6] = E2r - t2r; im[6] = E2i - t2i;
    re[3] = E3r + t3r; im[3] = E3i + t3i; re[7] = E3r - t3r; im[7] = E3i - t3i;
}

constexpr int TSTRIDE = 72;                       // words per transpose row (64 + 8: spreads rows over the 32 banks)
constexpr int WAVE_WORDS = 2 * 8 * TSTRIDE;       // re and im images of one wave

__global__ __launch_bounds__(64 * FRAMES_PER_BLOCK) void logmel512_kernel(LogmelParams p) {
    constexpr int SPAN_MAX = 1024;                                  // samples the block's frames cover: (F-1)*hop + 512
    __shared__ float lds[512 + FRAMES_PER_BLOCK * WAVE_WORDS + SPAN_MAX];   // W512^m (m < 256): cos | sin, the waves' images,
    float* tw_c = lds;                                              // and the block's pre-emphasised samples
    float* tw_s = lds + 256;
    float* ys = lds + 512 + FRAMES_PER_BLOCK * WAVE_WORDS;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* Are = lds + 512 + wave * WAVE_WORDS;
    float* Aim = Are + 8 * TSTRIDE;
    const int n = blockIdx.y;
    const int t = blockIdx.x * FRAMES_PER_BLOCK + wave;
    const int L = p.n_samples[n];
    const int T_n = 1 + L / p.hop;
    const bool live = t < T_n && t < p.Tmax;
    {
        float s_, c_;
        sincospif(-(float)threadIdx.x / 256.0f, &s_, &c_);          // exp(-2 pi i m / 512), m = threadIdx.x
        tw_c[threadIdx.x] = c_;
        tw_s[threadIdx.x] = s_;
    }
    auto twiddle = [&](int m, float& wr, float& wi) {                // W512^m for m < 512
        const int k = m & 255;
        wr = tw_c[k];
        wi = tw_s[k];
        if (m & 256) { wr = -wr; wi = -wi; }
    };
    float re[8], im[8];
    // ---- the block's frames overlap (hop < window): dither + pre-emphasis + reflect padding once per sample, into LDS
    const float* x = p.audio + (int64_t)n * p.audio_stride;
    const float* z = p.noise ? p.noise + (int64_t)n * p.audio_stride : nullptr;
    const int woff = (512 - p.win_length) >> 1;
    const int span = (FRAMES_PER_BLOCK - 1) * p.hop + 512;
    const bool staged = span <= SPAN_MAX;                            // (always, for hop <= 170; otherwise straight from memory)
    const int jb = blockIdx.x * FRAMES_PER_BLOCK * p.hop - 256;      // padded-signal position of ys[0]
    auto sample = [&](int j) {
        const int m = reflect_index(j, L);
        float s0 = x[m];
        if (z) s0 = __fadd_rn(s0, __fmul_rn(z[m], p.dither));
        if (m > 0) {
            float s1 = x[m - 1];
            if (z) s1 = __fadd_rn(s1, __fmul_rn(z[m - 1], p.dither));
            s0 = __fsub_rn(s0, __fmul_rn(p.preemph, s1));
        }
        return s0;
    };
    if (staged && blockIdx.x * FRAMES_PER_BLOCK < T_n)
        for (int i = threadIdx.x; i < span; i += blockDim.x) ys[i] = sample(jb + i);
    __syncthreads();                                                 // twiddle table and samples complete
    // ---- samples n = 64a + b (lane = b), windowed
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        const int i = 64 * a + lane;
        const int wi = i - woff;
        float v = 0.f;
        if (live && wi >= 0 && wi < p.win_length)
            v = (staged ? ys[wave * p.hop + i] : sample(t * p.hop - 256 + i)) * p.window[wi];
        re[a] = v;
        im[a] = 0.f;
    }
    dft8(re, im);                                                    // Y_b[c]
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        float wr, wi_;
        twiddle(lane * c, wr, wi_);
        const float r_ = re[c] * wr - im[c] * wi_, i_ = re[c] * wi_ + im[c] * wr;
        Are[c * TSTRIDE + lane] = r_;
        Aim[c * TSTRIDE + lane] = i_;
    }
    __builtin_amdgcn_wave_barrier();
    const int c1 = lane >> 3, f = lane & 7;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        re[e] = Are[c1 * TSTRIDE + 8 * e + f];
        im[e] = Aim[c1 * TSTRIDE + 8 * e + f];
    }
    dft8(re, im);                                                    // U_cf[g]
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        float wr, wi_;
        twiddle(8 * f * g, wr, wi_);
        const float r_ = re[g] * wr - im[g] * wi_, i_ = re[g] * wi_ + im[g] * wr;
        Are[c1 * TSTRIDE + 8 * g + ((f + g) & 7)] = r_;
        Aim[c1 * TSTRIDE + 8 * g + ((f + g) & 7)] = i_;
    }
    __builtin_amdgcn_wave_barrier();
    const int g1 = lane & 7;
#pragma unroll
    for (int ff = 0; ff < 8; ++ff) {
        re[ff] = Are[c1 * TSTRIDE + 8 * g1 + ((ff + g1) & 7)];
        im[ff] = Aim[c1 * TSTRIDE + 8 * g1 + ((ff + g1) & 7)];
    }
    dft8(re, im);                                                    // X[c1 + 8 g1 + 64 h]
    __builtin_amdgcn_wave_barrier();
    // power spectrum, as the reference forms it (sqrt, then squared), bins 0..256 into the wave's image
    float* P = Are;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const float mag = sqrtf(re[h] * re[h] + im[h] * im[h]);
        P[c1 + 8 * g1 + 64 * h] = mag * mag;
    }
    if (lane == 0) {
        const float mag = sqrtf(re[4] * re[4] + im[4] * im[4]);
        P[256] = mag * mag;
    }
    __builtin_amdgcn_wave_barrier();
    if (t < p.Tmax) {
        float* out = p.logmel + ((int64_t)n * p.Tmax + t) * p.n_mels;
        for (int m = lane; m < p.n_mels; m += 64) {
            float acc = 0.f;
            if (live) {
                const int k0 = p.fb_range ? p.fb_range[2 * m] : 0;
                const int k1 = p.fb_range ? p.fb_range[2 * m + 1] : 257;
                // eight filter weights in flight per round trip (the loads are unconditional on a clamped index; a
                // load under a per-element condition would be waited for one by one); same summation order as a plain loop
                for (int k = k0; k < k1; k += 8) {
                    float w[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) w[u] = p.fbT[min(k + u, k1 - 1) * p.n_mels + m];
#pragma unroll
                    for (int u = 0; u < 8; ++u) acc += (k + u < k1 ? w[u] : 0.f) * P[min(k + u, 256)];
                }
                if (p.take_log) acc = log1pf(acc + p.guard);
            }
            out[m] = acc;
        }
    }
}

// per (utterance, feature): mean and unbiased std over the utterance's frames (torch.Tensor.std), two-pass.
// One block of 16 waves per (utterance, 64 features): lane = feature, wave = time phase, 8 loads in flight per lane
// (the block is alone with a [T][n_mels] slab that other XCDs wrote: every first touch is an L2 miss).
constexpr int STAT_WAVES = 16;
__global__ __launch_bounds__(64 * STAT_WAVES) void feature_stats_kernel(const float* logmel, const int32_t* n_samples, int hop,
                                                                        int Tmax, int n_mels, float eps, float* mean_out,
                                                                        float* std_out) {
    __shared__ float part[STAT_WAVES][64];
    const int n = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int T_n = 1 + n_samples[n] / hop;
    T_n = T_n < Tmax ? T_n : Tmax;
    const int m = blockIdx.y * 64 + lane;
    const bool ok = m < n_mels;
    const float* base = logmel + (int64_t)n * Tmax * n_mels + (ok ? m : 0);
    auto reduce = [&](float v) {
        part[wave][lane] = v;
        __syncthreads();
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < STAT_WAVES; ++w) s += part[w][lane];
        __syncthreads();
        return s;
    };
    float s = 0.f;
    int t = wave;
    for (; t + 7 * STAT_WAVES < T_n; t += 8 * STAT_WAVES) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = base[(int64_t)(t + u * STAT_WAVES) * n_mels];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; t < T_n; t += STAT_WAVES) s += base[(int64_t)t * n_mels];
    const float mean = reduce(s) / (float)T_n;
    float q = 0.f;
    t = wave;
    for (; t + 7 * STAT_WAVES < T_n; t += 8 * STAT_WAVES) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = base[(int64_t)(t + u * STAT_WAVES) * n_mels];
#pragma unroll
        for (int u = 0; u < 8; ++u) q += (v[u] - mean) * (v[u] - mean);
    }
    for (; t < T_n; t += STAT_WAVES) {
        const float d = base[(int64_t)t * n_mels] - mean;
        q += d * d;
    }
    const float var = reduce(q) / (float)(T_n - 1);
    if (wave == 0 && ok) {
        mean_out[n * n_mels + m] = mean;
        std_out[n * n_mels + m] = sqrtf(var) + eps;          // T_n == 1 -> NaN, as torch.std of one sample
    }
}

// (x - mean) / std, transposed from [N][Tmax][n_mels] into the collate layout [N][n_mels][Tmax]; frames past an
// utterance's own length stay 0 (np.pad(..., mode='constant'), data_loader.py:154)
__global__ __launch_bounds__(256) void feature_apply_kernel(const float* logmel, const int32_t* n_samples, int hop, int Tmax,
                                                            int n_mels, const float* mean, const float* stdv, float* out) {
    __shared__ float tile[64][65];
    const int n = blockIdx.z, t0 = blockIdx.x * 64, m0 = blockIdx.y * 64;
    int T_n = 1 + n_samples[n] / hop;
    T_n = T_n < Tmax ? T_n : Tmax;
    const int a = threadIdx.x & 63, b = threadIdx.x >> 6;
    for (int r = b; r < 64; r += 4) {
        const int t = t0 + r, m = m0 + a;
        float v = 0.f;
        if (t < T_n && m < n_mels) v = (logmel[((int64_t)n * Tmax + t) * n_mels + m] - mean[n * n_mels + m]) / stdv[n * n_mels + m];
        tile[r][a] = v;
    }
    __syncthreads();
    for (int r = b; r < 64; r += 4) {
        const int m = m0 + r, t = t0 + a;
        if (m < n_mels && t < Tmax) out[((int64_t)n * n_mels + m) * Tmax + t] = tile[a][r];
    }
}

// x[n][f0:f1][t0:t1] = 0 for every rectangle (n, f0, f1, t0, t1); one block per rectangle
__global__ __launch_bounds__(256) void zero_rects_kernel(float* x, int N, int C, int T, const int32_t* rects, int R) {
    const int32_t* r = rects + blockIdx.x * 5;
    const int n = r[0];
    const int f0 = max(r[1], 0), f1 = min(r[2], C), t0 = max(r[3], 0), t1 = min(r[4], T);   // clipped: never out of bounds
    const int w = t1 - t0, h = f1 - f0;
    if (n < 0 || n >= N || w <= 0 || h <= 0) return;
    float* base = x + ((int64_t)n * C + f0) * T + t0;
    for (int i = threadIdx.x; i < w * h; i += blockDim.x) {
        const int f = i / w, t = i - f * w;
        base[(int64_t)f * T + t] = 0.f;
    }
}

}  // namespace

extern "C" int w2l_logmel(const float* audio, const int32_t* n_samples, const float* noise, float dither, float preemph, int N,
                          int64_t audio_stride, const float* window, int win_length, int n_fft, int hop, const float* fbT,
                          const int32_t* fb_range, int n_mels, int take_log, float log_guard, float* logmel, int Tmax, void* stream) {
    W2L_CHECK_ARG(audio && n_samples && window && fbT && logmel, "logmel: null pointer");
    W2L_CHECK_ARG(N > 0 && Tmax > 0 && hop > 0 && n_mels > 0, "logmel: bad sizes");
    W2L_CHECK_ARG(n_fft >= 64 && n_fft <= 1024 && (n_fft & (n_fft - 1)) == 0, "logmel: n_fft=%d must be a power of two in [64, 1024]",
                  n_fft);
    W2L_CHECK_ARG(win_length > 0 && win_length <= n_fft, "logmel: win_length=%d must be in (0, n_fft]", win_length);
    LogmelParams p;
    p.audio = audio; p.n_samples = n_samples; p.noise = noise; p.window = window; p.fbT = fbT; p.fb_range = fb_range; p.logmel = logmel;
    p.audio_stride = audio_stride; p.dither = dither; p.preemph = preemph; p.guard = log_guard;
    p.N = N; p.win_length = win_length; p.n_fft = n_fft; p.hop = hop; p.n_mels = n_mels; p.Tmax = Tmax; p.take_log = take_log;
    p.log2n = 0;
    while ((1 << p.log2n) < n_fft) ++p.log2n;
    const size_t lds = (size_t)(n_fft + FRAMES_PER_BLOCK * 2 * n_fft) * sizeof(float);
    dim3 grid((Tmax + FRAMES_PER_BLOCK - 1) / FRAMES_PER_BLOCK, N);
    if (n_fft == 512)
        hipLaunchKernelGGL(logmel512_kernel, grid, dim3(64 * FRAMES_PER_BLOCK), 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(logmel_kernel, grid, dim3(64 * FRAMES_PER_BLOCK), lds, (hipStream_t)stream, p);
    W2L_CHECK_LAUNCH();
    return 0;
}

extern "C" int w2l_feature_normalize(const float* logmel, const int32_t* n_samples, int hop, int N, int Tmax, int n_mels,
                                     float eps, float* mean_ws, float* std_ws, float* out_nct, void* stream) {
    W2L_CHECK_ARG(logmel && n_samples && mean_ws && std_ws && out_nct, "feature_normalize: null pointer");
    W2L_CHECK_ARG(N > 0 && Tmax > 0 && n_mels > 0 && hop > 0, "feature_normalize: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(feature_stats_kernel, dim3(N, (n_mels + 63) / 64), dim3(64 * STAT_WAVES), 0, st, logmel, n_samples, hop, Tmax, n_mels, eps, mean_ws, std_ws);
    W2L_CHECK_LAUNCH();
    dim3 grid((Tmax + 63) / 64, (n_mels + 63) / 64, N);
    hipLaunchKernelGGL(feature_apply_kernel, grid, dim3(256), 0, st, logmel, n_samples, hop, Tmax, n_mels, mean_ws, std_ws, out_nct);
    W2L_CHECK_LAUNCH();
    return 0;
}

extern "C" int w2l_zero_rects(float* x, int N, int C, int T, const int32_t* rects, int R, void* stream) {
    W2L_CHECK_ARG(x && N > 0 && C > 0 && T > 0 && R >= 0, "zero_rects: bad arguments");
    if (R == 0) return 0;
    W2L_CHECK_ARG(rects != nullptr, "zero_rects: null rectangle list");
    hipLaunchKernelGGL(zero_rects_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, x, N, C, T, rects, R);
    W2L_CHECK_LAUNCH();
    return 0;
}
