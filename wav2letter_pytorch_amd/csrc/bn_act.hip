// BatchNorm1d (train / eval) + Dropout + activation, forward and backward, over
// channels-last activations (gfx950).  HBM-bound: every pass moves 16 bytes per
// lane, per-channel reductions stay in registers -> LDS -> one partial row per
// block (deterministic two-stage sums, no atomics).
//
// Replaces, fused: nn.BatchNorm1d(momentum=.9, eps=1e-3) + nn.Dropout + torch.clamp(0,20)
// (wav2letter.py:37-38,43-46) and nn.BatchNorm1d(eps=1e-3, momentum=.1) + ReLU + Dropout +
// residual add + MaskedConv1d's masked_fill (jasper.py:116-119,363,376,409-410,448), and the
// reflect-padding of the next conv's input (wav2letter.py:28-34,41) incl. its backward fold.
#include "common.h"
#include "../../include/w2l_hip.h"

namespace {

// ---------------------------------------------------------------- Philox4x32-10
#ifndef W2L_PHILOX_ROUNDS
#define W2L_PHILOX_ROUNDS 10           // (experiment switch: 7, the fewest rounds that pass the Crush batteries, measured -0.6 us of 14.9)
#endif
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int i = 0; i < W2L_PHILOX_ROUNDS; ++i) {
        // one 32 x 32 -> 64 multiply per product (v_mad_u64_u32) instead of a mul_lo / mul_hi pair: the multiplies are the
        // forward BN-apply kernel's longest dependent chain
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// keep-bits for the 8 channels of group index `gidx`: bit j set <=> channel j kept.
// One Philox call gives eight 16-bit uniforms; keep iff u16 >= p * 65536.
__device__ __forceinline__ uint32_t dropout_bits(uint64_t seed, uint64_t offset, uint64_t gidx, uint32_t thresh) {
    uint32_t r[4];
    philox4x32_10((uint32_t)gidx, (uint32_t)(gidx >> 32), (uint32_t)offset, (uint32_t)(offset >> 32), (uint32_t)seed,
                  (uint32_t)(seed >> 32), r);
    uint32_t bits = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        bits |= ((r[j] & 0xFFFFu) >= thresh ? 1u : 0u) << (2 * j);
        bits |= ((r[j] >> 16) >= thresh ? 1u : 0u) << (2 * j + 1);
    }
    return bits;
}

// ---------------------------------------------------------------- 8-channel loads / stores
template <bool F32>
__device__ __forceinline__ void load8(const void* base, int64_t off, float v[8]) {
    if (F32) {
        const f32x4* p = reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + off);
        const f32x4 a = p[0], b = p[1];
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = a[j]; v[4 + j] = b[j]; }
    } else {
        const u16x8 a = *reinterpret_cast<const u16x8*>(reinterpret_cast<const bf16_raw*>(base) + off);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = bf16_bits_to_f32(a[j]);
    }
}
__device__ __forceinline__ void loadp8(const float* p, int c, float v[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p + c), b = *reinterpret_cast<const f32x4*>(p + c + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = a[j]; v[4 + j] = b[j]; }
}
__device__ __forceinline__ void store8_split(bf16_raw* hi, bf16_raw* lo, int64_t off, const float v[8]) {
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

__device__ __forceinline__ int pad_src_row(int r, int T, int pad_l, int pad_r, int pad_mode) {
    int t = r - pad_l;
    if (t >= 0 && t < T) return t;
    if (pad_mode != 1 || t < -pad_l || t >= T + pad_r) return -1;
    t = t < 0 ? -t : 2 * (T - 1) - t;
    return (t >= 0 && t < T) ? t : -1;
}

// per-channel constants of one BatchNorm branch for the 8 channels of a group
struct Chan {
    float sc[8], sh[8], m[8], is[8];
};
__device__ __forceinline__ void load_chan(Chan& ch, const float* scale, const float* shift, const float* mean,
                                          const float* invstd, int c) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { ch.sc[j] = 1.f; ch.sh[j] = 0.f; ch.m[j] = 0.f; ch.is[j] = 0.f; }
    if (scale) { loadp8(scale, c, ch.sc); loadp8(shift, c, ch.sh); }
    if (mean) { loadp8(mean, c, ch.m); loadp8(invstd, c, ch.is); }
}

// pre-activation value z = dropout(bn(y) [+ bn2(y2)]) for 8 channels; returns the keep bits
template <bool F32, bool HAS2>
__device__ __forceinline__ uint32_t preact8(const w2l_bnact_t& d, const Chan& c1, const Chan& c2, int64_t row /* n*T+t */,
                                            int cg, int G, float z[8], float y1[8], float y2v[8], uint32_t thresh,
                                            float inv_keep, bool gen_mask, bool write_mask) {
    const int64_t off = row * d.C + cg * 8;
    load8<F32>(d.y, off, y1);
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] = y1[j] * c1.sc[j] + c1.sh[j];
    if (HAS2) {
        load8<F32>(d.y2, off, y2v);
#pragma unroll
        for (int j = 0; j < 8; ++j) z[j] += y2v[j] * c2.sc[j] + c2.sh[j];
    }
    uint32_t bits = 0xFFu;
    if (d.drop_p > 0.f) {
        const int64_t gidx = row * G + cg;
        if (gen_mask) {      // forward: every row (halo copies too) regenerates its bits; the primary row records them
            bits = dropout_bits(d.seed, d.offset + (d.offset_dev ? *d.offset_dev : 0ull), (uint64_t)gidx, thresh);
            if (write_mask) d.mask[gidx] = (uint8_t)bits;
        } else {             // backward: replay the recorded bits
            bits = d.mask[gidx];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) z[j] = (bits >> j) & 1u ? z[j] * inv_keep : 0.f;
    }
    return bits;
}

__device__ __forceinline__ float activate(float z, int act) {
    if (act == 1) return fminf(fmaxf(z, 0.f), 20.f);
    if (act == 2) return fmaxf(z, 0.f);
    return z;
}
// (the apply kernels take the activation as a template parameter: a run-time switch costs them scalar branches per element)
template <int ACT>
__device__ __forceinline__ float activate_t(float z) {
    if (ACT == 1) return fminf(fmaxf(z, 0.f), 20.f);
    if (ACT == 2) return fmaxf(z, 0.f);
    return z;
}
// gradient gate: torch.clamp passes 1 on the CLOSED interval [0,20]; ReLU passes on z > 0
__device__ __forceinline__ bool act_pass(float z, int act) {
    if (act == 1) return z >= 0.f && z <= 20.f;
    if (act == 2) return z > 0.f;
    return true;
}

// device-side amax of dy (fp8 mode): W2L_AMAX_SLOTS partial maxima per tensor, the slot picked by the block index -- one
// word for a whole launch serialises 16 000 atomics at one L2 address (measured: the dy kernel 33 -> 171 us)
constexpr int AMAX_SLOTS = W2L_AMAX_SLOTS;
// a wave's maximum onto its slot: same-address device-scope atomics run one after the other at ~0.1 us each (the dy pass of a
// wide unit with one row group per wave makes 28 000 of them: +53 us per unit, measured as +1.07 ms per fp8 step), so a wave
// first LOOKS (one device-coherent load: parallel, never stale in the unsafe direction -- the slot only grows) and adds its
// atomic only if it would raise the slot: after the first few waves of a slot nearly none does
__device__ __forceinline__ void amax_publish(float* amax, int slot, float mx) {
    unsigned* p = reinterpret_cast<unsigned*>(amax) + slot;
    const unsigned v = __float_as_uint(mx);
    if (v > __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(p, v);
}
constexpr int BWD_SLAB = 64;             // channels per wave of the slab-form kernels (one 128-byte line per row)

// 8 floats -> 8 OCP e4m3 bytes (round to nearest even, saturating at +-448), v * scale
__device__ __forceinline__ uint2 quant8_e4m3(const float v[8], float scale) {
    float q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) q[j] = fminf(fmaxf(v[j] * scale, -448.f), 448.f);
    int lo = __builtin_amdgcn_cvt_pk_fp8_f32(q[0], q[1], 0, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(q[2], q[3], lo, true);
    int hi = __builtin_amdgcn_cvt_pk_fp8_f32(q[4], q[5], 0, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(q[6], q[7], hi, true);
    return make_uint2((unsigned)lo, (unsigned)hi);
}

// ---------------------------------------------------------------- forward
// out_q (optional): the same padded activation once more as e4m3 bytes, a * q_scale -- the operand of the next
// convolution's forward pass in fp8 mode (w2l_conv1d_igemm_fp8); the bf16 copy stays the weight gradient's operand
template <bool F32, bool HAS2, int ACT>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(w2l_bnact_t d, bf16_raw* out_hi, bf16_raw* out_lo, int R,
                                                          int pad_l, int pad_r, int pad_mode, uint32_t thresh,
                                                          float inv_keep, uint8_t* out_q, float q_scale, float inv_g) {
    // grid = (groups of one utterance / 256, N): the utterance is the block's y index, and the one remaining division (by the
    // number of 8-channel groups per row) is a float reciprocal with a fix-up -- exact, since rows x groups of ONE utterance is
    // below 2^24 (checked by the launcher).  The two integer divisions this replaces were a sixth of the kernel's VALU work.
    const int G = d.C >> 3;
    const int n = blockIdx.y;
    unsigned clipped = 0;                                    // e4m3 copy: elements beyond the format's range at this scale
    const float q_limit = 448.f / q_scale;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx < R * G) {
        int r = (int)((float)idx * inv_g);
        r -= (r * G > idx) ? 1 : 0;
        r += ((r + 1) * G <= idx) ? 1 : 0;
        const int cg = idx - r * G;
        const unsigned orow = (unsigned)n * R + r;
        const int t = pad_src_row(r, d.T, pad_l, pad_r, pad_mode);
        float a[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = 0.f;
        if (t >= 0 && (!d.lens || t < d.lens[n])) {
            float z[8], y1[8], y2v[8];
            Chan c1, c2;
            load_chan(c1, d.scale, d.shift, nullptr, nullptr, cg * 8);
            if (HAS2) load_chan(c2, d.scale2, d.shift2, nullptr, nullptr, cg * 8);
            preact8<F32, HAS2>(d, c1, c2, (int64_t)n * d.T + t, cg, G, z, y1, y2v, thresh, inv_keep, /*gen_mask=*/true,
                         /*write_mask=*/r - pad_l == t);
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = activate_t<ACT>(z[j]);
        }
        store8_split(out_hi, out_lo, (int64_t)orow * d.C + cg * 8, a);
        if (out_q) {
            *reinterpret_cast<uint2*>(out_q + (int64_t)orow * d.C + cg * 8) = quant8_e4m3(a, q_scale);
            if (d.q_clipped && r - pad_l == t) {             // (halo copies of a frame are not counted twice)
#pragma unroll
                for (int j = 0; j < 8; ++j) clipped += fabsf(a[j]) > q_limit ? 1u : 0u;
            }
        }
    }
    if (out_q && d.q_clipped && __any(clipped != 0)) {       // rare: one atomic per wave that saw any
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) clipped += __shfl_xor(clipped, m, 64);
        if ((threadIdx.x & 63) == 0) atomicAdd(reinterpret_cast<unsigned long long*>(d.q_clipped), (unsigned long long)clipped);
    }
}

// ---- forward with the statistics finalize folded in ("slab" form; round 5).  The separate bn_finalize launch is a 5 us kernel
// that costs its layer 17-20 us (two more stream boundaries on the critical path).  Here a block owns a
// 64-channel slab (one 128-byte line per row) of a range of OUTPUT rows; its prologue sums the few partial rows of ITS 64
// channels (w2l_conv_stats_mode(S): the convolution's epilogue folds its per-tile sums onto S <= 64 rows with fp32 atomics --
// 2 x S x 64 floats per block instead of 2 x 125 x C), derives mean / invstd / scale / shift exactly as bn_finalize_kernel
// does, and the blocks of row chunk 0 publish them (the backward pass reads them) and update the running statistics.  Then
// each wave walks its rows eight at a time in batches whose loads are all issued before the first use.  Measured
// (profiles/r05_bn_kernels.txt, C = 896): 19.6-21 us whatever the batch size (1 / 2 / 4 row groups) and the blocks per launch
// (1 280 / 2 560) -- 4.5 us more than the flat apply pass alone (15.1), i.e. the slab form runs at 2.9 instead of 3.8 TB/s; what
// the fold saves is the finalize launch (6 us + a stream boundary).  bf16 y only (the fp32 parity mode keeps the deterministic
// two-kernel path).
struct FinBranch {
    const float* partial;      // [rows][2][C] sums / sums of squares; NULL: take scale / shift from the descriptor
    int rows;
    double count;
    const float* gamma;
    const float* beta;
    float eps, momentum;
    float* running_mean;
    float* running_var;
    float* mean;
    float* invstd;
    float* scale;
    float* shift;
};

#ifndef W2L_FWD_FIN_U
#define W2L_FWD_FIN_U 1
#endif
constexpr int FWD_FIN_U = W2L_FWD_FIN_U;               // row groups (of 8 rows) per wave of bn_act_fwd_fin_kernel
#ifndef W2L_FWD_FIN_BLOCKS
#define W2L_FWD_FIN_BLOCKS 2560
#endif
__host__ __device__ inline int fwd_rows_per_block(int64_t rows, int C) {
    // batches of 4 waves x U groups x 8 rows; as many batches per block as leave ~W2L_FWD_FIN_BLOCKS blocks in the launch (one
    // round of resident blocks: the prologue -- the statistics of the block's slab -- is paid once per block)
    const int batch = 4 * FWD_FIN_U * 8;
    int64_t nb = (rows * (C / BWD_SLAB) + (int64_t)batch * W2L_FWD_FIN_BLOCKS - 1) / ((int64_t)batch * W2L_FWD_FIN_BLOCKS);
    nb = nb < 1 ? 1 : (nb > 16 ? 16 : nb);
    return (int)nb * batch;
}

template <bool HAS2, int ACT>
__global__ __launch_bounds__(256) void bn_act_fwd_fin_kernel(w2l_bnact_t d, FinBranch f1, FinBranch f2, bf16_raw* out_hi, int R,
                                                              int pad_l, int pad_r, int pad_mode, uint32_t thresh, float inv_keep,
                                                              uint8_t* out_q, float q_scale, int rpb) {
    __shared__ double red[2][2][BWD_SLAB];
    __shared__ float s_sc[2][BWD_SLAB], s_sh[2][BWD_SLAB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = d.C >> 3;
    const int nslabs = d.C / BWD_SLAB;
    const int slab = blockIdx.x % nslabs, chunk = blockIdx.x / nslabs;
    const int cgl = lane & 7, rr = lane >> 3;
    const int cg = slab * (BWD_SLAB / 8) + cgl, c = cg * 8;
    const int T = d.T;
    const int64_t rows = (int64_t)d.N * R;
    constexpr int U = FWD_FIN_U;                       // row groups per batch: a wave walks rpb / 4 rows in batches of U x 8
    const int nbatch = rpb / (4 * U * 8);
    int64_t q0 = (int64_t)chunk * rpb + (int64_t)wave * (U * 8) + rr;        // (batch b of wave w: rows + b * 4 * U * 8)
    // ---- every load of a batch is issued FIRST; the first batch's before the prologue: they do not depend on the statistics,
    // and the prologue's own round trip (the partial rows, two barriers) hides behind them
    u16x8 ya[U], yb[U];
    int nn[U], tt[U], rw[U];
    bool ok[U];
    auto load_batch = [&]() {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t q = q0 + 8 * u;
            ok[u] = false;
            nn[u] = 0; tt[u] = 0; rw[u] = 0;
            if (q < rows) {
                const int n = (int)(q / R), r = (int)(q - (int64_t)n * R);
                const int t = pad_src_row(r, T, pad_l, pad_r, pad_mode);
                nn[u] = n; tt[u] = t; rw[u] = r;
                ok[u] = t >= 0 && (!d.lens || t < d.lens[n]);
                if (ok[u]) {
                    const int64_t src = ((int64_t)n * T + t) * d.C + c;
                    ya[u] = *reinterpret_cast<const u16x8*>(reinterpret_cast<const bf16_raw*>(d.y) + src);
                    if (HAS2) yb[u] = *reinterpret_cast<const u16x8*>(reinterpret_cast<const bf16_raw*>(d.y2) + src);
                }
            }
        }
    };
    load_batch();
    // ---- prologue: this slab's statistics
    {
        const int cc = tid & 63, ch = slab * BWD_SLAB + cc;
        const int br = tid >> 7, k = (tid >> 6) & 1;              // branch, component (sum / sum of squares)
        const FinBranch& f = br ? f2 : f1;
        if ((br == 0 || HAS2) && f.partial) {
            double a = 0.0;
            int j = 0;
            for (; j + 8 <= f.rows; j += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = f.partial[((int64_t)(j + u) * 2 + k) * d.C + ch];
#pragma unroll
                for (int u = 0; u < 8; ++u) a += v[u];
            }
            for (; j < f.rows; ++j) a += f.partial[((int64_t)j * 2 + k) * d.C + ch];
            red[br][k][cc] = a;
        }
        __syncthreads();
        if (tid < 128 && (tid < 64 || HAS2)) {
            const int b = tid >> 6;
            const FinBranch& g = b ? f2 : f1;
            float sc = 1.f, sh = 0.f;
            if (g.partial) {
                const double mu = red[b][0][cc] / g.count;
                double var = red[b][1][cc] / g.count - mu * mu;   // biased: normalisation uses it
                if (var < 0.0) var = 0.0;
                const float m = (float)mu, istd = (float)(1.0 / sqrt(var + (double)g.eps));
                const float ga = g.gamma ? g.gamma[ch] : 1.f, be = g.beta ? g.beta[ch] : 0.f;
                sc = ga * istd;
                sh = be - m * ga * istd;
                if (chunk == 0) {
                    if (g.running_mean) {
                        const double unbiased = g.count > 1.0 ? var * g.count / (g.count - 1.0) : var;
                        g.running_mean[ch] = (1.f - g.momentum) * g.running_mean[ch] + g.momentum * m;
                        g.running_var[ch] = (1.f - g.momentum) * g.running_var[ch] + g.momentum * (float)unbiased;
                    }
                    if (g.mean) { g.mean[ch] = m; g.invstd[ch] = istd; }
                    g.scale[ch] = sc;
                    g.shift[ch] = sh;
                }
            } else {
                const float* scp = b ? d.scale2 : d.scale;
                const float* shp = b ? d.shift2 : d.shift;
                if (scp) { sc = scp[ch]; sh = shp[ch]; }
            }
            s_sc[b][cc] = sc;
            s_sh[b][cc] = sh;
        }
        __syncthreads();
    }
    float sc1[8], sh1[8], sc2[8], sh2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        sc1[j] = s_sc[0][cgl * 8 + j]; sh1[j] = s_sh[0][cgl * 8 + j];
        sc2[j] = HAS2 ? s_sc[1][cgl * 8 + j] : 0.f; sh2[j] = HAS2 ? s_sh[1][cgl * 8 + j] : 0.f;
    }
    unsigned clipped = 0;
    const float q_limit = 448.f / q_scale;
    const bool drop = d.drop_p > 0.f;
    const uint64_t off = drop ? d.offset + (d.offset_dev ? *d.offset_dev : 0ull) : 0ull;
    for (int bt = 0; bt < nbatch; ++bt) {
    if (bt) load_batch();
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t q = q0 + 8 * u;
        if (q >= rows) break;
        float a[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = 0.f;
        if (ok[u]) {
            float z[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                z[j] = bf16_bits_to_f32(ya[u][j]) * sc1[j] + sh1[j];
                if (HAS2) z[j] += bf16_bits_to_f32(yb[u][j]) * sc2[j] + sh2[j];
            }
            if (drop) {                            // every row (halo copies too) regenerates its bits; the primary row records them
                const int64_t gidx = ((int64_t)nn[u] * T + tt[u]) * G + cg;
                const uint32_t bits = dropout_bits(d.seed, off, (uint64_t)gidx, thresh);
                if (rw[u] - pad_l == tt[u]) d.mask[gidx] = (uint8_t)bits;
#pragma unroll
                for (int j = 0; j < 8; ++j) z[j] = (bits >> j) & 1u ? z[j] * inv_keep : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = activate_t<ACT>(z[j]);
        }
        store8_split(out_hi, nullptr, q * d.C + c, a);
        if (out_q) {
            *reinterpret_cast<uint2*>(out_q + q * d.C + c) = quant8_e4m3(a, q_scale);
            if (d.q_clipped && rw[u] - pad_l == tt[u]) {
#pragma unroll
                for (int j = 0; j < 8; ++j) clipped += fabsf(a[j]) > q_limit ? 1u : 0u;
            }
        }
    }
    q0 += 4 * U * 8;
    }
    if (out_q && d.q_clipped && __any(clipped != 0)) {
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) clipped += __shfl_xor(clipped, m, 64);
        if (lane == 0) atomicAdd(reinterpret_cast<unsigned long long*>(d.q_clipped), (unsigned long long)clipped);
    }
}

__global__ __launch_bounds__(256) void quantize_e4m3_kernel(const void* src, int src_f32, int64_t ngroups, float scale,
                                                             uint8_t* dst) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < ngroups; i += (int64_t)gridDim.x * 256) {
        float v[8];
        if (src_f32) load8<true>(src, i * 8, v);
        else load8<false>(src, i * 8, v);
        *reinterpret_cast<uint2*>(dst + i * 8) = quant8_e4m3(v, scale);
    }
}

// ---------------------------------------------------------------- backward helpers
template <bool GF32>
__device__ __forceinline__ void add_grad8(const w2l_gradsrc_t& s, int n, int t, int T, int C, int cg, float g[8]) {
    const int64_t base = (int64_t)n * s.rows;
    float v[8];
    load8<GF32>(s.dxp, (base + t + s.pad_l) * C + cg * 8, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] += v[j];
    if (s.pad_mode == 1) {                       // fold the reflected halo rows back onto their source frame
        if (t >= 1 && t <= s.pad_l) {
            load8<GF32>(s.dxp, (base + s.pad_l - t) * C + cg * 8, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) g[j] += v[j];
        }
        if (t <= T - 2 && t >= T - 1 - s.pad_r) {
            load8<GF32>(s.dxp, (base + s.pad_l + 2 * (T - 1) - t) * C + cg * 8, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) g[j] += v[j];
        }
    }
}

struct BwdRow {
    float g[8];       // gradient wrt bn output(s) after activation / dropout gates
    float xh1[8];     // normalised input of branch 1
    float xh2[8];
};

template <bool F32, bool GF32, bool HAS2>
__device__ __forceinline__ void bwd_row(const w2l_bnact_t& d, const Chan& c1, const Chan& c2, const w2l_gradsrc_t& g1,
                                        const w2l_gradsrc_t& g2, int has_g2, int n, int t, int cg, int G, float inv_keep,
                                        BwdRow& o) {
    float z[8], y1[8], y2v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { o.g[j] = 0.f; o.xh1[j] = 0.f; o.xh2[j] = 0.f; y2v[j] = 0.f; }
    const uint32_t bits = preact8<F32, HAS2>(d, c1, c2, (int64_t)n * d.T + t, cg, G, z, y1, y2v, 0, inv_keep, false, false);
    // masked_fill (jasper.py:116-119) zeroes the OUTPUT frame t >= len: no gradient flows through it, but the
    // frame still is a BatchNorm sample, so its normalised value enters dy through the -xhat*sum(g*xhat)/M term
    const bool masked = d.lens && t >= d.lens[n];
    float g[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] = 0.f;
    if (!masked) {
        add_grad8<GF32>(g1, n, t, d.T, d.C, cg, g);
        if (has_g2) add_grad8<GF32>(g2, n, t, d.T, d.C, cg, g);
    }
    const float gk = d.drop_p > 0.f ? inv_keep : 1.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const bool keep = (bits >> j) & 1u;
        o.g[j] = (!masked && keep && act_pass(z[j], d.act)) ? g[j] * gk : 0.f;
        o.xh1[j] = (y1[j] - c1.m[j]) * c1.is[j];           // is == 0 when the branch has no BatchNorm
        if (HAS2) o.xh2[j] = (y2v[j] - c2.m[j]) * c2.is[j];
    }
}

// ---- backward reduction: partial[chunk][ncomp][C] = sum over the chunk's rows of g, g*xh1 [, g, g*xh2 with a residual
// branch].  One WAVE owns a 64-channel slab (8 lanes x 16 bytes = one 128-byte line per row) of one row chunk and walks it
// 8 rows at a time; the eight row-lanes are combined with lane shuffles and lanes 0..7 store the slab's sums.  No LDS and
// no block barrier: the kernel's occupancy is bounded by registers only, so its blocks fit next to whatever else is
// resident (the LDS version -- 24 KB per block -- ran at one block per CU beside the weight-gradient kernel, which takes
// 132 of a CU's 160 KB).  The four waves of a block take four adjacent slabs of the same rows.  Deterministic.
__host__ __device__ inline int bwd_rows_per_wave(int64_t rows, int C) {
    // ~4096 wave tasks per launch: 16 rows for the narrow layers, up to 64 for the wide ones
    // (rounded UP to whole 8-row steps: rounding down left 4 600-6 000 tasks, i.e. a second, mostly empty round of waves behind
    // the 4 096 that are resident at four per SIMD)
    int64_t rw = (rows * (C / BWD_SLAB) + 4095) / 4096;
    rw = (rw + 7) / 8 * 8;
    return (int)(rw < 16 ? 16 : (rw > 64 ? 64 : rw));
}

template <bool F32, bool GF32, bool HAS2>
__global__ __launch_bounds__(256) void bn_act_bwd_reduce_kernel(w2l_bnact_t d, w2l_gradsrc_t g1, w2l_gradsrc_t g2,
                                                                 int has_g2, float* partial, float inv_keep, int rw,
                                                                 int nchunks) {
    const int G = d.C >> 3;
    const int nslabs = d.C / BWD_SLAB;
    const int lane = threadIdx.x & 63;
    const int task = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (task >= nslabs * nchunks) return;                     // (whole waves leave: no barrier below)
    const int chunk = task / nslabs, slab = task - chunk * nslabs;
    const int cg = slab * (BWD_SLAB / 8) + (lane & 7);        // 8-channel group of this lane
    const int rr = lane >> 3;                                 // row lane 0..7
    constexpr int ncomp = HAS2 ? 4 : 2;
    const int64_t rows = (int64_t)d.N * d.T;
    const int64_t row0 = (int64_t)chunk * rw;
    int64_t rend = row0 + rw;
    if (rend > rows) rend = rows;
    float s0[8], s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s0[j] = 0.f; s1[j] = 0.f; s2[j] = 0.f; }
    Chan c1, c2;
    load_chan(c1, d.scale, d.shift, d.mean, d.invstd, cg * 8);
    if (HAS2) load_chan(c2, d.scale2, d.shift2, d.mean2, d.invstd2, cg * 8);
    int n = (int)((row0 + rr) / d.T), t = (int)((row0 + rr) - (int64_t)n * d.T);
#pragma unroll 2
    for (int64_t row = row0 + rr; row < rend; row += 8, t += 8) {
        while (t >= d.T) { t -= d.T; ++n; }
        BwdRow o;
        bwd_row<F32, GF32, HAS2>(d, c1, c2, g1, g2, has_g2, n, t, cg, G, inv_keep, o);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            s0[j] += o.g[j];
            s1[j] += o.g[j] * o.xh1[j];
            if (HAS2) s2[j] += o.g[j] * o.xh2[j];
        }
    }
#pragma unroll
    for (int m = 8; m < 64; m <<= 1)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            s0[j] += __shfl_xor(s0[j], m, 64);
            s1[j] += __shfl_xor(s1[j], m, 64);
            if (HAS2) s2[j] += __shfl_xor(s2[j], m, 64);
        }
    if (rr == 0) {
        float* dst = partial + (int64_t)chunk * ncomp * d.C + cg * 8;
        auto put = [&](float* q, const float* v) {
            *reinterpret_cast<f32x4*>(q) = f32x4{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(q + 4) = f32x4{v[4], v[5], v[6], v[7]};
        };
        put(dst, s0);
        put(dst + d.C, s1);
        if (ncomp == 4) { put(dst + 2 * d.C, s0); put(dst + 3 * d.C, s2); }
    }
}

// column sums of partial[nblocks][ncols]: 32 columns x 8 row-lanes per block, coalesced 128-byte rows,
// 8 independent loads in flight per lane (the kernel is a latency chain otherwise: ~60 dependent L2 reads)
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* partial, int nblocks, int ncols, float* sums) {
    __shared__ float red[8][33];
    const int cx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    const int col = blockIdx.x * 32 + cx;
    float s = 0.f;
    if (col < ncols) {
        int b = ly;
        float a[8];
        for (; b + 56 < nblocks; b += 64) {
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] = partial[(int64_t)(b + 8 * u) * ncols + col];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += a[u];
        }
        for (; b < nblocks; b += 8) s += partial[(int64_t)b * ncols + col];
    }
    red[ly][cx] = s;
    __syncthreads();
    if (ly == 0 && col < ncols) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red[k][cx];
        sums[col] = t;
    }
}

// work items: [0, nvalid): one (n, t, channel-group) each -> compute dy (and dy2) and store into the
// shared-halo buffers; [nvalid, nvalid + z1): zero rows of dy; then z2 zero rows of dy2.
template <bool F32, bool GF32, bool HAS2>
__global__ __launch_bounds__(256) void bn_act_bwd_apply_kernel(w2l_bnact_t d, w2l_gradsrc_t g1, w2l_gradsrc_t g2,
                                                                int has_g2, const float* sums, bf16_raw* dy_hi,
                                                                bf16_raw* dy_lo, int h1, bf16_raw* dy2_hi,
                                                                bf16_raw* dy2_lo, int h2, float inv_keep, float* amax) {
    float mx1 = 0.f, mx2 = 0.f;                  // fp8 mode: running max |dy| (and |dy2|) of this thread
    const int G = d.C >> 3;
    const int T = d.T, N = d.N;
    const float invM = 1.f / ((float)N * (float)T);
    const unsigned nvalid = (unsigned)N * T * G;             // all < 2^31 (checked by the launcher)
    const unsigned z1 = (unsigned)h1 * (N + 1) * G;
    const unsigned z2 = dy2_hi ? (unsigned)h2 * (N + 1) * G : 0u;
    const unsigned total = nvalid + z1 + z2;
    for (unsigned it = blockIdx.x * 256u + threadIdx.x; it < total; it += gridDim.x * 256u) {
        if (it >= nvalid) {                       // zero-fill a halo row group
            unsigned k = it - nvalid;
            bf16_raw* hi = dy_hi; bf16_raw* lo = dy_lo; int h = h1;
            if (k >= z1) { k -= z1; hi = dy2_hi; lo = dy2_lo; h = h2; }
            const unsigned hr = k / (unsigned)G;   // index among the (N+1)*h halo rows
            const int cg = (int)(k - hr * G);
            const int gap = (int)(hr / (unsigned)h), r = (int)(hr - (unsigned)gap * h);
            const int64_t row = (int64_t)gap * (T + h) + r;
            float z[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) z[j] = 0.f;
            store8_split(hi, lo, row * d.C + cg * 8, z);
            continue;
        }
        const unsigned vrow = it / (unsigned)G;
        const int cg = (int)(it - vrow * G);
        const int n = (int)(vrow / (unsigned)T);
        const int t = (int)(vrow - (unsigned)n * T);
        float o1[8], o2[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { o1[j] = 0.f; o2[j] = 0.f; }
        BwdRow o;
        const int c = cg * 8;
        Chan c1, c2;
        load_chan(c1, d.scale, d.shift, d.mean, d.invstd, c);
        if (HAS2) load_chan(c2, d.scale2, d.shift2, d.mean2, d.invstd2, c);
        bwd_row<F32, GF32, HAS2>(d, c1, c2, g1, g2, has_g2, n, t, cg, G, inv_keep, o);
        if (d.mean) {
            float sg[8], sgx[8];
            loadp8(sums, c, sg);
            loadp8(sums + d.C, c, sgx);
#pragma unroll
            for (int j = 0; j < 8; ++j) o1[j] = c1.sc[j] * (o.g[j] - sg[j] * invM - o.xh1[j] * sgx[j] * invM);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) o1[j] = o.g[j] * c1.sc[j];
        }
        store8_split(dy_hi, dy_lo, ((int64_t)h1 + (int64_t)n * (T + h1) + t) * d.C + c, o1);
        if (amax) {
#pragma unroll
            for (int j = 0; j < 8; ++j) mx1 = fmaxf(mx1, fabsf(o1[j]));
        }
        if (HAS2 && dy2_hi) {
            if (d.mean2) {
                float sg[8], sgx[8];
                loadp8(sums + 2 * d.C, c, sg);
                loadp8(sums + 3 * d.C, c, sgx);
#pragma unroll
                for (int j = 0; j < 8; ++j) o2[j] = c2.sc[j] * (o.g[j] - sg[j] * invM - o.xh2[j] * sgx[j] * invM);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) o2[j] = o.g[j] * c2.sc[j];
            }
            store8_split(dy2_hi, dy2_lo, ((int64_t)h2 + (int64_t)n * (T + h2) + t) * d.C + c, o2);
            if (amax) {
#pragma unroll
                for (int j = 0; j < 8; ++j) mx2 = fmaxf(mx2, fabsf(o2[j]));
            }
        }
    }
    if (amax) {          // non-negative floats order like their bit patterns: one integer atomic max per wave
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            mx1 = fmaxf(mx1, __shfl_xor(mx1, m, 64));
            mx2 = fmaxf(mx2, __shfl_xor(mx2, m, 64));
        }
        if ((threadIdx.x & 63) == 0) {
            const int slot = blockIdx.x & (AMAX_SLOTS - 1);
            amax_publish(amax, slot, mx1);
            if (HAS2 && dy2_hi) amax_publish(amax, AMAX_SLOTS + slot, mx2);
        }
    }
}

// ---- backward apply with the finalize folded in ("slab" form).  A block owns a 64-channel slab (one 128-byte line per
// row) of a row range.  Its prologue re-reduces the per-tile partial sums of ITS 64 channels -- nb x ncomp x 64 floats,
// L2-resident -- in a fixed order, so every block of a slab arrives at bit-identical sums and the separate
// bn_bwd_finalize launch (13 us of kernel on the backward critical path, 25+ us of queueing behind the weight-gradient
// blocks) disappears; the blocks of row chunk 0 also publish the sums (d beta, d gamma).  Then each wave walks its rows
// eight at a time exactly like the reduction kernel above.  Halo rows of the shared-halo dy layout are zero-filled by
// the same blocks, a share per row chunk.
__host__ __device__ inline int apply_rows_per_block(int64_t rows, int C) {
    int64_t r = rows * (C / BWD_SLAB) / 1536;          // ~1536 blocks per launch
    r = (r + 31) / 32 * 32;
    return (int)(r < 64 ? 64 : (r > 1024 ? 1024 : r));
}

template <bool F32, bool GF32, bool HAS2>
__global__ __launch_bounds__(256) void bn_act_bwd_apply_fin_kernel(w2l_bnact_t d, w2l_gradsrc_t g1, w2l_gradsrc_t g2,
                                                                    int has_g2, const float* partial, int nb, float* sums_out,
                                                                    bf16_raw* dy_hi, bf16_raw* dy_lo, int h1, bf16_raw* dy2_hi,
                                                                    bf16_raw* dy2_lo, int h2, float inv_keep, float* amax,
                                                                    int rpb) {
    constexpr int ncomp = HAS2 ? 4 : 2;
    __shared__ float red[4][ncomp][BWD_SLAB];
    __shared__ float ssum[ncomp][BWD_SLAB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = d.C >> 3;
    const int nslabs = d.C / BWD_SLAB;
    const int slab = blockIdx.x % nslabs, chunk = blockIdx.x / nslabs;     // neighbours: adjacent slabs of the same rows
    const int T = d.T, N = d.N;
    // ---- prologue: column sums of partial[nb][ncomp][C] over this slab's channels (thread = channel x quarter of the rows)
    {
        const int c = slab * BWD_SLAB + (tid & 63);
        float a[ncomp];
#pragma unroll
        for (int k = 0; k < ncomp; ++k) a[k] = 0.f;
        int j = wave;
        for (; j + 12 < nb; j += 16) {                 // four rows in flight per component
            float v[4][ncomp];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < ncomp; ++k) v[u][k] = partial[((int64_t)(j + 4 * u) * ncomp + k) * d.C + c];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < ncomp; ++k) a[k] += v[u][k];
        }
        for (; j < nb; j += 4)
#pragma unroll
            for (int k = 0; k < ncomp; ++k) a[k] += partial[((int64_t)j * ncomp + k) * d.C + c];
#pragma unroll
        for (int k = 0; k < ncomp; ++k) red[wave][k][tid & 63] = a[k];
        __syncthreads();
        if (tid < ncomp * BWD_SLAB) {
            const int k = tid >> 6, cc = tid & 63;
            const float t4 = (red[0][k][cc] + red[1][k][cc]) + (red[2][k][cc] + red[3][k][cc]);
            ssum[k][cc] = t4;
            if (chunk == 0) sums_out[(int64_t)k * d.C + slab * BWD_SLAB + cc] = t4;
        }
        __syncthreads();
    }
    const int cgl = lane & 7, rr = lane >> 3;
    const int cg = slab * (BWD_SLAB / 8) + cgl, c = cg * 8;
    Chan c1, c2;
    load_chan(c1, d.scale, d.shift, d.mean, d.invstd, c);
    if (HAS2) load_chan(c2, d.scale2, d.shift2, d.mean2, d.invstd2, c);
    float sg[8], sgx[8], sg2[8], sgx2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        sg[j] = ssum[0][cgl * 8 + j];
        sgx[j] = ssum[1][cgl * 8 + j];
        sg2[j] = HAS2 ? ssum[2][cgl * 8 + j] : 0.f;
        sgx2[j] = HAS2 ? ssum[3][cgl * 8 + j] : 0.f;
    }
    const float invM = 1.f / ((float)N * (float)T);
    const int64_t rows = (int64_t)N * T;
    const int rpw = rpb >> 2;                          // rows per wave (multiple of 8)
    const int64_t row0 = (int64_t)chunk * rpb + (int64_t)wave * rpw;
    int64_t rend = row0 + rpw;
    if (rend > rows) rend = rows;
    float mx1 = 0.f, mx2 = 0.f;
    if (row0 + rr < rend) {
        int n = (int)((row0 + rr) / T), t = (int)((row0 + rr) - (int64_t)n * T);
#pragma unroll 2
        for (int64_t row = row0 + rr; row < rend; row += 8, t += 8) {
            while (t >= T) { t -= T; ++n; }
            BwdRow o;
            bwd_row<F32, GF32, HAS2>(d, c1, c2, g1, g2, has_g2, n, t, cg, G, inv_keep, o);
            float o1[8], o2[8];
            if (d.mean) {
#pragma unroll
                for (int j = 0; j < 8; ++j) o1[j] = c1.sc[j] * (o.g[j] - sg[j] * invM - o.xh1[j] * sgx[j] * invM);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) o1[j] = o.g[j] * c1.sc[j];
            }
            store8_split(dy_hi, dy_lo, ((int64_t)h1 + (int64_t)n * (T + h1) + t) * d.C + c, o1);
            if (amax) {
#pragma unroll
                for (int j = 0; j < 8; ++j) mx1 = fmaxf(mx1, fabsf(o1[j]));
            }
            if (HAS2 && dy2_hi) {
                if (d.mean2) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) o2[j] = c2.sc[j] * (o.g[j] - sg2[j] * invM - o.xh2[j] * sgx2[j] * invM);
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) o2[j] = o.g[j] * c2.sc[j];
                }
                store8_split(dy2_hi, dy2_lo, ((int64_t)h2 + (int64_t)n * (T + h2) + t) * d.C + c, o2);
                if (amax) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) mx2 = fmaxf(mx2, fabsf(o2[j]));
                }
            }
        }
    }
    // ---- halo rows of the shared-halo layout: (N+1) gaps of h rows each, this block's share (its slab's 128 bytes per row)
    {
        const int nchunks = gridDim.x / nslabs;
        float z[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) z[j] = 0.f;
        for (int pass = 0; pass < 2; ++pass) {
            bf16_raw* hi = pass ? dy2_hi : dy_hi;
            bf16_raw* lo = pass ? dy2_lo : dy_lo;
            const int h = pass ? h2 : h1;
            if (hi == nullptr || h == 0) continue;
            const int total = h * (N + 1);
            const int per = (total + nchunks - 1) / nchunks;
            int e = (chunk + 1) * per;
            if (e > total) e = total;
            for (int hr = chunk * per + wave * 8 + rr; hr < e; hr += 32) {
                const int gap = hr / h, r = hr - gap * h;
                store8_split(hi, lo, ((int64_t)gap * (T + h) + r) * d.C + c, z);
            }
        }
    }
    if (amax) {
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            mx1 = fmaxf(mx1, __shfl_xor(mx1, m, 64));
            mx2 = fmaxf(mx2, __shfl_xor(mx2, m, 64));
        }
        if (lane == 0) {
            const int slot = blockIdx.x & (AMAX_SLOTS - 1);
            amax_publish(amax, slot, mx1);
            if (HAS2 && dy2_hi) amax_publish(amax, AMAX_SLOTS + slot, mx2);
        }
    }
}

// ---- the backward chain's fast path (round 5): bf16 y, bf16 gradient from ONE source, one branch -- every Wav2Letter unit and
// every Jasper unit that is not a block end.  Two launches instead of three, both in the slab form (a wave = 64 channels x 8
// row lanes, one 128-byte line per row) with EVERY load of a wave's rows issued before the first use:
//   bn_bwd_reduce_fast_kernel: sums of g*gate and g*gate*xhat; the four waves of a block (four consecutive 32-row chunks of
//     one slab) combine in LDS and ADD their 2 x 64 sums onto row (chunk mod S) of a zero-filled [S][2][C] buffer (fp32
//     atomics: ~16 adds per address over the launch);
//   bn_bwd_apply_fast_kernel: every block re-reduces the S rows of its 64 channels (the finalize folded in; the blocks of
//     row chunk 0 publish d beta / d gamma), then forms dy.
// The separate finalize launch -- a 5 us kernel between two stream boundaries on the backward critical path -- is gone.
// Row groups (of 8 rows) per wave, measured (profiles/r05_bn_kernels.txt; C = 896: reduction 22.0 / 18.2 / 20.1 / 27.8 us and dy
// pass 24.4 / 25.2 / 28.7 / 34.3 us at 1 / 2 / 4 / 8 groups): more rows per wave = more loads in flight per lane but fewer waves per
// SIMD (their registers), and occupancy wins -- two groups for the reduction, one for the dy pass (two on the narrow layers).

struct FastRow {
    u16x8 y, g, ga, gb;                                // conv output, gradient of the frame, of its reflected images (left / right halo)
    unsigned bits;                                     // dropout keep bits
    bool live, fa, fb, masked;
    int n, t;
};

__device__ __forceinline__ void fast_row_load(const w2l_bnact_t& d, const w2l_gradsrc_t& s, int64_t row, int64_t rows, int c,
                                              int cg, int G, FastRow& o) {
    o.live = row < rows;
    o.fa = o.fb = o.masked = false;
    o.bits = 0xFFu;
    o.n = 0; o.t = 0;
    if (!o.live) return;
    const int T = d.T;
    const int n = (int)(row / T), t = (int)(row - (int64_t)n * T);
    o.n = n; o.t = t;
    o.y = *reinterpret_cast<const u16x8*>(reinterpret_cast<const bf16_raw*>(d.y) + row * d.C + c);
    if (d.drop_p > 0.f) o.bits = d.mask[row * G + cg];
    o.masked = d.lens && t >= d.lens[n];
    const bf16_raw* gp = reinterpret_cast<const bf16_raw*>(s.dxp);
    const int64_t base = (int64_t)n * s.rows;
    o.g = *reinterpret_cast<const u16x8*>(gp + (base + t + s.pad_l) * d.C + c);
    if (s.pad_mode == 1) {                             // the reflected halo rows' gradient folds back onto its source frame
        o.fa = t >= 1 && t <= s.pad_l;
        o.fb = t <= T - 2 && t >= T - 1 - s.pad_r;
        if (o.fa) o.ga = *reinterpret_cast<const u16x8*>(gp + (base + s.pad_l - t) * d.C + c);
        if (o.fb) o.gb = *reinterpret_cast<const u16x8*>(gp + (base + s.pad_l + 2 * (T - 1) - t) * d.C + c);
    }
}

// gated gradient and normalised input of one row's 8 channels.  The activation and whether there is dropout are template
// parameters: these kernels are bound by their instruction count, not by HBM (a plain copy of the same bytes runs at twice
// their rate: tools/probe/hbm_ceiling.py), and a run-time switch on d.act costs two scalar branches per ELEMENT
template <int ACT, bool DROP>
__device__ __forceinline__ void fast_row_eval(const FastRow& r, const Chan& ch, float inv_keep, float g[8], float xh[8]) {
    float gv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) gv[j] = bf16_bits_to_f32(r.g[j]);
    if (r.fa) {
#pragma unroll
        for (int j = 0; j < 8; ++j) gv[j] += bf16_bits_to_f32(r.ga[j]);
    }
    if (r.fb) {
#pragma unroll
        for (int j = 0; j < 8; ++j) gv[j] += bf16_bits_to_f32(r.gb[j]);
    }
    const float gk = DROP ? inv_keep : 1.f;
    const bool row_ok = !r.masked;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float y = bf16_bits_to_f32(r.y[j]);
        float z = y * ch.sc[j] + ch.sh[j];
        bool pass = row_ok;
        if (DROP) {
            pass = pass && ((r.bits >> j) & 1u);
            z *= inv_keep;                             // (a dropped element's z is never looked at: pass is false)
        }
        if (ACT == 1) pass = pass && z >= 0.f && z <= 20.f;      // torch.clamp passes 1 on the CLOSED interval
        else if (ACT == 2) pass = pass && z > 0.f;
        g[j] = pass ? gv[j] * gk : 0.f;
        xh[j] = (y - ch.m[j]) * ch.is[j];
    }
}

template <int U, int ACT, bool DROP>
__global__ __launch_bounds__(256) void bn_bwd_reduce_fast_kernel(w2l_bnact_t d, w2l_gradsrc_t g1, float* partial, float inv_keep,
                                                                  int slots) {
    __shared__ float red[4][2][BWD_SLAB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = d.C >> 3, nslabs = d.C / BWD_SLAB;
    const int slab = blockIdx.x % nslabs, chunk = blockIdx.x / nslabs;
    const int cgl = lane & 7, rr = lane >> 3;
    const int cg = slab * (BWD_SLAB / 8) + cgl, c = cg * 8;
    const int64_t rows = (int64_t)d.N * d.T;
    const int64_t q0 = ((int64_t)chunk * 4 + wave) * (U * 8) + rr;
    FastRow r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) fast_row_load(d, g1, q0 + 8 * u, rows, c, cg, G, r[u]);
    Chan ch;
    load_chan(ch, d.scale, d.shift, d.mean, d.invstd, c);
    float s0[8], s1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s0[j] = 0.f; s1[j] = 0.f; }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (!r[u].live) continue;
        float g[8], xh[8];
        fast_row_eval<ACT, DROP>(r[u], ch, inv_keep, g, xh);
#pragma unroll
        for (int j = 0; j < 8; ++j) { s0[j] += g[j]; s1[j] += g[j] * xh[j]; }
    }
#pragma unroll
    for (int m = 8; m < 64; m <<= 1)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            s0[j] += __shfl_xor(s0[j], m, 64);
            s1[j] += __shfl_xor(s1[j], m, 64);
        }
    if (rr == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { red[wave][0][cgl * 8 + j] = s0[j]; red[wave][1][cgl * 8 + j] = s1[j]; }
    }
    __syncthreads();
    if (tid < 2 * BWD_SLAB) {
        const int k = tid >> 6, cc = tid & 63;
        const float t4 = (red[0][k][cc] + red[1][k][cc]) + (red[2][k][cc] + red[3][k][cc]);
        atomicAdd(partial + ((int64_t)(chunk % slots) * 2 + k) * d.C + slab * BWD_SLAB + cc, t4);
    }
}

template <int U, int ACT, bool DROP>
__global__ __launch_bounds__(256) void bn_bwd_apply_fast_kernel(w2l_bnact_t d, w2l_gradsrc_t g1, const float* partial, int nb,
                                                                 float* sums_out, bf16_raw* dy_hi, int h1, float inv_keep,
                                                                 float* amax) {
    __shared__ float ssum[2][BWD_SLAB];
    __shared__ float smax[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = d.C >> 3, nslabs = d.C / BWD_SLAB;
    const int slab = blockIdx.x % nslabs, chunk = blockIdx.x / nslabs;
    const int cgl = lane & 7, rr = lane >> 3;
    const int cg = slab * (BWD_SLAB / 8) + cgl, c = cg * 8;
    const int T = d.T, N = d.N;
    const int64_t rows = (int64_t)N * T;
    const int64_t q0 = ((int64_t)chunk * 4 + wave) * (U * 8) + rr;
    FastRow r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) fast_row_load(d, g1, q0 + 8 * u, rows, c, cg, G, r[u]);     // before the prologue's round trip
    if (tid < 2 * BWD_SLAB) {                          // column sums of partial[nb][2][C] over this slab's channels, fixed order
        const int k = tid >> 6, cc = tid & 63;
        float a = 0.f;
        int j = 0;
        for (; j + 8 <= nb; j += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partial[((int64_t)(j + u) * 2 + k) * d.C + slab * BWD_SLAB + cc];
#pragma unroll
            for (int u = 0; u < 8; ++u) a += v[u];
        }
        for (; j < nb; ++j) a += partial[((int64_t)j * 2 + k) * d.C + slab * BWD_SLAB + cc];
        ssum[k][cc] = a;
        if (chunk == 0) sums_out[(int64_t)k * d.C + slab * BWD_SLAB + cc] = a;
    }
    __syncthreads();
    Chan ch;
    load_chan(ch, d.scale, d.shift, d.mean, d.invstd, c);
    float sg[8], sgx[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sg[j] = ssum[0][cgl * 8 + j]; sgx[j] = ssum[1][cgl * 8 + j]; }
    const float invM = 1.f / ((float)N * (float)T);
    float mx = 0.f;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (!r[u].live) continue;
        float g[8], xh[8], o[8];
        fast_row_eval<ACT, DROP>(r[u], ch, inv_keep, g, xh);
        if (d.mean) {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = ch.sc[j] * (g[j] - sg[j] * invM - xh[j] * sgx[j] * invM);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = g[j] * ch.sc[j];
        }
        store8_split(dy_hi, nullptr, ((int64_t)h1 + (int64_t)r[u].n * (T + h1) + r[u].t) * d.C + c, o);
        if (amax) {
#pragma unroll
            for (int j = 0; j < 8; ++j) mx = fmaxf(mx, fabsf(o[j]));
        }
    }
    // ---- halo rows of the shared-halo layout: (N+1) gaps of h1 rows each, this block's share (its slab's 128 bytes per row)
    if (h1 > 0) {
        const int nchunks = gridDim.x / nslabs;
        float z[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) z[j] = 0.f;
        const int total = h1 * (N + 1);
        const int per = (total + nchunks - 1) / nchunks;
        int e = (chunk + 1) * per;
        if (e > total) e = total;
        for (int hr = chunk * per + wave * 8 + rr; hr < e; hr += 32) {
            const int gap = hr / h1, rw = hr - gap * h1;
            store8_split(dy_hi, nullptr, ((int64_t)gap * (T + h1) + rw) * d.C + c, z);
        }
    }
    if (amax) {
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) mx = fmaxf(mx, __shfl_xor(mx, m, 64));
        if (lane == 0) smax[wave] = mx;
        __syncthreads();                               // (amax is a kernel argument: uniform)
        if (tid == 0) amax_publish(amax, blockIdx.x & (AMAX_SLOTS - 1), fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3])));
    }
}

// ---- looped forms (round 5, late): the kernels above give a wave 8-32 rows and then pay, per wave, the channel constants, the
// index arithmetic, the prologue / the cross-lane reduction: 550-670 vector instructions for 8-16 elements per lane -- and 14 000-
// 28 000 such waves take as long to ISSUE on 1 024 SIMDs as the kernels run (17-28 us).  Here a wave walks `iters` row groups of
// its slab with the next group's loads in flight while it works on the current one, so the per-wave work is paid once per
// iters x 8 rows.
// row cursor of a lane: the row's index, (utterance, frame) and the three addresses it reads, advanced by 8 rows at a time with
// additions only (no division, no 64-bit multiply per row; T >= 8: at most one utterance boundary per step -- the launcher checks)
struct RowCur {
    int64_t row;
    int n, t;
    const bf16_raw* py;                                // y[row][c]
    const bf16_raw* pg;                                // gradient source row of frame t: dxp[n][pad_l + t][c]
    const uint8_t* pm;                                 // dropout keep bits of the row's channel group
};
__device__ __forceinline__ void rowcur_init(RowCur& q, const w2l_bnact_t& d, const w2l_gradsrc_t& s, int64_t row, int c, int cg, int G) {
    q.row = row;
    q.n = (int)(row / d.T);
    q.t = (int)(row - (int64_t)q.n * d.T);
    q.py = reinterpret_cast<const bf16_raw*>(d.y) + row * d.C + c;
    q.pg = reinterpret_cast<const bf16_raw*>(s.dxp) + ((int64_t)q.n * s.rows + q.t + s.pad_l) * d.C + c;
    q.pm = d.mask ? d.mask + row * G + cg : nullptr;
}
__device__ __forceinline__ void rowcur_step(RowCur& q, const w2l_bnact_t& d, const w2l_gradsrc_t& s, int G) {
    q.row += 8;
    q.t += 8;
    q.py += 8 * (int64_t)d.C;
    q.pg += 8 * (int64_t)d.C;
    if (q.pm) q.pm += 8 * G;
    if (q.t >= d.T) {
        q.t -= d.T;
        ++q.n;
        q.pg += (int64_t)(s.rows - d.T) * d.C;
    }
}
__device__ __forceinline__ void fast_row_load_cur(const w2l_bnact_t& d, const w2l_gradsrc_t& s, const RowCur& q, bool live, FastRow& o) {
    o.live = live;
    o.fa = o.fb = o.masked = false;
    o.bits = 0xFFu;
    o.n = q.n; o.t = q.t;
    if (!live) return;
    const int T = d.T, t = q.t;
    o.y = *reinterpret_cast<const u16x8*>(q.py);
    if (d.drop_p > 0.f) o.bits = *q.pm;
    o.masked = d.lens && t >= d.lens[q.n];
    o.g = *reinterpret_cast<const u16x8*>(q.pg);
    if (s.pad_mode == 1) {                             // the reflected halo rows' gradient folds back onto its source frame
        o.fa = t >= 1 && t <= s.pad_l;
        o.fb = t <= T - 2 && t >= T - 1 - s.pad_r;
        if (o.fa) o.ga = *reinterpret_cast<const u16x8*>(q.pg - 2 * (int64_t)t * d.C);                    // padded row pad_l - t
        if (o.fb) o.gb = *reinterpret_cast<const u16x8*>(q.pg + 2 * (int64_t)(T - 1 - t) * d.C);          // pad_l + 2 (T-1) - t
    }
}

template <int ACT, bool DROP>
__global__ __launch_bounds__(256) void bn_bwd_reduce_loop_kernel(w2l_bnact_t d, w2l_gradsrc_t g1, float* partial, float inv_keep,
                                                                  int slots, int iters) {
    __shared__ float red[4][2][BWD_SLAB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = d.C >> 3, nslabs = d.C / BWD_SLAB;
    const int slab = blockIdx.x % nslabs, chunk = blockIdx.x / nslabs;
    const int cgl = lane & 7, rr = lane >> 3;
    const int cg = slab * (BWD_SLAB / 8) + cgl, c = cg * 8;
    const int64_t rows = (int64_t)d.N * d.T;
    RowCur q;
    rowcur_init(q, d, g1, ((int64_t)chunk * 4 + wave) * ((int64_t)iters * 8) + rr, c, cg, G);
    FastRow cur, nxt;
    fast_row_load_cur(d, g1, q, q.row < rows, cur);
    Chan ch;
    load_chan(ch, d.scale, d.shift, d.mean, d.invstd, c);
    float s0[8], s1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s0[j] = 0.f; s1[j] = 0.f; }
    for (int it = 0; it < iters; ++it) {
        rowcur_step(q, d, g1, G);
        fast_row_load_cur(d, g1, q, it + 1 < iters && q.row < rows, nxt);
        if (cur.live) {
            float g[8], xh[8];
            fast_row_eval<ACT, DROP>(cur, ch, inv_keep, g, xh);
#pragma unroll
            for (int j = 0; j < 8; ++j) { s0[j] += g[j]; s1[j] += g[j] * xh[j]; }
        }
        cur = nxt;
    }
#pragma unroll
    for (int m = 8; m < 64; m <<= 1)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            s0[j] += __shfl_xor(s0[j], m, 64);
            s1[j] += __shfl_xor(s1[j], m, 64);
        }
    if (rr == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { red[wave][0][cgl * 8 + j] = s0[j]; red[wave][1][cgl * 8 + j] = s1[j]; }
    }
    __syncthreads();
    if (tid < 2 * BWD_SLAB) {
        const int k = tid >> 6, cc = tid & 63;
        const float t4 = (red[0][k][cc] + red[1][k][cc]) + (red[2][k][cc] + red[3][k][cc]);
        atomicAdd(partial + ((int64_t)(chunk % slots) * 2 + k) * d.C + slab * BWD_SLAB + cc, t4);
    }
}

template <int ACT, bool DROP>
__global__ __launch_bounds__(256) void bn_bwd_apply_loop_kernel(w2l_bnact_t d, w2l_gradsrc_t g1, const float* partial, int nb,
                                                                 float* sums_out, bf16_raw* dy_hi, int h1, float inv_keep,
                                                                 float* amax, int iters) {
    __shared__ float ssum[2][BWD_SLAB];
    __shared__ float smax[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = d.C >> 3, nslabs = d.C / BWD_SLAB;
    const int slab = blockIdx.x % nslabs, chunk = blockIdx.x / nslabs;
    const int cgl = lane & 7, rr = lane >> 3;
    const int cg = slab * (BWD_SLAB / 8) + cgl, c = cg * 8;
    const int T = d.T, N = d.N;
    const int64_t rows = (int64_t)N * T;
    RowCur q;
    rowcur_init(q, d, g1, ((int64_t)chunk * 4 + wave) * ((int64_t)iters * 8) + rr, c, cg, G);
    bf16_raw* pdy = dy_hi + ((int64_t)h1 + (int64_t)q.n * (T + h1) + q.t) * d.C + c;      // dy row of the cursor's frame
    FastRow cur, nxt;
    fast_row_load_cur(d, g1, q, q.row < rows, cur);                                // before the prologue's round trip
    if (tid < 2 * BWD_SLAB) {                          // column sums of partial[nb][2][C] over this slab's channels, fixed order
        const int k = tid >> 6, cc = tid & 63;
        float a = 0.f;
        int j = 0;
        for (; j + 8 <= nb; j += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partial[((int64_t)(j + u) * 2 + k) * d.C + slab * BWD_SLAB + cc];
#pragma unroll
            for (int u = 0; u < 8; ++u) a += v[u];
        }
        for (; j < nb; ++j) a += partial[((int64_t)j * 2 + k) * d.C + slab * BWD_SLAB + cc];
        ssum[k][cc] = a;
        if (chunk == 0) sums_out[(int64_t)k * d.C + slab * BWD_SLAB + cc] = a;
    }
    __syncthreads();
    Chan ch;
    load_chan(ch, d.scale, d.shift, d.mean, d.invstd, c);
    const float invM = 1.f / ((float)N * (float)T);
    float sg[8], sgx[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sg[j] = ssum[0][cgl * 8 + j] * invM; sgx[j] = ssum[1][cgl * 8 + j] * invM; }
    float mx = 0.f;
    for (int it = 0; it < iters; ++it) {
        bf16_raw* const pdy_cur = pdy;
        pdy += (int64_t)(q.t + 8 >= T ? 8 + h1 : 8) * d.C;                           // (an utterance boundary skips its halo rows)
        rowcur_step(q, d, g1, G);
        fast_row_load_cur(d, g1, q, it + 1 < iters && q.row < rows, nxt);
        if (cur.live) {
            float g[8], xh[8], o[8];
            fast_row_eval<ACT, DROP>(cur, ch, inv_keep, g, xh);
            if (d.mean) {
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = ch.sc[j] * (g[j] - sg[j] - xh[j] * sgx[j]);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = g[j] * ch.sc[j];
            }
            store8_split(pdy_cur, nullptr, 0, o);
            if (amax) {
#pragma unroll
                for (int j = 0; j < 8; ++j) mx = fmaxf(mx, fabsf(o[j]));
            }
        }
        cur = nxt;
    }
    if (h1 > 0) {                                      // halo rows of the shared-halo layout: this block's share
        const int nchunks = gridDim.x / nslabs;
        float z[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) z[j] = 0.f;
        const int total = h1 * (N + 1);
        const int per = (total + nchunks - 1) / nchunks;
        int e = (chunk + 1) * per;
        if (e > total) e = total;
        for (int hr = chunk * per + wave * 8 + rr; hr < e; hr += 32) {
            const int gap = hr / h1, rw = hr - gap * h1;
            store8_split(dy_hi, nullptr, ((int64_t)gap * (T + h1) + rw) * d.C + c, z);
        }
    }
    if (amax) {
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) mx = fmaxf(mx, __shfl_xor(mx, m, 64));
        if (lane == 0) smax[wave] = mx;
        __syncthreads();
        if (tid == 0) amax_publish(amax, blockIdx.x & (AMAX_SLOTS - 1), fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3])));
    }
}

// e4m3 quantisation with the scale taken from a device-resident amax (no host round trip): scale = the power of two that
// puts amax at <= 224 (one binade of head-room below e4m3's 448); inv_scale[0] = 1 / scale for the consumer's epilogue
__global__ __launch_bounds__(256) void quantize_e4m3_dyn_kernel(const bf16_raw* src, int64_t ngroups, const float* amax,
                                                                 uint8_t* dst, float* inv_scale) {
    float a = amax[threadIdx.x & (AMAX_SLOTS - 1)];          // the tensor's amax = max over its slots (a wave covers all 64)
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) a = fmaxf(a, __shfl_xor(a, m, 64));
    const float scale = a > 0.f ? exp2f(floorf(log2f(224.f / a))) : 1.f;
    if (blockIdx.x == 0 && threadIdx.x == 0) inv_scale[0] = 1.f / scale;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < ngroups; i += (int64_t)gridDim.x * 256) {
        float v[8];
        load8<false>(src, i * 8, v);
        *reinterpret_cast<uint2*>(dst + i * 8) = quant8_e4m3(v, scale);
    }
}

// ---------------------------------------------------------------- statistics finalize (forward)
// 32 channels x 8 tile-lanes per block: the per-tile partial sums are read as coalesced rows
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* partial, int ntiles, int C, double count,
                                                          const float* gamma, const float* beta, float eps, float momentum,
                                                          float* running_mean, float* running_var, float* mean,
                                                          float* invstd, float* scale, float* shift) {
    __shared__ double red1[8][33], red2[8][33];
    const int cx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cx;
    double s1 = 0.0, s2 = 0.0;
    if (partial && c < C) {
        // 16 loads in flight per lane: the loop is a chain of L2 round trips otherwise (16 of them at N*T/128 = 128 tiles)
        int t = ly;
        for (; t + 56 < ntiles; t += 64) {
            float a[8], b[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a[u] = partial[(int64_t)(t + 8 * u) * 2 * C + c];
                b[u] = partial[(int64_t)(t + 8 * u) * 2 * C + C + c];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { s1 += a[u]; s2 += b[u]; }
        }
        for (; t < ntiles; t += 8) {
            s1 += partial[(int64_t)t * 2 * C + c];
            s2 += partial[(int64_t)t * 2 * C + C + c];
        }
    }
    red1[ly][cx] = s1;
    red2[ly][cx] = s2;
    __syncthreads();
    if (ly != 0 || c >= C) return;
    float m, istd;
    if (partial) {
        s1 = 0.0; s2 = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) { s1 += red1[k][cx]; s2 += red2[k][cx]; }
        const double mu = s1 / count;
        double var = s2 / count - mu * mu;           // biased: normalisation uses it
        if (var < 0.0) var = 0.0;
        m = (float)mu;
        istd = (float)(1.0 / sqrt(var + (double)eps));
        if (running_mean) {
            const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * m;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
        }
    } else {                                         // eval: running statistics
        m = running_mean[c];
        istd = 1.f / sqrtf(running_var[c] + eps);
    }
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    if (mean) { mean[c] = m; invstd[c] = istd; }
    scale[c] = g * istd;
    shift[c] = b - m * g * istd;
}

int elementwise_blocks(int64_t items) {
    int64_t b = (items + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

int check_desc(const w2l_bnact_t* d, const char* who) {
    W2L_CHECK_ARG(d && d->y, "%s: null descriptor / y", who);
    W2L_CHECK_ARG(d->N > 0 && d->T > 0 && d->C > 0 && d->C % 8 == 0 && d->C <= 2048, "%s: bad N/T/C (%d,%d,%d)", who,
                  d->N, d->T, d->C);
    W2L_CHECK_ARG(d->drop_p >= 0.f && d->drop_p < 1.f, "%s: dropout p must be in [0,1)", who);
    W2L_CHECK_ARG(d->drop_p == 0.f || d->mask, "%s: dropout needs a mask buffer", who);
    W2L_CHECK_ARG((d->scale == nullptr) == (d->shift == nullptr), "%s: scale/shift must come together", who);
    return 0;
}

}  // namespace

extern "C" int w2l_bn_finalize(const float* partial, int ntiles, int C, int64_t count, const float* gamma,
                               const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                               float* mean, float* invstd, float* scale, float* shift, void* stream) {
    W2L_CHECK_ARG(scale && shift && C > 0, "bn_finalize: null output");
    W2L_CHECK_ARG(partial || (running_mean && running_var), "bn_finalize: eval mode needs running stats");
    W2L_CHECK_ARG(!partial || (ntiles > 0 && count > 0), "bn_finalize: bad tile count");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 31) / 32), dim3(256), 0, (hipStream_t)stream, partial, ntiles, C,
                       (double)count, gamma, beta, eps, momentum, running_mean, running_var, mean, invstd, scale, shift);
    W2L_CHECK_LAUNCH();
    return 0;
}

extern "C" int w2l_bn_act_fwd(const w2l_bnact_t* d, void* out_hi, void* out_lo, int out_rows, int pad_l, int pad_r,
                              int pad_mode, void* stream) {
    return w2l_bn_act_fwd_q(d, out_hi, out_lo, nullptr, 1.f, out_rows, pad_l, pad_r, pad_mode, stream);
}

extern "C" int w2l_quantize_e4m3(const void* src, int src_f32, int64_t n, float scale, void* dst, void* stream) {
    W2L_CHECK_ARG(src && dst && n > 0 && n % 8 == 0 && scale > 0.f, "quantize_e4m3: bad arguments (n must be a multiple of 8)");
    hipLaunchKernelGGL(quantize_e4m3_kernel, dim3(elementwise_blocks(n / 8)), dim3(256), 0, (hipStream_t)stream, src, src_f32,
                       n / 8, scale, (uint8_t*)dst);
    W2L_CHECK_LAUNCH();
    return 0;
}

extern "C" int w2l_bn_act_fwd_q(const w2l_bnact_t* d, void* out_hi, void* out_lo, void* out_q, float q_scale, int out_rows,
                                int pad_l, int pad_r, int pad_mode, void* stream) {
    if (int e = check_desc(d, "bn_act_fwd")) return e;
    W2L_CHECK_ARG(!out_q || q_scale > 0.f, "bn_act_fwd: the e4m3 copy needs a positive scale");
    W2L_CHECK_ARG(out_hi && out_rows >= pad_l + d->T + pad_r && pad_l >= 0 && pad_r >= 0, "bn_act_fwd: bad output geometry");
    W2L_CHECK_ARG(pad_mode != 1 || (pad_l < d->T && pad_r < d->T), "bn_act_fwd: reflect pad (%d,%d) needs pad < T=%d",
                  pad_l, pad_r, d->T);
    const uint32_t thresh = (uint32_t)(d->drop_p * 65536.f);
    const float inv_keep = 1.f / (1.f - d->drop_p);
    W2L_CHECK_ARG((int64_t)d->N * out_rows * (d->C / 8) < (1LL << 31), "bn_act_fwd: tensor too large for 32-bit indexing");
    const int64_t per_utt = (int64_t)out_rows * (d->C / 8);
    W2L_CHECK_ARG(per_utt < (1 << 24) && d->N <= 65535, "bn_act_fwd: more than 2^24 channel groups per utterance or N > 65535");
    const dim3 grid((unsigned)((per_utt + 255) / 256), (unsigned)d->N);
    const float inv_g = 1.f / (float)(d->C / 8);
#define W2L_FWD_A(F, H, A)                                                                                   \
    hipLaunchKernelGGL((bn_act_fwd_kernel<F, H, A>), grid, dim3(256), 0, (hipStream_t)stream, *d,               \
                       (bf16_raw*)out_hi, (bf16_raw*)out_lo, out_rows, pad_l, pad_r, pad_mode, thresh, inv_keep,    \
                       (uint8_t*)out_q, q_scale, inv_g)
#define W2L_FWD(F, H) do { if (d->act == 1) W2L_FWD_A(F, H, 1); else if (d->act == 2) W2L_FWD_A(F, H, 2); else W2L_FWD_A(F, H, 0); } while (0)
    if (d->y_f32) { if (d->y2) W2L_FWD(true, true); else W2L_FWD(true, false); }
    else { if (d->y2) W2L_FWD(false, true); else W2L_FWD(false, false); }
#undef W2L_FWD
#undef W2L_FWD_A
    W2L_CHECK_LAUNCH();
    return 0;
}

extern "C" int w2l_bn_act_fwd_fin(const w2l_bnact_t* d, const w2l_bnfin_t* f1, const w2l_bnfin_t* f2, void* out_hi, void* out_q,
                                  float q_scale, int out_rows, int pad_l, int pad_r, int pad_mode, void* stream) {
    if (int e = check_desc(d, "bn_act_fwd_fin")) return e;
    W2L_CHECK_ARG(!d->y_f32, "bn_act_fwd_fin: bf16 y only (the fp32 mode takes w2l_bn_finalize + w2l_bn_act_fwd)");
    W2L_CHECK_ARG(d->C % BWD_SLAB == 0, "bn_act_fwd_fin: C=%d must be a multiple of %d", d->C, BWD_SLAB);
    W2L_CHECK_ARG(f1 && (f2 != nullptr) == (d->y2 != nullptr), "bn_act_fwd_fin: one finalize record per branch");
    W2L_CHECK_ARG(!out_q || q_scale > 0.f, "bn_act_fwd_fin: the e4m3 copy needs a positive scale");
    W2L_CHECK_ARG(out_hi && out_rows >= pad_l + d->T + pad_r && pad_l >= 0 && pad_r >= 0, "bn_act_fwd_fin: bad output geometry");
    W2L_CHECK_ARG(pad_mode != 1 || (pad_l < d->T && pad_r < d->T), "bn_act_fwd_fin: reflect pad (%d,%d) needs pad < T=%d", pad_l,
                  pad_r, d->T);
    FinBranch b[2];
    const w2l_bnfin_t* fs[2] = {f1, f2};
    for (int i = 0; i < 2; ++i) {
        FinBranch& o = b[i];
        const w2l_bnfin_t* f = fs[i];
        o = FinBranch{nullptr, 0, 1.0, nullptr, nullptr, 0.f, 0.f, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        if (!f) continue;
        W2L_CHECK_ARG(!f->partial || (f->rows > 0 && f->count > 0 && f->scale && f->shift), "bn_act_fwd_fin: bad finalize record");
        o = FinBranch{f->partial, f->rows, (double)f->count, f->gamma, f->beta, f->eps, f->momentum, f->running_mean, f->running_var,
                      f->mean, f->invstd, f->scale, f->shift};
    }
    const uint32_t thresh = (uint32_t)(d->drop_p * 65536.f);
    const float inv_keep = 1.f / (1.f - d->drop_p);
    const int64_t rows = (int64_t)d->N * out_rows;
    W2L_CHECK_ARG(rows * (d->C / 8) < (1LL << 31), "bn_act_fwd_fin: tensor too large for 32-bit indexing");
    const int rpb = fwd_rows_per_block(rows, d->C);
    const int nchunks = (int)((rows + rpb - 1) / rpb);
    const dim3 grid((unsigned)(nchunks * (d->C / BWD_SLAB)));
#define W2L_FIN_A(H, A)                                                                                                      \
    hipLaunchKernelGGL((bn_act_fwd_fin_kernel<H, A>), grid, dim3(256), 0, (hipStream_t)stream, *d, b[0], b[1], (bf16_raw*)out_hi, \
                       out_rows, pad_l, pad_r, pad_mode, thresh, inv_keep, (uint8_t*)out_q, q_scale, rpb)
#define W2L_FIN(H) do { if (d->act == 1) W2L_FIN_A(H, 1); else if (d->act == 2) W2L_FIN_A(H, 2); else W2L_FIN_A(H, 0); } while (0)
    if (d->y2) W2L_FIN(true); else W2L_FIN(false);
#undef W2L_FIN
#undef W2L_FIN_A
    W2L_CHECK_LAUNCH();
    return 0;
}

extern "C" int w2l_bn_bwd_blocks(int N, int T, int C) {
    if (N <= 0 || T <= 0 || C < BWD_SLAB) return 0;
    const int64_t rows = (int64_t)N * T;
    const int rw = bwd_rows_per_wave(rows, C);
    return (int)((rows + rw - 1) / rw);
}

#define W2L_DISPATCH_BWD2(KERNEL, H2, ...)                                                                 \
    do {                                                                                                   \
        if (d->y_f32 && g1->f32) hipLaunchKernelGGL((KERNEL<true, true, H2>), __VA_ARGS__);                \
        else if (d->y_f32) hipLaunchKernelGGL((KERNEL<true, false, H2>), __VA_ARGS__);                     \
        else if (g1->f32) hipLaunchKernelGGL((KERNEL<false, true, H2>), __VA_ARGS__);                      \
        else hipLaunchKernelGGL((KERNEL<false, false, H2>), __VA_ARGS__);                                  \
    } while (0)
#define W2L_DISPATCH_BWD(KERNEL, ...)                                                                      \
    do {                                                                                                   \
        if (d->y2) W2L_DISPATCH_BWD2(KERNEL, true, __VA_ARGS__);                                           \
        else W2L_DISPATCH_BWD2(KERNEL, false, __VA_ARGS__);                                                \
    } while (0)

extern "C" int w2l_bn_act_bwd_reduce(const w2l_bnact_t* d, const w2l_gradsrc_t* g1, const w2l_gradsrc_t* g2,
                                     float* partial, void* stream) {
    if (int e = check_desc(d, "bn_act_bwd_reduce")) return e;
    W2L_CHECK_ARG(g1 && g1->dxp && partial, "bn_act_bwd_reduce: null pointer");
    W2L_CHECK_ARG(!g2 || g2->f32 == g1->f32, "bn_act_bwd_reduce: gradient sources must share a dtype");
    W2L_CHECK_ARG(g1->rows >= g1->pad_l + d->T + g1->pad_r && (!g2 || g2->rows >= g2->pad_l + d->T + g2->pad_r),
                  "bn_act_bwd_reduce: gradient source has too few rows per utterance");
    W2L_CHECK_ARG(d->C % BWD_SLAB == 0, "bn_act_bwd_reduce: C=%d must be a multiple of %d", d->C, BWD_SLAB);
    const int64_t rows = (int64_t)d->N * d->T;
    const int rw = bwd_rows_per_wave(rows, d->C);
    const int nchunks = w2l_bn_bwd_blocks(d->N, d->T, d->C);       // rows of `partial`
    const int tasks = nchunks * (d->C / BWD_SLAB);
    const float inv_keep = 1.f / (1.f - d->drop_p);
    w2l_gradsrc_t g2v = g2 ? *g2 : *g1;
    W2L_DISPATCH_BWD(bn_act_bwd_reduce_kernel, dim3((tasks + 3) / 4), dim3(256), 0, (hipStream_t)stream, *d, *g1, g2v,
                     g2 ? 1 : 0, partial, inv_keep, rw, nchunks);
    W2L_CHECK_LAUNCH();
    return 0;
}

// the fast path's preconditions (bn_bwd_reduce_fast_kernel / bn_bwd_apply_fast_kernel)
static bool bwd_fast_ok(const w2l_bnact_t* d, const w2l_gradsrc_t* g1, const w2l_gradsrc_t* g2) {
    return !d->y_f32 && !d->y2 && g1 && !g1->f32 && g2 == nullptr && d->C % BWD_SLAB == 0 && d->scale != nullptr;
}

// row groups per wave of the looped backward kernels; 0 = the one-shot kernels.  Measured (tools/bench_elem.py, N x T = 16 000 rows;
// reduce / dy pass, us): C = 256 one-shot 6.3 / 7.2, looped (4) 6.5 / 7.8; C = 512 11.5 / 15.6 -> 10.9 / 12.7; C = 896 16.6 / 23.6 ->
// 16.9 / 20.7; C = 1 024 17.2 / 24.7 -> 17.8 / 21.9; 8 groups no better, 16 worse; in the step -0.07 ms at N = 32, +0.07 on Jasper
// 10x5 N = 16 (8 000 rows) when everything is looped: looped from 8 M elements per activation and 12 000 rows on.
// W2L_BN_LOOP_ITERS=<n> forces n everywhere (0 = never): experiment switch.
static int bn_loop_iters(int64_t rows, int T, int C) {
    static const int forced = getenv("W2L_BN_LOOP_ITERS") ? atoi(getenv("W2L_BN_LOOP_ITERS")) : -1;
    int it = forced >= 0 ? forced : ((rows >= 12000 && rows * C >= 8000000) ? 4 : 0);
    if (T < 8) it = 0;                                  // (the row cursor steps 8 rows with at most one utterance boundary)
    while (it > 1 && rows < 4LL * it * 8 * 4) it >>= 1; // short activations: keep at least a few chunks
    return it;
}

extern "C" int w2l_bn_bwd_fast_ok(const w2l_bnact_t* d, const w2l_gradsrc_t* g1, const w2l_gradsrc_t* g2) {
    return d && bwd_fast_ok(d, g1, g2) ? 1 : 0;
}

extern "C" int w2l_bn_act_bwd_reduce_slots(const w2l_bnact_t* d, const w2l_gradsrc_t* g1, float* partial, int slots, void* stream) {
    if (int e = check_desc(d, "bn_act_bwd_reduce_slots")) return e;
    W2L_CHECK_ARG(g1 && g1->dxp && partial && slots >= 1 && slots <= 64, "bn_act_bwd_reduce_slots: null pointer / slots not in 1..64");
    W2L_CHECK_ARG(bwd_fast_ok(d, g1, nullptr), "bn_act_bwd_reduce_slots: bf16 y and gradient, one branch, one source, C %% 64 == 0 only");
    W2L_CHECK_ARG(g1->rows >= g1->pad_l + d->T + g1->pad_r, "bn_act_bwd_reduce_slots: gradient source has too few rows per utterance");
    const int64_t rows = (int64_t)d->N * d->T;
    const float inv_keep = 1.f / (1.f - d->drop_p);
    const bool drop = d->drop_p > 0.f;
    const int iters = bn_loop_iters(rows, d->T, d->C);
    if (iters > 0) {
        const int nch = (int)((rows + 4LL * iters * 8 - 1) / (4LL * iters * 8));
        const dim3 lgrid((unsigned)(nch * (d->C / BWD_SLAB)));
#define W2L_REDL(A, D) hipLaunchKernelGGL((bn_bwd_reduce_loop_kernel<A, D>), lgrid, dim3(256), 0, (hipStream_t)stream, *d, *g1, partial, inv_keep, slots, iters)
        switch (d->act) {
            case 1: if (drop) W2L_REDL(1, true); else W2L_REDL(1, false); break;
            case 2: if (drop) W2L_REDL(2, true); else W2L_REDL(2, false); break;
            default: if (drop) W2L_REDL(0, true); else W2L_REDL(0, false); break;
        }
#undef W2L_REDL
        W2L_CHECK_LAUNCH();
        return 0;
    }
    constexpr int U = 2;
    const int nchunks = (int)((rows + 4 * U * 8 - 1) / (4 * U * 8));
    const dim3 grid((unsigned)(nchunks * (d->C / BWD_SLAB)));
#define W2L_RED(A, D) hipLaunchKernelGGL((bn_bwd_reduce_fast_kernel<U, A, D>), grid, dim3(256), 0, (hipStream_t)stream, *d, *g1, partial, inv_keep, slots)
    switch (d->act) {
        case 1: if (drop) W2L_RED(1, true); else W2L_RED(1, false); break;
        case 2: if (drop) W2L_RED(2, true); else W2L_RED(2, false); break;
        default: if (drop) W2L_RED(0, true); else W2L_RED(0, false); break;
    }
#undef W2L_RED
    W2L_CHECK_LAUNCH();
    return 0;
}

extern "C" int w2l_bn_act_bwd_apply_slots(const w2l_bnact_t* d, const w2l_gradsrc_t* g1, const float* partial, int nrows, float* sums,
                                          void* dy_hi, int halo, float* amax, void* stream) {
    if (int e = check_desc(d, "bn_act_bwd_apply_slots")) return e;
    W2L_CHECK_ARG(g1 && g1->dxp && dy_hi && partial && sums && nrows > 0 && halo >= 0, "bn_act_bwd_apply_slots: null pointer / no partial rows");
    W2L_CHECK_ARG(bwd_fast_ok(d, g1, nullptr), "bn_act_bwd_apply_slots: bf16 y and gradient, one branch, one source, C %% 64 == 0 only");
    W2L_CHECK_ARG(g1->rows >= g1->pad_l + d->T + g1->pad_r, "bn_act_bwd_apply_slots: gradient source has too few rows per utterance");
    const int64_t rows = (int64_t)d->N * d->T;
    W2L_CHECK_ARG((rows + (int64_t)halo * (d->N + 1)) * (d->C / 8) < (1LL << 31), "bn_act_bwd_apply_slots: tensor too large for 32-bit indexing");
    // (fp8 mode, amax: four groups -- a block ends in an atomic on one of W2L_AMAX_SLOTS words, and same-address atomics take
    // ~0.1 us each one after the other: the fewer blocks the better; measured 9.0 against 10.1 ms per fp8 step)
    const float inv_keep = 1.f / (1.f - d->drop_p);
    const int iters = bn_loop_iters(rows, d->T, d->C);
    if (iters > 0) {
        const int nch = (int)((rows + 4LL * iters * 8 - 1) / (4LL * iters * 8));
        const dim3 lgrid((unsigned)(nch * (d->C / BWD_SLAB)));
        const bool dropl = d->drop_p > 0.f;
#define W2L_APPL(A, D) hipLaunchKernelGGL((bn_bwd_apply_loop_kernel<A, D>), lgrid, dim3(256), 0, (hipStream_t)stream, *d, *g1, partial, nrows, sums, (bf16_raw*)dy_hi, halo, inv_keep, amax, iters)
        switch (d->act) {
            case 1: if (dropl) W2L_APPL(1, true); else W2L_APPL(1, false); break;
            case 2: if (dropl) W2L_APPL(2, true); else W2L_APPL(2, false); break;
            default: if (dropl) W2L_APPL(0, true); else W2L_APPL(0, false); break;
        }
#undef W2L_APPL
        W2L_CHECK_LAUNCH();
        return 0;
    }
    const int U = amax ? 4 : (d->C <= 384 ? 2 : 1);
    const int nchunks = (int)((rows + 4 * U * 8 - 1) / (4 * U * 8));
    const dim3 grid((unsigned)(nchunks * (d->C / BWD_SLAB)));
#define W2L_APP(UU, A, D) hipLaunchKernelGGL((bn_bwd_apply_fast_kernel<UU, A, D>), grid, dim3(256), 0, (hipStream_t)stream, *d, *g1, partial, nrows, sums, (bf16_raw*)dy_hi, halo, inv_keep, amax)
#define W2L_APP_U(A, D) do { if (U == 4) W2L_APP(4, A, D); else if (U == 2) W2L_APP(2, A, D); else W2L_APP(1, A, D); } while (0)
    const bool drop = d->drop_p > 0.f;
    switch (d->act) {
        case 1: if (drop) W2L_APP_U(1, true); else W2L_APP_U(1, false); break;
        case 2: if (drop) W2L_APP_U(2, true); else W2L_APP_U(2, false); break;
        default: if (drop) W2L_APP_U(0, true); else W2L_APP_U(0, false); break;
    }
#undef W2L_APP_U
#undef W2L_APP
    W2L_CHECK_LAUNCH();
    return 0;
}

extern "C" int w2l_bn_bwd_finalize(const float* partial, int nblocks, int C, int ncomp, float* sums, void* stream) {
    W2L_CHECK_ARG(partial && sums && nblocks > 0 && C > 0 && (ncomp == 2 || ncomp == 4), "bn_bwd_finalize: bad arguments");
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((ncomp * C + 31) / 32), dim3(256), 0, (hipStream_t)stream, partial,
                       nblocks, ncomp * C, sums);
    W2L_CHECK_LAUNCH();
    return 0;
}

extern "C" int w2l_bn_act_bwd_apply(const w2l_bnact_t* d, const w2l_gradsrc_t* g1, const w2l_gradsrc_t* g2,
                                    const float* sums, void* dy_hi, void* dy_lo, int halo, void* dy2_hi, void* dy2_lo,
                                    int halo2, void* stream) {
    return w2l_bn_act_bwd_apply_amax(d, g1, g2, sums, dy_hi, dy_lo, halo, dy2_hi, dy2_lo, halo2, nullptr, stream);
}

extern "C" int w2l_bn_act_bwd_apply_fin(const w2l_bnact_t* d, const w2l_gradsrc_t* g1, const w2l_gradsrc_t* g2,
                                        const float* partial, int nblocks, float* sums, void* dy_hi, void* dy_lo, int halo,
                                        void* dy2_hi, void* dy2_lo, int halo2, float* amax, void* stream) {
    if (int e = check_desc(d, "bn_act_bwd_apply_fin")) return e;
    W2L_CHECK_ARG(g1 && g1->dxp && dy_hi && partial && sums && nblocks > 0, "bn_act_bwd_apply_fin: null pointer / no partial rows");
    W2L_CHECK_ARG(d->C % BWD_SLAB == 0, "bn_act_bwd_apply_fin: C=%d must be a multiple of %d", d->C, BWD_SLAB);
    W2L_CHECK_ARG(!g2 || g2->f32 == g1->f32, "bn_act_bwd_apply_fin: gradient sources must share a dtype");
    W2L_CHECK_ARG(halo >= 0 && halo2 >= 0, "bn_act_bwd_apply_fin: negative halo");
    W2L_CHECK_ARG(g1->rows >= g1->pad_l + d->T + g1->pad_r && (!g2 || g2->rows >= g2->pad_l + d->T + g2->pad_r),
                  "bn_act_bwd_apply_fin: gradient source has too few rows per utterance");
    const int64_t rows = (int64_t)d->N * d->T;
    W2L_CHECK_ARG((rows + (int64_t)(halo + halo2) * (d->N + 1)) * (d->C / 8) < (1LL << 31),
                  "bn_act_bwd_apply_fin: tensor too large for 32-bit indexing");
    const int rpb = apply_rows_per_block(rows, d->C);
    const int nchunks = (int)((rows + rpb - 1) / rpb);
    const int blocks = nchunks * (d->C / BWD_SLAB);
    const float inv_keep = 1.f / (1.f - d->drop_p);
    w2l_gradsrc_t g2v = g2 ? *g2 : *g1;
    W2L_DISPATCH_BWD(bn_act_bwd_apply_fin_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, *d, *g1, g2v, g2 ? 1 : 0,
                     partial, nblocks, sums, (bf16_raw*)dy_hi, (bf16_raw*)dy_lo, halo, (bf16_raw*)dy2_hi, (bf16_raw*)dy2_lo,
                     halo2, inv_keep, amax, rpb);
    W2L_CHECK_LAUNCH();
    return 0;
}

extern "C" int w2l_quantize_e4m3_dyn(const void* src_bf16, int64_t n, const float* amax, void* dst, float* inv_scale,
                                     void* stream) {
    W2L_CHECK_ARG(src_bf16 && dst && amax && inv_scale && n > 0 && n % 8 == 0, "quantize_e4m3_dyn: bad arguments");
    hipLaunchKernelGGL(quantize_e4m3_dyn_kernel, dim3(elementwise_blocks(n / 8)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_raw*)src_bf16, n / 8, amax, (uint8_t*)dst, inv_scale);
    W2L_CHECK_LAUNCH();
    return 0;
}

extern "C" int w2l_bn_act_bwd_apply_amax(const w2l_bnact_t* d, const w2l_gradsrc_t* g1, const w2l_gradsrc_t* g2,
                                         const float* sums, void* dy_hi, void* dy_lo, int halo, void* dy2_hi, void* dy2_lo,
                                         int halo2, float* amax, void* stream) {
    if (int e = check_desc(d, "bn_act_bwd_apply")) return e;
    W2L_CHECK_ARG(g1 && g1->dxp && dy_hi, "bn_act_bwd_apply: null pointer");
    W2L_CHECK_ARG(!d->mean || sums, "bn_act_bwd_apply: BatchNorm backward needs the reduced sums");
    W2L_CHECK_ARG(!g2 || g2->f32 == g1->f32, "bn_act_bwd_apply: gradient sources must share a dtype");
    W2L_CHECK_ARG(halo >= 0 && halo2 >= 0, "bn_act_bwd_apply: negative halo");
    W2L_CHECK_ARG(g1->rows >= g1->pad_l + d->T + g1->pad_r && (!g2 || g2->rows >= g2->pad_l + d->T + g2->pad_r),
                  "bn_act_bwd_apply: gradient source has too few rows per utterance");
    const int G = d->C / 8;
    const int64_t items = (int64_t)d->N * d->T * G + (int64_t)halo * (d->N + 1) * G +
                          (dy2_hi ? (int64_t)halo2 * (d->N + 1) * G : 0);
    W2L_CHECK_ARG(items < (1LL << 31), "bn_act_bwd_apply: tensor too large for 32-bit indexing");
    const int blocks = elementwise_blocks(items);
    const float inv_keep = 1.f / (1.f - d->drop_p);
    w2l_gradsrc_t g2v = g2 ? *g2 : *g1;
    W2L_DISPATCH_BWD(bn_act_bwd_apply_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, *d, *g1, g2v, g2 ? 1 : 0,
                     sums, (bf16_raw*)dy_hi, (bf16_raw*)dy_lo, halo, (bf16_raw*)dy2_hi, (bf16_raw*)dy2_lo, halo2,
                     inv_keep, amax);
    W2L_CHECK_LAUNCH();
    return 0;
}
