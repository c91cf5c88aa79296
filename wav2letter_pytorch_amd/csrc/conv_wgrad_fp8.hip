// Conv1d weight gradient on OCP e4m3 operands (gfx950): the fp8 mode's counterpart of conv_wgrad.hip.
//
//   dw[kw][co][ci] (+)= descale * sum_{n,t} dyq[n][t][co] * xq[n][t + kw*d][ci]          (stride 1)
//
// dyq / xq are the e4m3 copies the fp8 step already makes for its data gradients and forward convolutions (one byte per
// element, channels-last), so the reduction index t is again the slow axis of both operands.  The MFMA is
// v_mfma_scale_f32_16x16x128_f8f6f4 (unit scales; twice the bf16 rate): one instruction reduces over 128 frames, so a K
// step is a [128 t][128 channel] byte tile -- the same 16 KiB as the bf16 kernel's [64 t][128 channel] tile.  A lane
// holds 32 reduction bytes per operand; they come from four ds_read_b64_tr_b8, CDNA4's transposing LDS read for bytes.
// Measured semantics (tools/probe/tr8_probe.hip; the ISA manual is not in this image): within 16 lanes, lane i supplies
// the address of 8 bytes at (row i/2, byte offset 8*(i%2)) and receives column i of that [8 rows][16 bytes] block, rows
// in byte order.  Lane group g of an operand takes rows 32h + 8g + 0..7 (h = 0..3: one read each); A and B use the same
// assignment, so their bytes pair up on the same frame whatever order the instruction walks its k range in.
// Block = [128 co x 128 ci] x KWB taps x a slice of (n, t) (split-K, fp32 atomics), as in the bf16 kernel.
//
// Replaces the weight-gradient half of aten::convolution_backward (wav2letter.py:35-36,42 / jasper.py:96-105,127) in
// the fp8 mode (BASELINE config 5).
#include "common.h"
#include <map>
#include <mutex>
#include <tuple>
#include <type_traits>

namespace {

typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v8i __attribute__((ext_vector_type(8)));

constexpr int BM = 128;        // co per block
constexpr int BNC = 128;       // ci per block
constexpr int BT = 128;        // frames per K step (= the MFMA's reduction length)
constexpr int ROWB = 128;      // bytes per LDS row (128 channels, one byte each)

struct WgradF8Params {
    const unsigned char* dy;
    const unsigned char* x;
    float* dw;
    const float* descale_dev;  // optional device factor (1 / scale of dyq, written by w2l_quantize_e4m3_dyn)
    float descale;             // host factor (1 / scale of xq)
    int64_t dy_rows_per_utt, x_rows_per_utt, x_max_row;
    int N, Cin, Cout, Tout, Kw, dil;
    int tiles_m, tiles_n, kgroups, tsteps, total_steps, steps_per_split, atomic, order, accumulate, xrows_lds;
};

__device__ __forceinline__ void glds16(const void* gbase_uniform, unsigned voff, unsigned lds_wave_base) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :
                 : "v"(voff), "s"(gbase_uniform), "s"(lds_wave_base)
                 : "memory");
}

__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p;
}

// swizzle key of an LDS row: one half-wave's transposing read touches 16 consecutive rows (two lane groups x 8 rows), 16
// bytes each; rows of equal parity share the 128-byte half of a bank line, so their eight 16-byte chunks must differ:
// chunk' = chunk ^ ((row >> 1) & 7).  Bits 1..3 only: adding 32h rows (the k-substep immediates) leaves the key alone.
__device__ __forceinline__ int row_key(int r) { return (r >> 1) & 7; }

__device__ __forceinline__ v2i tr8_read(unsigned lds_byte_addr) {
    return __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) v2i*)(size_t)lds_byte_addr);
}

__device__ __forceinline__ f32x4 mfma_e4m3(const v8i& a, const v8i& b, f32x4 c) {
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
}

template <int KWB>
__global__ __launch_bounds__(256, 2) void conv_wgrad_fp8_kernel(WgradF8Params p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // ---- block -> (tile, split); tile -> (tap group, co tile, ci tile): as in conv_wgrad.hip
    const int lin = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int split = lin / gridDim.x;
    int tile = lin - split * gridDim.x;
    int kw0, tm, tn;
    if (p.order) {
        kw0 = (tile % p.kgroups) * KWB;
        tile /= p.kgroups;
        tm = tile % p.tiles_m;
        tn = tile / p.tiles_m;
    } else {
        tm = tile % p.tiles_m;
        tile /= p.tiles_m;
        tn = tile % p.tiles_n;
        kw0 = (tile / p.tiles_n) * KWB;
    }
    const int ntaps = (p.Kw - kw0) < KWB ? (p.Kw - kw0) : KWB;
    const int m0 = tm * BM, c0 = tn * BNC;
    const int d = p.dil;
    const int shift = kw0 * d;
    const int xrows = p.xrows_lds;                 // BT + (KWB-1)*d rounded up to 8
    const int step_begin = split * p.steps_per_split;
    int step_end = step_begin + p.steps_per_split;
    if (step_end > p.total_steps) step_end = p.total_steps;

    char* abuf0 = smem;                            // dyq tile  [BT][128 co]
    char* abuf1 = smem + BT * ROWB;
    char* bbuf0 = smem + 2 * BT * ROWB;            // xq window [xrows][128 ci]
    char* bbuf1 = bbuf0 + xrows * ROWB;

    // ---- staging: one wave-instruction fills 8 rows of 128 B by LDS-DMA; per-lane offsets are computed once
    const int srow = lane >> 3;                    // 0..7
    const int schunk = lane & 7;                   // physical 16-byte chunk of the row
    constexpr int AG = BT / 8 / 4;                 // eight-row groups of the dy tile per wave
    constexpr int XG = 5;                          // ... of the x window (up to 160 rows)
    unsigned a_voff[AG], x_voff[XG];
    const unsigned x_max_row = (unsigned)p.x_max_row;
#pragma unroll
    for (int i = 0; i < AG; ++i) {
        const int r = (wave * AG + i) * 8 + srow;
        int co = m0 + ((schunk ^ row_key(r)) << 4);
        co = co < p.Cout ? co : p.Cout - 16;
        a_voff[i] = (unsigned)r * (unsigned)p.Cout + (unsigned)co;
    }
#pragma unroll
    for (int i = 0; i < XG; ++i) {
        const int r = (wave + 4 * i) * 8 + srow;
        int ci = c0 + ((schunk ^ row_key(r)) << 4);
        ci = ci < p.Cin ? ci : p.Cin - 16;
        x_voff[i] = (unsigned)r * (unsigned)p.Cin + (unsigned)ci;
    }
    auto stage = [&](char* adst, char* bdst, int n, int ts) {
        const int t0 = ts * BT;
        const char* abase = reinterpret_cast<const char*>(p.dy) + ((int64_t)n * p.dy_rows_per_utt + t0) * p.Cout;
        const unsigned a_lds = __builtin_amdgcn_readfirstlane(lds_addr(adst) + wave * AG * 1024);
#pragma unroll
        for (int i = 0; i < AG; ++i) glds16(abase, a_voff[i], a_lds + i * 1024);
        const unsigned b_lds = __builtin_amdgcn_readfirstlane(lds_addr(bdst));
        const unsigned xrow0 = (unsigned)(n * p.x_rows_per_utt) + (unsigned)(t0 + shift);
        const int ngrp = xrows >> 3;
        if (xrow0 + (unsigned)xrows - 1u <= x_max_row) {           // (wave-uniform) the whole window exists
            const char* xbase = reinterpret_cast<const char*>(p.x) + (uint64_t)xrow0 * (unsigned)p.Cin;
#pragma unroll
            for (int i = 0; i < XG; ++i)
                if (wave + 4 * i < ngrp) glds16(xbase, x_voff[i], b_lds + (wave + 4 * i) * 1024);
            return;
        }
        for (int grp = wave; grp < ngrp; grp += 4) {
            const int r = grp * 8 + srow;
            unsigned fr = xrow0 + (unsigned)r;
            fr = fr < x_max_row ? fr : x_max_row;
            int ci = c0 + ((schunk ^ row_key(r)) << 4);
            ci = ci < p.Cin ? ci : p.Cin - 16;
            glds16(p.x, fr * (unsigned)p.Cin + (unsigned)ci, b_lds + grp * 1024);
        }
    };

    f32x4 acc[KWB][4][4];
#pragma unroll
    for (int tp = 0; tp < KWB; ++tp)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[tp][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- transposing-read geometry: lane (g = lane/16, i = lane%16) addresses row 8g + i/2 (+ 32h by immediate, + tap*d
    // for the x window), byte 8*(i%2) of the fragment's 16-byte chunk, and receives channel i of that chunk
    const int g = lane >> 4, i16 = lane & 15;
    const int lrow = g * 8 + (i16 >> 1);
    unsigned pa[4], pb[KWB][4];
    const unsigned abase0 = lds_addr(abuf0), bbase0 = lds_addr(bbuf0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        pa[i] = abase0 + lrow * ROWB + (((wm * 4 + i) ^ row_key(lrow)) << 4) + ((i16 & 1) << 3);
#pragma unroll
        for (int tp = 0; tp < KWB; ++tp) {
            const int rb = lrow + tp * d;
            pb[tp][i] = bbase0 + rb * ROWB + (((wn * 4 + i) ^ row_key(rb)) << 4) + ((i16 & 1) << 3);
        }
    }
    int a_toggle = BT * ROWB, b_toggle = xrows * ROWB;
    auto toggle = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            pa[i] += a_toggle;
#pragma unroll
            for (int tp = 0; tp < KWB; ++tp) pb[tp][i] += b_toggle;
        }
        a_toggle = -a_toggle;
        b_toggle = -b_toggle;
    };
    auto load_frag = [&](v8i& dst, unsigned ptr) {
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const v2i v = tr8_read(ptr + h * 32 * ROWB);
            dst[2 * h] = v[0];
            dst[2 * h + 1] = v[1];
        }
    };

    // ---- K loop.  A step = NT x 4 fragment groups (tap-major, ci-subtile-minor) of 4 MFMAs; the dy fragments a[0..3] serve
    // all of them, the x fragment of group q+1 is read while group q computes.  The block barrier sits before the LAST
    // group of a step (all of this step's x reads have been issued and returned by then): after it the LDS-DMA of step+2
    // goes into the buffers just read, the first x fragment of step+1 is requested from the other buffers, and each dy
    // fragment is refilled in place for step+1 right after its last MFMA (a second set of dy registers does not fit:
    // 275 spills).
    v8i a[4], b[2];
    auto advance = [&](int& n, int& ts) {
        if (++ts == p.tsteps) { ts = 0; ++n; }
    };
    auto step_body = [&](auto nt_tag, auto last_tag, char* adst, char* bdst, int n_nn, int ts_nn, bool have_nn) {
        constexpr int NT = decltype(nt_tag)::value;
        constexpr bool LAST = decltype(last_tag)::value;
        constexpr int NG = NT * 4;
#pragma unroll
        for (int q = 0; q < NG; ++q) {
            const int tp = q / 4, ni = q % 4;
            const bool lastq = q + 1 == NG;
            if (!lastq) load_frag(b[(q + 1) & 1], pb[(q + 1) / 4][(q + 1) % 4]);
            if (lastq && !LAST) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                toggle();
                if (have_nn) stage(adst, bdst, n_nn, ts_nn);
                load_frag(b[(q + 1) & 1], pb[0][0]);
            }
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                acc[tp][mi][ni] = mfma_e4m3(a[mi], b[q & 1], acc[tp][mi][ni]);
                if (lastq && !LAST) load_frag(a[mi], pa[mi]);
            }
        }
    };
    auto run = [&](auto nt_tag) {
        if (step_begin >= step_end) return;
        int n = step_begin / p.tsteps;
        int ts = step_begin - n * p.tsteps;
        stage(abuf0, bbuf0, n, ts);
        advance(n, ts);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (step_begin + 1 < step_end) stage(abuf1, bbuf1, n, ts);
#pragma unroll
        for (int i = 0; i < 4; ++i) load_frag(a[i], pa[i]);
        load_frag(b[0], pb[0][0]);
        for (int step = step_begin; step + 1 < step_end; ++step) {
            const int par = (step - step_begin) & 1;
            advance(n, ts);                               // now step+2
            step_body(nt_tag, std::false_type{}, par ? abuf1 : abuf0, par ? bbuf1 : bbuf0, n, ts, step + 2 < step_end);
        }
        step_body(nt_tag, std::true_type{}, nullptr, nullptr, 0, 0, false);
    };
    if (KWB == 1 || ntaps == KWB) run(std::integral_constant<int, KWB>{});
    else run(std::integral_constant<int, 1>{});

    // ---- epilogue: acc[tp][mi][ni][r] = dw[kw0+tp][co = m0+wm*64+mi*16+fq*4+r][ci = c0+wn*64+ni*16+fr] / scales
    float scale = p.descale;
    if (p.descale_dev != nullptr) scale *= *p.descale_dev;
    const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int tp = 0; tp < KWB; ++tp) {
        if (tp >= ntaps) break;
        float* base = p.dw + (int64_t)(kw0 + tp) * p.Cout * p.Cin;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int ci = c0 + wn * 64 + ni * 16 + fr;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = m0 + wm * 64 + mi * 16 + fq * 4 + r;
                    if (co < p.Cout && ci < p.Cin) {
                        float* dst = base + (int64_t)co * p.Cin + ci;
                        const float v = acc[tp][mi][ni][r] * scale;
                        if (p.atomic) atomicAdd(dst, v);
                        else if (p.accumulate) *dst += v;
                        else *dst = v;
                    }
                }
            }
    }
}

typedef std::tuple<int, int, int, int, int> F8ShapeKey;
std::map<F8ShapeKey, int> g_f8_tuned;          // shape -> split count | block order << 16
std::mutex g_f8_mu;
thread_local int g_f8_force_splits = 0;        // set by the tuner around its own launches only
thread_local int g_f8_force_order = -1;

int f8_plan(int N, int Cin, int Cout, int Tout, int Kw, int* order_out) {
    const int tsteps = (Tout + BT - 1) / BT;
    const int total = N * tsteps;
    *order_out = g_f8_force_order >= 0 ? g_f8_force_order : 1;
    if (g_f8_force_splits > 0) return g_f8_force_splits <= total ? g_f8_force_splits : total;
    {
        std::lock_guard<std::mutex> lock(g_f8_mu);
        auto it = g_f8_tuned.find(F8ShapeKey(N, Cin, Cout, Tout, Kw));
        if (it != g_f8_tuned.end()) {
            if (g_f8_force_order < 0) *order_out = it->second >> 16;
            return it->second & 0xffff;
        }
    }
    // untuned: fill whole rounds of the 512 resident blocks, keep >= 4 steps per block
    const int kwb = Kw > 1 ? 2 : 1;
    const long tiles = (long)((Cout + BM - 1) / BM) * ((Cin + BNC - 1) / BNC) * ((Kw + kwb - 1) / kwb);
    int best = 1;
    double best_cost = 1e30;
    for (int s = 1; s <= 16 && s <= total; ++s) {
        const int steps = (total + s - 1) / s;
        if (s > 1 && steps < 4) break;
        const long rounds = (tiles * s + 511) / 512;
        const double cost = (double)rounds * steps + (s > 1 ? 0.05 * steps * s : 0.0);
        if (cost < best_cost * 0.97) { best_cost = cost; best = s; }
    }
    return best;
}

}  // namespace

extern "C" int w2l_wgrad_fp8_needs_zero(int N, int Cin, int Cout, int Tout, int Kw) {
    int order = 0;
    return f8_plan(N, Cin, Cout, Tout, Kw, &order) > 1;
}

extern "C" int w2l_conv1d_wgrad_fp8(const void* dyq, int64_t dy_bstride, const void* xq, int64_t x_bstride,
                                    int64_t x_rows_total, float* dw, int N, int Cin, int Cout, int Tout, int Kw, int dil,
                                    float descale, const float* descale_dev, int accumulate, void* stream) {
    W2L_CHECK_ARG(dyq && xq && dw, "conv1d_wgrad_fp8: null pointer");
    W2L_CHECK_ARG(N > 0 && Tout > 0 && Kw > 0 && dil > 0, "conv1d_wgrad_fp8: bad sizes");
    W2L_CHECK_ARG(Cin % 64 == 0 && Cout % 64 == 0 && Cin > 0 && Cout > 0,
                  "conv1d_wgrad_fp8: channels (%d,%d) must be positive multiples of 64", Cin, Cout);
    W2L_CHECK_ARG(dy_bstride % Cout == 0 && x_bstride % Cin == 0, "conv1d_wgrad_fp8: batch strides must be whole rows");
    W2L_CHECK_ARG(x_rows_total * (int64_t)Cin < (1LL << 32), "conv1d_wgrad_fp8: activation buffer exceeds 32-bit byte offsets");
    const int kwb = Kw > 1 ? 2 : 1;
    W2L_CHECK_ARG((kwb - 1) * dil <= 32, "conv1d_wgrad_fp8: dilation %d exceeds the staged window", dil);
    WgradF8Params p;
    p.dy = (const unsigned char*)dyq;
    p.x = (const unsigned char*)xq;
    p.dw = dw;
    p.descale = descale;
    p.descale_dev = descale_dev;
    p.dy_rows_per_utt = dy_bstride / Cout;
    p.x_rows_per_utt = x_bstride / Cin;
    p.x_max_row = x_rows_total - 1;
    p.N = N; p.Cin = Cin; p.Cout = Cout; p.Tout = Tout; p.Kw = Kw; p.dil = dil;
    p.tiles_m = (Cout + BM - 1) / BM;
    p.tiles_n = (Cin + BNC - 1) / BNC;
    p.tsteps = (Tout + BT - 1) / BT;
    p.total_steps = N * p.tsteps;
    int order = 1;
    const int splits = f8_plan(N, Cin, Cout, Tout, Kw, &order);
    p.order = order & 1;
    p.steps_per_split = (p.total_steps + splits - 1) / splits;
    p.atomic = splits > 1;
    p.accumulate = accumulate;
    p.kgroups = (Kw + kwb - 1) / kwb;
    p.xrows_lds = (BT + (kwb - 1) * dil + 7) & ~7;
    const size_t lds = 2 * BT * ROWB + 2 * (size_t)p.xrows_lds * ROWB;
    dim3 grid(p.tiles_m * p.tiles_n * p.kgroups, splits), block(256);
    if (kwb == 2) {
        W2L_CHECK_HIP(w2l_allow_big_lds((const void*)conv_wgrad_fp8_kernel<2>));
        hipLaunchKernelGGL((conv_wgrad_fp8_kernel<2>), grid, block, lds, (hipStream_t)stream, p);
    } else {
        W2L_CHECK_HIP(w2l_allow_big_lds((const void*)conv_wgrad_fp8_kernel<1>));
        hipLaunchKernelGGL((conv_wgrad_fp8_kernel<1>), grid, block, lds, (hipStream_t)stream, p);
    }
    W2L_CHECK_LAUNCH();
    return 0;
}

// Measure split counts x block orders for this problem and remember the fastest (SYNCHRONISING; warm-up only).
extern "C" int w2l_conv1d_wgrad_fp8_tune(const void* dyq, int64_t dy_bstride, const void* xq, int64_t x_bstride,
                                         int64_t x_rows_total, float* dw_scratch, int N, int Cin, int Cout, int Tout, int Kw,
                                         int dil, int reps, void* stream) {
    const F8ShapeKey key(N, Cin, Cout, Tout, Kw);
    {
        std::lock_guard<std::mutex> lock(g_f8_mu);
        if (g_f8_tuned.count(key)) return 0;
    }
    hipEvent_t e0, e1;
    W2L_CHECK_HIP(hipEventCreate(&e0));
    W2L_CHECK_HIP(hipEventCreate(&e1));
    hipStream_t st = (hipStream_t)stream;
    const int total = N * ((Tout + BT - 1) / BT);
    const int cands[] = {1, 2, 3, 4, 5, 6, 8, 10, 12, 16};
    int best = -1;
    float best_ms = 1e30f;
    if (reps < 1) reps = 1;
    const size_t bytes = (size_t)Kw * Cout * Cin * sizeof(float);
    for (int ci = 0; ci < 2 * (int)(sizeof(cands) / sizeof(cands[0])); ++ci) {
        const int s = cands[ci >> 1], order = ci & 1;
        if (s > total || (s > 1 && total / s < 4)) break;
        g_f8_force_splits = s;
        g_f8_force_order = order;
        int rc = w2l_conv1d_wgrad_fp8(dyq, dy_bstride, xq, x_bstride, x_rows_total, dw_scratch, N, Cin, Cout, Tout, Kw, dil,
                                      1.0f, nullptr, 0, stream);
        if (rc != 0) continue;
        (void)hipEventRecord(e0, st);
        for (int r = 0; r < reps; ++r) {
            if (s > 1) (void)hipMemsetAsync(dw_scratch, 0, bytes, st);          // the fill is part of a split launch's cost
            w2l_conv1d_wgrad_fp8(dyq, dy_bstride, xq, x_bstride, x_rows_total, dw_scratch, N, Cin, Cout, Tout, Kw, dil, 1.0f,
                                 nullptr, 0, stream);
        }
        (void)hipEventRecord(e1, st);
        if (hipEventSynchronize(e1) != hipSuccess) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess) continue;
        if (ms < best_ms) { best_ms = ms; best = s | (order << 16); }
    }
    g_f8_force_splits = 0;
    g_f8_force_order = -1;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    W2L_CHECK_ARG(best >= 1, "conv1d_wgrad_fp8_tune: no candidate ran");
    std::lock_guard<std::mutex> lock(g_f8_mu);
    g_f8_tuned[key] = best;
    return 0;
}

// Tuning-cache (de)serialisation used by w2l_tune_save / w2l_tune_load (runtime.hip).
void w2l_wgrad_fp8_tune_dump(FILE* f) {
    std::lock_guard<std::mutex> lock(g_f8_mu);
    for (const auto& kv : g_f8_tuned) {
        const F8ShapeKey& k = kv.first;
        fprintf(f, "wgradf8 %d %d %d %d %d %d %d\n", std::get<0>(k), std::get<1>(k), std::get<2>(k), std::get<3>(k),
                std::get<4>(k), kv.second & 0xffff, kv.second >> 16);
    }
}

bool w2l_wgrad_fp8_tune_put(const int* v) {      // v[0..4] = key, v[5] = split count, v[6] = block order
    if (v[5] < 1 || v[5] > 0xffff || v[0] < 1 || v[3] < 1 || v[6] < 0 || v[6] > 1) return false;
    const int ts = (v[3] + BT - 1) / BT;
    if ((int64_t)v[5] > (int64_t)v[0] * ts) return false;
    std::lock_guard<std::mutex> lock(g_f8_mu);
    g_f8_tuned[F8ShapeKey(v[0], v[1], v[2], v[3], v[4])] = v[5] | (v[6] << 16);
    return true;
}
