// Conv1d forward / data-gradient as an implicit GEMM on CDNA4 MFMA (gfx950).
//
//   y[n][t][co] (+)= bias[co] + sum_{kw} sum_{ci} w[kw][co][ci] * xp[n][t*s + kw*d][ci]
//
// GEMM view: M = co (A = weights, K-contiguous), N = t (B = activations,
// K-contiguous because activations are channels-last), K = (kw, ci).
// The B operand is never materialised (no im2col): for one 64-channel chunk a
// block stages ONE time window of (BN-1)*s + (Kw-1)*d + 1 rows in LDS and every
// tap kw reads it at a row offset kw*d, so activation traffic is amortised over
// the Kw taps.  Weights stream one [BM x 64] tile per (chunk, tap) step.
// Both tiles are filled by LDS-DMA (global_load_lds_dwordx4), double-buffered,
// with the 16-byte-chunk XOR swizzle applied on the *source* address and on the
// ds_read_b128 address (the LDS image itself is lane-linear).
//
// Replaces nn.Conv1d at wav2letter.py:35-36,42 / jasper.py:96-105,127 (forward) and
// its autograd data gradient.  Padding is physical: the producer kernel writes the
// reflect (wav2letter.py:28-34) or zero halo, so the hot loop has no padding logic.
#include "common.h"
#include "../../include/w2l_hip.h"
#include <algorithm>
#include <map>
#include <vector>
#include <mutex>
#include <tuple>

namespace {

constexpr int BK = 64;                 // channels per K chunk (128-byte LDS rows)
constexpr int ROWB = BK * 2;           // bytes per LDS row

struct IgemmParams {
    const bf16_raw* x;
    const bf16_raw* w;
    void* y;
    const float* bias;
    float* stats;
    int stats_slots;          // 0: one statistics row per 128-column tile (plain stores); S: added onto row (tile mod S) with fp32 atomics
    int64_t x_rows_per_utt;   // x_bstride / Cin
    int64_t x_max_row;        // last readable flat row
    int N, Cin, Cout, Tout, Kw, stride, dil;
    int tiles_t, ncols, xrows_lds;
    int y_f32, accumulate;
    // split-K (w2l_conv1d_igemm_ws): `splits` blocks share one output tile, each reducing a contiguous range of the
    // (chunk, tap) steps; partial tiles go through fp32 slabs, the block that draws the last ticket sums and stores
    int splits;
    float* slabs;             // [tiles][splits][BM*BN]
    unsigned* tickets;        // [tiles], zero between launches
    // stream-K (SK kernels): the grid is sk_ranges blocks; block r owns steps [W*r/G, W*(r+1)/G) of the tile-major
    // (tile, step) space, W = sk_total = tiles * steps per tile, and walks the tiles that range touches one after the other.
    // A tile cut by range boundaries is combined exactly like a split-K tile (slab id = range + tile: unique, and the pieces
    // of one tile are consecutive), a whole tile inside one range goes straight to the epilogue
    int sk_ranges, sk_total;
    float descale;            // F8 kernels: y = acc * descale (+ bias), descale = 1 / (activation scale * weight scale)
    const float* descale_dev; // optional further factor in device memory (scale derived from a device-side amax)
    // EPI == 1 (data gradient fused with the BatchNorm-backward reduction of the layer that produced the conv's input):
    // the output tile IS the gradient wrt that layer's padded activation, so the epilogue also forms, per channel,
    // sum g*gate and sum g*gate*xhat over its rows (what bn_act_bwd_reduce_kernel computes in a pass of its own)
    const bf16_raw* bn_y;     // that layer's conv output [N][T][C] (C = this launch's Cout)
    const float* bn_scale;    // its BatchNorm scale / shift / mean / invstd
    const float* bn_shift;
    const float* bn_mean;
    const float* bn_invstd;
    const uint8_t* bn_mask;   // dropout keep bits (one byte per 8 channels) or NULL
    const int32_t* bn_lens;   // optional [N] valid lengths
    int bn_T, bn_pad_l, bn_pad_mode, bn_per, bn_Tp, bn_act;
    float bn_gk;              // 1 / (1 - p) with dropout, else 1
};

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
// one 16x16x128 MFMA on OCP e4m3 operands (cbsz = blgp = 0) with unit E8M0 block scales (127 = 2^0): per-tensor scaling
// is applied by the producer kernels and undone in the epilogue.  The 32 operand bytes of a lane are two ds_read_b128
// chunks; A and B take the same 32 channels of a row per lane group, so the products pair up whatever order the
// instruction walks its k range in.
__device__ __forceinline__ f32x4 mfma_e4m3_k128(bf16x8 a_lo, bf16x8 a_hi, bf16x8 b_lo, bf16x8 b_hi, f32x4 c) {
    const v4i al = __builtin_bit_cast(v4i, a_lo), ah = __builtin_bit_cast(v4i, a_hi);
    const v4i bl = __builtin_bit_cast(v4i, b_lo), bh = __builtin_bit_cast(v4i, b_hi);
    const v8i av = __builtin_shufflevector(al, ah, 0, 1, 2, 3, 4, 5, 6, 7);
    const v8i bv = __builtin_shufflevector(bl, bh, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
}

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst_wave_base, 16, 0, 0);
}

// MW x NW waves, each owning MS x NS MFMA tiles of 16x16: block tile BM = 16*MW*MS (co) x BN = 16*NW*NS (t)
// S = conv stride (compile time: the fragment reads then use immediate LDS offsets)
// F8: operands are OCP e4m3 bytes (x [.][rows][Cin], w [Kw][Cout][Cin], one byte per element): a 128-byte LDS row is 128
// channels, a K step is one tap of a 128-channel chunk, and its MS x NS MFMAs are v_mfma_scale_f32_16x16x128_f8f6f4 (twice
// the bf16 rate); LDS-DMA, swizzle, window reuse, split-K and the epilogue are the bf16 kernel's.  PIPE = 0 only.
// One piece of a stream-K range: the tile the range's cursor w_cur is in, the steps [s_begin, s_end) of that tile the range owns,
// how many ranges share the tile (nsplit), this range's place among them (split) and the first of the tile's consecutive slab ids.
// Range r owns steps [W r / G, W (r + 1) / G) of the tile-major space, W = tiles * S_; the range holding step w is
// ceil((w + 1) G / W) - 1.  Host and device run this same function (w2l_conv_streamk_pieces: the CPU test of the decomposition).
struct SkPiece { int tile, s_begin, s_end, nsplit, split; int64_t slab_base; };
__host__ __device__ inline SkPiece sk_piece(int W, int G, int S_, int r, int w_cur, int w_end) {
    SkPiece q;
    q.tile = w_cur / S_;
    q.s_begin = w_cur - q.tile * S_;
    q.s_end = q.s_begin + (w_end - w_cur) < S_ ? q.s_begin + (w_end - w_cur) : S_;
    const int r_lo = (int)((((int64_t)q.tile * S_ + 1) * G + W - 1) / W) - 1;
    const int r_hi = (int)((((int64_t)q.tile * S_ + S_) * G + W - 1) / W) - 1;
    q.nsplit = r_hi - r_lo + 1;
    q.split = r - r_lo;
    q.slab_base = (int64_t)r_lo + q.tile;
    return q;
}

template <int MW, int NW, int MS, int NS, int S, int PIPE, bool F8 = false, int EPI = 0, bool SK = false>
__global__ __launch_bounds__(64 * MW * NW, 2) void conv_igemm_kernel(IgemmParams p) {
    static_assert(!SK || (!F8 && EPI == 0 && S == 1), "stream-K is built for the plain bf16 stride-1 kernels");
    static_assert(!(F8 && PIPE != 0), "the e4m3 kernel is built for K-loop structure 0 only");
    static_assert(EPI == 0 || (!F8 && S == 1), "the fused BatchNorm-backward epilogue belongs to bf16 data gradients");
    constexpr int ESZ = F8 ? 1 : 2;                // bytes per operand element
    constexpr int BKE = ROWB / ESZ;                // channels per K chunk (one 128-byte LDS row)
    constexpr int BM = 16 * MW * MS, BN = 16 * NW * NS, NWAVES = MW * NW, NT = 64 * NWAVES;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / NW, wn = wave % NW;

    // consecutive logical ids share an XCD (xcd_remap): the `splits` blocks of one tile run next to each other in time and
    // place, so their slabs meet in that XCD's L2
    const int lin = xcd_remap(blockIdx.x, gridDim.x);
    // (consecutive ranges of a stream-K launch share tiles: the same remap keeps them on one XCD)
    int w_cur = 0, w_end = 0;
    if constexpr (SK) {
        w_cur = (int)(((int64_t)p.sk_total * lin) / p.sk_ranges);
        w_end = (int)(((int64_t)p.sk_total * (lin + 1)) / p.sk_ranges);
    }
    do {                                               // one pass per tile this block works on (exactly one unless SK)
    int split, tile, nsplit, sk_begin = 0, sk_end = 0;
    int64_t slab_base;
    if constexpr (SK) {
        __syncthreads();                               // the previous tile's LDS (ticket word, statistics scratch) is dead
        const SkPiece q = sk_piece(p.sk_total, p.sk_ranges, (p.Cin / (ROWB / (F8 ? 1 : 2))) * p.Kw, lin, w_cur, w_end);
        tile = q.tile;
        sk_begin = q.s_begin;
        sk_end = q.s_end;
        w_cur += sk_end - sk_begin;
        nsplit = q.nsplit;
        split = q.split;
        slab_base = q.slab_base;
    } else {
        nsplit = p.splits;
        split = nsplit > 1 ? lin % nsplit : 0;
        tile = nsplit > 1 ? lin / nsplit : lin;
        slab_base = (int64_t)tile * nsplit;
    }
    const int tm = tile / p.ncols;
    const int col = tile - tm * p.ncols;
    const int n = col / p.tiles_t;
    const int tt = col - n * p.tiles_t;
    const int m0 = tm * BM;
    const int t0 = tt * BN;

    constexpr int s = S;
    const int d = p.dil, Kw = p.Kw, Cin = p.Cin;
    const int xrows = p.xrows_lds;                 // multiple of 8
    char* wbuf0 = smem;
    char* wbuf1 = smem + BM * ROWB;
    char* xbuf0 = smem + 2 * BM * ROWB;
    char* xbuf1 = xbuf0 + xrows * ROWB;

    const int64_t xrow0 = (int64_t)n * p.x_rows_per_utt + (int64_t)t0 * s;

    // ---- staging (each wave-instruction writes 8 LDS rows = 1 KiB by LDS-DMA) ----
    // The main loop is issue-bound (VALU + MFMA share a SIMD's issue port), so everything that does not
    // change from step to step is hoisted: a weight tile's source address is a wave-uniform slab base
    // (tap, chunk) plus a per-lane 32-bit offset computed once.
    const int srow = lane >> 3;                    // row within the 8-row group
    const int schunk = lane & 7;                   // LDS chunk this lane fills
    const int gchunk = schunk ^ srow;              // source chunk (groups are 8-row aligned: row&7 == srow)
    constexpr int WGROUPS = BM / 8;
    constexpr int W_PER_WAVE = (WGROUPS + NWAVES - 1) / NWAVES;
    unsigned w_voff[W_PER_WAVE];
#pragma unroll
    for (int i = 0; i < W_PER_WAVE; ++i) {
        int co = m0 + (wave + i * NWAVES) * 8 + srow;
        co = co < p.Cout ? co : p.Cout - 1;
        w_voff[i] = (unsigned)co * (unsigned)Cin * (unsigned)ESZ + (unsigned)gchunk * 16u;
    }
    const int64_t w_tap_bytes = (int64_t)p.Cout * Cin * ESZ;
    auto stage_w = [&](char* dst, int kw, int c) {
        W2L_DIAG_SKIP_DMA(p);
        const char* slab = reinterpret_cast<const char*>(p.w) + kw * w_tap_bytes + c * (BK * 2);   // wave-uniform
#pragma unroll
        for (int i = 0; i < W_PER_WAVE; ++i) {
            const int grp = wave + i * NWAVES;
            if ((WGROUPS % NWAVES == 0) || grp < WGROUPS) glds16(slab + w_voff[i], dst + grp * 1024);
        }
    };
    auto stage_x = [&](char* dst, int c) {
        W2L_DIAG_SKIP_DMA(p);
        const int ngrp = xrows >> 3;
        for (int grp = wave; grp < ngrp; grp += NWAVES) {
            int64_t r = xrow0 + grp * 8 + srow;
            r = r < p.x_max_row ? r : p.x_max_row;
            const char* src = reinterpret_cast<const char*>(p.x) + (r * Cin + (int64_t)c * BKE) * ESZ + gchunk * 16;
            glds16(src, dst + grp * 1024);
        }
    };

    f32x4 acc[MS][NS];
#pragma unroll
    for (int i = 0; i < MS; ++i)
#pragma unroll
        for (int j = 0; j < NS; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nchunks = Cin / BKE;
    // this block's share of the nchunks * Kw (chunk-major) steps; a range may start in the middle of a chunk
    const int total_steps = nchunks * Kw;
    const int s_begin = SK ? sk_begin : (nsplit > 1 ? (int)(((int64_t)total_steps * split) / nsplit) : 0);
    const int s_end = SK ? sk_end : (nsplit > 1 ? (int)(((int64_t)total_steps * (split + 1)) / nsplit) : total_steps);
    const int nsteps = s_end - s_begin;
    const int c_first = s_begin / Kw, kw_first = s_begin - c_first * Kw;

    // per-lane fragment offsets (constant over the whole K loop)
    const int fr = lane & 15, fq = lane >> 4;
    // bf16: k-substep 0 reads 16-byte chunk fq of the row, k-substep 1 chunk 4+fq.  e4m3: the lane group's 32 channels
    // are chunks 2fq and 2fq+1, both operands of the one MFMA of the step.
    constexpr int CH1_XOR = F8 ? 16 : 64;          // byte distance (under the XOR swizzle) between the two chunks of a lane
    const int ch0 = F8 ? 2 * fq : fq;
    const int a_lane0 = (wm * MS * 16 + fr) * ROWB + ((ch0 ^ (fr & 7)) << 4);
    const int a_lane1 = a_lane0 ^ CH1_XOR;
    const int b_row0 = (wn * NS * 16 + fr) * s;      // tile rows ni*16*s further down share (row & 7): NS reads per base

    // Two K-loop structures are built for every block shape; which one is faster depends on the shape and the
    // pass (measured by w2l_conv1d_igemm_tune): PIPE = 0 keeps the barrier at the top of a step, PIPE = 1 moves it
    // to the middle so that the next step's first fragment reads overlap this step's last MFMAs.
    if constexpr (PIPE == 0) {
        stage_x((c_first & 1) ? xbuf1 : xbuf0, c_first);
        stage_w(wbuf0, kw_first, c_first);

        int kw = kw_first, c = c_first;
        for (int step = 0; step < nsteps; ++step) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            // prefetch the next step's tiles into the other buffers
            int kw_n = kw + 1, c_n = c;
            if (kw_n == Kw) { kw_n = 0; c_n = c + 1; }
            if (step + 1 < nsteps) {
                stage_w((step & 1) ? wbuf0 : wbuf1, kw_n, c_n);
                if (kw_n == 0) stage_x((c_n & 1) ? xbuf1 : xbuf0, c_n);
            }
            const char* wb = (step & 1) ? wbuf1 : wbuf0;
            const char* xb = (c & 1) ? xbuf1 : xbuf0;
            const char* A0 = wb + a_lane0;
            const char* A1 = wb + a_lane1;
            const int tsh = b_row0 + kw * d;             // LDS row of this lane's first B tile for this tap
            const int sw0 = ((ch0 ^ (tsh & 7)) << 4);
            const char* Bb = xb + (tsh << 7);
            const char* B0 = Bb + sw0;
            const char* B1 = Bb + (sw0 ^ CH1_XOR);       // bf16: chunk 4+fq == (chunk fq) ^ 4; e4m3: chunk 2fq+1
            // both k-substeps' fragments are requested up front (16 ds_read_b128 in flight); the second half lands
            // while the first half's MFMAs run.  sched_barrier pins that order against the register-pressure scheduler.
            bf16x8 a0[MS], b0[NS], a1[MS], b1[NS];
    #pragma unroll
            for (int mi = 0; mi < MS; ++mi) a0[mi] = *reinterpret_cast<const bf16x8*>(A0 + mi * 16 * ROWB);
    #pragma unroll
            for (int ni = 0; ni < NS; ++ni) b0[ni] = *reinterpret_cast<const bf16x8*>(B0 + ni * 16 * S * ROWB);
    #pragma unroll
            for (int mi = 0; mi < MS; ++mi) a1[mi] = *reinterpret_cast<const bf16x8*>(A1 + mi * 16 * ROWB);
    #pragma unroll
            for (int ni = 0; ni < NS; ++ni) b1[ni] = *reinterpret_cast<const bf16x8*>(B1 + ni * 16 * S * ROWB);
            if constexpr (F8) {
    #pragma unroll
                for (int mi = 0; mi < MS; ++mi)
    #pragma unroll
                    for (int ni = 0; ni < NS; ++ni)
                        acc[mi][ni] = mfma_e4m3_k128(a0[mi], a1[mi], b0[ni], b1[ni], acc[mi][ni]);
            } else {
    #pragma unroll
                for (int mi = 0; mi < MS; ++mi)
    #pragma unroll
                    for (int ni = 0; ni < NS; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[mi], b0[ni], acc[mi][ni], 0, 0, 0);
    #pragma unroll
                for (int mi = 0; mi < MS; ++mi)
    #pragma unroll
                    for (int ni = 0; ni < NS; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[mi], b1[ni], acc[mi][ni], 0, 0, 0);
                // schedule: ks0 fragment reads, then ks1 reads slotted one per MFMA into the ks0 MFMAs, then the rest
                __builtin_amdgcn_sched_group_barrier(0x100, MS + NS, 0);
    #pragma unroll
                for (int i = 0; i < MS + NS; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 2 * MS * NS - (MS + NS), 0);
            }
            kw = kw_n;
            c = c_n;
        }

    } else {
        // fragments of k-substep ks (0/1) of the step whose weights sit in W buffer (stp & 1), for tap kk of chunk cc
        auto load_frags = [&](int ks, int stp, int cc, int kk, bf16x8* a, bf16x8* b) {
            const char* wb = (stp & 1) ? wbuf1 : wbuf0;
            const char* xb = (cc & 1) ? xbuf1 : xbuf0;
            const char* A = wb + (ks ? a_lane1 : a_lane0);
            const int tsh = b_row0 + kk * d;             // LDS row of this lane's first B tile for this tap
            const int sw0 = ((ch0 ^ (tsh & 7)) << 4);
            const char* B = xb + (tsh << 7) + (ks ? (sw0 ^ CH1_XOR) : sw0);      // chunk 4+fq == (chunk fq) ^ 4
            W2L_DIAG_SKIP_FRAGS(p, stp);
    #pragma unroll
            for (int mi = 0; mi < MS; ++mi) a[mi] = *reinterpret_cast<const bf16x8*>(A + mi * 16 * ROWB);
    #pragma unroll
            for (int ni = 0; ni < NS; ++ni) b[ni] = *reinterpret_cast<const bf16x8*>(B + ni * 16 * S * ROWB);
        };
        auto mfma_all = [&](const bf16x8* a, const bf16x8* b) {
            W2L_DIAG_SKIP_MFMA(p);
    #pragma unroll
            for (int mi = 0; mi < MS; ++mi)
    #pragma unroll
                for (int ni = 0; ni < NS; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
        };
        // schedule of one half step: MS+NS fragment reads slotted one per MFMA into the first MFMAs, then the rest
        auto sched_half = [&]() {
    #pragma unroll
            for (int i = 0; i < MS + NS; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, MS * NS - (MS + NS), 0);
        };
        auto advance = [&](int kk, int cc, int& kk2, int& cc2) {
            kk2 = kk + 1;
            cc2 = cc;
            if (kk2 == Kw) { kk2 = 0; cc2 = cc + 1; }
        };

        // ---- K loop.  The block-wide barrier sits in the MIDDLE of a step: while the MFMAs of k-substep 1 run, the LDS-DMA
        // for step+2 is issued and the k-substep-0 fragments of step+1 are already being read, so neither the barrier
        // skew nor the first fragment reads of a step are exposed (they used to be, once per step, right after the barrier).
        //   top of step:  a0/b0 (substep 0 of this step) requested; DMA of step+1 in flight into the other buffers
        //   1. request a1/b1 (substep 1, same buffers), run the substep-0 MFMAs
        //   2. vmcnt(0) + barrier: every wave has finished READING this step's buffers, and step+1's tiles have landed
        //   3. DMA step+2 into this step's (now dead) buffers; request a0/b0 of step+1; run the substep-1 MFMAs
        stage_x((c_first & 1) ? xbuf1 : xbuf0, c_first);
        stage_w(wbuf0, kw_first, c_first);
        int kw = kw_first, c = c_first, kw_n, c_n;
        advance(kw, c, kw_n, c_n);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (nsteps > 1) {
            stage_w(wbuf1, kw_n, c_n);
            if (kw_n == 0) stage_x((c_n & 1) ? xbuf1 : xbuf0, c_n);
        }
        bf16x8 a0[MS], b0[NS], a1[MS], b1[NS];
        load_frags(0, 0, c, kw, a0, b0);
        W2L_DIAG_CLK_BEGIN();
        for (int step = 0; step + 1 < nsteps; ++step) {
            load_frags(1, step, c, kw, a1, b1);
            mfma_all(a0, b0);
            sched_half();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            int kw_nn, c_nn;
            advance(kw_n, c_n, kw_nn, c_nn);
            if (step + 2 < nsteps) {
                stage_w((step & 1) ? wbuf1 : wbuf0, kw_nn, c_nn);
                if (kw_nn == 0) stage_x((c_nn & 1) ? xbuf1 : xbuf0, c_nn);
            }
            load_frags(0, step + 1, c_n, kw_n, a0, b0);
            mfma_all(a1, b1);
            sched_half();
            kw = kw_n; c = c_n;
            kw_n = kw_nn; c_n = c_nn;
        }
        W2L_DIAG_CLK_END(tid);
        load_frags(1, nsteps - 1, c, kw, a1, b1);
        mfma_all(a0, b0);
        mfma_all(a1, b1);

    }

    // ---- split-K: publish the partial tile, draw a ticket; the last arriver sums all partials IN SPLIT ORDER (so the result
    // does not depend on which block arrived last) and goes on to the epilogue.  No block ever waits for another one.
    // (agent-scope release before the ticket, agent-scope acquire after it: correct for any placement of the blocks on XCDs)
    if (nsplit > 1) {
        float* slab = p.slabs + (slab_base + split) * (BM * BN);
#pragma unroll
        for (int mi = 0; mi < MS; ++mi)
#pragma unroll
            for (int ni = 0; ni < NS; ++ni)
                *reinterpret_cast<f32x4*>(slab + ((mi * NS + ni) * NT + tid) * 4) = acc[mi][ni];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                               // every wave's stores are issued and complete; main-loop LDS is dead
        unsigned* flag = reinterpret_cast<unsigned*>(smem);
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            *flag = __hip_atomic_fetch_add(&p.tickets[tile], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        const unsigned ticket = *flag;
        if (ticket != (unsigned)(nsplit - 1)) continue;        // (not the last piece of this tile: on to the next tile, or out)
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(&p.tickets[tile], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // clean for the next launch
        }
        __syncthreads();
        const float* base = p.slabs + slab_base * (BM * BN);
#pragma unroll
        for (int mi = 0; mi < MS; ++mi)
#pragma unroll
            for (int ni = 0; ni < NS; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int sp = 0; sp < nsplit; ++sp) {
            const float* sl = base + (int64_t)sp * (BM * BN);
#pragma unroll
            for (int mi = 0; mi < MS; ++mi)
#pragma unroll
                for (int ni = 0; ni < NS; ++ni)
                    acc[mi][ni] += *reinterpret_cast<const f32x4*>(sl + ((mi * NS + ni) * NT + tid) * 4);
        }
        __syncthreads();                               // `flag` shares LDS with the statistics scratch below
    }

    // ---- epilogue: bias, optional accumulate, store, BatchNorm partial statistics ----
    // acc[mi][ni][r] = y[co = m0 + (wm*MS+mi)*16 + fq*4 + r][t = t0 + (wn*NS+ni)*16 + fr]
    const int Cout = p.Cout, Tout = p.Tout;
    const float descale = F8 ? p.descale * (p.descale_dev ? p.descale_dev[0] : 1.f) : 1.f;
    float s1[MS][4], s2[MS][4];
#pragma unroll
    for (int mi = 0; mi < MS; ++mi) {
        const int co = m0 + (wm * MS + mi) * 16 + fq * 4;
        const bool co_ok = co < Cout;
        f32x4 bias4 = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.bias && co_ok) bias4 = *reinterpret_cast<const f32x4*>(p.bias + co);
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1[mi][r] = 0.f; s2[mi][r] = 0.f; }
#pragma unroll
        for (int ni = 0; ni < NS; ++ni) {
            const int t = t0 + (wn * NS + ni) * 16 + fr;
            const bool ok = co_ok && t < Tout;
            f32x4 v = F8 ? acc[mi][ni] * descale + bias4 : acc[mi][ni] + bias4;
            const int64_t off = ((int64_t)n * Tout + t) * Cout + co;
            if (ok) {
                if (p.y_f32) {
                    float* yp = reinterpret_cast<float*>(p.y) + off;
                    if (p.accumulate) v += *reinterpret_cast<const f32x4*>(yp);
                    *reinterpret_cast<f32x4*>(yp) = v;
                } else {
                    u16x4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = f32_to_bf16_bits(v[r]);
                    *reinterpret_cast<u16x4*>(reinterpret_cast<bf16_raw*>(p.y) + off) = o;
                }
                if constexpr (EPI == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { s1[mi][r] += v[r]; s2[mi][r] += v[r] * v[r]; }
                }
            }
        }
    }
    if constexpr (EPI == 1) {
        // ---- BatchNorm-backward sums of the layer that produced this conv's input (see IgemmParams).  Flat output row t is
        // padded row u of utterance nu; its gradient belongs to source frame ts (itself, or its mirror image when the row is
        // a reflected halo row -- the fold of bn_act.hip add_grad8 is linear, so every padded row simply counts with its
        // source frame's gate and xhat).  The y / mask loads of CH channel blocks are issued together, from clamped (always
        // valid) addresses and outside any branch, so they overlap instead of forming one dependent chain per element.
        int64_t yrow[NS];
        bool live[NS];
#pragma unroll
        for (int ni = 0; ni < NS; ++ni) {
            const int t = t0 + (wn * NS + ni) * 16 + fr;
            const int tc = t < Tout ? t : Tout - 1;
            const int nu = (int)((unsigned)tc / (unsigned)p.bn_per);
            const int u = tc - nu * p.bn_per;
            int ts = u - p.bn_pad_l;
            bool lv = t < Tout && u < p.bn_Tp;
            if (ts < 0 || ts >= p.bn_T) {
                if (p.bn_pad_mode == 1) ts = ts < 0 ? -ts : 2 * (p.bn_T - 1) - ts;
                else lv = false;
            }
            ts = ts < 0 ? 0 : (ts >= p.bn_T ? p.bn_T - 1 : ts);
            if (lv && p.bn_lens && ts >= p.bn_lens[nu]) lv = false;          // masked frame: no gradient through it
            live[ni] = lv;
            yrow[ni] = (int64_t)nu * p.bn_T + ts;
        }
        constexpr int CH = MS % 4 == 0 ? 4 : (MS % 2 == 0 ? 2 : 1);
        __syncthreads();                               // main-loop LDS is dead from here: the sums go straight into it
        float* red = reinterpret_cast<float*>(smem);   // [NW][2][BM]
#pragma unroll
        for (int m0i = 0; m0i < MS; m0i += CH) {
            u16x4 yv[CH][NS];
            unsigned mb[CH][NS];
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                int co = m0 + (wm * MS + m0i + c) * 16 + fq * 4;
                co = co < Cout ? co : Cout - 4;
#pragma unroll
                for (int ni = 0; ni < NS; ++ni) {
                    yv[c][ni] = *reinterpret_cast<const u16x4*>(p.bn_y + yrow[ni] * Cout + co);
                    mb[c][ni] = p.bn_mask ? (unsigned)p.bn_mask[yrow[ni] * (Cout >> 3) + (co >> 3)] : 0xFFu;
                }
            }
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const int mi = m0i + c;
                const int co = m0 + (wm * MS + mi) * 16 + fq * 4;
                const bool co_ok = co < Cout;
                const int cc = co_ok ? co : Cout - 4;
                f32x4 bsc = f32x4{1.f, 1.f, 1.f, 1.f}, bsh = f32x4{0.f, 0.f, 0.f, 0.f}, bmu = bsh, bis = bsh;
                if (p.bn_scale) { bsc = *reinterpret_cast<const f32x4*>(p.bn_scale + cc); bsh = *reinterpret_cast<const f32x4*>(p.bn_shift + cc); }
                if (p.bn_mean) { bmu = *reinterpret_cast<const f32x4*>(p.bn_mean + cc); bis = *reinterpret_cast<const f32x4*>(p.bn_invstd + cc); }
#pragma unroll
                for (int ni = 0; ni < NS; ++ni) {
                    const bool on = live[ni] && co_ok;
                    const unsigned bits = mb[c][ni] >> (co & 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float yf = bf16_bits_to_f32(yv[c][ni][r]);
                        const bool keep = (bits >> r) & 1u;
                        float z = yf * bsc[r] + bsh[r];
                        z = keep ? z * p.bn_gk : 0.f;
                        bool pass = keep && on;
                        if (p.bn_act == 1) pass = pass && z >= 0.f && z <= 20.f;
                        else if (p.bn_act == 2) pass = pass && z > 0.f;
                        const float g = pass ? acc[mi][ni][r] * p.bn_gk : 0.f;
                        s1[mi][r] += g;
                        s2[mi][r] += g * ((yf - bmu[r]) * bis[r]);
                    }
                }
                // this channel block is complete: combine the 16 column lanes and park the sums in LDS right away (keeping
                // all MS blocks' sums in registers next to the accumulators spills)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float a = s1[mi][r], b = s2[mi][r];
#pragma unroll
                    for (int m = 1; m < 16; m <<= 1) {
                        a += __shfl_xor(a, m, 64);
                        b += __shfl_xor(b, m, 64);
                    }
                    if (fr == 0) {
                        const int cl = (wm * MS + mi) * 16 + fq * 4 + r;
                        red[(wn * 2 + 0) * BM + cl] = a;
                        red[(wn * 2 + 1) * BM + cl] = b;
                    }
                }
            }
        }
    }
    if (p.stats) {
        // statistics rows are laid out per 128-column tile (w2l_conv_stat_tiles); a 256-column block owns two
        constexpr int HALVES = BN / 128;
        constexpr int WPH = NW / HALVES;               // N-waves per 128-column half
        static_assert(BN % 128 == 0 || true, "");
        float* red = reinterpret_cast<float*>(smem);   // [NW][2][BM]
        if constexpr (EPI == 0) {
            __syncthreads();                           // main-loop LDS is dead from here
#pragma unroll
            for (int mi = 0; mi < MS; ++mi)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float a = s1[mi][r], b = s2[mi][r];
#pragma unroll
                    for (int m = 1; m < 16; m <<= 1) {
                        a += __shfl_xor(a, m, 64);
                        b += __shfl_xor(b, m, 64);
                    }
                    if (fr == 0) {
                        const int cl = (wm * MS + mi) * 16 + fq * 4 + r;
                        red[(wn * 2 + 0) * BM + cl] = a;
                        red[(wn * 2 + 1) * BM + cl] = b;
                    }
                }
        }
        __syncthreads();
        if (p.stats_slots) {
            // w2l_conv_stats_mode(S): the block's sums -- all its NW column waves, whatever the block shape -- ADDED onto row
            // (column tile mod S) of the zero-filled [S][2][Cout] buffer; its reader re-reduces the S rows
            float* dst = p.stats + (((int64_t)n * p.tiles_t + tt) % p.stats_slots) * 2 * Cout;
            for (int cl = tid; cl < BM; cl += NT) {
                if (m0 + cl < Cout) {
                    float a = 0.f, b = 0.f;
#pragma unroll
                    for (int w = 0; w < NW; ++w) {
                        a += red[(w * 2 + 0) * BM + cl];
                        b += red[(w * 2 + 1) * BM + cl];
                    }
                    atomicAdd(dst + m0 + cl, a);
                    atomicAdd(dst + Cout + m0 + cl, b);
                }
            }
        } else {
        const int tiles128 = (Tout + 127) / 128;
        for (int idx = tid; idx < BM * HALVES; idx += NT) {
            const int h = idx / BM, cl = idx - h * BM;
            const int trow = tt * HALVES + h;          // 128-column tile index inside the utterance
            if (m0 + cl < Cout && trow < tiles128) {
                float a = 0.f, b = 0.f;
#pragma unroll
                for (int w = 0; w < WPH; ++w) {
                    a += red[((h * WPH + w) * 2 + 0) * BM + cl];
                    b += red[((h * WPH + w) * 2 + 1) * BM + cl];
                }
                const int64_t srow = (int64_t)n * tiles128 + trow;
                float* dst = p.stats + srow * 2 * Cout;
                dst[m0 + cl] = a;
                dst[Cout + m0 + cl] = b;
            }
        }
        }
    }
    } while (SK && w_cur < w_end);
}

struct TileCfg { int mw, nw, ms, ns; float eff; };
// candidate block shapes (BM = 16*mw*ms output channels x BN = 16*nw*ns time rows) with their measured
// relative MFMA efficiency at full occupancy (tools/bench_conv.py --sweep, MI355X)
constexpr TileCfg kCfgs[] = {
    {2, 2, 2, 4, 0.83f}, {2, 2, 3, 4, 1.00f}, {2, 2, 4, 4, 1.00f}, {2, 2, 5, 4, 1.01f},   // BN 128, 4 waves
    {4, 2, 3, 4, 0.90f}, {4, 2, 4, 4, 0.92f},                                               // BN 128, 8 waves
    {2, 3, 2, 3, 0.80f}, {2, 3, 3, 3, 0.97f}, {2, 3, 4, 3, 0.97f}, {2, 3, 5, 3, 0.98f},   // BN 144, 6 waves
    {2, 4, 2, 4, 0.85f}, {2, 4, 3, 4, 0.97f}, {2, 4, 4, 4, 1.00f}, {2, 4, 5, 4, 1.04f},   // BN 256, 8 waves, 1 block/CU
    {2, 4, 6, 4, 1.10f}, {2, 4, 7, 4, 1.06f}, {2, 4, 8, 4, 1.12f},
    {2, 3, 4, 6, 1.00f}, {2, 3, 5, 6, 1.00f}, {2, 3, 6, 6, 1.00f}, {2, 3, 7, 6, 1.00f},   // BN 288, 6 waves, 1 block/CU
    // BN 384, 8 waves (round 3): a weight tile serves 384 columns instead of 256 -- a third less LDS-DMA per MFMA, the
    // instruction a K step spends a third of its issue time on (in-kernel stamps).  Useful where the column count
    // tiles well: the flat data gradients of the 640-wide layers +12-14 %, 768-wide +1-4 %; never the per-utterance forward
    // pass (500 columns = 2 tiles of 384)
    {2, 4, 4, 6, 1.00f}, {2, 4, 5, 6, 1.00f}, {2, 4, 6, 6, 1.00f},
    {2, 4, 4, 7, 1.00f}, {2, 4, 5, 7, 1.00f},                                               // BN 448, 8 waves
};
constexpr int kNumCfgs = sizeof(kCfgs) / sizeof(kCfgs[0]);

template <int MW, int NW, int MS, int NS, int PIPE>
int launch_cfg1(const IgemmParams& p, int tiles_m, size_t lds, hipStream_t stream, int epi = 0) {
    if (epi == 1) {
        // statistics rows are per 128-column tile -- unless the sums are added onto slot rows (w2l_conv_stats_mode): any shape then
        if ((16 * NW * NS) % 128 == 0 && 16 * NW * NS <= 256 ? true : p.stats_slots > 0) {
            auto kern1 = conv_igemm_kernel<MW, NW, MS, NS, 1, PIPE, false, 1>;
            W2L_CHECK_HIP(w2l_allow_big_lds((const void*)kern1));
            hipLaunchKernelGGL(kern1, dim3(tiles_m * p.ncols), dim3(64 * MW * NW), lds, stream, p);
            W2L_CHECK_LAUNCH();
            return 0;
        } else {
            w2l_set_error("conv1d_igemm: the fused BatchNorm-backward epilogue needs a block shape of 128-column tiles");
            return 1;
        }
    }
    if (p.stride == 2) {
        if constexpr (MW == 2 && NW == 2 && MS == 4 && NS == 4) {
            auto kern2 = conv_igemm_kernel<2, 2, 4, 4, 2, PIPE>;
            W2L_CHECK_HIP(w2l_allow_big_lds((const void*)kern2));
            hipLaunchKernelGGL(kern2, dim3(tiles_m * p.ncols), dim3(256), lds, stream, p);
            W2L_CHECK_LAUNCH();
            return 0;
        } else {
            w2l_set_error("conv1d_igemm: stride 2 is only built for the 128x128 block shape");
            return 1;
        }
    }
    if (p.sk_ranges > 0) {
        auto kern_sk = conv_igemm_kernel<MW, NW, MS, NS, 1, PIPE, false, 0, true>;
        W2L_CHECK_HIP(w2l_allow_big_lds((const void*)kern_sk));
        hipLaunchKernelGGL(kern_sk, dim3(p.sk_ranges), dim3(64 * MW * NW), lds, stream, p);
        W2L_CHECK_LAUNCH();
        return 0;
    }
    auto kern = conv_igemm_kernel<MW, NW, MS, NS, 1, PIPE>;
    W2L_CHECK_HIP(w2l_allow_big_lds((const void*)kern));
    hipLaunchKernelGGL(kern, dim3(tiles_m * p.ncols), dim3(64 * MW * NW), lds, stream, p);
    W2L_CHECK_LAUNCH();
    return 0;
}

template <int MW, int NW, int MS, int NS>
int launch_cfg(const IgemmParams& p, int tiles_m, size_t lds, hipStream_t stream, int pipe, int epi = 0) {
    return pipe ? launch_cfg1<MW, NW, MS, NS, 1>(p, tiles_m, lds, stream, epi)
                : launch_cfg1<MW, NW, MS, NS, 0>(p, tiles_m, lds, stream, epi);
}

// e4m3 launches: K-loop structure 0, stride 1, a subset of the block shapes (indices into kCfgs)
constexpr int kF8Cfgs[] = {2, 5, 12, 14, 16, 1, 3, 8, 9, 11, 13, 18, 19, 21, 24};     // (the first five were round 2's first set;
                                                                              //  21 / 24: the 384- / 448-column shapes, round 3)
constexpr int kNumF8Cfgs = sizeof(kF8Cfgs) / sizeof(kF8Cfgs[0]);

template <int MW, int NW, int MS, int NS>
int launch_f8(const IgemmParams& p, int tiles_m, size_t lds, hipStream_t stream) {
    auto kern = conv_igemm_kernel<MW, NW, MS, NS, 1, 0, true>;
    W2L_CHECK_HIP(w2l_allow_big_lds((const void*)kern));
    hipLaunchKernelGGL(kern, dim3(tiles_m * p.ncols), dim3(64 * MW * NW), lds, stream, p);
    W2L_CHECK_LAUNCH();
    return 0;
}

}  // namespace

// A configuration index is (block shape) + kNumCfgs * (K-loop structure PIPE) + 2 * kNumCfgs * (split-K option).
// Split option 0 = stream-K (round 5): one block per resident slot, each an equal share of ALL the launch's (tile, step) pairs
constexpr int kSplits[] = {1, 2, 3, 4, 5, 6, 8, 0};
constexpr int kNumSplits = sizeof(kSplits) / sizeof(kSplits[0]);
constexpr int kMaxSplit = 8;
constexpr int kSkMinSteps = 8;                          // a stream-K range is at least this many steps
constexpr int kBaseCfgs = 2 * kNumCfgs;
constexpr size_t kTicketBytes = 64 * 1024;              // head of the split-K workspace: one counter per output tile
// thread-local: the tuner (and the test hook below) force a configuration for launches made by the CALLING thread only --
// a backward running on an autograd worker thread while another thread tunes never sees a forced index
static thread_local int g_force_cfg = -1;
static thread_local int g_stats_slots = 0;      // w2l_conv_stats_mode
extern "C" void w2l_conv_stats_mode(int slots) { g_stats_slots = slots < 0 ? 0 : (slots > 64 ? 64 : slots); }
extern "C" void w2l_conv_force_tile_config(int idx) { g_force_cfg = idx; }
W2L_DIAG_IGEMM_EXPORTS

static inline int cfg_xrows(const TileCfg& c, int stride, int Kw, int dil) {
    const int bn = 16 * c.nw * c.ns;
    return ((bn - 1) * stride + (Kw - 1) * dil + 1 + 7) & ~7;
}

// measured choices (w2l_conv1d_igemm_tune): shape -> block-shape index
typedef std::tuple<int, int, int, int, int, int, int, int> ShapeKey;
static std::map<ShapeKey, int> g_tuned;
static std::mutex g_tuned_mu;

static bool cfg_feasible(int idx, int Kw, int stride, int dil, bool need_bn128) {
    if (idx < 0 || idx >= kBaseCfgs * kNumSplits) return false;
    const int i = idx % kNumCfgs;
    const TileCfg& c = kCfgs[i];
    const int bm = 16 * c.mw * c.ms, bn = 16 * c.nw * c.ns;
    if (need_bn128 && (bn % 128 != 0 || bn > 256)) return false;   // the statistics epilogue assumes whole waves per 128-column tile
    if (stride != 1 && i != 2) return false;
    if (i == 20) return false;                               // spills (kept only for index stability)
    const size_t lds = 2 * (size_t)bm * ROWB + 2 * (size_t)cfg_xrows(c, stride, Kw, dil) * ROWB;
    return lds <= 160 * 1024;
}

// pick the block shape: a measured choice if this shape was tuned, else a cost model (whole rounds of
// resident blocks on the 256 CUs, larger tiles preferred)
// statistics flag of a shape key: 0 none, 1 one row per 128-column tile (block shapes of 128 / 256 columns only), 2 added onto
// slot rows (w2l_conv_stats_mode: every block shape)
static int stats_flag(const float* stats_partial) { return stats_partial == nullptr ? 0 : (g_stats_slots > 0 ? 2 : 1); }

static int choose_cfg(int N, int Cin, int Cout, int Tout, int Kw, int stride, int dil, int sflag) {
    const bool need_bn128 = sflag == 1;
    if (g_force_cfg < 0) {
        std::lock_guard<std::mutex> lock(g_tuned_mu);
        auto it = g_tuned.find(ShapeKey(N, Cin, Cout, Tout, Kw, stride, dil, sflag));
        if (it != g_tuned.end()) return it->second;
    }
    if (g_force_cfg >= 0) return cfg_feasible(g_force_cfg, Kw, stride, dil, need_bn128) ? g_force_cfg : -1;
    int best = -1;                                     // the cost model only ranks the PIPE = 0 variants
    double best_cost = 1e30;
    for (int i = 0; i < kNumCfgs; ++i) {
        const TileCfg& c = kCfgs[i];
        const int bm = 16 * c.mw * c.ms, bn = 16 * c.nw * c.ns;
        if (need_bn128 && (bn % 128 != 0 || bn > 256)) continue;     // BatchNorm partial statistics are per 128-column tile
        if (stride != 1 && i != 2) continue;          // strided convs (first layer only) use the 128x128 shape
        if (i == 20) continue;
        const size_t lds = 2 * (size_t)bm * ROWB + 2 * (size_t)cfg_xrows(c, stride, Kw, dil) * ROWB;
        if (lds > 160 * 1024) continue;
        const int waves = c.mw * c.nw;
        int per_cu = (int)((160 * 1024) / lds);
        const int wave_cap = 16 / waves;                 // <= 16 waves per CU at the VGPR budget of these kernels
        if (per_cu > wave_cap) per_cu = wave_cap;
        if (per_cu < 1) continue;
        const long blocks = (long)((Cout + bm - 1) / bm) * N * ((Tout + bn - 1) / bn);
        const long slots = 256L * per_cu;
        const long rounds = (blocks + slots - 1) / slots;
        // time ~ rounds x (work per block) x (blocks sharing a CU) / efficiency
        double eff = c.eff * (per_cu * waves >= 8 ? 1.0 : 0.75);
        double cost = (double)rounds * bm * bn * per_cu / eff;
        cost *= 1.0 + 1e-3 * i;                          // stable tie-break
        if (cost < best_cost) { best_cost = cost; best = i; }
    }
    return best;
}

extern "C" int w2l_conv_stat_tiles(int N, int Tout) { return N * ((Tout + 127) / 128); }

// bytes of split-K workspace that let every configuration of this problem run (slabs of the largest split + the tickets)
static size_t splitk_bytes(int cfg_i, int splits, int N, int Cout, int Tout) {
    const TileCfg& c = kCfgs[cfg_i];
    const int bm = 16 * c.mw * c.ms, bn = 16 * c.nw * c.ns;
    const size_t tiles = (size_t)((Cout + bm - 1) / bm) * N * ((Tout + bn - 1) / bn);
    return kTicketBytes + tiles * splits * bm * bn * sizeof(float);
}

// blocks of a stream-K launch of this shape: the slots the chip has for it (one or two blocks per CU)
static int sk_ranges(const TileCfg& c, int stride, int Kw, int dil) {
    const size_t lds = 2 * (size_t)(16 * c.mw * c.ms) * ROWB + 2 * (size_t)cfg_xrows(c, stride, Kw, dil) * ROWB;
    int per_cu = (int)((160 * 1024) / lds);
    if (per_cu > 8 / (c.mw * c.nw)) per_cu = 8 / (c.mw * c.nw);          // (eight waves per CU: what two 4-wave blocks hold)
    return 256 * (per_cu < 1 ? 1 : (per_cu > 2 ? 2 : per_cu));
}

static bool sk_feasible(int cfg_i, int N, int Cin, int Cout, int Tout, int Kw, int stride, int dil, const void* ws, int64_t ws_bytes) {
    if (ws == nullptr || stride != 1) return false;
    const TileCfg& c = kCfgs[cfg_i];
    const int bm = 16 * c.mw * c.ms, bn = 16 * c.nw * c.ns;
    const int64_t tiles = (int64_t)((Cout + bm - 1) / bm) * N * ((Tout + bn - 1) / bn);
    const int64_t steps = (int64_t)(Cin / BK) * Kw, G = sk_ranges(c, stride, Kw, dil);
    if (tiles * steps / G < kSkMinSteps || (tiles * steps + steps) * G >= (1LL << 31)) return false;
    return tiles * (int64_t)sizeof(unsigned) <= (int64_t)kTicketBytes &&
           (int64_t)kTicketBytes + (tiles + G) * bm * bn * (int64_t)sizeof(float) <= ws_bytes;
}

static bool split_feasible(int cfg_i, int splits, int N, int Cin, int Cout, int Tout, int Kw, const void* ws, int64_t ws_bytes) {
    if (splits == 1) return true;
    if (ws == nullptr || (Cin / BK) * Kw < 2 * splits) return false;
    const TileCfg& c = kCfgs[cfg_i];
    const int bm = 16 * c.mw * c.ms, bn = 16 * c.nw * c.ns;
    const size_t tiles = (size_t)((Cout + bm - 1) / bm) * N * ((Tout + bn - 1) / bn);
    return tiles * sizeof(unsigned) <= kTicketBytes && splitk_bytes(cfg_i, splits, N, Cout, Tout) <= (size_t)ws_bytes;
}

// the BatchNorm-backward side of a fused data-gradient launch (w2l_conv1d_dgrad_bnreduce_ws); NULL: plain launch
struct BnBwdArgs {
    const w2l_bnact_t* d;
    int pad_l, pad_r, pad_mode, per;
};

static int igemm_launch(const void* xp, int64_t x_bstride, int64_t x_rows_total, const void* w, void* y, int y_f32,
                        int accumulate, const float* bias, float* stats_partial, int N, int Cin, int Cout, int Tout, int Kw,
                        int stride, int dil, void* splitk_ws, int64_t splitk_ws_bytes, void* stream, const BnBwdArgs* bb);

extern "C" int w2l_conv1d_igemm_ws(const void* xp, int64_t x_bstride, int64_t x_rows_total, const void* w, void* y,
                                   int y_f32, int accumulate, const float* bias, float* stats_partial, int N, int Cin,
                                   int Cout, int Tout, int Kw, int stride, int dil, void* splitk_ws, int64_t splitk_ws_bytes,
                                   void* stream) {
    return igemm_launch(xp, x_bstride, x_rows_total, w, y, y_f32, accumulate, bias, stats_partial, N, Cin, Cout, Tout, Kw,
                        stride, dil, splitk_ws, splitk_ws_bytes, stream, nullptr);
}

// thread-local hand-over of the fused launch's descriptor to the tuner's inner launches (same thread, see g_force_cfg)
static thread_local const BnBwdArgs* g_tune_bb = nullptr;

static int igemm_launch(const void* xp, int64_t x_bstride, int64_t x_rows_total, const void* w, void* y, int y_f32,
                        int accumulate, const float* bias, float* stats_partial, int N, int Cin, int Cout, int Tout, int Kw,
                        int stride, int dil, void* splitk_ws, int64_t splitk_ws_bytes, void* stream, const BnBwdArgs* bb) {
    if (bb == nullptr) bb = g_tune_bb;
    W2L_CHECK_ARG(xp && w && y, "conv1d_igemm: null pointer");
    W2L_CHECK_ARG(N > 0 && Tout > 0 && Kw > 0 && (stride == 1 || stride == 2) && dil > 0,
                  "conv1d_igemm: bad sizes (stride must be 1 or 2)");
    W2L_CHECK_ARG(Cin % 64 == 0 && Cin > 0, "conv1d_igemm: Cin=%d must be a positive multiple of 64", Cin);
    W2L_CHECK_ARG(Cout % 64 == 0 && Cout > 0, "conv1d_igemm: Cout=%d must be a positive multiple of 64", Cout);
    W2L_CHECK_ARG(x_bstride % Cin == 0, "conv1d_igemm: x_bstride must be a multiple of Cin");
    W2L_CHECK_ARG(!(accumulate && !y_f32), "conv1d_igemm: accumulate needs fp32 output");
    IgemmParams p;
    p.x = (const bf16_raw*)xp;
    p.w = (const bf16_raw*)w;
    p.y = y;
    p.bias = bias;
    p.stats = stats_partial;
    p.stats_slots = stats_partial ? g_stats_slots : 0;
    p.x_rows_per_utt = x_bstride / Cin;
    p.x_max_row = x_rows_total - 1;
    p.N = N; p.Cin = Cin; p.Cout = Cout; p.Tout = Tout; p.Kw = Kw; p.stride = stride; p.dil = dil;
    p.y_f32 = y_f32; p.accumulate = accumulate;
    p.descale = 1.f;
    p.descale_dev = nullptr;
    int epi = 0;
    if (bb != nullptr) {
        const w2l_bnact_t* d = bb->d;
        W2L_CHECK_ARG(d && d->y && !d->y_f32 && !d->y2 && stats_partial && !y_f32 && !accumulate && !bias && N == 1 && stride == 1,
                      "conv1d_dgrad_bnreduce: needs a bf16 single-branch layer, a statistics buffer and a flat bf16 output");
        W2L_CHECK_ARG(d->C == Cout && bb->per > 0 && bb->pad_l >= 0 && bb->pad_r >= 0 &&
                      (int64_t)d->N * bb->per <= Tout && bb->per >= bb->pad_l + d->T + bb->pad_r,
                      "conv1d_dgrad_bnreduce: geometry mismatch (C=%d vs %d, per=%d, N=%d, rows=%d)", d->C, Cout, bb->per,
                      d->N, Tout);
        W2L_CHECK_ARG(d->drop_p == 0.f || d->mask, "conv1d_dgrad_bnreduce: dropout needs the recorded mask");
        p.bn_y = (const bf16_raw*)d->y;
        p.bn_scale = d->scale; p.bn_shift = d->shift; p.bn_mean = d->mean; p.bn_invstd = d->invstd;
        p.bn_mask = d->drop_p > 0.f ? d->mask : nullptr;
        p.bn_lens = d->lens;
        p.bn_T = d->T; p.bn_pad_l = bb->pad_l; p.bn_pad_mode = bb->pad_mode; p.bn_per = bb->per;
        p.bn_Tp = bb->pad_l + d->T + bb->pad_r;
        p.bn_act = d->act;
        p.bn_gk = d->drop_p > 0.f ? 1.f / (1.f - d->drop_p) : 1.f;
        epi = 1;
    }
    // the last valid output row must only need rows that exist in the padded buffer
    const int64_t need = (int64_t)(N - 1) * p.x_rows_per_utt + (int64_t)(Tout - 1) * stride + (int64_t)(Kw - 1) * dil;
    W2L_CHECK_ARG(need <= p.x_max_row, "conv1d_igemm: padded input too small (need row %lld, have %lld)",
                  (long long)need, (long long)p.x_max_row);
    // BatchNorm partial statistics are laid out per 128-row column tile (w2l_conv_stat_tiles)
    const int ci = choose_cfg(N, Cin, Cout, Tout, Kw, stride, dil, stats_flag(stats_partial));     // (a fused data gradient
    // shares the table with forward launches: its N = 1, Tout = flat rows shape never coincides with one of theirs)
    W2L_CHECK_ARG(ci >= 0, "conv1d_igemm: no block shape fits LDS (Kw=%d dil=%d stride=%d)", Kw, dil, stride);
    const int pipe = (ci % kBaseCfgs) / kNumCfgs;
    const TileCfg& c = kCfgs[ci % kNumCfgs];
    const int bm = 16 * c.mw * c.ms, bn = 16 * c.nw * c.ns;
    p.tiles_t = (Tout + bn - 1) / bn;
    p.ncols = N * p.tiles_t;
    p.xrows_lds = cfg_xrows(c, stride, Kw, dil);
    // a split-K choice (measured with a workspace) silently degrades to one block per tile when the caller brings none
    int splits = kSplits[ci / kBaseCfgs];
    p.sk_ranges = 0;
    p.sk_total = 0;
    if (splits == 0) {                                   // stream-K (never with the fused epilogue: cfg_feasible)
        if (epi == 0 && sk_feasible(ci % kNumCfgs, N, Cin, Cout, Tout, Kw, stride, dil, splitk_ws, splitk_ws_bytes)) {
            p.sk_ranges = sk_ranges(c, stride, Kw, dil);
            p.sk_total = ((Cout + bm - 1) / bm) * p.ncols * ((Cin / BK) * Kw);
        }
        splits = 1;
    }
    if (!split_feasible(ci % kNumCfgs, splits, N, Cin, Cout, Tout, Kw, splitk_ws, splitk_ws_bytes)) splits = 1;
    p.splits = splits;
    p.tickets = (unsigned*)splitk_ws;
    p.slabs = splitk_ws ? (float*)((char*)splitk_ws + kTicketBytes) : nullptr;
    const int tiles_m = ((Cout + bm - 1) / bm) * splits;        // grid = tiles x splits (launch_cfg multiplies by ncols)
    const size_t lds = 2 * (size_t)bm * ROWB + 2 * (size_t)p.xrows_lds * ROWB;
    hipStream_t st = (hipStream_t)stream;
    switch (ci % kNumCfgs) {
        case 0: return launch_cfg<2, 2, 2, 4>(p, tiles_m, lds, st, pipe, epi);
        case 1: return launch_cfg<2, 2, 3, 4>(p, tiles_m, lds, st, pipe, epi);
        case 2: return launch_cfg<2, 2, 4, 4>(p, tiles_m, lds, st, pipe, epi);
        case 3: return launch_cfg<2, 2, 5, 4>(p, tiles_m, lds, st, pipe, epi);
        case 4: return launch_cfg<4, 2, 3, 4>(p, tiles_m, lds, st, pipe, epi);
        case 5: return launch_cfg<4, 2, 4, 4>(p, tiles_m, lds, st, pipe, epi);
        case 6: return launch_cfg<2, 3, 2, 3>(p, tiles_m, lds, st, pipe, epi);
        case 7: return launch_cfg<2, 3, 3, 3>(p, tiles_m, lds, st, pipe, epi);
        case 8: return launch_cfg<2, 3, 4, 3>(p, tiles_m, lds, st, pipe, epi);
        case 9: return launch_cfg<2, 3, 5, 3>(p, tiles_m, lds, st, pipe, epi);
        case 10: return launch_cfg<2, 4, 2, 4>(p, tiles_m, lds, st, pipe, epi);
        case 11: return launch_cfg<2, 4, 3, 4>(p, tiles_m, lds, st, pipe, epi);
        case 12: return launch_cfg<2, 4, 4, 4>(p, tiles_m, lds, st, pipe, epi);
        case 13: return launch_cfg<2, 4, 5, 4>(p, tiles_m, lds, st, pipe, epi);
        case 14: return launch_cfg<2, 4, 6, 4>(p, tiles_m, lds, st, pipe, epi);
        case 15: return launch_cfg<2, 4, 7, 4>(p, tiles_m, lds, st, pipe, epi);
        case 16: return launch_cfg<2, 4, 8, 4>(p, tiles_m, lds, st, pipe, epi);
        case 17: return launch_cfg<2, 3, 4, 6>(p, tiles_m, lds, st, pipe, epi);
        case 18: return launch_cfg<2, 3, 5, 6>(p, tiles_m, lds, st, pipe, epi);
        case 19: return launch_cfg<2, 3, 6, 6>(p, tiles_m, lds, st, pipe, epi);
        case 20: return launch_cfg<2, 3, 7, 6>(p, tiles_m, lds, st, pipe, epi);
        case 21: return launch_cfg<2, 4, 4, 6>(p, tiles_m, lds, st, pipe, epi);
        case 22: return launch_cfg<2, 4, 5, 6>(p, tiles_m, lds, st, pipe, epi);
        case 23: return launch_cfg<2, 4, 6, 6>(p, tiles_m, lds, st, pipe, epi);
        case 24: return launch_cfg<2, 4, 4, 7>(p, tiles_m, lds, st, pipe, epi);
        default: return launch_cfg<2, 4, 5, 7>(p, tiles_m, lds, st, pipe, epi);
    }
}

extern "C" int w2l_conv1d_dgrad_bnreduce_ws(const void* dy, int64_t dy_rows_total, const void* w_dgr, void* dxp, float* partial,
                                            const w2l_bnact_t* d, int pad_l, int pad_r, int pad_mode, int per, int Cconv_out,
                                            int flat_rows, int Kw, int dil, void* splitk_ws, int64_t splitk_ws_bytes,
                                            void* stream) {
    W2L_CHECK_ARG(d != nullptr, "conv1d_dgrad_bnreduce: null descriptor");
    const BnBwdArgs bb{d, pad_l, pad_r, pad_mode, per};
    return igemm_launch(dy, dy_rows_total * Cconv_out, dy_rows_total, w_dgr, dxp, 0, 0, nullptr, partial, 1, Cconv_out, d->C,
                        flat_rows, Kw, 1, dil, splitk_ws, splitk_ws_bytes, stream, &bb);
}

extern "C" int w2l_conv1d_igemm(const void* xp, int64_t x_bstride, int64_t x_rows_total, const void* w, void* y,
                                int y_f32, int accumulate, const float* bias, float* stats_partial, int N, int Cin,
                                int Cout, int Tout, int Kw, int stride, int dil, void* stream) {
    return w2l_conv1d_igemm_ws(xp, x_bstride, x_rows_total, w, y, y_f32, accumulate, bias, stats_partial, N, Cin, Cout, Tout,
                               Kw, stride, dil, nullptr, 0, stream);
}

// blocks of the stream-K launch configuration idx would make of this problem with a workspace of ws_bytes; 0: the launch
// falls back to one block per tile (no stream-K form of that configuration, ranges too short, workspace too small)
extern "C" int w2l_conv_streamk_ranges(int idx, int N, int Cin, int Cout, int Tout, int Kw, int stride, int dil, int64_t ws_bytes) {
    if (!cfg_feasible(idx, Kw, stride, dil, false) || kSplits[idx / kBaseCfgs] != 0) return 0;
    static const char dummy = 0;
    if (!sk_feasible(idx % kNumCfgs, N, Cin, Cout, Tout, Kw, stride, dil, &dummy, ws_bytes)) return 0;
    return sk_ranges(kCfgs[idx % kNumCfgs], stride, Kw, dil);
}

// the pieces of a stream-K launch of `tiles` tiles x `steps` steps over G ranges, in range order, as the kernel walks them:
// out[7 * i] = range, tile, first step, end step, ranges sharing the tile, this range's place among them, slab id (-1: the
// tile is whole).  Returns the piece count, -1 if cap is too small or the sizes are out of the kernel's 32-bit range.
extern "C" int w2l_conv_streamk_pieces(int tiles, int steps, int G, int* out, int cap) {
    if (tiles <= 0 || steps <= 0 || G <= 0 || ((int64_t)tiles * steps + steps) * G >= (1LL << 31)) return -1;
    const int W = tiles * steps;
    int n = 0;
    for (int r = 0; r < G; ++r) {
        int w_cur = (int)(((int64_t)W * r) / G);
        const int w_end = (int)(((int64_t)W * (r + 1)) / G);
        while (w_cur < w_end) {
            const SkPiece q = sk_piece(W, G, steps, r, w_cur, w_end);
            if (n >= cap) return -1;
            int* o = out + 7 * n++;
            o[0] = r; o[1] = q.tile; o[2] = q.s_begin; o[3] = q.s_end; o[4] = q.nsplit; o[5] = q.split;
            o[6] = q.nsplit > 1 ? (int)(q.slab_base + q.split) : -1;
            w_cur += q.s_end - q.s_begin;
        }
    }
    return n;
}

extern "C" int64_t w2l_conv_splitk_workspace_bytes(int N, int Cout, int Tout) {
    size_t need = 0;
    for (int i = 0; i < kNumCfgs; ++i) {
        const size_t b = splitk_bytes(i, kMaxSplit, N, Cout, Tout);
        if (b > need) need = b;
    }
    return (int64_t)need;
}

// Measure every feasible block shape for this problem on the caller's device and remember the fastest.
// EXPLICITLY synchronising (hipEventSynchronize): call it once per shape during warm-up, never inside a
// captured / latency-critical region.  Only for accumulate == 0 launches (the output is simply rewritten).
extern "C" int w2l_conv1d_igemm_tune_ws(const void* xp, int64_t x_bstride, int64_t x_rows_total, const void* w, void* y,
                                        int y_f32, const float* bias, float* stats_partial, int N, int Cin, int Cout, int Tout,
                                        int Kw, int stride, int dil, int reps, void* splitk_ws, int64_t splitk_ws_bytes,
                                        void* stream) {
    const bool need128 = stats_flag(stats_partial) == 1;
    const ShapeKey key(N, Cin, Cout, Tout, Kw, stride, dil, stats_flag(stats_partial));
    {
        std::lock_guard<std::mutex> lock(g_tuned_mu);
        if (g_tuned.count(key)) return 0;
    }
    hipEvent_t e0, e1;
    W2L_CHECK_HIP(hipEventCreate(&e0));
    W2L_CHECK_HIP(hipEventCreate(&e1));
    hipStream_t st = (hipStream_t)stream;
    const int saved = g_force_cfg;
    if (reps < 1) reps = 1;
    if (splitk_ws) (void)hipMemsetAsync(splitk_ws, 0, kTicketBytes, st);     // tickets start from zero whatever ran before
    // time `n` back-to-back launches of candidate i after one warm-up launch (which also validates it); < 0: it did not run
    auto time_cfg = [&](int i, int n) -> float {
        g_force_cfg = i;
        int rc = w2l_conv1d_igemm_ws(xp, x_bstride, x_rows_total, w, y, y_f32, 0, bias, stats_partial, N, Cin, Cout, Tout, Kw,
                                     stride, dil, splitk_ws, splitk_ws_bytes, stream);
        if (rc != 0) return -1.f;
        (void)hipEventRecord(e0, st);
        for (int r = 0; r < n; ++r)
            w2l_conv1d_igemm_ws(xp, x_bstride, x_rows_total, w, y, y_f32, 0, bias, stats_partial, N, Cin, Cout, Tout, Kw,
                                stride, dil, splitk_ws, splitk_ws_bytes, stream);
        (void)hipEventRecord(e1, st);
        if (hipEventSynchronize(e1) != hipSuccess) return -1.f;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess) return -1.f;
        // a stream-K form has to win by 3 %: level with the best one-block-per-tile form it buys the step nothing (plan files
        // with and without it: 12.79 / 12.82 ms) and moves 40-45 MB more per launch through its slabs
        if (kSplits[i / kBaseCfgs] == 0) ms *= 1.03f;
        return ms;
    };
    std::vector<std::pair<float, int>> timed;
    for (int i = 0; i < kBaseCfgs * kNumSplits; ++i) {
        if (!cfg_feasible(i, Kw, stride, dil, need128)) continue;
        const int ci = i % kNumCfgs, splits = kSplits[i / kBaseCfgs];
        if (splits == 0) {
            // stream-K where one block per tile fills the last round to 85 % or less (and never under the fused epilogue)
            if (g_tune_bb != nullptr || !sk_feasible(ci, N, Cin, Cout, Tout, Kw, stride, dil, splitk_ws, splitk_ws_bytes)) continue;
            const TileCfg& c = kCfgs[ci];
            const int bm = 16 * c.mw * c.ms, bn = 16 * c.nw * c.ns;
            const long slots = sk_ranges(c, stride, Kw, dil);
            const long blocks = (long)((Cout + bm - 1) / bm) * N * ((Tout + bn - 1) / bn);
            const double util = (double)blocks / (double)(((blocks + slots - 1) / slots) * slots);
            if (util > 0.85 || blocks > 4 * slots) continue;       // (at 92 % fill it measured 3 % ahead alone and level in the step,
                                                                   //  for 48 MB more traffic per launch: every finisher's acquire empties its XCD's L2)
        } else if (splits > 1) {
            // split-K is only a candidate where one block per tile leaves CUs idle (a partly filled last round, or fewer
            // tiles than CUs) and where it does not flood the chip with short blocks
            if (!split_feasible(ci, splits, N, Cin, Cout, Tout, Kw, splitk_ws, splitk_ws_bytes)) continue;
            const TileCfg& c = kCfgs[ci];
            const int bm = 16 * c.mw * c.ms, bn = 16 * c.nw * c.ns;
            const size_t lds = 2 * (size_t)bm * ROWB + 2 * (size_t)cfg_xrows(c, stride, Kw, dil) * ROWB;
            int per_cu = (int)((160 * 1024) / lds);
            if (per_cu > 16 / (c.mw * c.nw)) per_cu = 16 / (c.mw * c.nw);
            const long slots = 256L * (per_cu < 1 ? 1 : per_cu);
            const long blocks = (long)((Cout + bm - 1) / bm) * N * ((Tout + bn - 1) / bn);
            const double util = (double)blocks / (double)(((blocks + slots - 1) / slots) * slots);
            if (util > 0.92 || blocks * splits > 6 * slots || (Cin / BK) * Kw / splits < 8) continue;
        }
        const float ms = time_cfg(i, reps);
        if (ms >= 0.f) timed.emplace_back(ms, i);
    }
    // the first pass ranks ~50 candidates on `reps` launches each -- the clock the chip holds drifts over such a sweep by more
    // than the best candidates differ --, so the kFinalists fastest are timed again, interleaved (common.h)
    std::sort(timed.begin(), timed.end());
    int best = timed.empty() ? -1 : timed[0].second;
    const int finalists = timed.size() < kFinalists ? (int)timed.size() : kFinalists;
    if (finalists > 1) {
        float total[kFinalists] = {};
        for (int round = 0; round < kFinalRounds; ++round)
            for (int k = 0; k < finalists; ++k) {
                const float ms = time_cfg(timed[k].second, 4 * reps);
                total[k] += ms >= 0.f ? ms : 1e30f;
            }
        int kb = 0;
        for (int k = 1; k < finalists; ++k)
            if (total[k] < total[kb]) kb = k;
        best = timed[kb].second;
    }
    g_force_cfg = saved;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    W2L_CHECK_ARG(best >= 0, "conv1d_igemm_tune: no feasible block shape");
    std::lock_guard<std::mutex> lock(g_tuned_mu);
    g_tuned[key] = best;
    return 0;
}

// measure-and-pick for a fused data gradient: the candidates run with the fused epilogue (SYNCHRONISING; warm-up only)
extern "C" int w2l_conv1d_dgrad_bnreduce_tune_ws(const void* dy, int64_t dy_rows_total, const void* w_dgr, void* dxp,
                                                 float* partial, const w2l_bnact_t* d, int pad_l, int pad_r, int pad_mode,
                                                 int per, int Cconv_out, int flat_rows, int Kw, int dil, int reps,
                                                 void* splitk_ws, int64_t splitk_ws_bytes, void* stream) {
    W2L_CHECK_ARG(d != nullptr, "conv1d_dgrad_bnreduce_tune: null descriptor");
    const BnBwdArgs bb{d, pad_l, pad_r, pad_mode, per};
    g_tune_bb = &bb;
    const int rc = w2l_conv1d_igemm_tune_ws(dy, dy_rows_total * Cconv_out, dy_rows_total, w_dgr, dxp, 0, nullptr, partial, 1,
                                            Cconv_out, d->C, flat_rows, Kw, 1, dil, reps, splitk_ws, splitk_ws_bytes, stream);
    g_tune_bb = nullptr;
    return rc;
}

extern "C" int w2l_conv1d_igemm_tune(const void* xp, int64_t x_bstride, int64_t x_rows_total, const void* w, void* y,
                                     int y_f32, const float* bias, float* stats_partial, int N, int Cin, int Cout, int Tout,
                                     int Kw, int stride, int dil, int reps, void* stream) {
    return w2l_conv1d_igemm_tune_ws(xp, x_bstride, x_rows_total, w, y, y_f32, bias, stats_partial, N, Cin, Cout, Tout, Kw,
                                    stride, dil, reps, nullptr, 0, stream);
}

// ---- e4m3 operands (BASELINE config 5: fp8 MFMA) -------------------------------------------------------------------
// measured choices of the e4m3 kernel: shape -> index into kF8Cfgs
static std::map<ShapeKey, int> g_tuned_f8;
static thread_local int g_force_f8 = -1;
// testing / profiling hook (per calling thread): pin the e4m3 kernel's block shape (index into kF8Cfgs), -1 = automatic
extern "C" void w2l_conv_force_fp8_config(int idx) { g_force_f8 = idx; }

static bool f8_feasible(int k, int Kw, int dil, bool need_bn128) {
    if (k < 0 || k >= kNumF8Cfgs) return false;
    const TileCfg& c = kCfgs[kF8Cfgs[k]];
    if (need_bn128 && ((16 * c.nw * c.ns) % 128 != 0 || 16 * c.nw * c.ns > 256)) return false;
    const size_t lds = 2 * (size_t)(16 * c.mw * c.ms) * ROWB + 2 * (size_t)cfg_xrows(c, 1, Kw, dil) * ROWB;
    return lds <= 160 * 1024;
}

static int choose_f8(int N, int Cin, int Cout, int Tout, int Kw, int dil, bool need_bn128) {
    if (g_force_f8 >= 0) return f8_feasible(g_force_f8, Kw, dil, need_bn128) ? g_force_f8 : -1;
    {
        std::lock_guard<std::mutex> lock(g_tuned_mu);
        auto it = g_tuned_f8.find(ShapeKey(N, Cin, Cout, Tout, Kw, 1, dil, need_bn128 ? 1 : 0));
        if (it != g_tuned_f8.end()) return it->second;
    }
    int best = -1;
    double best_cost = 1e30;
    for (int k = 0; k < kNumF8Cfgs; ++k) {            // cost model: whole rounds of resident blocks, larger tiles preferred
        if (!f8_feasible(k, Kw, dil, need_bn128)) continue;
        const TileCfg& c = kCfgs[kF8Cfgs[k]];
        const int bm = 16 * c.mw * c.ms, bn = 16 * c.nw * c.ns, waves = c.mw * c.nw;
        const size_t lds = 2 * (size_t)bm * ROWB + 2 * (size_t)cfg_xrows(c, 1, Kw, dil) * ROWB;
        int per_cu = (int)((160 * 1024) / lds);
        if (per_cu > 16 / waves) per_cu = 16 / waves;
        if (per_cu < 1) continue;
        const long blocks = (long)((Cout + bm - 1) / bm) * N * ((Tout + bn - 1) / bn);
        const long rounds = (blocks + 256L * per_cu - 1) / (256L * per_cu);
        const double cost = (double)rounds * bm * bn * per_cu / c.eff * (1.0 + 1e-3 * k);
        if (cost < best_cost) { best_cost = cost; best = k; }
    }
    return best;
}

extern "C" int w2l_conv1d_igemm_fp8(const void* xq, int64_t x_bstride, int64_t x_rows_total, const void* wq, void* y, int y_f32,
                                    float descale, const float* descale_dev, const float* bias, float* stats_partial, int N,
                                    int Cin, int Cout, int Tout, int Kw, int dil, void* stream) {
    W2L_CHECK_ARG(xq && wq && y, "conv1d_igemm_fp8: null pointer");
    W2L_CHECK_ARG(N > 0 && Tout > 0 && Kw > 0 && dil > 0, "conv1d_igemm_fp8: bad sizes");
    W2L_CHECK_ARG(Cin % 128 == 0 && Cin > 0, "conv1d_igemm_fp8: Cin=%d must be a positive multiple of 128", Cin);
    W2L_CHECK_ARG(Cout % 64 == 0 && Cout > 0, "conv1d_igemm_fp8: Cout=%d must be a positive multiple of 64", Cout);
    W2L_CHECK_ARG(x_bstride % Cin == 0, "conv1d_igemm_fp8: x_bstride must be a multiple of Cin");
    W2L_CHECK_ARG(descale > 0.f, "conv1d_igemm_fp8: descale must be positive");
    IgemmParams p;
    p.x = (const bf16_raw*)xq;
    p.w = (const bf16_raw*)wq;
    p.y = y;
    p.bias = bias;
    p.stats = stats_partial;
    p.stats_slots = stats_partial ? g_stats_slots : 0;
    p.x_rows_per_utt = x_bstride / Cin;
    p.x_max_row = x_rows_total - 1;
    p.N = N; p.Cin = Cin; p.Cout = Cout; p.Tout = Tout; p.Kw = Kw; p.stride = 1; p.dil = dil;
    p.y_f32 = y_f32; p.accumulate = 0;
    p.splits = 1; p.slabs = nullptr; p.tickets = nullptr;
    p.sk_ranges = 0; p.sk_total = 0;
    p.descale = descale;
    p.descale_dev = descale_dev;
    const int64_t need = (int64_t)(N - 1) * p.x_rows_per_utt + (int64_t)(Tout - 1) + (int64_t)(Kw - 1) * dil;
    W2L_CHECK_ARG(need <= p.x_max_row, "conv1d_igemm_fp8: padded input too small (need row %lld, have %lld)",
                  (long long)need, (long long)p.x_max_row);
    const int k = choose_f8(N, Cin, Cout, Tout, Kw, dil, stats_partial != nullptr);
    W2L_CHECK_ARG(k >= 0, "conv1d_igemm_fp8: no block shape fits LDS (Kw=%d dil=%d)", Kw, dil);
    const TileCfg& c = kCfgs[kF8Cfgs[k]];
    const int bm = 16 * c.mw * c.ms, bn = 16 * c.nw * c.ns;
    p.tiles_t = (Tout + bn - 1) / bn;
    p.ncols = N * p.tiles_t;
    p.xrows_lds = cfg_xrows(c, 1, Kw, dil);
    const int tiles_m = (Cout + bm - 1) / bm;
    const size_t lds = 2 * (size_t)bm * ROWB + 2 * (size_t)p.xrows_lds * ROWB;
    hipStream_t st = (hipStream_t)stream;
    switch (kF8Cfgs[k]) {                       // block shapes of kCfgs
        case 2: return launch_f8<2, 2, 4, 4>(p, tiles_m, lds, st);
        case 5: return launch_f8<4, 2, 4, 4>(p, tiles_m, lds, st);
        case 12: return launch_f8<2, 4, 4, 4>(p, tiles_m, lds, st);
        case 14: return launch_f8<2, 4, 6, 4>(p, tiles_m, lds, st);
        case 16: return launch_f8<2, 4, 8, 4>(p, tiles_m, lds, st);
        case 1: return launch_f8<2, 2, 3, 4>(p, tiles_m, lds, st);
        case 3: return launch_f8<2, 2, 5, 4>(p, tiles_m, lds, st);
        case 8: return launch_f8<2, 3, 4, 3>(p, tiles_m, lds, st);
        case 9: return launch_f8<2, 3, 5, 3>(p, tiles_m, lds, st);
        case 11: return launch_f8<2, 4, 3, 4>(p, tiles_m, lds, st);
        case 13: return launch_f8<2, 4, 5, 4>(p, tiles_m, lds, st);
        case 18: return launch_f8<2, 3, 5, 6>(p, tiles_m, lds, st);
        case 19: return launch_f8<2, 3, 6, 6>(p, tiles_m, lds, st);
        case 21: return launch_f8<2, 4, 4, 6>(p, tiles_m, lds, st);
        default: return launch_f8<2, 4, 4, 7>(p, tiles_m, lds, st);
    }
}

// measure the e4m3 block shapes for this problem and remember the fastest (SYNCHRONISING: warm-up only)
extern "C" int w2l_conv1d_igemm_fp8_tune(const void* xq, int64_t x_bstride, int64_t x_rows_total, const void* wq, void* y,
                                         int y_f32, const float* bias, float* stats_partial, int N, int Cin, int Cout, int Tout,
                                         int Kw, int dil, int reps, void* stream) {
    const bool need128 = stats_partial != nullptr;
    const ShapeKey key(N, Cin, Cout, Tout, Kw, 1, dil, need128 ? 1 : 0);
    {
        std::lock_guard<std::mutex> lock(g_tuned_mu);
        if (g_tuned_f8.count(key)) return 0;
    }
    hipEvent_t e0, e1;
    W2L_CHECK_HIP(hipEventCreate(&e0));
    W2L_CHECK_HIP(hipEventCreate(&e1));
    hipStream_t st = (hipStream_t)stream;
    int best = -1;
    float best_ms = 1e30f;
    const int saved = g_force_f8;
    if (reps < 1) reps = 1;
    for (int k = 0; k < kNumF8Cfgs; ++k) {
        if (!f8_feasible(k, Kw, dil, need128)) continue;
        g_force_f8 = k;
        if (w2l_conv1d_igemm_fp8(xq, x_bstride, x_rows_total, wq, y, y_f32, 1.f, nullptr, bias, stats_partial, N, Cin, Cout, Tout, Kw,
                                 dil, stream) != 0)
            continue;
        (void)hipEventRecord(e0, st);
        for (int r = 0; r < reps; ++r)
            w2l_conv1d_igemm_fp8(xq, x_bstride, x_rows_total, wq, y, y_f32, 1.f, nullptr, bias, stats_partial, N, Cin, Cout, Tout, Kw,
                                 dil, stream);
        (void)hipEventRecord(e1, st);
        if (hipEventSynchronize(e1) != hipSuccess) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess) continue;
        if (ms < best_ms) { best_ms = ms; best = k; }
    }
    g_force_f8 = saved;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    W2L_CHECK_ARG(best >= 0, "conv1d_igemm_fp8_tune: no feasible block shape");
    std::lock_guard<std::mutex> lock(g_tuned_mu);
    g_tuned_f8[key] = best;
    return 0;
}

// Tuning-cache (de)serialisation used by w2l_tune_save / w2l_tune_load (runtime.hip).
void w2l_igemm_tune_dump(FILE* f) {
    std::lock_guard<std::mutex> lock(g_tuned_mu);
    for (const auto& kv : g_tuned) {
        const ShapeKey& k = kv.first;
        fprintf(f, "igemm %d %d %d %d %d %d %d %d %d\n", std::get<0>(k), std::get<1>(k), std::get<2>(k), std::get<3>(k),
                std::get<4>(k), std::get<5>(k), std::get<6>(k), std::get<7>(k), kv.second);
    }
}

// the e4m3 kernel's choices: same key (stride always 1), value = index into kF8Cfgs
void w2l_igemm_fp8_tune_dump(FILE* f) {
    std::lock_guard<std::mutex> lock(g_tuned_mu);
    for (const auto& kv : g_tuned_f8) {
        const ShapeKey& k = kv.first;
        fprintf(f, "igemmf8 %d %d %d %d %d %d %d %d %d\n", std::get<0>(k), std::get<1>(k), std::get<2>(k), std::get<3>(k),
                std::get<4>(k), std::get<5>(k), std::get<6>(k), std::get<7>(k), kv.second);
    }
}

bool w2l_igemm_fp8_tune_put(const int* v) {      // v[0..7] = key, v[8] = index into kF8Cfgs
    if (v[5] != 1 || v[0] < 1 || v[3] < 1 || !f8_feasible(v[8], v[4], v[6], v[7] != 0)) return false;
    std::lock_guard<std::mutex> lock(g_tuned_mu);
    g_tuned_f8[ShapeKey(v[0], v[1], v[2], v[3], v[4], 1, v[6], v[7] != 0 ? 1 : 0)] = v[8];
    return true;
}

bool w2l_igemm_tune_put(const int* v) {          // v[0..7] = key, v[8] = block-shape index
    if (v[7] < 0 || v[7] > 2 || !cfg_feasible(v[8], v[4], v[5], v[6], v[7] == 1)) return false;
    std::lock_guard<std::mutex> lock(g_tuned_mu);
    g_tuned[ShapeKey(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7])] = v[8];
    return true;
}
