// The weight-gradient kernel body shared by conv_wgrad.hip (the library's ordinary kernels) and conv_wgrad3_dev.hip (the
// three-tap kernel with its 192 accumulator registers in AGPRs, compiled on its own: see that file and the Makefile).
#pragma once
#include "common.h"
#include <type_traits>

#ifndef W2L_AG_NOP
#define W2L_AG_NOP 1
#endif
#if W2L_AG_NOP
#define W2L_AG_PRE "s_nop 1\n\t"
#else
#define W2L_AG_PRE ""
#endif
#ifndef W2L_AG_ADBUF
#define W2L_AG_ADBUF 1
#endif

namespace w2l_wgrad {

constexpr int BM = 128;       // co per block
constexpr int BNC = 128;      // ci per block
constexpr int BT = 64;        // t rows per K step
constexpr int ROWB = 256;     // bytes per LDS row (128 bf16)

#define W2L_WGRAD_MAX_RANGES 1024
constexpr int KWB_DEFAULT = W2L_DIAG_WGRAD_TAPS;   // taps per 4-wave block: 2 (common.h; a diagnostic build may probe 1)

#define W2L_WGRAD_MAX_LAYERS 8
// One layer of a launch.  A launch may cover SEVERAL layers (a "group": same N, Tout, stride 1, dilation and block form; any
// Cin / Cout / Kw): their [128 co x 128 ci] x tap-group tiles form ONE pool of equal-shaped work items, so layers whose own
// tile count fills a fraction of the 512 block slots (640 -> 640, k21: 175 three-tap tiles) fill whole rounds together --
// no split, no partial tiles, no atomics, no zero-filled dw.
struct WgradLayer {
    const bf16_raw* dy;
    const bf16_raw* x;
    float* dw;
    int64_t dy_rows_per_utt;  // dy_bstride / Cout
    int64_t x_rows_per_utt;
    int64_t x_max_row;
    int Cin, Cout, Kw;
    int tiles_m, tiles_n, kgroups;
    int tile0;                // first tile of this layer in the launch's pool
    int pad_;
};

struct WgradParams {
    WgradLayer layers[W2L_WGRAD_MAX_LAYERS];
    int nlayers, tiles_total;
    int N, Tout, stride, dil;
    int tsteps, total_steps, steps_per_split, atomic;
    int order;                 // block order inside a split: 1 = tap group fastest, 0 = co tile fastest
    int streamk;               // 1: `gridDim.x` persistent blocks share the (tile, K step) space in equal contiguous ranges
    int xrows_lds;
    // split-K through a workspace (w2l_conv1d_wgrad_ws): partial tiles go to fp32 slabs, the block that draws a tile's last
    // ticket sums them in split order and writes dw with plain stores -- no atomics, no zero fill, bit-reproducible
    int splits, accumulate;
    float* slabs;              // [tiles][splits][KWB*128*128]; NULL: fp32 atomics into a zero-filled dw
    unsigned* tickets;         // [tiles], zero between launches
    // "dealt" stream-K (needs the workspace): the (tile, K step) space is cut into `dealt` equal ranges, one per resident
    // block slot.  A range crosses at most one tile boundary (tiles <= dealt), i.e. it is one or two SEGMENTS, and every
    // segment is a block of its own -- the kernel stays the one-segment classic kernel, no persistent loop and none of its
    // live scalars.  Blocks 0 .. dealt-1 take the first segment of their range and start together; blocks dealt .. take the
    // second segments, LONGEST FIRST (dealt_perm, sorted by the launcher per XCD: block G + j runs on the XCD of blocks j % 8
    // and takes a second segment of THAT XCD's ranges, 0xffff = none left): the hardware hands the next block to the slot that
    // frees first, which is the one whose first segment was shortest, so every slot ends up with about one range of work
    // and the chip stays full whatever the tile count.  A segment that covers its tile alone stores it; every other segment
    // writes its partial tile to slab (range + tile) and draws the tile's ticket; the last arriver sums the tile's slabs
    // in range order (bit-reproducible, no atomics, no zero-filled dw).
    int dealt, dealt_b;
    unsigned short dealt_perm[W2L_WGRAD_MAX_RANGES];
};

// One segment of the dealt stream-K decomposition (WgradParams::dealt; shared by the kernel, the launcher and the host-side
// test hook w2l_wgrad_dealt_segments): range r of G over the W = tiles * S (tile, K step) space is [W r / G, W (r + 1) / G);
// its first segment ends at the next tile boundary, its second one (if any) starts there.  nsplit = how many segments (of
// consecutive ranges r_lo .. r_hi) touch the segment's tile, split = this one's place among them, slab0 = r_lo + tile: the
// segment of range r in tile t owns slab r + t (ranges and tile boundaries interleave, so r + t counts segments).
// 32-bit arithmetic: the launcher checks W * (G + 2) < 2^31.
struct DealtSeg { int w, w_end, nsplit, split, slab0; };
__host__ __device__ inline DealtSeg dealt_segment(unsigned tiles, unsigned S, unsigned G, unsigned r, bool second) {
    const unsigned W = tiles * S;
    const unsigned w0 = W * r / G, w1 = W * (r + 1u) / G;
    const unsigned bound = (w0 / S + 1u) * S;
    DealtSeg o;
    o.w = (int)(second ? bound : w0);
    o.w_end = (int)(second ? w1 : (w1 < bound ? w1 : bound));
    // the ranges that touch tile t: the last one that starts at or before the tile's first step .. the last one that starts
    // before its end
    const unsigned t = (unsigned)o.w / S;
    const unsigned r_lo = ((t * S + 1u) * G + W - 1u) / W - 1u;
    const unsigned r_hi = ((t + 1u) * S * G + W - 1u) / W - 1u;
    o.nsplit = (int)(r_hi - r_lo + 1u);
    o.split = (int)(r - r_lo);
    o.slab0 = (int)(r_lo + t);
    return o;
}

// 16 B per lane global -> LDS (LDS-DMA): lane l lands at lds_wave_base + 16*l.  Written as inline asm on purpose: with
// the builtin the compiler (which cannot prove that the DMA destination and the buffer being read are different
// halves of the double buffer) puts s_waitcnt vmcnt(0) in front of the ds_read_b64_tr_b16 that FOLLOW the prefetch,
// i.e. it waits for the loads just issued and the prefetch hides nothing.  The loop's own vmcnt(0) + barrier at the
// top of every step is what orders the DMA against its readers.
__device__ __forceinline__ void glds16(const void* gbase_uniform, unsigned voff, unsigned lds_wave_base) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :
                 : "v"(voff), "s"(gbase_uniform), "s"(lds_wave_base)
                 : "memory");
}

__device__ __forceinline__ uint64_t uni64(uint64_t v) {     // a wave-uniform 64-bit value, provably so
    return ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}

__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p;
}

// swizzle key of an LDS row: spreads the 8 rows one tr-read half-wave touches
// (r..r+3 and r+8..r+11) over the 8 aligned 32-byte column pairs of a bank row.
__device__ __forceinline__ int row_key(int r) { return (r & 3) | (((r >> 3) & 1) << 2); }

__device__ __forceinline__ bf16x4 tr_read(unsigned lds_byte_addr) {      // 32-bit LDS address
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(size_t)lds_byte_addr);
}

// S1: the conv stride is 1 (every layer but a model's first): the row offsets of the k-substeps become ds_read immediates
// instead of per-step VALU adds (the K loop is issue-bound).
// SK: the stream-K decomposition (a block walks up to three tile segments); compiled separately so that the classic
// one-segment kernel keeps its scalar-register budget (the segment loop's extra live scalars cost it 16 v_readlane per step).
// TG: tap groups per block.  TG = 2: ONE 8-wave block per CU instead of two 4-wave blocks; waves 0-3 own taps kw0, kw0+1 and
// waves 4-7 taps kw0+2, kw0+3 of the same [128 co x 128 ci] tile, all reading ONE dy tile and ONE x window (3*d rows longer).
// Same waves per CU, same registers and LDS reads per wave, but half the LDS-DMA instructions per MFMA: timing-only builds
// price the K loop's DMA at 27 % of the kernel (1075 -> 1366 TFLOP/s without it; without the transposing reads 1250;
// neither 1788) -- a DMA piece costs the issuing SIMD 60-185 cycles beside MFMAs, and each wave issues 8 per 64 MFMAs.
// M32: the fragments feed v_mfma_f32_32x32x16_bf16 instead of v_mfma_f32_16x16x32_bf16 -- the same FLOPs, LDS bytes and
// transposing reads per step from half as many MFMA instructions.  An MFMA holds its SIMD's issue port for 8 cycles whatever
// its shape (MI355X_MICROARCH.md, 'vector-instruction ISSUE cost'), and this K loop is issue-bound (per 16x16x32 MFMA: 0.75
// transposing reads, 0.125 LDS-DMA pieces at 60-185 cycles each, ~1.4 VALU and ~1.6 SALU instructions, times two waves per
// SIMD, against 16 cycles of matrix pipe), so 32 instead of 64 MFMAs per wave and step give 256 issue cycles back.  The
// LDS swizzle key becomes (row & 3) << 1: a half-wave of the 32x32 operand read touches FOUR rows in TWO adjacent 16-channel
// blocks (the 16x16 operand: eight rows, one block), and a key that only depends on row & 3 also makes every k-substep
// offset (multiples of 4 rows) an immediate: 6 address registers instead of 20, and 24 fewer fragment registers.
// AGACC: the 32x32x16 accumulators live in AGPRs -- every MFMA is an inline-asm statement with an "a"-constrained accumulator
// operand -- so that THREE taps per block (192 accumulator registers per lane) fit beside the fragments and addresses of two
// waves per SIMD.  hipcc splits a 256-register budget 128 / 128 between the two files unless the kernel carries the
// "amdgpu-agpr-alloc" function attribute, which no source-level attribute sets: the Makefile compiles conv_wgrad3_dev.hip to
// LLVM IR, adds the attribute and builds a code object that the library embeds and loads (conv_wgrad.hip: wgrad3_function).
template <int KWB, bool S1, bool SK, int TG = 1, bool M32 = false, bool AGACC = false>
__device__ __forceinline__ void conv_wgrad_body(const WgradParams& p, char* smem) {
    static_assert(!M32 || (S1 && !SK && (TG == 1 || AGACC)), "the 32x32x16 variant is built for stride 1, classic split-K, one tap group (the AGPR form: one or two)");
    static_assert(!AGACC || M32, "AGPR accumulators are built for the 32x32x16 form");
    auto rkey = [](int r) { return M32 ? ((r & 3) << 1) : row_key(r); };
    constexpr int NWV = 4 * TG;                    // waves per block
    constexpr int KWBLK = KWB * TG;                // taps per block
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = TG == 1 ? wave_all : wave_all & 3;   // place in the 2 x 2 arrangement over the tile
    const int tg = TG == 1 ? 0 : wave_all >> 2;           // tap group of this wave (a constant for the 4-wave kernel)
    const int wm = wave >> 1, wn = wave & 1;

    // ---- work decomposition.  The launch's work is the (tile, K step) space, tile-major; a block owns ONE contiguous range
    // [w, w_end) of it:
    //  * classic: grid = (tiles, splits); block (tile, split) owns steps [split*steps_per_split, ...) of its tile -- one
    //    segment.  Whole rounds of the 512 resident blocks need tiles*splits to be a multiple of 512, which the big layers
    //    miss (896x896x29: 735 tiles = 1.44 rounds at split 1; split 2 pays 186 MB of fp32 atomics for 93 MB of dw);
    //  * stream-K (p.streamk): gridDim.x persistent blocks (one per resident slot) cut the whole space into equal ranges.  A
    //    range spans at most three tiles; a tile a block covers completely is written with plain stores, the pieces of a
    //    shared tile are added atomically (into the zero-filled dw), so the chip stays full for the whole launch and only
    //    one piece per block -- not every tile times the split factor -- goes through atomics.
    // The XCD remap runs on the linear block id because that is what the hardware deals round-robin to the XCDs: the ~64
    // blocks resident on one XCD are then consecutive tiles.  order 1 (tap group fastest): all tap groups of a few co tiles
    // of one ci tile, i.e. the same few dy tiles and overlapping x windows stream through that XCD's L2; order 0 (co tile
    // fastest, then ci tile, then tap group): all (co, ci) tiles of one or two tap groups.  Measured per shape.
    const int lin = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int S = p.total_steps;
    int w, w_end;                                  // (tiles * S < 2^31 is checked by the launcher)
    int split = 0, nsplit = p.splits, slab0 = 0;   // this block's place among the nsplit partial tiles of its tile; the tile's first slab
    if constexpr (SK) {
        const int64_t W = (int64_t)p.tiles_total * S;
        w = (int)(W * lin / gridDim.x);
        w_end = (int)(W * (lin + 1) / gridDim.x);
    } else {
        if (p.dealt) {
            const unsigned G = (unsigned)p.dealt, b = blockIdx.x;
            const bool second = b >= G;
            const unsigned r = __builtin_amdgcn_readfirstlane(second ? (unsigned)p.dealt_perm[b - G] : (unsigned)xcd_remap((int)b, (int)G));
            if (r == 0xffffu) return;                  // a padding position of the per-XCD deal (whole block, before any barrier)
            const DealtSeg sg = dealt_segment((unsigned)p.tiles_total, (unsigned)S, G, r, second);
            // (the divisions run on the vector unit: hand the results back to scalar registers, or every tile coordinate and
            // staging address of the K loop becomes a vector value -- measured: 15 % of the kernel)
            w = __builtin_amdgcn_readfirstlane(sg.w);
            w_end = __builtin_amdgcn_readfirstlane(sg.w_end);
            nsplit = __builtin_amdgcn_readfirstlane(sg.nsplit);
            split = __builtin_amdgcn_readfirstlane(sg.split);
            slab0 = __builtin_amdgcn_readfirstlane(sg.slab0);
        } else {
            split = lin / gridDim.x;
            const int tl = lin - split * gridDim.x;
            w = tl * S + split * p.steps_per_split;
            w_end = w + p.steps_per_split;
            if (w_end > (tl + 1) * S) w_end = (tl + 1) * S;
            slab0 = tl * p.splits;
        }
    }
    int tile_id = 0, kw0 = 0, ntaps = KWB, m0 = 0, c0 = 0, shift = 0;
    const int s = S1 ? 1 : p.stride, d = p.dil;
    const int xrows = p.xrows_lds;                 // (BT-1)*s + (KWBLK-1)*d + 1 rounded up to 4

    char* abuf0 = smem;                            // dy tile  [BT][128 co]
    char* abuf1 = smem + BT * ROWB;
    char* bbuf0 = smem + 2 * BT * ROWB;            // x window [xrows][128 ci]
    char* bbuf1 = bbuf0 + xrows * ROWB;

    // ---- staging: one wave-instruction fills 4 rows of 256 B by LDS-DMA.  The K loop is issue-bound, so
    // per-lane offsets are computed once; a step only adds wave-uniform bases (scalar) to them.
    const int srow = lane >> 4;                    // 0..3
    const int schunk = lane & 15;                  // LDS 16-byte chunk
    constexpr int AG = BT / 4 / NWV;               // four-row groups of the dy tile per wave
    unsigned a_voff[AG];                           // byte offset of this lane's 16 B inside the step's dy tile
    unsigned x_max_row = 0;                                    // (of the tile's layer; rows * Cin * 2 < 2^32 is checked by the launcher)
    WgradLayer L = p.layers[0];                                // the layer of the current tile (set_tile)
    // the x window likewise (stride 1, <= 20 four-row groups): per-lane offsets inside the window, computed once per tile
    constexpr int XG = TG == 1 ? 5 : 3;
    const bool x_fast = S1 && (xrows >> 2) <= NWV * XG;
    unsigned x_voff[XG];
    // tile id -> (tap group, co tile, ci tile) and everything that depends on them
    auto set_tile = [&](int tile) {
        tile = __builtin_amdgcn_readfirstlane(tile);   // (wave-uniform, but a stream-K range start comes out of a vector division)
        tile_id = tile;                                // (of the launch's pool: tickets and slabs are numbered by it)
        if (p.nlayers > 1) {                           // a group launch: which layer's tile is this?  (wave-uniform: scalar loads)
            int li = 0;
            for (int i = 1; i < p.nlayers; ++i) li = tile >= p.layers[i].tile0 ? i : li;
            const WgradLayer& g = p.layers[li];
            // (field by field through readfirstlane: the values ARE uniform, and the LDS-DMA's base pointer must be a scalar operand)
            L.dy = reinterpret_cast<const bf16_raw*>(uni64(reinterpret_cast<uint64_t>(g.dy)));
            L.x = reinterpret_cast<const bf16_raw*>(uni64(reinterpret_cast<uint64_t>(g.x)));
            L.dw = reinterpret_cast<float*>(uni64(reinterpret_cast<uint64_t>(g.dw)));
            L.dy_rows_per_utt = (int64_t)uni64((uint64_t)g.dy_rows_per_utt);
            L.x_rows_per_utt = (int64_t)uni64((uint64_t)g.x_rows_per_utt);
            L.x_max_row = (int64_t)uni64((uint64_t)g.x_max_row);
            L.Cin = __builtin_amdgcn_readfirstlane(g.Cin);
            L.Cout = __builtin_amdgcn_readfirstlane(g.Cout);
            L.Kw = __builtin_amdgcn_readfirstlane(g.Kw);
            L.tiles_m = __builtin_amdgcn_readfirstlane(g.tiles_m);
            L.tiles_n = __builtin_amdgcn_readfirstlane(g.tiles_n);
            L.kgroups = __builtin_amdgcn_readfirstlane(g.kgroups);
            tile -= __builtin_amdgcn_readfirstlane(g.tile0);
        }
        x_max_row = (unsigned)L.x_max_row;
        int tm, tn;
        if (p.order) {
            kw0 = (tile % L.kgroups) * KWBLK;          // first tap of this block's group
            tile /= L.kgroups;
            tm = tile % L.tiles_m;
            tn = tile / L.tiles_m;
        } else {
            tm = tile % L.tiles_m;
            tile /= L.tiles_m;
            tn = tile % L.tiles_n;
            kw0 = (tile / L.tiles_n) * KWBLK;
        }
        shift = kw0 * d;                               // the block's x window starts at its first tap
        kw0 += tg * KWB;                               // from here on: the first tap of THIS WAVE
        ntaps = L.Kw - kw0;                            // live taps of this wave: 0 (TG > 1 only) .. KWB
        ntaps = ntaps < 0 ? 0 : (ntaps < KWB ? ntaps : KWB);
        m0 = tm * BM;
        c0 = tn * BNC;
#pragma unroll
        for (int i = 0; i < AG; ++i) {
            const int r = (wave_all * AG + i) * 4 + srow;
            const int g = schunk ^ (rkey(r) << 1);
            int co = m0 + g * 8;
            co = co < L.Cout ? co : L.Cout - 8;
            a_voff[i] = ((unsigned)r * (unsigned)L.Cout + (unsigned)co) * 2u;
        }
#pragma unroll
        for (int i = 0; i < XG; ++i) {
            const int r = (wave_all + NWV * i) * 4 + srow;
            const int g = schunk ^ (rkey(r) << 1);
            int ci = c0 + g * 8;
            ci = ci < L.Cin ? ci : L.Cin - 8;
            x_voff[i] = ((unsigned)r * (unsigned)L.Cin + (unsigned)ci) * 2u;
        }
    };
    auto stage = [&](char* adst, char* bdst, int n, int ts) {
        const int t0 = ts * BT;
        // dy rows t0..t0+63 (rows >= Tout are zero by contract)
        const char* abase = reinterpret_cast<const char*>(L.dy) + ((int64_t)n * L.dy_rows_per_utt + t0) * L.Cout * 2;
        const unsigned a_lds = __builtin_amdgcn_readfirstlane(lds_addr(adst) + wave_all * AG * 1024);
        // (the 32x32x16 form's swizzle key depends on row & 3 only, i.e. on the lane: the pieces of a wave differ by a wave-uniform
        // number of rows, which goes into the scalar base -- one offset register for the dy tile, one for the x window)
#pragma unroll
        for (int i = 0; i < AG; ++i) {
            if constexpr (M32) glds16(abase + (int64_t)i * 4 * L.Cout * 2, a_voff[0], a_lds + i * 1024);
            else glds16(abase, a_voff[i], a_lds + i * 1024);
        }
        const unsigned b_lds = __builtin_amdgcn_readfirstlane(lds_addr(bdst));
        const unsigned xrow0 = (unsigned)(n * L.x_rows_per_utt) + (unsigned)(t0 * s + shift);
        const int ngrp = xrows >> 2;
        if (x_fast && xrow0 + (unsigned)xrows - 1u <= x_max_row) {     // (wave-uniform) the whole window exists: no clamping
            const char* xbase = reinterpret_cast<const char*>(L.x) + (uint64_t)xrow0 * (unsigned)L.Cin * 2u;
#pragma unroll
            for (int i = 0; i < XG; ++i)
                if (wave_all + NWV * i < ngrp) {
                    if constexpr (M32) glds16(xbase + (int64_t)i * NWV * 4 * L.Cin * 2, x_voff[0], b_lds + (wave_all + NWV * i) * 1024);
                    else glds16(xbase, x_voff[i], b_lds + (wave_all + NWV * i) * 1024);
                }
            return;
        }
        for (int grp = wave_all; grp < ngrp; grp += NWV) {
            const int r = grp * 4 + srow;
            const int g = schunk ^ (rkey(r) << 1);
            unsigned fr = xrow0 + (unsigned)r;
            fr = fr < x_max_row ? fr : x_max_row;
            int ci = c0 + g * 8;
            ci = ci < L.Cin ? ci : L.Cin - 8;
            const unsigned voff = (fr * (unsigned)L.Cin + (unsigned)ci) * 2u;
            glds16(L.x, voff, b_lds + grp * 1024);
        }
    };

    f32x4 acc[KWB][4][4];                          // 16x16x32 form: [tap][16 co][16 ci] tiles of the wave's 64 x 64
    f32x16 acc32[KWB][2][2];                       // 32x32x16 form (M32): [tap][32 co][32 ci] tiles; only one of the two is live
    // the wave's accumulators of tap tp as 16 chunks of four registers (the unit the split-K slabs are stored in)
    auto chunk_get = [&](int tp, int c) -> f32x4 {
        if constexpr (M32) {
            const f32x16& v = acc32[tp][c >> 3][(c >> 2) & 1];
            const int o = (c & 3) * 4;
            return f32x4{v[o], v[o + 1], v[o + 2], v[o + 3]};
        } else {
            return acc[tp][c >> 2][c & 3];
        }
    };
    auto chunk_set = [&](int tp, int c, f32x4 x) {
        if constexpr (M32) {
            f32x16& v = acc32[tp][c >> 3][(c >> 2) & 1];
            const int o = (c & 3) * 4;
            v[o] = x[0]; v[o + 1] = x[1]; v[o + 2] = x[2]; v[o + 3] = x[3];
        } else {
            acc[tp][c >> 2][c & 3] = x;
        }
    };
    int step_begin = 0, step_end = 0;              // this segment's K steps inside its tile

    // ---- tr-read lane geometry: within a 16-lane group, lane 4q+pp supplies row q, columns 4pp..4pp+3.
    // LDS row of fragment (ks, h) = ks*32 + h*4 + (kgrp*8 + q) [+ tap*d for the x window]; the swizzle key only
    // depends on the lane part (and, for x, on the carry of h*4 into bit 3), so every address is
    //   buffer base + per-lane constant + compile-time immediate.
    const int l16 = lane & 15;
    const int q = l16 >> 2, pp = l16 & 3;
    const int kgrp = lane >> 4;                    // k octet of this lane group
    const int lrow = kgrp * 8 + q;
    // read pointers into buffer 0; toggled in place between the two buffers after every step (no second copy of the
    // lane constants stays live: the kernel is VGPR-bound)
    unsigned pa[4];
    unsigned pb[KWB][2][4];
    const unsigned abase0 = lds_addr(abuf0), bbase0 = lds_addr(bbuf0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int acol = (wm * 64 + i * 16 + pp * 4) * 2;
        pa[i] = abase0 + lrow * ROWB + (acol ^ (row_key(lrow) << 5));
#pragma unroll
        for (int tp = 0; tp < KWB; ++tp)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int rb = (lrow + h * 4) * s + (tg * KWB + tp) * d;      // row at ks = 0 (window row 0 = the block's first tap)
                const int bcol = (wn * 64 + i * 16 + pp * 4) * 2;
                pb[tp][h][i] = bbase0 + rb * ROWB + (bcol ^ (row_key(rb) << 5));   // ks*32*s rows further: same key (multiple of 16)
            }
    }
    // M32 lane geometry: 16-lane group gi covers channels cb*16.. of a 32-channel block and the k half kh of a 16-deep
    // substep; LDS row of fragment (kk, hh) = kk*16 + hh*4 + (kh*8 + q) [+ tap*d] -- everything but the lane part is an immediate
    unsigned pa32[2];
    unsigned pb32[KWB][2];
    if constexpr (M32) {
        const int cb = kgrp & 1, kh = kgrp >> 1;
        const int lrow32 = kh * 8 + q;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int acol = (wm * 64 + i * 32 + cb * 16 + pp * 4) * 2;
            pa32[i] = abase0 + lrow32 * ROWB + (acol ^ (rkey(lrow32) << 5));
#pragma unroll
            for (int tp = 0; tp < KWB; ++tp) {
                const int rb = lrow32 + (tg * KWB + tp) * d;
                const int bcol = (wn * 64 + i * 32 + cb * 16 + pp * 4) * 2;
                pb32[tp][i] = bbase0 + rb * ROWB + (bcol ^ (rkey(rb) << 5));
            }
        }
    }
    int a_toggle = BT * ROWB, b_toggle = xrows * ROWB;

    // ---- K loop, software-pipelined inside the wave and across the block barrier -------------------------------
    // A step (64 rows) is NG = 2*NT fragment groups (ks-major, tap-minor), 16 MFMAs each.
    //  * the transposing reads of group g+1 are issued before the MFMAs of group g (B fragments double-buffered;
    //    A fragments are refilled in place, row by row, as the last group that uses them retires each row);
    //  * the block barrier sits BEFORE THE LAST GROUP of a step: by then every read of the step's buffers has been
    //    issued and completed, so after the barrier the LDS-DMA for step+2 goes into those buffers, the first
    //    fragments of step+1 are requested from the other buffers, and all of it hides behind the last 16 MFMAs.
    // NT / LAST are compile-time so that each body is straight-line code with in-place accumulators (the compiler
    // does not move reads across a branch).
    W2L_DIAG_STAMP_DECL();
    bf16x8 a[4], b[2][4];
    auto load_a1 = [&](int i, int ks) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const bf16x4 va = tr_read(pa[i] + (ks * 32 + h * 4) * ROWB);
#pragma unroll
            for (int e = 0; e < 4; ++e) a[i][h * 4 + e] = va[e];
        }
    };
    auto load_b = [&](bf16x8* dst, int tp, int ks) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bf16x4 vb = tr_read(pb[tp][h][i] + ks * 32 * s * ROWB);
#pragma unroll
                for (int e = 0; e < 4; ++e) dst[i][h * 4 + e] = vb[e];
            }
    };
    auto toggle = [&]() {
        if constexpr (M32) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                pa32[i] += a_toggle;
#pragma unroll
                for (int tp = 0; tp < KWB; ++tp) pb32[tp][i] += b_toggle;
            }
            a_toggle = -a_toggle;
            b_toggle = -b_toggle;
            return;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            pa[i] += a_toggle;
#pragma unroll
            for (int tp = 0; tp < KWB; ++tp)
#pragma unroll
                for (int h = 0; h < 2; ++h) pb[tp][h][i] += b_toggle;
        }
        a_toggle = -a_toggle;
        b_toggle = -b_toggle;
    };
    // one step; on entry a[] = A(ks 0) and b[0] = B(tap 0, ks 0) of this step are requested
    auto step_body = [&](auto nt_tag, auto last_tag, char* adst, char* bdst, int n_nn, int ts_nn, bool have_nn) {
        constexpr int NT = decltype(nt_tag)::value;
        constexpr bool LAST = decltype(last_tag)::value;
        constexpr int NG = (BT / 32) * NT;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int ks = g / NT, tp = g % NT;
            const bool lastg = g + 1 == NG;
            const int ks2 = (g + 1) / NT, tp2 = (g + 1) % NT;
            if (!lastg) load_b(b[(g + 1) & 1], tp2, ks2);
            if (lastg && !LAST) {
                W2L_DIAG_STAMP_STEP0();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // step+1's tiles (this wave's part) have landed
                W2L_DIAG_STAMP_STEP1();
                __syncthreads();                                    // ... everyone's; and nobody reads this step's buffers any more
                W2L_DIAG_STAMP_STEP2();
                toggle();                                           // read pointers -> step+1's buffers
                if (have_nn) stage(adst, bdst, n_nn, ts_nn);        // step+2 -> this step's buffers
                W2L_DIAG_STAMP_STEP3();
                load_b(b[(g + 1) & 1], 0, 0);
            }
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    acc[tp][mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mi], b[g & 1][ni], acc[tp][mi][ni], 0, 0, 0);
                if (!lastg && ks2 != ks) load_a1(mi, ks2);
                if (lastg && !LAST) load_a1(mi, 0);
            }
        }
    };
    // ---- the same step with 32x32x16 fragments: 4 k-substeps of 16 rows, NG = 4*NT groups of 2 x 2 MFMAs
    bf16x8 a32[2][2], b32[2][2];       // A fragments double-buffered by the parity of the k-substep (see the MFMA statement)
    auto load_a32 = [&](int i, int kk) {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const bf16x4 va = tr_read(pa32[i] + (kk * 16 + hh * 4) * ROWB);
#pragma unroll
            for (int e = 0; e < 4; ++e) a32[W2L_AG_ADBUF ? (kk & 1) : 0][i][hh * 4 + e] = va[e];
        }
    };
    auto load_b32 = [&](bf16x8* dst, int tp, int kk) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const bf16x4 vb = tr_read(pb32[tp][i] + (kk * 16 + hh * 4) * ROWB);
#pragma unroll
                for (int e = 0; e < 4; ++e) dst[i][hh * 4 + e] = vb[e];
            }
    };
    // AGACC: every MFMA is an asm statement whose accumulator is an "a"-constrained operand, and hipcc neither pads the hazards
    // of an instruction inside an asm string nor knows that one is an MFMA.  So the accumulators are never touched by compiler
    // code while MFMAs are in flight: the FIRST k-substep of a segment takes the constant 0 as C (an output-only operand: no
    // v_accvgpr_write fill in front of an MFMA), the LAST one carries the wait states behind which its result may be read
    // (16 passes: 18+) inside its own statement, and every statement opens with s_nop 1 (a fragment register the compiler has
    // just written with a VALU copy is an MFMA operand two wait states later at the earliest; behind an MFMA that is still
    // executing the two cycles cost nothing).
    auto step_body32 = [&](auto nt_tag, auto last_tag, auto first_tag, char* adst, char* bdst, int n_nn, int ts_nn, bool have_nn) {
        constexpr int NT = decltype(nt_tag)::value;
        constexpr bool LAST = decltype(last_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value;
        constexpr int NG = (BT / 16) * NT;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int kk = g / NT, tp = g % NT;
            const bool lastg = g + 1 == NG;
            const int kk2 = (g + 1) / NT, tp2 = (g + 1) % NT;
            if (!lastg) load_b32(b32[(g + 1) & 1], tp2, kk2);
            if (lastg && !LAST) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                toggle();
                if (have_nn) stage(adst, bdst, n_nn, ts_nn);
                load_b32(b32[(g + 1) & 1], 0, 0);
            }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    if constexpr (AGACC) {
                        if constexpr (FIRST) {
                            if (kk == 0) {
                                asm volatile(W2L_AG_PRE "v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0"
                                             : "=a"(acc32[tp][mi][ni]) : "v"(a32[W2L_AG_ADBUF ? (kk & 1) : 0][mi]), "v"(b32[g & 1][ni]));
                                continue;
                            }
                        }
                        if (LAST && kk == BT / 16 - 1)
                            asm volatile(W2L_AG_PRE "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7"
                                         : "+a"(acc32[tp][mi][ni]) : "v"(a32[W2L_AG_ADBUF ? (kk & 1) : 0][mi]), "v"(b32[g & 1][ni]));
                        else
                            asm volatile(W2L_AG_PRE "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0"
                                         : "+a"(acc32[tp][mi][ni]) : "v"(a32[W2L_AG_ADBUF ? (kk & 1) : 0][mi]), "v"(b32[g & 1][ni]));
                    } else
                        acc32[tp][mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a32[W2L_AG_ADBUF ? (kk & 1) : 0][mi], b32[g & 1][ni], acc32[tp][mi][ni], 0, 0, 0);
                if (!lastg && kk2 != kk) load_a32(mi, kk2);
                if (lastg && !LAST) load_a32(mi, 0);
            }
        }
    };
    auto advance = [&](int& n, int& ts) {
        if (++ts == p.tsteps) { ts = 0; ++n; }
    };
    // the whole K loop exists once per live-tap count (branch outside the loop)
    auto run = [&](auto nt_tag) {
        if (step_begin >= step_end) return;
        int n = step_begin / p.tsteps;
        int ts = step_begin - n * p.tsteps;
        stage(abuf0, bbuf0, n, ts);
        advance(n, ts);                                   // (n, ts): coordinates of step+1
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (step_begin + 1 < step_end) stage(abuf1, bbuf1, n, ts);
        if constexpr (M32) {
            load_a32(0, 0);
            load_a32(1, 0);
            load_b32(b32[0], 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) load_a1(i, 0);
            load_b(b[0], 0, 0);
        }
        W2L_DIAG_STAMP_LOOP_BEGIN();
        if constexpr (AGACC) {
            // the segment's first step zeroes the accumulators through C = 0: peeled, so that the steady loop is one body
            if (step_begin + 1 == step_end) {              // a one-step segment: first and last at once
                step_body32(nt_tag, std::true_type{}, std::true_type{}, nullptr, nullptr, 0, 0, false);
                return;
            }
            advance(n, ts);                                // now step+2
            step_body32(nt_tag, std::false_type{}, std::true_type{}, abuf0, bbuf0, n, ts, step_begin + 2 < step_end);
            for (int step = step_begin + 1; step + 1 < step_end; ++step) {
                const int par = (step - step_begin) & 1;
                advance(n, ts);
                step_body32(nt_tag, std::false_type{}, std::false_type{}, par ? abuf1 : abuf0, par ? bbuf1 : bbuf0, n, ts,
                            step + 2 < step_end);
            }
            step_body32(nt_tag, std::true_type{}, std::false_type{}, nullptr, nullptr, 0, 0, false);
            return;
        }
        for (int step = step_begin; step + 1 < step_end; ++step) {
            const int par = (step - step_begin) & 1;
            advance(n, ts);                               // now step+2
            if constexpr (M32)
                step_body32(nt_tag, std::false_type{}, std::false_type{}, par ? abuf1 : abuf0, par ? bbuf1 : bbuf0, n, ts,
                            step + 2 < step_end);
            else
                step_body(nt_tag, std::false_type{}, par ? abuf1 : abuf0, par ? bbuf1 : bbuf0, n, ts, step + 2 < step_end);
        }
        if constexpr (M32) step_body32(nt_tag, std::true_type{}, std::false_type{}, nullptr, nullptr, 0, 0, false);
        else step_body(nt_tag, std::true_type{}, nullptr, nullptr, 0, 0, false);
    };
    // a wave whose tap group lies beyond Kw (TG > 1, last tap group of an odd tap count): no MFMAs, but its share of the
    // staging and every barrier of run()
    auto run_idle = [&]() {
        if (step_begin >= step_end) return;
        int n = step_begin / p.tsteps;
        int ts = step_begin - n * p.tsteps;
        stage(abuf0, bbuf0, n, ts);
        advance(n, ts);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (step_begin + 1 < step_end) stage(abuf1, bbuf1, n, ts);
        for (int step = step_begin; step + 1 < step_end; ++step) {
            const int par = (step - step_begin) & 1;
            advance(n, ts);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            toggle();
            if (step + 2 < step_end) stage(par ? abuf1 : abuf0, par ? bbuf1 : bbuf0, n, ts);
        }
    };
    for (;;) {
    {   // ---- one segment: steps [step_begin, step_end) of tile w / S
        const int tile = w / S;
        step_begin = w - tile * S;
        const int left = w_end - w;
        step_end = (S - step_begin) < left ? S : step_begin + left;
        set_tile(tile);
    }
#pragma unroll
    for (int tp = 0; tp < KWB; ++tp)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (AGACC) {
                    // (no fill: the first MFMA of every accumulator takes the constant 0 as its C operand -- see step_body32)
                } else if constexpr (M32) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc32[tp][i >> 1][i & 1][j * 4 + e] = 0.f;
                } else {
                    acc[tp][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
    {
        if (TG > 1 && ntaps == 0) run_idle();
        else if (KWB == 1 || ntaps == KWB) run(std::integral_constant<int, KWB>{});
        else if (KWB == 3 && ntaps == 2) run(std::integral_constant<int, (KWB == 3 ? 2 : 1)>{});
        else run(std::integral_constant<int, 1>{});
    }
    W2L_DIAG_STAMP_STORE(lane, (blockIdx.y * gridDim.x + blockIdx.x) * NWV + wave_all);
    // the read pointers were toggled once per non-last step: bring them back to buffer 0 for a following segment
    if (step_end > step_begin && ((step_end - step_begin - 1) & 1)) toggle();
    // a piece of a tile (stream-K) is added atomically; a tile this block covered completely is stored
    // (with p.accumulate -- dw already holds earlier passes -- every piece adds)
    const bool piece = SK && (p.accumulate || !(step_begin == 0 && step_end == S));

    // ---- split-K through slabs: publish the partial tile, draw a ticket; only the last arriver goes on.  The eight XCDs' L2s
    // are not coherent inside a launch, so the slabs never live in a cache that could be stale: every slab byte is STORED
    // write-through (16-byte buffer stores with sc1) and LOADED with sc1 (past this CU's L1, which no plain load ever fills
    // with slab lines), every storing wave drains its stores (vmcnt(0)) before the block barrier behind which ONE lane adds
    // to the tile's agent-scope ticket, and the block whose add came last reads only behind another barrier.  No release
    // fence (a buffer_wbl2 of the XCD's whole L2 per block: 2-7 us each with a freshly written 192 KB slab in it -- rounds
    // 2-4 paid it on every partial tile) and no acquire (a buffer_inv per tile).  Correct wherever the tile's blocks ran.
    // The ISA guarantee relied on (gfx942 / gfx950 memory model, LLVM AMDGPUUsage "Memory Model gfx942"): an access carrying
    // sc1 IS an agent-scope access -- it is the code LLVM itself emits for `load / store atomic monotonic syncscope("agent")`
    // (global / buffer load sc1=1, store sc1=1): the store is written through to the level all XCDs share, the load is served
    // from that level, never from a non-coherent line.  A fence's buffer_wbl2 / buffer_inv exist to make OTHER, plain accesses
    // visible / fresh; there are none here -- every slab byte moves through sc1 instructions, ordered against the ticket by
    // s_waitcnt vmcnt(0) + the block barrier on both sides.  tests/test_gpu_kernels.py::test_wgrad_slabs_stress_bit_exact runs
    // hundreds of back-to-back split launches beside main-stream traffic and compares them bit for bit.
    if (!SK && p.slabs != nullptr && nsplit > 1) {            // (the stream-K launch has no workspace form)
        constexpr int TILE_F = KWBLK * BM * BNC;
        const int tid_t = tid & 255;                   // thread inside its tap group
        // buffer descriptor over the nsplit slabs of this tile (wave-uniform inputs only)
        const uint64_t tb = (uint64_t)(p.slabs + (int64_t)slab0 * TILE_F);
        const unsigned tb_lo = __builtin_amdgcn_readfirstlane((unsigned)tb), tb_hi = __builtin_amdgcn_readfirstlane((unsigned)(tb >> 32));
        const unsigned tbytes = __builtin_amdgcn_readfirstlane((unsigned)nsplit * (unsigned)(TILE_F * 4));
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)tb_hi << 32) | tb_lo), 0, tbytes, 0x00020000);
        const unsigned voff0 = (unsigned)((tg * KWB * 16 * 256 + tid_t) * 16);     // + (tp * 16 + c) * 4096 + slab * TILE_F * 4
        {
            const unsigned soff = __builtin_amdgcn_readfirstlane((unsigned)split * (unsigned)(TILE_F * 4));
#pragma unroll
            for (int tp = 0; tp < KWB; ++tp)
#pragma unroll
                for (int c = 0; c < 16; ++c)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, chunk_get(tp, c)), rsrc,
                                                           voff0 + (unsigned)((tp * 16 + c) * 4096), soff, 16 /* sc1 */);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // EVERY storing wave: its write-through stores have left
        __syncthreads();                               // ... all of them; the K loop's LDS is dead
        unsigned* flag = reinterpret_cast<unsigned*>(smem);
        if (tid == 0) *flag = __hip_atomic_fetch_add(&p.tickets[tile_id], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (*flag != (unsigned)(nsplit - 1)) return;
        if (tid == 0) __hip_atomic_store(&p.tickets[tile_id], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");      // (no instruction: the loads below stay below)
#pragma unroll
        for (int tp = 0; tp < KWB; ++tp)
#pragma unroll
            for (int c = 0; c < 16; ++c) chunk_set(tp, c, f32x4{0.f, 0.f, 0.f, 0.f});
        for (int sp = 0; sp < nsplit; ++sp) {          // in split order whoever came last: bit-reproducible
            const unsigned soff = __builtin_amdgcn_readfirstlane((unsigned)sp * (unsigned)(TILE_F * 4));
#pragma unroll
            for (int tp = 0; tp < KWB; ++tp)
#pragma unroll
                for (int c0_ = 0; c0_ < 16; c0_ += 8) {     // eight 16-byte loads in flight per lane
                    u32x4 t[8];
#pragma unroll
                    for (int c = 0; c < 8; ++c)
                        t[c] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff0 + (unsigned)((tp * 16 + c0_ + c) * 4096), soff, 16 /* sc1 */);
#pragma unroll
                    for (int c = 0; c < 8; ++c) chunk_set(tp, c0_ + c, chunk_get(tp, c0_ + c) + __builtin_bit_cast(f32x4, t[c]));
                }
        }
    }

    // ---- epilogue: acc[tp][mi][ni][r] = dw[kw0+tp][co = m0+wm*64+mi*16+fq*4+r][ci = c0+wn*64+ni*16+fr]; the 32x32 form:
    // acc32[tp][mi][ni][reg] = dw[kw0+tp][co = m0+wm*64+mi*32+(reg&3)+8*(reg>>2)+4*(lane>>5)][ci = c0+wn*64+ni*32+(lane&31)]
    // (a wave-instruction then covers two 128-byte row segments: the shape float atomics take at full rate) ----
    auto put = [&](float* dst, float v) {
        if (p.atomic || piece) atomicAdd(dst, v);              // several blocks per element (no workspace)
        else if (p.accumulate) *dst += v;                      // one block per element: plain read-add-store
        else *dst = v;
    };
    const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int tp = 0; tp < KWB; ++tp) {
        if (tp >= ntaps) break;
        float* base = L.dw + (int64_t)(kw0 + tp) * L.Cout * L.Cin;
        if constexpr (M32) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const int ci = c0 + wn * 64 + ni * 32 + (lane & 31);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int co = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        if (co < L.Cout && ci < L.Cin) put(base + (int64_t)co * L.Cin + ci, acc32[tp][mi][ni][r]);
                    }
                }
        } else {
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    const int ci = c0 + wn * 64 + ni * 16 + fr;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int co = m0 + wm * 64 + mi * 16 + fq * 4 + r;
                        if (co < L.Cout && ci < L.Cin) put(base + (int64_t)co * L.Cin + ci, acc[tp][mi][ni][r]);
                    }
                }
        }
    }
    if constexpr (!SK) {
        break;
    } else {
        w += step_end - step_begin;
        if (w >= w_end) break;
        __syncthreads();    // the next segment's LDS-DMA reuses buffers the slowest wave may still be reading
    }
    }
}

}  // namespace w2l_wgrad
