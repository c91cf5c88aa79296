// Conv1d weight gradient as an MFMA GEMM whose reduction runs over time (gfx950).
//
//   dw[kw][co][ci] (+)= sum_{n,t} dy[n][t][co] * xp[n][t*s + kw*d][ci]
//
// GEMM view per tap kw: M = co, N = ci, K = (n, t).  Both operands are
// channels-last, i.e. the reduction index t is the SLOW axis of both tiles, so
// the MFMA fragments (8 consecutive k per lane) are column reads of the LDS
// images: they are fetched with ds_read_b64_tr_b16, CDNA4's transposing LDS
// read, from row-major [t][channel] tiles that LDS-DMA fills straight from HBM.
// One block owns a [128 co x 128 ci] tile of KWB adjacent taps and a slice of the
// (n,t) range (split-K; partial tiles are combined with fp32 atomics).  The taps
// share the dy tile and read the same staged x window at row offsets tap*d, which
// halves the L2->LDS traffic per FLOP -- the kernel is fabric-bound, not MFMA-bound,
// with one tap per block (11.6 GB of tile reads for an 896x896x29 layer).
//
// Replaces the weight-gradient half of aten::convolution_backward for the
// nn.Conv1d call sites wav2letter.py:35-36,42 / jasper.py:96-105,127.
#include "conv_wgrad_kernel.h"
#include "../../include/w2l_hip.h"
#include <algorithm>
#include <array>
#include <atomic>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

namespace {
using namespace w2l_wgrad;

template <int KWB, bool S1, bool SK, int TG = 1, bool M32 = false>
__global__ __launch_bounds__(256 * TG, TG == 1 ? 2 : 1) void conv_wgrad_kernel(WgradParams p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    conv_wgrad_body<KWB, S1, SK, TG, M32>(p, smem);
}

}  // namespace

// ---- the three-tap kernel lives in a code object of its own (conv_wgrad3_dev.hip explains why), embedded here as bytes ----
extern "C" const unsigned char w2l_wgrad3_hsaco[];
extern "C" const unsigned int w2l_wgrad3_hsaco_len;

namespace {

// hipFunction_t of w2l_wgrad3_kernel on the calling thread's current device (module loaded once per device)
int wgrad3_function(hipFunction_t* fn, int which /* 0 four waves, 1 two tap groups (eight waves) */) {
    static std::mutex mu;
    static std::map<int, std::array<hipFunction_t, 2>> loaded;
    int dev = 0;
    W2L_CHECK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    auto it = loaded.find(dev);
    if (it == loaded.end()) {
        hipModule_t mod;
        W2L_CHECK_ARG(w2l_wgrad3_hsaco_len > 0, "conv1d_wgrad: the three-tap code object is missing from this build");
        W2L_CHECK_HIP(hipModuleLoadData(&mod, w2l_wgrad3_hsaco));
        std::array<hipFunction_t, 2> f;
        W2L_CHECK_HIP(hipModuleGetFunction(&f[0], mod, "w2l_wgrad3_kernel"));
        W2L_CHECK_HIP(hipModuleGetFunction(&f[1], mod, "w2l_wgrad3x2_kernel"));
        it = loaded.emplace(dev, f).first;
    }
    *fn = it->second[which];
    return 0;
}

// Split the (n,t) reduction over `splits` blocks per tile so that the grid fills whole rounds of the
// 512 resident blocks (256 CUs x 2): cost = rounds x steps-per-block (+ the fp32 atomic traffic of the
// extra partial tiles, ~1.3 TB/s chip-wide).  Non-power-of-two splits are allowed.
// measured choices (w2l_conv1d_wgrad_tune): shape -> split count | block order << 16
typedef std::tuple<int, int, int, int, int> WShapeKey;
std::map<WShapeKey, int> g_wtuned;
std::mutex g_wtuned_mu;
thread_local int g_force_splits = 0;      // per calling thread, like g_force_cfg of the implicit GEMM
thread_local int g_force_order = -1;
constexpr int kDefaultOrder = 1;

// `order` values: bit 0 = block order, bit 1 = stream-K decomposition (then the split count is not used), bit 2 = two tap
// groups per block (the 8-wave kernel, TG = 2; not combined with stream-K), bit 3 = 32x32x16 MFMA fragments
constexpr int kStreamK = 2;
constexpr int kTapGroups2 = 4;
constexpr int kMfma32 = 8;                 // bit 3: the 32x32x16 MFMA form (stride 1; not combined with bits 1 and 2)
constexpr int kTaps3 = 16;                 // bit 4: THREE taps per block, accumulators in AGPRs (conv_wgrad3_dev.hip; stride 1, dilation <= 4)
constexpr int kTaps3MaxDil = 4;
constexpr int kDealt = 32;                 // bit 5: dealt stream-K (WgradParams::dealt; the plan's split field holds the range count);
                                           // needs the workspace -- without one the launch falls back to a classic split of 2
constexpr int kAtomicSplit = 64;           // bit 6: a classic split adds its partial tiles atomically even when a workspace is given
constexpr int kOrderMask = 127;
constexpr int kDealtFallbackSplits = 2;
constexpr int kResidentBlocks = 512;       // 256 CUs x 2 blocks (LDS and registers both allow two)
thread_local bool g_last_dealt = false;    // did the calling thread's last launch take the dealt path? (the tuner asks)
// w2l_wgrad_deterministic(1): plans that add their partial tiles with fp32 atomics although a workspace is at hand (bit 6) lose
// that bit when resolved -- a plan cache written by a default-mode run must not switch atomics back on in a run that asked
// for bit-reproducible gradients.  Process-wide (the weight gradients are launched from autograd's worker thread).
std::atomic<int> g_deterministic{0};

// second-segment blocks of a dealt launch, longest first (WgradParams::dealt_perm); cached per geometry: 21 launches per step
// must not sort 512 ranges each
struct DealtPerm { int nb; std::vector<unsigned short> perm; };
std::map<std::tuple<int, int, int>, DealtPerm> g_dealt_perms;
std::mutex g_dealt_mu;

bool dealt_geometry_ok(int tiles, int S, int G) {
    if (G < 1 || G > W2L_WGRAD_MAX_RANGES || tiles < 1 || tiles > G) return false;
    const int64_t W = (int64_t)tiles * S;
    return W >= G && W * (G + 2) < (1LL << 31);
}

// fills p.dealt_perm / p.dealt_b; false: this geometry has no dealt form
bool dealt_fill(WgradParams& p, int tiles, int S, int G) {
    if (!dealt_geometry_ok(tiles, S, G)) return false;
    std::lock_guard<std::mutex> lock(g_dealt_mu);
    auto key = std::make_tuple(tiles, S, G);
    auto it = g_dealt_perms.find(key);
    if (it == g_dealt_perms.end()) {
        // Blocks are dealt round-robin to the 8 XCDs, so block G + j runs where blocks j % 8 ran: the first-segment blocks of
        // XCD x own the consecutive ranges xcd_remap gives them, and the second segments of THOSE ranges (the next tiles
        // along: same dy tiles and x windows in that XCD's L2) are dealt to the same XCD, longest first -- position j of the
        // list holds the (j / 8)-th longest second segment of XCD j % 8's ranges, 0xffff where that XCD has no more.
        std::vector<std::pair<int, int>> second[8];        // per XCD: (-steps, range)
        const int q = G >> 3, rem = G & 7;
        size_t longest = 0;
        for (int x = 0; x < 8; ++x) {
            const int base = x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q, cnt = x < rem ? q + 1 : q;
            for (int r = base; r < base + cnt; ++r) {
                const DealtSeg sg = dealt_segment((unsigned)tiles, (unsigned)S, (unsigned)G, (unsigned)r, true);
                if (sg.w_end > sg.w) second[x].emplace_back(-(sg.w_end - sg.w), r);
            }
            std::sort(second[x].begin(), second[x].end());
            longest = std::max(longest, second[x].size());
        }
        DealtPerm dp;
        for (size_t i = 0; i < longest; ++i)
            for (int x = 0; x < 8; ++x) dp.perm.push_back(i < second[x].size() ? (unsigned short)second[x][i].second : (unsigned short)0xffff);
        while (!dp.perm.empty() && dp.perm.back() == 0xffff) dp.perm.pop_back();
        dp.nb = (int)dp.perm.size();
        if (dp.nb > W2L_WGRAD_MAX_RANGES) return false;
        it = g_dealt_perms.emplace(key, std::move(dp)).first;
    }
    p.dealt = G;
    p.dealt_b = it->second.nb;
    std::copy(it->second.perm.begin(), it->second.perm.end(), p.dealt_perm);
    return true;
}

size_t dealt_ws_need(int tiles, int kwblk, int G) {
    return (size_t)64 * 1024 + (size_t)(G + tiles) * kwblk * BM * BNC * sizeof(float);
}

int plan_splits(int N, int Cin, int Cout, int Tout, int Kw, int* tsteps_out, int* order_out = nullptr) {
    const int kwb = Kw > 1 ? KWB_DEFAULT : 1;
    if (order_out) *order_out = g_force_order >= 0 ? g_force_order : kDefaultOrder;
    {
        const int ts = (Tout + BT - 1) / BT;
        if (tsteps_out) *tsteps_out = ts;
        if (g_force_splits > 0)        // (a dealt plan's "split count" is its range count: not bounded by the step count)
            return (g_force_order >= 0 && (g_force_order & kDealt)) || g_force_splits <= N * ts ? g_force_splits : N * ts;
        std::lock_guard<std::mutex> lock(g_wtuned_mu);
        auto it = g_wtuned.find(WShapeKey(N, Cin, Cout, Tout, Kw));
        if (it != g_wtuned.end()) {
            if (order_out && g_force_order < 0) *order_out = it->second >> 16;
            return it->second & 0xffff;
        }
    }
    const int tiles = ((Cout + BM - 1) / BM) * ((Cin + BNC - 1) / BNC) * ((Kw + kwb - 1) / kwb);
    const int tsteps = (Tout + BT - 1) / BT;
    const int total = N * tsteps;
    if (tsteps_out) *tsteps_out = tsteps;
    const double t_step_us = 2.7 * kwb / 2.0;                       // one 64-row K step of a block sharing its CU
    const double out_us = (double)Cout * Cin * Kw * 4.0 / 1.3e6;    // one full pass of fp32 atomics over dw
    int best = 1;
    double best_cost = 1e30;
    for (int s = 1; s <= 32 && s <= total; ++s) {
        const int steps = (total + s - 1) / s;
        if (s > 1 && steps < 4) break;
        const long blocks = (long)tiles * s;
        const long rounds = (blocks + 511) / 512;
        double per_step = t_step_us * (blocks <= 256 ? 0.6 : 1.0);  // a block alone on its CU runs faster
        double cost = rounds * steps * per_step + (s > 1 ? 0.5 * out_us * s : 0.0);
        if (cost < best_cost * 0.97) { best_cost = cost; best = s; }
    }
    // (the stream-K decomposition is a candidate of the measured selection only: its kernel carries ~10 % more instructions
    // per K step -- the segment loop's live scalars -- and beat the best split-K plan on one shape of the Wav2Letter table)
    return best;
}

}  // namespace

extern "C" void w2l_wgrad_deterministic(int on) { g_deterministic.store(on ? 1 : 0, std::memory_order_relaxed); }

// testing / profiling hook: pin the split count (0 = automatic) and the block order (-1 = automatic)
extern "C" void w2l_wgrad_force_plan(int splits, int order) {
    g_force_splits = splits > 0 ? splits : 0;
    // bit 0: block order, 1: stream-K, 2: two tap groups per block, 3: 32x32x16 MFMA, 4: three taps (AGPR), 5: dealt stream-K
    // (splits = the range count), 6: classic splits add atomically even with a workspace
    g_force_order = order >= 0 ? (order & kOrderMask) : -1;
}

// the plan a launch of this problem will take: its order bits (see w2l_wgrad_force_plan) | split or range count << 8
extern "C" int w2l_wgrad_plan(int N, int Cin, int Cout, int Tout, int Kw) {
    int order = 0;
    const int splits = plan_splits(N, Cin, Cout, Tout, Kw, nullptr, &order);
    return (order & 0xff) | (splits << 8);
}

extern "C" int w2l_wgrad_needs_zero(int N, int Cin, int Cout, int Tout, int Kw) {
    int order = 0;
    const int splits = plan_splits(N, Cin, Cout, Tout, Kw, nullptr, &order);
    if ((order & kDealt) && !(order & kStreamK)) return 1;      // no workspace: the dealt plan's fallback is an atomic split
    return splits > 1 || (order & kStreamK) ? 1 : 0;
}

constexpr size_t kWgradTicketBytes = 64 * 1024;

// slabs of one launch: tiles x splits x (taps per block) x 128 x 128 floats; the tap count rounds up to the block's tap
// group, so the 4-tap form can need a little more than the 2-tap form: the need is the larger of the two
static size_t wgrad_ws_need(int Cin, int Cout, int Kw, int splits) {
    const size_t tiles_mn = (size_t)((Cout + BM - 1) / BM) * ((Cin + BNC - 1) / BNC);
    size_t taps = 1;
    if (Kw > 1) {
        // (2, 4, 3 and 6 taps per block: the two-tap kernel, its two-tap-group form, the three-tap kernel and its two-tap-group form)
        const size_t t2 = (size_t)((Kw + 1) / 2) * 2, t4 = (size_t)((Kw + 3) / 4) * 4, t3 = (size_t)((Kw + 2) / 3) * 3;
        const size_t t6 = (size_t)((Kw + 5) / 6) * 6;
        taps = std::max(std::max(t2, t4), std::max(t3, t6));
    }
    return kWgradTicketBytes + tiles_mn * taps * splits * BM * BNC * sizeof(float);
}

// the largest workspace a dealt plan of this layer can ask for: over the block forms (2 / 3 taps per 4-wave block on 512
// or 256 ranges, 4 / 6 taps per 8-wave block on 256 ranges) that have one (tiles <= ranges)
extern "C" int64_t w2l_wgrad_dealt_workspace_bytes(int Cin, int Cout, int Kw) {
    const int tiles_mn = ((Cout + BM - 1) / BM) * ((Cin + BNC - 1) / BNC);
    size_t need = 0;
    const int forms[4][2] = {{2, kResidentBlocks}, {3, kResidentBlocks}, {4, kResidentBlocks / 2}, {6, kResidentBlocks / 2}};
    for (const auto& f : forms) {
        const int kwblk = Kw > 1 ? f[0] : 1, tiles = tiles_mn * ((Kw + kwblk - 1) / kwblk);
        if (tiles <= f[1]) need = std::max(need, dealt_ws_need(tiles, kwblk, f[1]));
    }
    return (int64_t)need;
}

// test hook (host only): the blocks of a dealt launch in launch order, six ints each -- range, tile, first step, end step,
// place among the tile's segments, number of the tile's segments; returns the block count, -1 if the geometry has no dealt form
extern "C" int w2l_wgrad_dealt_segments(int tiles, int S, int G, int* out, int max_blocks) {
    WgradParams p;
    if (!dealt_fill(p, tiles, S, G)) return -1;
    const int nblk = G + p.dealt_b;
    for (int b = 0; b < nblk && b < max_blocks; ++b) {
        const bool second = b >= G;
        const int q = G >> 3, rem = G & 7, xcd = b & 7;          // xcd_remap, host side
        const int r = second ? p.dealt_perm[b - G] : (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (b >> 3);
        int* o = out + 6 * b;
        if (r == 0xffff) {                                       // a padding position of the per-XCD deal: the block exits at once
            o[0] = -1; o[1] = o[2] = o[3] = o[4] = o[5] = 0;
            continue;
        }
        const DealtSeg sg = dealt_segment((unsigned)tiles, (unsigned)S, (unsigned)G, (unsigned)r, second);
        const int t = sg.w / S;
        o[0] = r; o[1] = t; o[2] = sg.w - t * S; o[3] = sg.w_end - t * S; o[4] = sg.split; o[5] = sg.nsplit;
    }
    return nblk;
}

static bool wgrad_ws_ok(int Cin, int Cout, int Kw, int splits, const void* ws, int64_t ws_bytes) {
    if (ws == nullptr) return false;
    const int kwb = Kw > 1 ? KWB_DEFAULT : 1;          // the 2-tap form has the most tiles (tickets)
    const size_t tiles = (size_t)((Cout + BM - 1) / BM) * ((Cin + BNC - 1) / BNC) * ((Kw + kwb - 1) / kwb);
    return tiles * sizeof(unsigned) <= kWgradTicketBytes && wgrad_ws_need(Cin, Cout, Kw, splits) <= (size_t)ws_bytes;
}

// with a workspace of ws_bytes: does the launch still add into dw with atomics (i.e. need a zero-filled dw)?
// What a launch of this problem will run as: the measured (or cost-model) plan, narrowed to what the layer's stride /
// dilation / tap count admit and to the workspace at hand.  Shared by the launcher and by w2l_wgrad_needs_zero_*.
struct WgradResolved {
    int splits, order, tsteps;
    bool streamk, tg2, m32, taps3, slabs, atomic;
    int kwb, kwblk, kgroups, tiles;
    int G;                       // > 0: the dealt stream-K path with G ranges
    bool needs_zero;             // the launch ADDS into dw (atomics): the caller must hand it a zero-filled (or accumulated) dw
};

static WgradResolved wgrad_resolve(int N, int Cin, int Cout, int Tout, int Kw, int stride, int dil, bool have_ws, int64_t ws_bytes) {
    WgradResolved r;
    r.order = 0;
    r.splits = plan_splits(N, Cin, Cout, Tout, Kw, &r.tsteps, &r.order);
    if (g_deterministic.load(std::memory_order_relaxed)) r.order &= ~kAtomicSplit;
    const int order = r.order;
    int G = 0;
    if ((order & kDealt) && !(order & kStreamK)) { G = r.splits; r.splits = kDealtFallbackSplits; }
    const int total_steps = N * r.tsteps;
    if (r.splits > total_steps) r.splits = total_steps;
    // (only split counts whose ranges are all non-empty: ceil(total / s) steps each leave ceil(total / that) ranges -- 32 asked of
    // 40 steps are 20 ranges of 2; an empty range would start in the NEXT tile and draw that tile's ticket)
    if (r.splits > 1) r.splits = (total_steps + (total_steps + r.splits - 1) / r.splits - 1) / ((total_steps + r.splits - 1) / r.splits);
    // stream-K is the atomic path's alternative to split-K: with a workspace (deterministic slabs) the classic plan runs,
    // unless the plan says that it adds atomically whatever it is given (bit 6)
    r.streamk = (order & kStreamK) && (!have_ws || (order & kAtomicSplit));
    if (r.streamk) r.splits = 1;
    // (with stream-K: 256 persistent 8-wave blocks, one per CU -- built for stride 1 only; the plan cache is keyed without the
    // stride, so a plan measured on a stride-1 layer may reach a strided one: that launch falls back to the 4-wave stream-K kernel)
    r.tg2 = (order & kTapGroups2) && Kw > 2 && !(r.streamk && stride != 1);
    r.m32 = (order & kMfma32) && !r.streamk && !r.tg2 && stride == 1;
    // three taps per block (conv_wgrad3_dev.hip): a plan measured on a layer that admits it may reach one that does not
    // (the plan cache is keyed without stride and dilation): such a launch takes the two-tap kernel
    r.taps3 = (order & kTaps3) && !(order & kMfma32) && !r.streamk && stride == 1 && dil <= kTaps3MaxDil && Kw >= 3;
    r.kwb = r.taps3 ? 3 : (Kw > 1 ? KWB_DEFAULT : 1);          // taps per wave
    r.kwblk = r.tg2 ? 2 * r.kwb : r.kwb;                       // taps per block
    r.kgroups = (Kw + r.kwblk - 1) / r.kwblk;
    r.tiles = ((Cout + BM - 1) / BM) * ((Cin + BNC - 1) / BNC) * r.kgroups;
    r.G = 0;
    if (G > 0 && !r.streamk && have_ws && dealt_geometry_ok(r.tiles, total_steps, G) &&
        dealt_ws_need(r.tiles, r.kwblk, G) <= (size_t)(ws_bytes > 0 ? ws_bytes : 0) && (size_t)r.tiles * sizeof(unsigned) <= 64 * 1024) {
        r.G = G;
        r.splits = 1;
    }
    r.slabs = r.G > 0 || (r.splits > 1 && have_ws && !(order & kAtomicSplit) && (size_t)r.tiles * sizeof(unsigned) <= 64 * 1024 &&
                          wgrad_ws_need(Cin, Cout, Kw, r.splits) <= (size_t)(ws_bytes > 0 ? ws_bytes : 0));
    r.atomic = r.splits > 1 && !r.slabs;
    r.needs_zero = r.atomic || r.streamk;
    return r;
}

// with a workspace of ws_bytes: does the launch still add into dw with atomics (i.e. need a zero-filled dw)?
// (w2l_wgrad_needs_zero_ws assumes a stride-1, dilation-1 layer; w2l_wgrad_needs_zero_x is exact)
extern "C" int w2l_wgrad_needs_zero_x(int N, int Cin, int Cout, int Tout, int Kw, int stride, int dil, int64_t ws_bytes) {
    return wgrad_resolve(N, Cin, Cout, Tout, Kw, stride, dil, ws_bytes > 0, ws_bytes).needs_zero ? 1 : 0;
}

extern "C" int w2l_wgrad_needs_zero_ws(int N, int Cin, int Cout, int Tout, int Kw, int64_t ws_bytes) {
    return w2l_wgrad_needs_zero_x(N, Cin, Cout, Tout, Kw, 1, 1, ws_bytes);
}

extern "C" int64_t w2l_wgrad_workspace_bytes(int Cin, int Cout, int Kw) {
    return (int64_t)wgrad_ws_need(Cin, Cout, Kw, 32);      // 32 = the largest split count the tuner tries
}

// the kernel of a resolved plan: (taps per wave, stride-1 specialisation, stream-K, tap groups per block, 32x32x16 fragments,
// three taps with AGPR accumulators)
static int wgrad_dispatch(const WgradParams& p, dim3 grid, bool tg2, bool m32, bool taps3, int kwb, int stride, size_t lds, void* stream) {
    dim3 block(tg2 ? 512 : 256);
    // experiment switch: W2L_WGRAD_LDS_PAD=<bytes> of unused dynamic LDS per block of the two-tap kernels -- with > 40 KB only ONE
    // four-wave block fits a CU, which leaves half its register file to the main stream's kernels
    static const size_t lds_pad = getenv("W2L_WGRAD_LDS_PAD") ? (size_t)atoll(getenv("W2L_WGRAD_LDS_PAD")) : 0;
    if (!taps3 && lds + lds_pad <= 160 * 1024) lds += lds_pad;
#define W2L_WGRAD_LAUNCH(K, S1_, SK_)                                                                     \
    do {                                                                                                  \
        W2L_CHECK_HIP(w2l_allow_big_lds((const void*)conv_wgrad_kernel<K, S1_, SK_>));                    \
        hipLaunchKernelGGL((conv_wgrad_kernel<K, S1_, SK_>), grid, block, lds, (hipStream_t)stream, p);   \
    } while (0)
    if (taps3) {
        hipFunction_t fn;
        if (int e = wgrad3_function(&fn, tg2 ? 1 : 0)) return e;
        W2L_CHECK_ARG(lds <= 2 * BT * ROWB + 2 * (size_t)(tg2 ? 88 : 72) * ROWB, "conv1d_wgrad: the three-tap kernel's LDS window is too small");
        void* args[] = {const_cast<WgradParams*>(&p)};
        W2L_CHECK_HIP(hipModuleLaunchKernel(fn, grid.x, grid.y, 1, tg2 ? 512 : 256, 1, 1, 0, (hipStream_t)stream, args, nullptr));
    } else if (m32) {
        if (kwb == 2) {
            W2L_CHECK_HIP(w2l_allow_big_lds((const void*)conv_wgrad_kernel<2, true, false, 1, true>));
            hipLaunchKernelGGL((conv_wgrad_kernel<2, true, false, 1, true>), grid, block, lds, (hipStream_t)stream, p);
        } else {
            W2L_CHECK_HIP(w2l_allow_big_lds((const void*)conv_wgrad_kernel<1, true, false, 1, true>));
            hipLaunchKernelGGL((conv_wgrad_kernel<1, true, false, 1, true>), grid, block, lds, (hipStream_t)stream, p);
        }
    } else if (tg2 && p.streamk) {
        W2L_CHECK_HIP(w2l_allow_big_lds((const void*)conv_wgrad_kernel<2, true, true, 2>));
        hipLaunchKernelGGL((conv_wgrad_kernel<2, true, true, 2>), grid, block, lds, (hipStream_t)stream, p);
    } else if (tg2) {
        if (stride == 1) {
            W2L_CHECK_HIP(w2l_allow_big_lds((const void*)conv_wgrad_kernel<2, true, false, 2>));
            hipLaunchKernelGGL((conv_wgrad_kernel<2, true, false, 2>), grid, block, lds, (hipStream_t)stream, p);
        } else {
            W2L_CHECK_HIP(w2l_allow_big_lds((const void*)conv_wgrad_kernel<2, false, false, 2>));
            hipLaunchKernelGGL((conv_wgrad_kernel<2, false, false, 2>), grid, block, lds, (hipStream_t)stream, p);
        }
    } else if (p.streamk) {
        if (kwb == 2) { if (stride == 1) W2L_WGRAD_LAUNCH(2, true, true); else W2L_WGRAD_LAUNCH(2, false, true); }
        else { if (stride == 1) W2L_WGRAD_LAUNCH(1, true, true); else W2L_WGRAD_LAUNCH(1, false, true); }
    } else {
        if (kwb == 2) { if (stride == 1) W2L_WGRAD_LAUNCH(2, true, false); else W2L_WGRAD_LAUNCH(2, false, false); }
        else { if (stride == 1) W2L_WGRAD_LAUNCH(1, true, false); else W2L_WGRAD_LAUNCH(1, false, false); }
    }
#undef W2L_WGRAD_LAUNCH
    W2L_CHECK_LAUNCH();
    return 0;
}

extern "C" int w2l_conv1d_wgrad_ws(const void* dy, int64_t dy_bstride, const void* xp, int64_t x_bstride,
                                   int64_t x_rows_total, float* dw, int N, int Cin, int Cout, int Tout, int Kw,
                                   int stride, int dil, int accumulate, void* ws, int64_t ws_bytes, void* stream) {
    W2L_CHECK_ARG(dy && xp && dw, "conv1d_wgrad: null pointer");
    W2L_CHECK_ARG(N > 0 && Tout > 0 && Kw > 0 && stride > 0 && dil > 0, "conv1d_wgrad: bad sizes");
    W2L_CHECK_ARG(Cin % 64 == 0 && Cout % 64 == 0 && Cin > 0 && Cout > 0,
                  "conv1d_wgrad: channels (%d,%d) must be positive multiples of 64", Cin, Cout);
    W2L_CHECK_ARG(dy_bstride % Cout == 0 && x_bstride % Cin == 0, "conv1d_wgrad: batch strides must be whole rows");
    W2L_CHECK_ARG(x_rows_total * (int64_t)Cin * 2 < (1LL << 32), "conv1d_wgrad: activation buffer exceeds 32-bit byte offsets");
    WgradParams p;
    p.nlayers = 1;
    WgradLayer& l0 = p.layers[0];
    l0.dy = (const bf16_raw*)dy;
    l0.x = (const bf16_raw*)xp;
    l0.dw = dw;
    l0.dy_rows_per_utt = dy_bstride / Cout;
    l0.x_rows_per_utt = x_bstride / Cin;
    l0.x_max_row = x_rows_total - 1;
    l0.Cin = Cin; l0.Cout = Cout; l0.Kw = Kw;
    l0.tiles_m = (Cout + BM - 1) / BM;
    l0.tiles_n = (Cin + BNC - 1) / BNC;
    l0.tile0 = 0;
    p.N = N; p.Tout = Tout; p.stride = stride; p.dil = dil;
    const WgradResolved rs = wgrad_resolve(N, Cin, Cout, Tout, Kw, stride, dil, ws != nullptr, ws_bytes);
    const int order = rs.order, splits = rs.splits;
    const bool tg2 = rs.tg2, m32 = rs.m32, taps3 = rs.taps3;
    const int kwb = rs.kwb, kwblk = rs.kwblk;
    p.tsteps = rs.tsteps;
    p.order = order & 1;
    p.streamk = rs.streamk ? 1 : 0;
    p.total_steps = N * p.tsteps;
    p.steps_per_split = (p.total_steps + splits - 1) / splits;
    p.splits = splits;
    p.accumulate = accumulate;
    p.tickets = rs.slabs ? (unsigned*)ws : nullptr;
    p.slabs = rs.slabs ? (float*)((char*)ws + kWgradTicketBytes) : nullptr;
    p.atomic = rs.atomic;
    l0.kgroups = rs.kgroups;
    p.tiles_total = rs.tiles;
    p.dealt = p.dealt_b = 0;
    const int xr = (BT - 1) * stride + (kwblk - 1) * dil + 1;
    p.xrows_lds = (xr + 3) & ~3;
    const size_t lds = 2 * BT * ROWB + 2 * (size_t)p.xrows_lds * ROWB;
    W2L_CHECK_ARG(lds <= 160 * 1024, "conv1d_wgrad: stride %d / dilation %d need %zu bytes of LDS", stride, dil, lds);
    dim3 grid(p.tiles_total, splits), block(tg2 ? 512 : 256);
    g_last_dealt = false;
    if (rs.G > 0) {
        W2L_CHECK_ARG(dealt_fill(p, rs.tiles, p.total_steps, rs.G), "conv1d_wgrad: no dealt form for %d tiles x %d steps on %d ranges",
                      rs.tiles, p.total_steps, rs.G);
        grid = dim3((unsigned)(p.dealt + p.dealt_b), 1);
        g_last_dealt = true;
    }
    if (p.streamk) {
        // one block per resident slot, but at least ~16 K steps each (short ranges are all prologue and epilogue)
        const int64_t W = (int64_t)grid.x * p.total_steps;
        W2L_CHECK_ARG(W < (1LL << 31), "conv1d_wgrad: (tile, step) space exceeds 32 bits");
        int64_t g = W / 16;
        const int64_t resident = tg2 ? kResidentBlocks / 2 : kResidentBlocks;
        if (g > resident) g = resident;
        if (g < 1) g = 1;
        grid = dim3((unsigned)g, 1);
    }
    return wgrad_dispatch(p, grid, tg2, m32, taps3, kwb, stride, lds, stream);
}

// ---- a GROUP of layers in one launch (WgradLayer): their tiles are one pool of equal-shaped work items.  Plain stores, one
// block per tile (no split: a group exists to fill the chip without one).  form: plan order bits 0 (block order inside a
// layer), 2 (two tap groups per 8-wave block), 3 (32x32x16 fragments), 4 (three taps per wave, AGPR accumulators).
extern "C" int w2l_wgrad_group_tiles(int Cin, int Cout, int Kw, int form) {
    const int kwb = (form & kTaps3) ? 3 : (Kw > 1 ? KWB_DEFAULT : 1);
    const int kwblk = (form & kTapGroups2) ? 2 * kwb : kwb;
    return ((Cout + BM - 1) / BM) * ((Cin + BNC - 1) / BNC) * ((Kw + kwblk - 1) / kwblk);
}

extern "C" int w2l_conv1d_wgrad_group(const w2l_wgrad_item_t* items, int nitems, int N, int Tout, int dil, int form, void* stream) {
    W2L_CHECK_ARG(items && nitems >= 1 && nitems <= W2L_WGRAD_MAX_LAYERS, "conv1d_wgrad_group: 1..%d layers", W2L_WGRAD_MAX_LAYERS);
    W2L_CHECK_ARG(N > 0 && Tout > 0 && dil > 0, "conv1d_wgrad_group: bad sizes");
    const bool taps3 = (form & kTaps3) != 0, tg2 = (form & kTapGroups2) != 0, m32 = (form & kMfma32) != 0 && !taps3 && !tg2;
    W2L_CHECK_ARG(!(form & (kStreamK | kDealt | kAtomicSplit)), "conv1d_wgrad_group: block forms only (order bits 0, 2, 3, 4)");
    W2L_CHECK_ARG(!taps3 || dil <= kTaps3MaxDil, "conv1d_wgrad_group: the three-tap form takes dilation <= %d", kTaps3MaxDil);
    WgradParams p;
    const int kwb = taps3 ? 3 : KWB_DEFAULT, kwblk = tg2 ? 2 * kwb : kwb;
    int tiles = 0;
    for (int i = 0; i < nitems; ++i) {
        const w2l_wgrad_item_t& it = items[i];
        W2L_CHECK_ARG(it.dy && it.xp && it.dw, "conv1d_wgrad_group: null pointer (layer %d)", i);
        W2L_CHECK_ARG(it.Cin % 64 == 0 && it.Cout % 64 == 0 && it.Cin > 0 && it.Cout > 0 && it.Kw > 0,
                      "conv1d_wgrad_group: channels (%d,%d) must be positive multiples of 64", it.Cin, it.Cout);
        W2L_CHECK_ARG(it.Kw > 1 || kwb == KWB_DEFAULT || taps3, "conv1d_wgrad_group: bad form");
        W2L_CHECK_ARG(it.dy_bstride % it.Cout == 0 && it.x_bstride % it.Cin == 0, "conv1d_wgrad_group: batch strides must be whole rows");
        W2L_CHECK_ARG(it.x_rows_total * (int64_t)it.Cin * 2 < (1LL << 32), "conv1d_wgrad_group: activation buffer exceeds 32-bit byte offsets");
        WgradLayer& l = p.layers[i];
        l.dy = (const bf16_raw*)it.dy;
        l.x = (const bf16_raw*)it.xp;
        l.dw = it.dw;
        l.dy_rows_per_utt = it.dy_bstride / it.Cout;
        l.x_rows_per_utt = it.x_bstride / it.Cin;
        l.x_max_row = it.x_rows_total - 1;
        l.Cin = it.Cin; l.Cout = it.Cout; l.Kw = it.Kw;
        l.tiles_m = (it.Cout + BM - 1) / BM;
        l.tiles_n = (it.Cin + BNC - 1) / BNC;
        l.kgroups = (it.Kw + kwblk - 1) / kwblk;
        l.tile0 = tiles;
        tiles += l.tiles_m * l.tiles_n * l.kgroups;
    }
    p.nlayers = nitems;
    p.tiles_total = tiles;
    p.N = N; p.Tout = Tout; p.stride = 1; p.dil = dil;
    p.tsteps = (Tout + BT - 1) / BT;
    p.total_steps = N * p.tsteps;
    W2L_CHECK_ARG((int64_t)tiles * p.total_steps < (1LL << 31), "conv1d_wgrad_group: (tile, step) space exceeds 32 bits");
    p.steps_per_split = p.total_steps;
    p.splits = 1;
    p.accumulate = 0;
    p.atomic = 0;
    p.order = form & 1;
    p.streamk = 0;
    p.slabs = nullptr;
    p.tickets = nullptr;
    p.dealt = p.dealt_b = 0;
    const int xr = (BT - 1) + (kwblk - 1) * dil + 1;
    p.xrows_lds = (xr + 3) & ~3;
    const size_t lds = 2 * BT * ROWB + 2 * (size_t)p.xrows_lds * ROWB;
    W2L_CHECK_ARG(lds <= 160 * 1024, "conv1d_wgrad_group: dilation %d needs %zu bytes of LDS", dil, lds);
    g_last_dealt = false;
    return wgrad_dispatch(p, dim3(tiles, 1), tg2, m32, taps3, kwb, 1, lds, stream);
}

extern "C" int w2l_conv1d_wgrad(const void* dy, int64_t dy_bstride, const void* xp, int64_t x_bstride,
                                int64_t x_rows_total, float* dw, int N, int Cin, int Cout, int Tout, int Kw,
                                int stride, int dil, int accumulate, void* stream) {
    return w2l_conv1d_wgrad_ws(dy, dy_bstride, xp, x_bstride, x_rows_total, dw, N, Cin, Cout, Tout, Kw, stride, dil, accumulate,
                               nullptr, 0, stream);
}

// Measure candidate split-K factors for this problem on the caller's device and remember the fastest
// (SYNCHRONISING; warm-up only).  `dw_scratch` is a throw-away [Kw][Cout][Cin] fp32 buffer.
// flags bit 0: classic split plans are measured (and will run) with fp32 atomics although a workspace is given -- the
// workspace then only serves the dealt stream-K plans (the default mode of the step engine; without the bit every split
// reduction goes through slabs: W2L_DETERMINISTIC=1)
extern "C" int w2l_conv1d_wgrad_tune_x(const void* dy, int64_t dy_bstride, const void* xp, int64_t x_bstride,
                                       int64_t x_rows_total, float* dw_scratch, int N, int Cin, int Cout, int Tout, int Kw,
                                       int stride, int dil, int reps, void* ws, int64_t ws_bytes, int flags, void* stream) {
    const WShapeKey key(N, Cin, Cout, Tout, Kw);
    {
        std::lock_guard<std::mutex> lock(g_wtuned_mu);
        if (g_wtuned.count(key)) return 0;
    }
    hipEvent_t e0, e1;
    W2L_CHECK_HIP(hipEventCreate(&e0));
    W2L_CHECK_HIP(hipEventCreate(&e1));
    hipStream_t st = (hipStream_t)stream;
    const int total = N * ((Tout + BT - 1) / BT);
    const int cands[] = {1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 24, 32};
    if (ws) (void)hipMemsetAsync(ws, 0, kWgradTicketBytes, st);
    if (reps < 1) reps = 1;
    const size_t bytes = (size_t)Kw * Cout * Cin * sizeof(float);
    const int ncand = 2 * (int)(sizeof(cands) / sizeof(cands[0]));
    // time `n` back-to-back launches of plan (s, order) after one warm-up launch (which also validates it); < 0: it did not run
    auto time_plan = [&](int s, int order, int n) -> float {
        g_force_splits = s;
        g_force_order = order;
        const bool zero = wgrad_resolve(N, Cin, Cout, Tout, Kw, stride, dil, ws != nullptr, ws_bytes).needs_zero;    // atomics need a zero-filled dw
        if (zero) (void)hipMemsetAsync(dw_scratch, 0, bytes, st);
        int rc = w2l_conv1d_wgrad_ws(dy, dy_bstride, xp, x_bstride, x_rows_total, dw_scratch, N, Cin, Cout, Tout, Kw, stride,
                                     dil, 0, ws, ws_bytes, stream);
        if (rc != 0) return -1.f;
        if ((order & kDealt) && !g_last_dealt) return -1.f;      // this form has no dealt geometry here: the launch was its fallback
        (void)hipEventRecord(e0, st);
        for (int r = 0; r < n; ++r) {
            if (zero) (void)hipMemsetAsync(dw_scratch, 0, bytes, st);       // that fill is part of the launch's cost
            w2l_conv1d_wgrad_ws(dy, dy_bstride, xp, x_bstride, x_rows_total, dw_scratch, N, Cin, Cout, Tout, Kw, stride, dil, 0,
                                ws, ws_bytes, stream);
        }
        (void)hipEventRecord(e1, st);
        if (hipEventSynchronize(e1) != hipSuccess) return -1.f;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess) return -1.f;
        // Plans that ADD their partial tiles with fp32 atomics are priced above their stand-alone time: back to back, the next
        // launch's zero fill runs under this launch's tail (in the step it is one more dependent launch in front of its kernel:
        // 8-9 us), and beside the data gradients of the step the atomics themselves run slower than alone -- 768 -> 896 at split 3
        // measured level with the unsplit six-tap form alone (picked in two of five selections) and cost the step +0.09 ms = 17 %
        // of that layer (tools/probe/plan_cmp.sh).  The narrow layers' split plans win by 30-100 % and are not affected.
        if (zero) ms = ms * 1.08f + 0.008f * n;
        return ms;
    };
    std::vector<std::pair<float, int>> timed;                 // (ms, split count | order << 16)
    // (a stream-K form of the three-tap kernel was built and measured 10-15 % behind its classic form on every shape of the
    // table -- nearly every block then adds its output atomically, and its K loop reloads ~30 spilled scalars per step --:
    // dropped; order bits 4 + 1 take the two-tap stream-K kernel)
    const int variants[] = {0, kTapGroups2, kMfma32, kTaps3, kTaps3 | kTapGroups2};
    static const bool no_taps3 = getenv("W2L_WGRAD_NO_TAPS3") != nullptr;        // A/B switch of the measured selection
    for (int vi = 0; vi < 5; ++vi) {
        const int tgbit = variants[vi];
        if ((tgbit == kTapGroups2 && Kw <= 2) || (tgbit == kMfma32 && stride != 1)) continue;
        if ((tgbit & kTaps3) && (no_taps3 || stride != 1 || dil > kTaps3MaxDil || Kw < 3 || ((tgbit & kTapGroups2) && Kw <= 3))) continue;
        const bool has_sk = tgbit == 0 || (tgbit == kTapGroups2 && stride == 1);
        for (int ci = has_sk ? -2 : 0; ci < ncand; ++ci) {
            // ci = -2, -1: the stream-K decomposition in both block orders (no workspace form: skipped in deterministic mode);
            // tgbit: the same split counts and block orders once more with two tap groups per block (the 8-wave kernel: classic
            // and stream-K -- 256 persistent blocks, the balanced form of a kernel whose 1-block-per-CU tiles quantise badly),
            // once more with 32x32x16 MFMA fragments, and with three taps per wave (4- and 8-wave blocks)
            const bool sk = ci < 0;
            if (sk && ws != nullptr && !(flags & 1)) continue;
            const int s = sk ? 1 : cands[ci >> 1];
            int order = sk ? (kStreamK | (ci & 1) | tgbit) : ((ci & 1) | tgbit);
            if ((sk || s > 1) && ws != nullptr && (flags & 1)) order |= kAtomicSplit;     // atomics although a workspace is there
            if (!sk && (s > total || s > 0xffff || (s > 1 && total / s < 4))) break;
            const float ms = time_plan(s, order, reps);
            if (ms >= 0.f) timed.emplace_back(ms, s | (order << 16));
        }
        // dealt stream-K (needs the workspace): one range per resident block slot -- 512 for the 4-wave forms (also tried:
        // 256, one block per CU at a time), 256 for the 8-wave forms -- in both block orders
        if (ws != nullptr) {
            const bool waves8 = (tgbit & kTapGroups2) != 0;
            const int ranges[2] = {waves8 ? kResidentBlocks / 2 : kResidentBlocks, waves8 ? 0 : kResidentBlocks / 2};
            for (int gi = 0; gi < 2; ++gi)
                for (int bo = 0; bo < 2 && ranges[gi] > 0; ++bo) {
                    const int order = bo | tgbit | kDealt;
                    const float ms = time_plan(ranges[gi], order, reps);
                    if (ms >= 0.f) timed.emplace_back(ms, ranges[gi] | (order << 16));
                }
        }
    }
    // (the first pass ranks ~100 plans on `reps` launches each while the chip's clock drifts: the kFinalists fastest are timed again,
    // interleaved: common.h)
    std::sort(timed.begin(), timed.end());
    int best = timed.empty() ? -1 : timed[0].second;
    const int finalists = timed.size() < kFinalists ? (int)timed.size() : kFinalists;
    if (finalists > 1) {
        float total_ms[kFinalists] = {};
        for (int round = 0; round < kFinalRounds; ++round)
            for (int k = 0; k < finalists; ++k) {
                const float ms = time_plan(timed[k].second & 0xffff, timed[k].second >> 16, 4 * reps);
                total_ms[k] += ms >= 0.f ? ms : 1e30f;
            }
        int kb = 0;
        for (int k = 1; k < finalists; ++k)
            if (total_ms[k] < total_ms[kb]) kb = k;
        best = timed[kb].second;
    }
    g_force_splits = 0;
    g_force_order = -1;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    W2L_CHECK_ARG(best >= 1, "conv1d_wgrad_tune: no candidate ran");
    std::lock_guard<std::mutex> lock(g_wtuned_mu);
    g_wtuned[key] = best;
    return 0;
}

extern "C" int w2l_conv1d_wgrad_tune_ws(const void* dy, int64_t dy_bstride, const void* xp, int64_t x_bstride,
                                        int64_t x_rows_total, float* dw_scratch, int N, int Cin, int Cout, int Tout, int Kw,
                                        int stride, int dil, int reps, void* ws, int64_t ws_bytes, void* stream) {
    return w2l_conv1d_wgrad_tune_x(dy, dy_bstride, xp, x_bstride, x_rows_total, dw_scratch, N, Cin, Cout, Tout, Kw, stride, dil,
                                   reps, ws, ws_bytes, 0, stream);
}

extern "C" int w2l_conv1d_wgrad_tune(const void* dy, int64_t dy_bstride, const void* xp, int64_t x_bstride,
                                     int64_t x_rows_total, float* dw_scratch, int N, int Cin, int Cout, int Tout, int Kw,
                                     int stride, int dil, int reps, void* stream) {
    return w2l_conv1d_wgrad_tune_ws(dy, dy_bstride, xp, x_bstride, x_rows_total, dw_scratch, N, Cin, Cout, Tout, Kw, stride, dil,
                                    reps, nullptr, 0, stream);
}

W2L_DIAG_WGRAD_EXPORTS

// Tuning-cache (de)serialisation used by w2l_tune_save / w2l_tune_load (runtime.hip).
void w2l_wgrad_tune_dump(FILE* f) {
    std::lock_guard<std::mutex> lock(g_wtuned_mu);
    for (const auto& kv : g_wtuned) {
        const WShapeKey& k = kv.first;
        fprintf(f, "wgrad %d %d %d %d %d %d %d\n", std::get<0>(k), std::get<1>(k), std::get<2>(k), std::get<3>(k),
                std::get<4>(k), kv.second & 0xffff, kv.second >> 16);
    }
}

bool w2l_wgrad_tune_put(const int* v) {          // v[0..4] = key, v[5] = split count, v[6] = block order
    if (v[5] < 1 || v[5] > 0xffff || v[0] < 1 || v[3] < 1 || v[6] < 0 || v[6] > kOrderMask || ((v[6] & 8) && (v[6] & 6)) || ((v[6] & 16) && (v[6] & 10)))
        return false;
    const int ts = (v[3] + BT - 1) / BT;
    if (v[6] & kDealt) {                      // the split field of a dealt plan is its range count
        if ((v[6] & kStreamK) || v[5] > W2L_WGRAD_MAX_RANGES) return false;
    } else if ((int64_t)v[5] > (int64_t)v[0] * ts) return false;
    std::lock_guard<std::mutex> lock(g_wtuned_mu);
    g_wtuned[WShapeKey(v[0], v[1], v[2], v[3], v[4])] = v[5] | (v[6] << 16);
    return true;
}
