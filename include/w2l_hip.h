/*
 * w2l_hip.h -- C ABI of libw2l_hip.so: the MI355X (gfx950) kernels under the
 * Wav2Letter / Jasper forward + CTC + backward training step.
 *
 * The reference (assafmu/wav2letter_pytorch) is 100% Python and has no native
 * interface of its own; every entry point below replaces the torch ATen op the
 * reference calls at the cited site.  A binding only needs plain pointers and
 * sizes (ctypes stub: wav2letter_pytorch_amd/_lib.py).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - activations are channels-last "NTC" [N][rows][C] with C the padded channel
 *     count (multiple of 64, pad channels are zero); bf16 unless stated;
 *   - `stream` is a hipStream_t passed as void*; all launches are asynchronous on
 *     it, nothing synchronises, nothing is allocated or freed;
 *   - return 0 on success, otherwise a hipError_t / 1 for bad arguments, with a
 *     message retrievable through w2l_last_error() (thread-local);
 *   - re-entrant: may be called concurrently from the Python main thread
 *     (forward) and autograd worker threads (backward).
 */
#ifndef W2L_HIP_H
#define W2L_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

const char* w2l_last_error(void);
int w2l_abi_version(void);            /* 2 (1 -> 2: w2l_bnact_t gained the trailing field q_clipped) */

/* ---- layout / packing ---------------------------------------------------- */

/* fp32 weights w[co][ci][kw] (arbitrary element strides) ->
 *   w_fwd  [Kw][CoutP][CinP] bf16 (hi, and lo = residual if w_fwd_lo != NULL)
 *   w_dgr  [Kw][CinP][CoutP] bf16 with taps flipped: w_dgr[k][ci][co] = w[co][ci][Kw-1-k]
 * pad rows/cols are zero.  Replaces nothing in the reference; it is the operand
 * layout for nn.Conv1d (wav2letter.py:35-36, jasper.py:96-105). */
int w2l_pack_weights(const float* w, int64_t s_co, int64_t s_ci, int64_t s_kw, int Cout, int Cin, int Kw,
                     int CoutP, int CinP, void* w_fwd_hi, void* w_fwd_lo, void* w_dgr_hi, void* w_dgr_lo,
                     void* stream);

/* torch.optim.SGD(momentum, nesterov, weight_decay; dampening 0) -- configuration/optimizer/exp_lr_optimizer.yaml:2-7
 * -- fused with w2l_pack_weights for a tap-major conv weight: p, g, m are dense fp32 [Kw][Cout][Cin];
 * first_step != 0 initialises the momentum buffer with the (decayed) gradient as torch does.  The bf16
 * outputs are the operands of the next step (no channel padding: Cout, Cin multiples of 64).  zero_grad != 0 writes
 * zeros back into g once it has been read: the buffer can then serve as the next step's dw without a fill launch
 * (split-K weight gradients accumulate with atomics into a zeroed buffer, see w2l_wgrad_needs_zero).  w_fwd_q / w_dgr_q
 * (optional, fp8 mode): the same two operands once more as e4m3 bytes, value * q_scale, so that the forward and the data
 * gradient of the next step need no quantisation pass over the weights. */
int w2l_sgd_pack(float* p, float* g, float* m, int first_step, float lr, float momentum, float weight_decay,
                 int nesterov, int zero_grad, int Cout, int Cin, int Kw, void* w_fwd_hi, void* w_fwd_lo, void* w_dgr_hi,
                 void* w_dgr_lo, void* w_fwd_q, void* w_dgr_q, float q_scale, void* stream);

/* input spectrogram fp32 [N][C][T] -> padded NTC bf16 [N][pad_l+T+pad_r][CP];
 * pad_mode 1 = reflect (nn.ReflectionPad1d, wav2letter.py:28-34,41), 0 = zeros
 * (Conv1d padding=, jasper.py:96-105).  lens (optional, [N] int32): rows t >= lens[n]
 * are zeroed first (MaskedConv1d, jasper.py:114-119). */
int w2l_nct_to_ntc(const float* x, int N, int C, int T, int CP, int pad_l, int pad_r, int pad_mode,
                   const int32_t* lens, void* out_hi, void* out_lo, void* stream);

/* "Shared-halo" gradient layout used for every dy buffer: rows = halo + N*(T + halo):
 *   [halo zero rows][utt 0: T rows][halo zero rows][utt 1: T rows] ... [utt N-1][halo zero rows]
 * One zero gap serves as trailing halo of utterance n and leading halo of n+1, so the data-gradient
 * convolution can run over the whole buffer as ONE long sequence (no per-utterance tile rounding) while
 * the weight-gradient kernel still sees zero rows after each utterance.
 *
 * fp32 dense [N][T][C] -> shared-halo bf16 [halo + N*(T+halo)][CP] (+ optional lo), and per-channel
 * column sums colsum[CP] (bias gradient of an un-normalised conv: the 1x1 classifier,
 * wav2letter.py:69, jasper.py:432-433). */
int w2l_pad_cast(const float* g, int N, int T, int C, int CP, int halo, void* out_hi, void* out_lo, float* colsum,
                 void* stream);

/* ---- Conv1d as implicit GEMM on MFMA (nn.Conv1d fwd: wav2letter.py:42, jasper.py:127;
 *      its dgrad: autograd of the same call sites) ---------------------------------
 * y[n][t][co] (+)= bias[co] + sum_{kw,ci} w[kw][co][ci] * xp[n][t*stride + kw*dil][ci]
 * xp is a physically padded buffer (reflect or zero halo written by the producer);
 * x_rows_total = number of rows of Cin elements readable from xp (reads are clamped).
 * y is dense [N][Tout][Cout] bf16 (y_f32=0) or fp32 (y_f32=1; accumulate=1 adds to y).
 * stats_partial (optional) [w2l_conv_stat_tiles(N,Tout)][2][Cout]: per column-tile sums of y
 * and y^2 over valid t (BatchNorm batch statistics, wav2letter.py:37,43).
 * dgrad = the same call with xp := zero-haloed dy, w := w_dgr, Cin<->Cout, stride 1. */
int w2l_conv_stat_tiles(int N, int Tout);
int w2l_conv1d_igemm(const void* xp, int64_t x_bstride, int64_t x_rows_total, const void* w, void* y, int y_f32,
                     int accumulate, const float* bias, float* stats_partial, int N, int Cin, int Cout, int Tout,
                     int Kw, int stride, int dil, void* stream);

/* Autotune: time every feasible block shape for this exact problem on the caller's device (HIP events on
 * `stream`, SYNCHRONISING) and remember the fastest for later w2l_conv1d_igemm calls of the same shape.
 * Call once per shape during warm-up; no-op when already tuned.  The output is written like a normal
 * accumulate=0 launch. */
int w2l_conv1d_igemm_tune(const void* xp, int64_t x_bstride, int64_t x_rows_total, const void* w, void* y, int y_f32,
                          const float* bias, float* stats_partial, int N, int Cin, int Cout, int Tout, int Kw, int stride,
                          int dil, int reps, void* stream);

/* The same two entry points with a split-K workspace (device memory, owned by the caller, >= 64 KiB, ZERO-FILLED once
 * when it is allocated; one workspace per stream of execution -- launches that may overlap must not share it).  With
 * it, the tuner may split the (chunk, tap) reduction of a tile over 2-8 blocks where one block per tile would leave CUs
 * idle (a partly filled last round -- the flat data gradient has N*(T + halo) rows, never a round number of tiles -- or
 * small batches with fewer tiles than CUs): partial tiles go through fp32 slabs in the workspace, the block that draws a
 * tile's last ticket sums them in split order (bit-reproducible) and runs the normal epilogue; no block waits for another.
 * w2l_conv_splitk_workspace_bytes: a size with which every configuration of the problem is available (any smaller size
 * just removes the split configurations that do not fit). */
int w2l_conv1d_igemm_ws(const void* xp, int64_t x_bstride, int64_t x_rows_total, const void* w, void* y, int y_f32,
                        int accumulate, const float* bias, float* stats_partial, int N, int Cin, int Cout, int Tout, int Kw,
                        int stride, int dil, void* splitk_ws, int64_t splitk_ws_bytes, void* stream);
int w2l_conv1d_igemm_tune_ws(const void* xp, int64_t x_bstride, int64_t x_rows_total, const void* w, void* y, int y_f32,
                             const float* bias, float* stats_partial, int N, int Cin, int Cout, int Tout, int Kw, int stride,
                             int dil, int reps, void* splitk_ws, int64_t splitk_ws_bytes, void* stream);
int64_t w2l_conv_splitk_workspace_bytes(int N, int Cout, int Tout);
/* Stream-K configurations (indices 52 * 7 .. 52 * 8 - 1 of w2l_conv_force_tile_config; the tuner measures them beside the
 * split-K ones wherever one block per tile fills the chip's last round badly): the launch is one block per resident slot,
 * block r works on steps [W*r/G, W*(r+1)/G) of the tile-major (tile, step) space -- every block the same number of MFMA
 * steps whatever the tile count -- and a tile cut by a range boundary is combined like a split-K tile (slabs summed in
 * range order by the block that draws the tile's last ticket: deterministic).  Returns G for this problem and workspace,
 * 0 where the launch would fall back to one block per tile. */
int w2l_conv_streamk_ranges(int idx, int N, int Cin, int Cout, int Tout, int Kw, int stride, int dil, int64_t ws_bytes);
/* the decomposition itself, run on the host by the same function the kernel runs: the pieces of `tiles` x `steps` over G
 * ranges in range order, out[7 * i] = range, tile, first step, end step, ranges sharing the tile, place among them, slab id
 * (-1: whole tile, no slab); returns the piece count or -1 (cap too small / sizes beyond the kernel's 32-bit arithmetic) */
int w2l_conv_streamk_pieces(int tiles, int steps, int G, int* out, int cap);

/* nn.Conv1d forward (wav2letter.py:35-36,42 / jasper.py:96-105) on OCP e4m3 operands -- BASELINE config 5 "fp8 MFMA":
 *   y[n][t][co] = descale * sum_{kw,ci} wq[kw][co][ci] * xq[n][t + kw*dil][ci] (+ bias[co]),
 * xq / wq one byte per element, same layouts as the bf16 entry point (x_bstride in elements), Cin a multiple of 128,
 * stride 1; the products run on v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales (2x the bf16 MFMA rate), fp32
 * accumulate; descale = 1 / (activation scale * weight scale) undoes the producers' per-tensor scaling (descale_dev, optional:
 * a further factor read from device memory -- the inverse scale w2l_quantize_e4m3_dyn derived from a device-side amax, as
 * for the data gradient, where xq is the e4m3 copy of dy and wq the flipped-tap operand).  Output bf16 or
 * fp32, optional BatchNorm partial statistics as w2l_conv1d_igemm.  The _tune form measures the block shapes once per
 * shape (synchronising; warm-up only). */
int w2l_conv1d_igemm_fp8(const void* xq, int64_t x_bstride, int64_t x_rows_total, const void* wq, void* y, int y_f32,
                         float descale, const float* descale_dev, const float* bias, float* stats_partial, int N, int Cin,
                         int Cout, int Tout, int Kw, int dil, void* stream);
int w2l_conv1d_igemm_fp8_tune(const void* xq, int64_t x_bstride, int64_t x_rows_total, const void* wq, void* y, int y_f32,
                              const float* bias, float* stats_partial, int N, int Cin, int Cout, int Tout, int Kw, int dil,
                              int reps, void* stream);

/* tuning hook: force configuration idx (>= 0) for every later w2l_conv1d_igemm call MADE BY THE CALLING THREAD (the
 * setting is thread-local: launches from other threads are never affected); -1 = automatic.
 * idx = block shape (0..20) + 21 * K-loop structure (0: barrier at the top of a step, 1: barrier mid-step);
 * a call whose problem the forced configuration cannot run returns an error. */
void w2l_conv_force_tile_config(int idx);
/* likewise for the e4m3 kernel: idx = index into its list of 13 block shapes; an infeasible one (statistics need 128-row
 * tiles, LDS) makes w2l_conv1d_igemm_fp8 fail with a message */
void w2l_conv_force_fp8_config(int idx);

/* Conv1d weight gradient (autograd of the same call sites):
 * dw[kw][co][ci] (+)= sum_{n,t} dy[n][t][co] * xp[n][t*stride + kw*dil][ci]
 * dy: bf16, pointer at utterance 0 row 0, dy_bstride elements between utterances; rows
 * [Tout, roundup(Tout,64)) of every utterance must be zero (shared-halo layout with halo >= that).
 * dw fp32 [Kw][Cout][Cin].  accumulate=1 adds to dw (fp32 atomics).  With accumulate=0 the
 * library may still split the (n,t) reduction over blocks and combine with atomics: when
 * w2l_wgrad_needs_zero() != 0 the caller must zero-fill dw first. */
int w2l_wgrad_needs_zero(int N, int Cin, int Cout, int Tout, int Kw);
int w2l_conv1d_wgrad(const void* dy, int64_t dy_bstride, const void* xp, int64_t x_bstride, int64_t x_rows_total,
                     float* dw, int N, int Cin, int Cout, int Tout, int Kw, int stride, int dil, int accumulate,
                     void* stream);

/* The weight gradient with a split-K workspace (device memory owned by the caller, >= 64 KiB, zero-filled once when
 * allocated; one per stream of execution): the partial tiles of a split reduction go through fp32 slabs and the block
 * that draws a tile's last ticket sums them in split order and writes dw with plain stores -- no atomics, no zero-filled
 * dw, bit-reproducible gradients.  A workspace too small for the chosen split falls back to the atomic path:
 * w2l_wgrad_needs_zero_ws tells whether dw must be zero-filled for this launch. */
int w2l_conv1d_wgrad_ws(const void* dy, int64_t dy_bstride, const void* xp, int64_t x_bstride, int64_t x_rows_total, float* dw,
                        int N, int Cin, int Cout, int Tout, int Kw, int stride, int dil, int accumulate, void* ws,
                        int64_t ws_bytes, void* stream);
int w2l_conv1d_wgrad_tune_ws(const void* dy, int64_t dy_bstride, const void* xp, int64_t x_bstride, int64_t x_rows_total,
                             float* dw_scratch, int N, int Cin, int Cout, int Tout, int Kw, int stride, int dil, int reps,
                             void* ws, int64_t ws_bytes, void* stream);
int w2l_wgrad_needs_zero_ws(int N, int Cin, int Cout, int Tout, int Kw, int64_t ws_bytes);
int64_t w2l_wgrad_workspace_bytes(int Cin, int Cout, int Kw);

/* Dealt stream-K (round 5; plan order bit 5, the plan's split field = the number of ranges G): the (tile, K step) space of
 * the launch is cut into G equal ranges, one per resident block slot (512 / 256), so the chip is full whatever the tile
 * count.  A range crosses at most one tile boundary; each of its one or two segments is a block of its own (the kernel stays
 * the one-segment kernel): the first G blocks take the first segments, the following blocks the second segments LONGEST FIRST,
 * so the slot that frees first gets the longest remainder.  Partial tiles go through fp32 slabs of the caller's workspace,
 * the last arriver of a tile sums them in range order: no atomics, no zero-filled dw, bit-reproducible.  The plan needs
 * w2l_wgrad_dealt_workspace_bytes(); a launch without it (or of a shape without a dealt form) falls back to an atomic split.
 * w2l_conv1d_wgrad_tune_x = w2l_conv1d_wgrad_tune_ws with flags: bit 0 = classic split / stream-K plans are measured and run
 * with fp32 atomics although a workspace is given (plan order bit 6) -- the workspace then only serves dealt plans.
 * w2l_wgrad_needs_zero_x: must dw be zero-filled for this launch?  Exact (w2l_wgrad_needs_zero_ws assumes stride 1, dilation 1).
 * w2l_wgrad_dealt_segments (host only, testing): the blocks of a dealt launch in launch order, six ints each (range, tile,
 * first step, end step, place among the tile's segments, number of them); returns the block count, -1: no dealt form. */
int64_t w2l_wgrad_dealt_workspace_bytes(int Cin, int Cout, int Kw);
int w2l_conv1d_wgrad_tune_x(const void* dy, int64_t dy_bstride, const void* xp, int64_t x_bstride, int64_t x_rows_total,
                            float* dw_scratch, int N, int Cin, int Cout, int Tout, int Kw, int stride, int dil, int reps,
                            void* ws, int64_t ws_bytes, int flags, void* stream);
int w2l_wgrad_needs_zero_x(int N, int Cin, int Cout, int Tout, int Kw, int stride, int dil, int64_t ws_bytes);
int w2l_wgrad_dealt_segments(int tiles, int steps, int ranges, int* out, int max_blocks);
/* the plan a launch of this problem will take (measured, or the cost model's): order bits | split / range count << 8 */
int w2l_wgrad_plan(int N, int Cin, int Cout, int Tout, int Kw);

/* A GROUP of layers' weight gradients in ONE launch (round 5): layers of the same N, Tout, stride 1 and dilation (any Cin,
 * Cout, Kw) whose [128 co x 128 ci] x tap-group tiles form one pool of equal-shaped work items -- layers whose own tile
 * count fills a fraction of the chip's block slots fill whole rounds together, with no split, no partial tiles, no atomics
 * and no zero-filled dw (plain stores, accumulate = 0).  form = the block form (plan order bits 0: block order inside a
 * layer, 2: two tap groups per 8-wave block, 3: 32x32x16 fragments, 4: three taps per wave with AGPR accumulators).
 * Operand layouts per item as for w2l_conv1d_wgrad.  w2l_wgrad_group_tiles: the number of tiles a layer adds to the pool.
 * Replaces the weight half of aten::convolution_backward for several nn.Conv1d call sites at once
 * (wav2letter.py:35-36,42 / jasper.py:96-105,127). */
typedef struct {
    const void* dy;
    int64_t dy_bstride;
    const void* xp;
    int64_t x_bstride;
    int64_t x_rows_total;
    float* dw;
    int Cin, Cout, Kw, pad_;
} w2l_wgrad_item_t;
#define W2L_WGRAD_GROUP_MAX 8
int w2l_conv1d_wgrad_group(const w2l_wgrad_item_t* items, int nitems, int N, int Tout, int dil, int form, void* stream);
int w2l_wgrad_group_tiles(int Cin, int Cout, int Kw, int form);

/* Testing / profiling hook: pin the split count (0 = automatic; a dealt plan: the range count) and the plan order (-1 =
 * automatic; bit 5 = dealt stream-K, bit 6 = atomics although a workspace is given, see above; bit 0 = block order,
 * bit 1 = stream-K decomposition, bit 2 = two tap groups per 8-wave block, bit 3 = 32x32x16 MFMA tiles, bit 4 = THREE taps
 * per wave with the 192 accumulator registers in AGPRs (stride 1, dilation <= 4; alone: two 4-wave blocks per CU, with bit
 * 2: one 8-wave block of six taps per CU -- the kernels of csrc/conv_wgrad3_dev.hip, a code object of their own embedded in
 * the library); a combination the shape does not admit falls back to the nearest built variant) for launches made by the
 * calling thread (thread-local, like w2l_conv_force_tile_config). */
void w2l_wgrad_force_plan(int splits, int order);
/* W2L_DETERMINISTIC=1 (engine.py): on != 0 makes every later launch of w2l_conv1d_wgrad_ws (and w2l_wgrad_needs_zero_x) ignore
 * plan bit 6, i.e. a split reduction handed a workspace goes through slabs summed in a fixed order whatever the measured plan
 * (or a plan cache written by a default-mode run) says: bit-reproducible weight gradients.  Process-wide, not thread-local:
 * the weight gradients are launched from autograd's worker threads. */
void w2l_wgrad_deterministic(int on);

/* Autotune of the split-K factor and block order, like w2l_conv1d_igemm_tune (SYNCHRONISING, warm-up only); dw_scratch is a
 * throw-away [Kw][Cout][Cin] fp32 buffer.  Call before w2l_wgrad_needs_zero() / w2l_conv1d_wgrad for the shape. */
int w2l_conv1d_wgrad_tune(const void* dy, int64_t dy_bstride, const void* xp, int64_t x_bstride, int64_t x_rows_total,
                          float* dw_scratch, int N, int Cin, int Cout, int Tout, int Kw, int stride, int dil, int reps,
                          void* stream);

/* fp8 mode (BASELINE config 5): the weight gradient on OCP e4m3 operands, v_mfma_scale_f32_16x16x128_f8f6f4 with
 * ds_read_b64_tr_b8 fragment reads (stride 1; conv_wgrad_fp8.hip).  dyq / xq: the one-byte copies of dy and of the padded
 * input that the fp8 step already holds (w2l_quantize_e4m3_dyn / w2l_bn_act_fwd_q), same row layouts as the bf16 entry
 * point (batch strides in elements = bytes); dw = descale * (*descale_dev, if given) * sum dyq * xq, fp32, tap-major.
 * Split-K partial tiles are added atomically: w2l_wgrad_fp8_needs_zero tells whether dw must be zero-filled.
 * Replaces aten::convolution_backward's weight half (wav2letter.py:35-36,42 / jasper.py:96-105,127) in that mode. */
int w2l_wgrad_fp8_needs_zero(int N, int Cin, int Cout, int Tout, int Kw);
int w2l_conv1d_wgrad_fp8(const void* dyq, int64_t dy_bstride, const void* xq, int64_t x_bstride, int64_t x_rows_total,
                         float* dw, int N, int Cin, int Cout, int Tout, int Kw, int dil, float descale,
                         const float* descale_dev, int accumulate, void* stream);
int w2l_conv1d_wgrad_fp8_tune(const void* dyq, int64_t dy_bstride, const void* xq, int64_t x_bstride, int64_t x_rows_total,
                              float* dw_scratch, int N, int Cin, int Cout, int Tout, int Kw, int dil, int reps, void* stream);

/* Persistence of the measured choices of w2l_conv1d_igemm_tune / w2l_conv1d_wgrad_tune (a small text file; the
 * analogue of a vendor library's find-db).  save: 0 on success.  load: number of entries taken, -1 on error; lines
 * that do not describe a feasible launch of THIS build are skipped. */
int w2l_tune_save(const char* path);
int w2l_tune_load(const char* path);

/* ---- depthwise Conv1d, groups == channels (nn.Conv1d inside MaskedConv1d, jasper.py:96-105,127,319-330) ----
 * weights fp32 tap-major w[k][c]; activations channels-last bf16 hi [+ lo]; fp32 arithmetic; HBM-bound.
 * fwd:   y[n][t][c] = sum_k w[k][c] * xp[n][t*stride + k*dil][c]; frames t >= lens[n] are written as 0
 *        (the next MaskedConv1d's masked_fill).  xp: [N][x_rows][C] zero-padded by the producer.
 * dgrad: dxp[n][v][c] = sum_k w[k][c] * dy[n][v - k*dil][c] (stride 1), v in [0, Tp); dy [N][dy_rows][C] with
 *        Tout valid frames, frames >= lens[n] count as 0.
 * wgrad: dw[k][c] += sum_{n, t < min(Tout, lens[n])} dy[n][t][c] * xp[n][t*stride + k*dil][c] (fp32 atomics into
 *        a zero-filled dw). */
int w2l_dwconv_fwd(const void* x_hi, const void* x_lo, int x_rows, const float* w, void* y_hi, void* y_lo, int N, int Tout,
                   int C, int K, int stride, int dil, const int32_t* lens, void* stream);
int w2l_dwconv_dgrad(const void* dy, int dy_f32, int dy_rows, const float* w, void* dxp, int dxp_f32, int N, int Tp, int Tout,
                     int C, int K, int dil, const int32_t* lens, void* stream);
int w2l_dwconv_wgrad(const void* dy, int dy_f32, int dy_rows, const void* x_hi, const void* x_lo, int x_rows, float* dw, int N,
                     int Tout, int C, int K, int stride, int dil, const int32_t* lens, void* stream);

/* ---- BatchNorm1d + Dropout + activation (wav2letter.py:43-46, jasper.py:363,376,448) ---- */

/* partial [ntiles][2][C] -> batch mean / biased var -> scale = gamma*invstd, shift = beta - mean*scale;
 * running stats updated with the UNBIASED variance: r = (1-momentum)*r + momentum*batch.
 * If partial == NULL (eval mode) scale/shift come from the running stats. */
int w2l_bn_finalize(const float* partial, int ntiles, int C, int64_t count, const float* gamma, const float* beta,
                    float eps, float momentum, float* running_mean, float* running_var, float* mean, float* invstd,
                    float* scale, float* shift, void* stream);

typedef struct {
    int32_t N, T, C;        /* valid rows per utterance; padded channel count */
    const void* y;          /* conv output, dense [N][T][C] */
    int32_t y_f32;          /* 0: bf16, 1: fp32 */
    const float* scale;     /* NULL => identity (no BatchNorm) */
    const float* shift;
    const float* mean;      /* backward only */
    const float* invstd;
    const void* y2;         /* optional residual branch (jasper.py:400-410), same layout/dtype as y */
    const float* scale2;
    const float* shift2;
    const float* mean2;
    const float* invstd2;
    int32_t act;            /* 0 none, 1 clamp[0,20] (wav2letter.py:46), 2 ReLU (jasper.py:448) */
    float drop_p;           /* nn.Dropout p; 0 => no mask is read or written */
    uint64_t seed, offset;  /* Philox4x32-10 key / stream offset */
    uint8_t* mask;          /* keep bits, one byte per 8 channels: [N*T*C/8] */
    const int32_t* lens;    /* optional [N]: rows t >= lens[n] produce 0 / receive 0 gradient */
    const uint64_t* offset_dev; /* optional device word ADDED to `offset` when the mask is drawn: a step counter that lives
                               in device memory, so that a captured hipGraph of the step draws fresh masks at every replay */
    uint64_t* q_clipped;    /* optional (w2l_bn_act_fwd_q with out_q): device counter incremented by the number of activation
                               elements whose e4m3 copy SATURATED (|a * q_scale| > 448): the per-tensor activation scales of
                               fp8 mode are fixed, so an unbounded activation (ReLU after a residual sum) clips silently
                               otherwise.  NULL: not counted.  (ABI version 2) */
} w2l_bnact_t;

/* a = act(dropout(y*scale+shift [+ y2*scale2+shift2])) written to a padded buffer
 * [N][pad_l+T+pad_r+tail][C] for the NEXT conv: halo rows are reflected copies (pad_mode 1) or zeros. */
int w2l_bn_act_fwd(const w2l_bnact_t* d, void* out_hi, void* out_lo, int out_rows, int pad_l, int pad_r,
                   int pad_mode, void* stream);
/* the same with a second copy of the padded activation as OCP e4m3 bytes, a * q_scale (saturating at +-448): the
 * forward operand of the next nn.Conv1d in fp8 mode (BASELINE config 5).  out_q == NULL: plain w2l_bn_act_fwd. */
int w2l_bn_act_fwd_q(const w2l_bnact_t* d, void* out_hi, void* out_lo, void* out_q, float q_scale, int out_rows,
                     int pad_l, int pad_r, int pad_mode, void* stream);
/* w2l_bn_finalize + w2l_bn_act_fwd(_q) in ONE launch (round 5; bf16 y, C a multiple of 64): every block first sums the
 * partial statistics rows of its own 64 channels and derives mean / invstd / scale / shift exactly as w2l_bn_finalize does;
 * the blocks of the first row range publish them (f->mean, invstd, scale, shift: the backward pass reads them) and update
 * the running statistics.  Meant for FEW partial rows: w2l_conv_stats_mode(S) makes the convolutions fold their per-tile
 * statistics onto S rows.  One record per branch (f2 iff d->y2); a record with partial == NULL takes scale / shift from the
 * descriptor (a branch without BatchNorm, or eval mode after w2l_bn_finalize).  The descriptor's own scale / shift pointers
 * are ignored for a branch with a record.  Same output contract as w2l_bn_act_fwd_q (out_lo: none, bf16 mode only).
 * Replaces nn.BatchNorm1d's statistics + normalisation + Dropout + clamp / ReLU (+ residual add, length mask, the next
 * convolution's reflect padding): wav2letter.py:28-34,37-38,43-46 / jasper.py:116-119,363,376,409-410,448. */
typedef struct {
    const float* partial;   /* [rows][2][C]: sums, sums of squares (w2l_conv1d_igemm* statistics rows) */
    int32_t rows;
    int64_t count;          /* N * T: elements per channel */
    const float* gamma;     /* NULL: 1 */
    const float* beta;      /* NULL: 0 */
    float eps, momentum;
    float* running_mean;    /* NULL: not tracked */
    float* running_var;
    float* mean;            /* outputs, [C] each (mean / invstd may be NULL) */
    float* invstd;
    float* scale;
    float* shift;
} w2l_bnfin_t;
int w2l_bn_act_fwd_fin(const w2l_bnact_t* d, const w2l_bnfin_t* f1, const w2l_bnfin_t* f2, void* out_hi, void* out_q,
                       float q_scale, int out_rows, int pad_l, int pad_r, int pad_mode, void* stream);
/* How the convolutions lay out their statistics rows, for launches made by the calling thread (thread-local, like
 * w2l_conv_force_tile_config): 0 (default) = one row per 128-column tile, plain stores (w2l_conv_stat_tiles rows,
 * bit-reproducible); S = 1..64: the per-tile sums are ADDED (fp32 atomics) onto row (tile mod S) of a [S][2][C] buffer the
 * caller zero-filled -- a handful of rows that w2l_bn_act_fwd_fin / w2l_bn_act_bwd_apply_fin re-reduce per block instead of
 * a finalize launch of their own.  Applies to w2l_conv1d_igemm*, w2l_conv1d_igemm_fp8 and w2l_conv1d_dgrad_bnreduce_ws.
 * With S > 0 the bf16 kernels may take ANY block shape for a launch with statistics (a block adds its whole tile's sums
 * onto row (column tile mod S)); with 0 only the shapes of 128 or 256 columns, whose waves line up with the 128-column rows. */
void w2l_conv_stats_mode(int slots);

/* dst[i] = e4m3(src[i] * scale), src bf16 (src_f32 = 0) or fp32, n a multiple of 8: per-tensor quantisation of the conv
 * weights (and of the spectrogram) for w2l_conv1d_igemm_fp8. */
int w2l_quantize_e4m3(const void* src, int src_f32, int64_t n, float scale, void* dst, void* stream);

typedef struct {
    const void* dxp;        /* gradient wrt the padded activation buffer [N][rows][C], bf16 or fp32; the first
                               pad_l+T+pad_r rows of each utterance are valid */
    int32_t f32;
    int32_t pad_l, pad_r, pad_mode;   /* fold reflected halo rows back (mode 1) or skip the halo (mode 0) */
    int32_t rows;           /* rows per utterance in dxp (>= pad_l+T+pad_r) */
} w2l_gradsrc_t;

/* sums over (n,t) of g and g*xhat per channel for each branch:
 * partial [nblocks][ncomp][C] = {sum g, sum g*xhat1 [, sum g (branch2), sum g*xhat2]}; ncomp = 4 with a residual
 * branch (d->y2 != NULL), else 2; nblocks = w2l_bn_bwd_blocks(). */
int w2l_bn_bwd_blocks(int N, int T, int C);
int w2l_bn_act_bwd_reduce(const w2l_bnact_t* d, const w2l_gradsrc_t* g1, const w2l_gradsrc_t* g2, float* partial,
                          void* stream);
/* partial -> sums [ncomp][C] (sum_g = d beta, sum_gx = d gamma), written at the front of a [4][C] buffer */
int w2l_bn_bwd_finalize(const float* partial, int nblocks, int C, int ncomp, float* sums, void* stream);
/* dy = scale*(g - sum_g/M - xhat*sum_gx/M) into shared-halo buffers [halo + N*(T+halo)][C] (hi[,lo]);
 * dy2 likewise for the residual branch (NULL if none).  The conv bias gradient under BatchNorm is
 * sum(dy) == 0 identically (the reference's value is fp32 rounding noise); it is not computed. */
int w2l_bn_act_bwd_apply(const w2l_bnact_t* d, const w2l_gradsrc_t* g1, const w2l_gradsrc_t* g2, const float* sums,
                         void* dy_hi, void* dy_lo, int halo, void* dy2_hi, void* dy2_lo, int halo2, void* stream);
/* w2l_bn_bwd_finalize + w2l_bn_act_bwd_apply(_amax) in ONE launch: every block re-reduces the partial rows of its own 64
 * channels (fixed order: all blocks of a slab get bit-identical sums), so the finalize launch on the backward critical path
 * disappears; sums [ncomp][C] (d beta, d gamma, as w2l_bn_bwd_finalize writes them) are published as a by-product.
 * partial = what w2l_bn_act_bwd_reduce or w2l_conv1d_dgrad_bnreduce_ws produced; amax optional as below. */
int w2l_bn_act_bwd_apply_fin(const w2l_bnact_t* d, const w2l_gradsrc_t* g1, const w2l_gradsrc_t* g2, const float* partial,
                             int nblocks, float* sums, void* dy_hi, void* dy_lo, int halo, void* dy2_hi, void* dy2_lo,
                             int halo2, float* amax, void* stream);
/* The backward chain in TWO launches (round 5; the fast path: bf16 y, bf16 gradient from one source, one branch, BatchNorm
 * present, C a multiple of 64 -- w2l_bn_bwd_fast_ok says whether a unit qualifies): w2l_bn_act_bwd_reduce_slots ADDS the sums
 * of g*gate and g*gate*xhat onto `slots` rows of a zero-filled partial [slots][2][C] (fp32 atomics, one 2 x 64 row segment
 * per 128-row block), w2l_bn_act_bwd_apply_slots re-reduces those rows per block (the finalize folded in; sums [2][C] = d
 * beta, d gamma published as by w2l_bn_bwd_finalize) and writes dy into its shared-halo buffer (halo rows zero-filled).  Both
 * issue every load of a wave's rows before the first use.  amax as for w2l_bn_act_bwd_apply_amax (row 0 only). */
int w2l_bn_bwd_fast_ok(const w2l_bnact_t* d, const w2l_gradsrc_t* g1, const w2l_gradsrc_t* g2);
int w2l_bn_act_bwd_reduce_slots(const w2l_bnact_t* d, const w2l_gradsrc_t* g1, float* partial, int slots, void* stream);
int w2l_bn_act_bwd_apply_slots(const w2l_bnact_t* d, const w2l_gradsrc_t* g1, const float* partial, int nrows, float* sums,
                               void* dy_hi, int halo, float* amax, void* stream);
/* the same, also leaving max |dy| and max |dy2| in device memory -- the scale of the e4m3 copy of dy that the data gradient
 * reads in fp8 mode.  amax is [2][W2L_AMAX_SLOTS] floats, zeroed by the caller: slot (block index mod W2L_AMAX_SLOTS) of row
 * 0 (dy) / row 1 (dy2) takes an integer atomic max of the bit patterns; the tensor's amax is the max over a row's slots
 * (one word per tensor would serialise every wave of the launch on one L2 address). */
#define W2L_AMAX_SLOTS 64
int w2l_bn_act_bwd_apply_amax(const w2l_bnact_t* d, const w2l_gradsrc_t* g1, const w2l_gradsrc_t* g2, const float* sums,
                              void* dy_hi, void* dy_lo, int halo, void* dy2_hi, void* dy2_lo, int halo2, float* amax,
                              void* stream);
/* dst[i] = e4m3(src[i] * s), s = 2^floor(log2(224 / max(amax[0..W2L_AMAX_SLOTS)))) read from DEVICE memory (no host round
 * trip; s = 1 when amax is 0); inv_scale[0] = 1 / s for the consumer (w2l_conv1d_igemm_fp8 descale_dev).  src bf16, n a multiple of 8. */
int w2l_quantize_e4m3_dyn(const void* src_bf16, int64_t n, const float* amax, void* dst, float* inv_scale, void* stream);

/* The data gradient of a stride-1 nn.Conv1d FUSED with w2l_bn_act_bwd_reduce of the layer that produced the conv's input
 * (d describes that layer: bf16 y, one branch).  dxp [flat_rows][d->C] bf16 is the gradient wrt the padded activation,
 * computed over the shared-halo dy buffer as one sequence exactly like w2l_conv1d_igemm_ws(N = 1, Tout = flat_rows, w =
 * the flipped-tap operand); row v of it is padded row v % per of utterance v / per (per >= pad_l + d->T + pad_r).  The
 * epilogue, which holds that gradient in registers, also writes partial[tile][2][d->C] -- per 128-row tile
 * (w2l_conv_stat_tiles(1, flat_rows) rows) the sums of g*gate and g*gate*xhat, every padded row counted with the gate
 * and xhat of its source frame (itself, or its mirror under reflect padding) -- so the separate reduction pass over dxp
 * and y (and its launch on the backward critical path) disappears; w2l_bn_bwd_finalize(partial, tiles, C, 2, ...) follows.
 * Replaces the reduction half of autograd's batch_norm backward at wav2letter.py:43 / jasper.py:363. */
int w2l_conv1d_dgrad_bnreduce_ws(const void* dy, int64_t dy_rows_total, const void* w_dgr, void* dxp, float* partial,
                                 const w2l_bnact_t* d, int pad_l, int pad_r, int pad_mode, int per, int Cconv_out,
                                 int flat_rows, int Kw, int dil, void* splitk_ws, int64_t splitk_ws_bytes, void* stream);
/* measure-and-pick of the block shape for that launch (SYNCHRONISING; warm-up only) */
int w2l_conv1d_dgrad_bnreduce_tune_ws(const void* dy, int64_t dy_rows_total, const void* w_dgr, void* dxp, float* partial,
                                      const w2l_bnact_t* d, int pad_l, int pad_r, int pad_mode, int per, int Cconv_out,
                                      int flat_rows, int Kw, int dil, int reps, void* splitk_ws, int64_t splitk_ws_bytes,
                                      void* stream);

/* ---- log_softmax + CTC (wav2letter.py:86-87, jasper.py:469-473, base_asr_models.py:23,81,90) ---- */
/* logits fp32 [N][T][CP] (first C valid) -> out fp32 [N][T][C]; mode 0 log_softmax, 1 softmax */
int w2l_log_softmax_fwd(const float* logits, int N, int T, int C, int CP, int mode, float* out, void* stream);
/* grad wrt logits (dense [N][T][C]) from grad wrt out and out itself */
int w2l_log_softmax_bwd(const float* gout, const float* out, int N, int T, int C, int mode, float* glogits,
                        void* stream);
/* workspace bytes for w2l_ctc_loss */
int64_t w2l_ctc_workspace_bytes(int N, int T, int Smax);
/* nn.CTCLoss(blank, reduction='mean', zero_infinity): log_probs [N][T][C] fp32 (batch-major),
 * targets [N][Smax] int32 (padded), lengths int32 [N].  Outputs: nll[N] (0 where infinite and zero_infinity),
 * loss[1] = mean_n(nll_n / max(S_n,1)), grad [N][T][C] = d loss / d log_probs in torch's convention
 * (exp(lp) - posterior) * grad_scale / (N*max(S_n,1)), zero for t >= input_lengths[n]). */
int w2l_ctc_loss(const float* log_probs, const int32_t* targets, const int32_t* input_lengths,
                 const int32_t* target_lengths, int N, int T, int C, int Smax, int blank, int zero_infinity,
                 float* nll, float* loss, float* grad, void* workspace, void* stream);

/* ---- greedy decode (decoder.py:136) + Levenshtein (decoder.py:49,60) ---- */
/* argmax over the last dim, ties -> lowest index (torch.max): probs fp32 [rows][C] -> idx int32 [rows] */
int w2l_argmax(const float* probs, int64_t rows, int C, int32_t* idx, void* stream);
/* host-side edit distance over int32 symbol arrays */
int w2l_levenshtein_host(const int32_t* a_host, int na, const int32_t* b_host, int nb);
/* ConvCTCASR.add_string_metrics (base_asr_models.py:53-69) for one batch in ONE host call (no device work, no interpreter
 * lock held): idx int32 [N][T] = the step's argmax indices on the host, sizes[n] (NULL: T) the valid frames of utterance n.
 * Greedy collapse as GreedyDecoder.process_string(remove_repetitions=True) (decoder.py:104-119: blanks dropped, a frame equal
 * to the previous frame dropped), labels mapped through lut[n_labels] (one Unicode code point per label), then against the
 * reference transcripts ref_cp[ref_off[n] .. ref_off[n+1]) (code points): CER numerator = edit distance with ' ' removed
 * from both (decoder.py:51-63), denominator = len(expected without ' '); WER numerator = edit distance over the words of
 * str.split() (decoder.py:31-49,65-66), denominator = the number of expected words.  totals[5] = {cer_err, cer_ref, wer_err,
 * wer_ref, sum of decoded lengths}; hyp_cp (>= N*T ints) / hyp_off (N+1), both optional, receive the decoded code points. */
int w2l_greedy_score_host(const int32_t* idx_host, int N, int T, const int32_t* sizes_host, int blank, const int32_t* lut,
                          int n_labels, const int32_t* ref_cp, const int64_t* ref_off, int64_t* totals, int32_t* hyp_cp,
                          int64_t* hyp_off);

/* Novograd step of one conv weight (novograd.py:86-112) fused with the bf16 operand pack, tap-major [Kw][Cout][Cin] fp32
 * p / g / exp_avg like w2l_sgd_pack:  v = ||g||^2 on the first step (exp_avg_sq == 0), else beta2*v + (1-beta2)*||g||^2;
 * amsgrad (max_exp_avg_sq != NULL): v <- running max;  g' = g/(sqrt(v)+eps) + weight_decay*p;  g' *= (1-beta1) if
 * grad_averaging;  exp_avg = beta1*exp_avg + g';  p -= lr*exp_avg.  exp_avg_sq / max_exp_avg_sq are device scalars;
 * scratch: >= 2 floats of device memory (1025 lets the norm use 1024 blocks).  Three launches, no host sync. */
int w2l_novograd_pack(float* p, const float* g, float* exp_avg, float* exp_avg_sq, float* max_exp_avg_sq, float* scratch,
                      int scratch_floats, float lr, float beta1, float beta2, float eps, float weight_decay, int grad_averaging,
                      int Cout, int Cin, int Kw, void* w_fwd_hi, void* w_fwd_lo, void* w_dgr_hi, void* w_dgr_lo, void* stream);

/* ---- feature front-end (SpectrogramExtractor, data/data_loader.py:33-88) and augmentation masks ----------------
 * w2l_logmel: per utterance n (n_samples[n] samples of audio[n*audio_stride ...], fp32, device):
 *   x = audio + dither * noise (noise = N(0,1) draws, may be NULL: no dither)   data_loader.py:67
 *   x[i] -= preemph * x[i-1] for i >= 1                                         data_loader.py:68
 *   torch.stft(n_fft, hop, win_length, window, center=True [reflect pad n_fft/2]) -> 1 + n_samples/hop frames   :55-63,69
 *   power = (sqrt(re^2 + im^2))^2; mel = fb . power  (_get_spect's result: take_log == 0)            :70-72
 *   logmel = log1p(mel + log_guard)                  (take_log != 0)                                  :78-79
 * fbT is the filterbank transposed, [n_fft/2 + 1][n_mels]; fb_range (optional) [n_mels][2] int32 = first and one-past-last
 * non-zero bin of each filter (only that run is summed; NULL = all bins).  Output logmel[N][Tmax][n_mels]; frames past an utterance's
 * own count are written as 0.  Requires n_samples[n] > n_fft/2 (torch's reflect-pad rule), n_fft a power of two <= 1024.
 * w2l_feature_normalize: per (utterance, feature) mean and UNBIASED std over the utterance's frames, std += eps,
 *   out[n][m][t] = (logmel - mean) / std for t < frames(n), 0 beyond: the right-zero-padded batch layout of _collator
 *   (data_loader.py:80-88,149-158).  mean_ws / std_ws: [N][n_mels] fp32 workspaces (returned filled). */
int w2l_logmel(const float* audio, const int32_t* n_samples, const float* noise, float dither, float preemph, int N,
               int64_t audio_stride, const float* window, int win_length, int n_fft, int hop, const float* fbT,
               const int32_t* fb_range, int n_mels, int take_log, float log_guard, float* logmel, int Tmax, void* stream);
int w2l_feature_normalize(const float* logmel, const int32_t* n_samples, int hop, int N, int Tmax, int n_mels, float eps,
                          float* mean_ws, float* std_ws, float* out_nct, void* stream);
/* x[n][f0:f1][t0:t1] = 0 for each of R rectangles rects[r] = {n, f0, f1, t0, t1} (int32, device; clipped to the tensor):
 * the masked_fill of SpecAugment.forward / SpecCutout.forward (data/augmentations.py:56,97); x fp32 [N][C][T]. */
int w2l_zero_rects(float* x, int N, int C, int T, const int32_t* rects, int R, void* stream);

/* ---- stream concurrency probe -----------------------------------------------------------------------------------------
 * Launches a chip-filling spin kernel (`rounds` waves of 2 blocks per CU, `spin_us` each) on stream_a, then a one-wave
 * kernel on stream_b that records when it started.  stamps_dev: 3 x int64, zero before the call (caller synchronises
 * before and after): [0] start of the fill, [1] end of the fill, [2] start of the stamp kernel, in 100 MHz ticks.
 * (stamps[2] - stamps[0]) / (stamps[1] - stamps[0]) ~ 0: kernels of the two streams run side by side; ~ 1: they share a
 * hardware queue / pipe and serialise.  The host uses it to choose the side streams of the step (weight gradients,
 * optimizer updates, gradient collectives) -- see streams.py; no reference counterpart (stream placement is below torch). */
int w2l_stream_probe(void* stream_a, void* stream_b, void* stamps_dev, int rounds, int spin_us);

/* ---- RCCL helpers: the exchange step of the data-parallel path ------------------------------------------------------
 * One process per GPU; every rank holds a full replica and its own utterances, the only exchange is the average of the
 * parameter gradients (what PL's Trainer(gpus=N) / torch DDP would do for the reference, README.md:40; BatchNorm stays
 * per rank, wav2letter.py:37).  For hosts without torch.distributed: rank 0 draws an id, the host ships its
 * W2L_RCCL_ID_BYTES to the other ranks by whatever channel it has (file, socket, env), every rank calls w2l_rccl_init
 * with its device current (a COLLECTIVE over the ranks), then all-reduces each gradient buffer in place on a stream
 * of its choice (asynchronous, like every other launch here), and destroys the communicator at the end.
 * RCCL itself is resolved at run time (the copy already mapped into the process, else librccl.so.1): without it these
 * entry points -- and only these -- fail with a message.  Errors: 1 = bad argument / RCCL absent, 1000 + ncclResult_t.
 *   dtype: 0 = fp32, 1 = bf16; average != 0: ncclAvg (sum / world), else ncclSum.
 *   w2l_rccl_broadcast: raw bytes from `root` (identical replicas at step 0, DDP's construction-time broadcast). */
#define W2L_RCCL_ID_BYTES 128
int w2l_rccl_available(void);
const char* w2l_rccl_library(void);          /* path of the RCCL shared object in use ("" if none) */
int w2l_rccl_unique_id(void* id_host);
int w2l_rccl_init(const void* id_host, int rank, int world, void** comm_out);
int w2l_rccl_world(void* comm, int* world_out);
int w2l_rccl_all_reduce(void* comm, void* buf, int64_t count, int dtype, int average, void* stream);
int w2l_rccl_broadcast(void* comm, void* buf, int64_t bytes, int root, void* stream);
int w2l_rccl_destroy(void* comm);

/* ---- recorded launch lists (round 6) -----------------------------------------------------------------------------------
 * The reference's host loop is PyTorch's dispatcher (one Python -> ATen transition per op of wav2letter.py:40-47,84-92 and
 * of autograd's backward); the step engine here made one Python -> ctypes transition per launch, 200-500 per step.  After
 * the tuned warm-up the engine records what a step calls -- entry point, argument values, stream, and the event records /
 * waits between its streams -- and replays every phase of the step (forward, backward, optimizer, held-back weight
 * gradients) with ONE w2l_replay call: a C loop over the very same entry points.  Not a hipGraph: each launch goes to the
 * stream it was recorded on, so side streams and the cross-step overlap of the updates behave as in the eager step.
 *   w2l_slot_t    one argument: p for pointers (and pointers to the by-reference structs, whose copies the RECORDER owns),
 *                 i for every integer type, d for float / double (narrowed to the parameter's type at the call);
 *   w2l_call_t    op = w2l_replay_op("w2l_..."), nargs = that entry point's arity, a[] = its arguments in order;
 *   w2l_replay    runs calls[0..n) in order on the calling thread (thread-local library state -- w2l_conv_stats_mode -- is
 *                 that thread's); stops at the first failing call: returns its code, *failed_at = its index, the message
 *                 is the entry point's own (w2l_last_error).  Ownership as everywhere: the caller keeps every buffer the
 *                 recorded pointers name alive and unmoved for as long as it replays the list. */
typedef union { void* p; int64_t i; double d; } w2l_slot_t;
#define W2L_REPLAY_MAX_ARGS 24
typedef struct { int32_t op; int32_t nargs; w2l_slot_t a[W2L_REPLAY_MAX_ARGS]; } w2l_call_t;
int w2l_replay_op(const char* entry_point_name);      /* -1: not a replayable entry point */
int w2l_replay_arity(int op);
int w2l_replay(const w2l_call_t* calls, int n, int* failed_at);

/* Stream order as entry points, so that it can be recorded like a launch (the eager engine used torch.cuda.Event /
 * Stream.wait_event here): events are created without timing; w2l_event_query: 0 fired, 1 not yet;
 * w2l_stream_wait_stream(waiter, signaler): `waiter` waits for everything enqueued on `signaler` so far. */
int w2l_event_create(void** event_out);
int w2l_event_destroy(void* event);
int w2l_event_record(void* event, void* stream);
int w2l_stream_wait_event(void* stream, void* event);
int w2l_event_query(void* event);
int w2l_event_synchronize(void* event);
int w2l_stream_wait_stream(void* waiter, void* signaler);

/* What the eager engine left to torch ops between its launches, as entry points: tensor.zero_() of the statistics / slot /
 * gradient pools (hipMemsetAsync); the zero- / one-padded copy of a per-channel vector (conv bias of the 29-label classifier
 * padded to 64, wav2letter.py:69); `+= delta` on a device int64 (the dropout step counter of a replayed step: the Philox
 * offset of nn.Dropout, wav2letter.py:38,44, must move every step); num_batches_tracked += 1 of every BatchNorm1d of the
 * stack in one launch (table_dev: n device pointers to int64 scalars; wav2letter.py:37); torch.optim.SGD's update of all
 * the small parameters -- biases, BatchNorm gamma / beta -- in one launch (exp_lr_optimizer.yaml:2-7; dampening 0;
 * m == NULL: no momentum buffer). */
int w2l_fill_zero(void* p, int64_t bytes, void* stream);
int w2l_pad_vec_f32(const float* src, int n, float* dst, int cp, float fill, void* stream);
int w2l_counter_add(void* counter_i64, int64_t delta, void* stream);
int w2l_add_i64_multi(void* table_dev, int n, int64_t delta, void* stream);
typedef struct { float* p; const float* g; float* m; int32_t n; int32_t pad_; } w2l_sgd_small_t;
int w2l_sgd_small_multi(const w2l_sgd_small_t* items_dev, int nitems, int max_n /* the largest items[i].n */, float lr,
                        float momentum, float weight_decay, int nesterov, void* stream);

#ifdef __cplusplus
}
#endif
#endif
