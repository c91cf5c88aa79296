// Diagnostic builds only -- NEVER part of the shipped library.  The kernels of conv_igemm.hip / conv_wgrad.hip carry named hook
// points (W2L_DIAG_*; common.h defines every one of them empty); a diagnostic build puts bodies behind them by force-including
// this file:
//   make -C wav2letter_pytorch_amd/csrc BUILD=build_abl4 OUT=../libw2l_hip_abl4.so EXTRA='-include diag/hooks.h -DW2L_ABLATE=4'
//   make -C wav2letter_pytorch_amd/csrc BUILD=build_stamp OUT=../libw2l_hip_stamp.so EXTRA='-include diag/hooks.h -DW2L_STAMP'
// (tools/ablate_igemm.py, tools/stamp_wgrad.py).  Neither bench.py's source stamp nor the default Makefile target sees this file.
//
// W2L_ABLATE bits (implicit GEMM, PIPE = 1 loop): 1 = no LDS-DMA is issued, 2 = operand fragments are read from LDS once only,
// 4 = no MFMA, 8 = wave 0 of every block stamps s_memtime / s_memrealtime around its K loop.  Builds 1..7 compute garbage: only
// their run time and clock mean something.
// W2L_STAMP (weight gradient, 16x16x32 two-tap kernel): s_memtime stamps around the segments of every K step, summed in scalar
// registers and stored once per wave into a buffer nothing else reads (MI355X guide, "In-kernel stamps").
// W2L_KWB=1: the one-tap weight-gradient form (140-153 VGPRs: other kernels' waves fit beside it) -- a co-residency probe.
#pragma once
#include <hip/hip_runtime.h>

#ifdef W2L_ABLATE
#if W2L_ABLATE & 1
#define W2L_DIAG_SKIP_DMA(p) if ((p).Kw > 0) return
#endif
#if W2L_ABLATE & 2
#define W2L_DIAG_SKIP_FRAGS(p, stp) if ((p).Kw > 0 && (stp) > 0) return
#endif
#if W2L_ABLATE & 4
#define W2L_DIAG_SKIP_MFMA(p) if ((p).Kw > 0) return
#endif
#if W2L_ABLATE & 8
__device__ unsigned long long g_igemm_clk[2 * 4096];
#define W2L_DIAG_CLK_AT(c, r)                                                                         \
    do {                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c), "=s"(r)::"memory"); \
        __builtin_amdgcn_sched_barrier(0);                                                            \
    } while (0)
#define W2L_DIAG_CLK_BEGIN() unsigned long long clk_c0, clk_r0, clk_c1, clk_r1; W2L_DIAG_CLK_AT(clk_c0, clk_r0)
#define W2L_DIAG_CLK_END(tid)                                                                         \
    do {                                                                                              \
        W2L_DIAG_CLK_AT(clk_c1, clk_r1);                                                              \
        if ((tid) == 0 && blockIdx.x < 4096) {                                                        \
            g_igemm_clk[2 * blockIdx.x] = clk_c1 - clk_c0;                                            \
            g_igemm_clk[2 * blockIdx.x + 1] = clk_r1 - clk_r0;                                        \
        }                                                                                             \
    } while (0)
#define W2L_DIAG_IGEMM_EXPORTS                                                                        \
    extern "C" int w2l_igemm_read_clock(unsigned long long* dst, int n) {                             \
        W2L_CHECK_HIP(hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_igemm_clk), sizeof(unsigned long long) * (size_t)(n < 8192 ? n : 8192))); \
        return 0;                                                                                     \
    }
#endif
#endif  // W2L_ABLATE

#ifdef W2L_STAMP
__device__ unsigned long long g_wgrad_stamps[8 * 8192];
#define W2L_STAMP_AT(var)                                                                            \
    do {                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                           \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");                \
        __builtin_amdgcn_sched_barrier(0);                                                           \
    } while (0)
#define W2L_DIAG_STAMP_DECL() unsigned long long st_sum[6] = {0, 0, 0, 0, 0, 0}, st_prev = 0
// the four stamps of a step: (0) fragment reads + MFMAs of groups 0 .. NG-2, (1) wait for this wave's LDS-DMA,
// (2) block barrier, (3) pointer toggles + issuing the next LDS-DMA pieces
#define W2L_DIAG_STAMP_STEP0() unsigned long long t0_, t1_, t2_, t3_; W2L_STAMP_AT(t0_); st_sum[0] += t0_ - st_prev
#define W2L_DIAG_STAMP_STEP1() W2L_STAMP_AT(t1_); st_sum[1] += t1_ - t0_
#define W2L_DIAG_STAMP_STEP2() W2L_STAMP_AT(t2_); st_sum[2] += t2_ - t1_
#define W2L_DIAG_STAMP_STEP3() W2L_STAMP_AT(t3_); st_sum[3] += t3_ - t2_; st_prev = t3_; st_sum[5] += 1
#define W2L_DIAG_STAMP_LOOP_BEGIN() W2L_STAMP_AT(st_prev)
#define W2L_DIAG_STAMP_STORE(lane, slot_expr)                                                        \
    do {                                                                                             \
        if ((lane) == 0) {                                                                           \
            const int slot_ = (slot_expr);                                                           \
            if (slot_ < 8192) {                                                                      \
                for (int i_ = 0; i_ < 6; ++i_) g_wgrad_stamps[slot_ * 8 + i_] = st_sum[i_];          \
            }                                                                                        \
        }                                                                                            \
    } while (0)
#define W2L_DIAG_WGRAD_EXPORTS                                                                       \
    extern "C" int w2l_wgrad_read_stamps(unsigned long long* dst, int nwords) {                      \
        const size_t n = (size_t)(nwords < 8 * 8192 ? nwords : 8 * 8192) * sizeof(unsigned long long); \
        W2L_CHECK_HIP(hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_wgrad_stamps), n));                      \
        return 0;                                                                                    \
    }
#endif  // W2L_STAMP

#ifdef W2L_KWB
#define W2L_DIAG_WGRAD_TAPS W2L_KWB
#endif
