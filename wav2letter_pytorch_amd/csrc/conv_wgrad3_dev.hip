// Three taps per block for the weight gradient (gfx950): the 32x32x16 form of conv_wgrad_kernel.h with its 3 x 2 x 2 x 16 = 192
// accumulator registers per lane in AGPRs ("a"-constrained inline-asm MFMAs), the remaining 64 VGPRs for fragments, addresses and
// LDS-DMA offsets -- two 4-wave blocks per CU as before.  Per 64-row K step a block stages the same dy tile and an x window two
// rows (d) longer than the two-tap form's for 1.5 x the MFMAs: a third less LDS-DMA per FLOP, which is what the K loop of the
// two-tap kernel spends a third of its time issuing ("Where a K step of the weight gradient goes").  29 taps make
// 10 groups: the 896-wide layers' 7 x 7 x 10 = 490 tiles fit ONE round of the 512 resident blocks -- no split, no atomics, no
// zero-filled dw.
//
// DEVICE CODE ONLY.  This file is not part of the fat binary hipcc builds for the library: hipcc splits a 256-register budget
// 128 VGPRs / 128 AGPRs whatever the kernel needs (and then spills 129 registers here) unless the kernel function carries the
// LLVM attribute "amdgpu-agpr-alloc", which no HIP source attribute sets.  The Makefile therefore compiles this file to LLVM IR
// (hipcc --cuda-device-only -emit-llvm), adds "amdgpu-agpr-alloc"="192,192" to the kernel's attribute group, builds a gfx950
// code object from the IR and embeds it in libw2l_hip.so as a byte array; conv_wgrad.hip loads it with hipModuleLoadData and
// launches w2l_wgrad3_kernel with hipModuleLaunchKernel.
//
// Replaces the weight-gradient half of aten::convolution_backward (wav2letter.py:35-36,42 / jasper.py:96-105,127), stride 1.
#include "conv_wgrad_kernel.h"

using namespace w2l_wgrad;

// LDS: two dy tiles [64][128 co] + two x windows of up to W2L_WGRAD3_XROWS rows [.][128 ci] (rows of 256 bytes), static so that
// the module launch needs no dynamic-LDS attribute: 32 KiB + 2 x 18 KiB = 68 KiB per block, two blocks per CU.
#define W2L_WGRAD3_XROWS 72          /* >= 63 + 2 * dilation + 1, rounded up to 4: dilation <= 4 */

extern "C" __global__ __launch_bounds__(256, 2) void w2l_wgrad3_kernel(WgradParams p) {
    __shared__ __attribute__((aligned(1024))) char smem[2 * BT * ROWB + 2 * W2L_WGRAD3_XROWS * ROWB];
    conv_wgrad_body<3, true, false, 1, true, true>(p, smem);
}

// two tap groups per block: ONE 8-wave block per CU, waves 0-3 taps kw0 .. kw0+2, waves 4-7 taps kw0+3 .. kw0+5 of the same
// [128 co x 128 ci] tile, all reading one dy tile and one x window (5 d rows longer than a tap's) -- half the LDS-DMA per MFMA
// once more.  29 taps = five groups of six: 7 x 7 x 5 = 245 tiles of the 896-wide layers fit the 256 CUs in one round.
#define W2L_WGRAD3X2_XROWS 88        /* >= 63 + 5 * dilation + 1, rounded up to 4: dilation <= 4 */
extern "C" __global__ __launch_bounds__(512, 2) void w2l_wgrad3x2_kernel(WgradParams p) {
    __shared__ __attribute__((aligned(1024))) char smem[2 * BT * ROWB + 2 * W2L_WGRAD3X2_XROWS * ROWB];
    conv_wgrad_body<3, true, false, 2, true, true>(p, smem);
}
