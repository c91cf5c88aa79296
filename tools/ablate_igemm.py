#!/usr/bin/env python3
"""What bounds the implicit-GEMM K loop: timing experiments on diagnostic builds (never the shipped library).

  make -C wav2letter_pytorch_amd/csrc BUILD=build_abl$V EXTRA='-include diag/hooks.h -DW2L_ABLATE='$V OUT=../libw2l_hip_abl$V.so
  W2L_LIB=$PWD/wav2letter_pytorch_amd/libw2l_hip_abl$V.so python3 tools/ablate_igemm.py [Cin Cout Kw dil [cfg]]

W2L_ABLATE bits (csrc/conv_igemm.hip, PIPE = 1 loop): 1 = no LDS-DMA is issued, 2 = operand fragments are read from LDS
once only, 4 = no MFMA, 8 = wave 0 of every block stamps s_memtime / s_memrealtime around its K loop (the clock the loop
ran at = ratio x 100 MHz).  The results of builds 1..7 are garbage; only the run time and the clock mean something.
The launch is repeated for ~2 s first so that the chip sits at the clock it sustains; operands are random."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wav2letter_pytorch_amd import _lib as L  # noqa: E402


def main():
    a = [int(v) for v in sys.argv[1:]]
    cin, cout, kw, d = a[:4] if len(a) >= 4 else (768, 768, 25, 1)
    cfg = a[4] if len(a) > 4 else 40                  # 192 x 256 block, PIPE = 1
    zeros = len(a) > 5 and a[5] == 1
    N, Tout = 32, 500
    rows = Tout + (kw - 1) * d
    x = torch.randn(N, rows, cin, device='cuda').to(torch.bfloat16)
    w = (torch.randn(kw, cout, cin, device='cuda') * 0.05).to(torch.bfloat16)
    if zeros:
        x.zero_()
        w.zero_()
    y = torch.empty(N, Tout, cout, dtype=torch.bfloat16, device='cuda')
    st = L.stream_ptr()
    L.lib.w2l_conv_force_tile_config(cfg)

    def run():
        L.check(L.lib.w2l_conv1d_igemm_ws(L.ptr(x), rows * cin, N * rows, L.ptr(w), L.ptr(y), 0, 0, None, None, N,
                                          cin, cout, Tout, kw, 1, d, None, 0, st))
    t0 = time.time()
    while time.time() - t0 < 2.0:
        for _ in range(20):
            run()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    flops = 2.0 * N * Tout * cout * cin * kw
    out = '%-28s %dx%d k%d cfg %d%s: %.4f ms  %6.0f TFLOP/s-equivalent' % (
        os.path.basename(L.LIB_PATH), cin, cout, kw, cfg, ' ZEROS' if zeros else '', ms, flops / ms / 1e9)
    raw = C.CDLL(L.LIB_PATH)
    if hasattr(raw, 'w2l_igemm_read_clock'):
        buf = np.zeros(2 * 4096, dtype=np.uint64)
        raw.w2l_igemm_read_clock.argtypes = [C.c_void_p, C.c_int]
        assert raw.w2l_igemm_read_clock(buf.ctypes.data, buf.size) == 0
        c, r = buf[0::2].astype(np.float64), buf[1::2].astype(np.float64)
        ok = r > 0
        ghz = c[ok] / r[ok] * 0.1
        out += '   K-loop clock: median %.3f GHz (min %.3f, max %.3f, %d blocks; loop %.1f us)' % (
            np.median(ghz), ghz.min(), ghz.max(), ok.sum(), np.median(r[ok]) / 100.0)
    print(out)


if __name__ == '__main__':
    main()
