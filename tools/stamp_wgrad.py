#!/usr/bin/env python3
"""Where a K step of the weight-gradient kernel spends its cycles (diagnostic build, never the shipped library).

  make -C wav2letter_pytorch_amd/csrc BUILD=build_stamp EXTRA='-include diag/hooks.h -DW2L_STAMP' OUT=../libw2l_hip_stamp.so
  W2L_LIB=$PWD/wav2letter_pytorch_amd/libw2l_hip_stamp.so python3 tools/stamp_wgrad.py [Cin Cout Kw dil [splits [order]]]

The stamp build brackets four segments of every step of the 16x16x32 two-tap kernel with s_memtime (csrc/conv_wgrad.hip,
W2L_STAMP): (A) the MFMA groups with their fragment reads, (B) the wait for this wave's LDS-DMA, (C) the block barrier,
(D) pointer toggles + issuing the next LDS-DMA pieces.  Its run time is NOT the kernel's (the stamps fence overlaps the real
kernel has): read the SHARES.  The launch is repeated for ~2 s first so that the chip sits at its sustained clock."""
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
    cin, cout, kw, d = (a + [896, 896, 29, 2])[:4] if len(a) >= 4 else (896, 896, 29, 2)
    splits = a[4] if len(a) > 4 else 2
    order = a[5] if len(a) > 5 else 1
    N, Tout = 32, 500
    rows = Tout + (kw - 1) * d
    x = torch.randn(N, rows, cin, device='cuda').to(torch.bfloat16)
    hb = (kw - 1) * d
    h = max(hb, (Tout + 63) // 64 * 64 - Tout)
    per = Tout + h
    dy = torch.zeros(h + N * per, cout, dtype=torch.bfloat16, device='cuda')
    dy[h:].view(N, per, cout)[:, :Tout] = torch.randn(N, Tout, cout, device='cuda').to(torch.bfloat16)
    dw = torch.zeros(kw, cout, cin, device='cuda')
    st = L.stream_ptr()
    L.lib.w2l_wgrad_force_plan(splits, order)

    def run():
        L.check(L.lib.w2l_conv1d_wgrad_ws(C.c_void_p(dy.data_ptr() + h * cout * 2), per * cout, L.ptr(x), rows * cin, N * rows,
                                          L.ptr(dw), N, cin, cout, Tout, kw, 1, d, 1, None, 0, st))
    t0 = time.time()
    while time.time() - t0 < 2.0:
        for _ in range(20):
            run()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    raw = C.CDLL(L.LIB_PATH)
    if not hasattr(raw, 'w2l_wgrad_read_stamps'):
        raise SystemExit('this library was not built with -DW2L_STAMP (see the header of this file)')
    buf = np.zeros(8 * 8192, dtype=np.uint64)
    raw.w2l_wgrad_read_stamps.argtypes = [C.c_void_p, C.c_int]
    assert raw.w2l_wgrad_read_stamps(buf.ctypes.data, buf.size) == 0
    v = buf.reshape(8192, 8).astype(np.float64)
    v = v[v[:, 5] > 0]
    steps = v[:, 5]
    names = ('MFMA groups + fragment reads', 'wait for own LDS-DMA (vmcnt 0)', 'block barrier', 'toggle + issue LDS-DMA pieces')
    per_step = v[:, :4] / steps[:, None]
    tot = per_step.sum(1)
    flops = 2.0 * N * Tout * cout * cin * kw
    print(f'{cin}->{cout} k{kw} d{d}, splits {splits}, order {order}: stamp build {ms:.3f} ms/launch = {flops / ms / 1e9:.0f} TFLOP/s '
          f'(NOT the kernel\'s rate), {len(v)} waves, {steps.mean():.0f} steps per wave')
    print(f'cycles per K step (64 MFMAs of 16 cycles = 1024 of matrix pipe per wave): median {np.median(tot):.0f}')
    for i, n in enumerate(names):
        print(f'  {n:34s} median {np.median(per_step[:, i]):7.0f}  mean {per_step[:, i].mean():7.0f}  share {per_step[:, i].mean() / tot.mean() * 100:5.1f} %')


if __name__ == '__main__':
    main()
