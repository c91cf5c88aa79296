#!/usr/bin/env python3
"""Range-count sweep of the dealt stream-K weight gradient on one layer (diagnostic): TFLOP/s per (order, G)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from wav2letter_pytorch_amd import _lib as L  # noqa: E402


def main():
    cin, cout, kw, d = (int(v) for v in sys.argv[1:5])
    orders = [int(v) for v in sys.argv[5].split(',')]
    Gs = [int(v) for v in sys.argv[6].split(',')]
    N, T = 32, 500
    p = (kw - 1) * d
    rows = T + p
    Tout = T
    x = torch.randn(N, rows, cin, device='cuda').to(torch.bfloat16)
    hb = p
    h = max(hb, (Tout + 63) // 64 * 64 - Tout)
    per = Tout + h
    dy = torch.zeros(h + N * per, cout, dtype=torch.bfloat16, device='cuda')
    dy[h:].view(N, per, cout)[:, :Tout] = torch.randn(N, Tout, cout, device='cuda').to(torch.bfloat16)
    dw = torch.zeros(kw, cout, cin, device='cuda')
    ws = torch.zeros(1 << 30, dtype=torch.uint8, device='cuda')
    flops = 2.0 * N * Tout * cout * cin * kw
    st = L.stream_ptr()

    def run():
        L.check(L.lib.w2l_conv1d_wgrad_ws(C.c_void_p(dy.data_ptr() + h * cout * 2), per * cout, L.ptr(x), rows * cin, N * rows,
                                          L.ptr(dw), N, cin, cout, Tout, kw, 1, d, 0, L.ptr(ws), ws.numel(), st))
    import time
    t0 = time.time()                 # the first second of launches in a process runs slow (clocks): keep it out of the table
    while time.time() - t0 < 3.0:
        for _ in range(20):
            run()
        torch.cuda.synchronize()
    for order in orders:
        out = []
        for G in Gs:
            L.lib.w2l_wgrad_force_plan(G, order)
            if L.lib.w2l_wgrad_needs_zero_x(N, cin, cout, Tout, kw, 1, d, ws.numel()):
                out.append(f'{G}: n/a')
                continue
            for _ in range(3):
                run()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20):
                run()
            e.record()
            torch.cuda.synchronize()
            out.append(f'{G}: {flops / (s.elapsed_time(e) / 20) / 1e9:.0f}')
        print(f'order {order}: ' + ' | '.join(out), flush=True)
    L.lib.w2l_wgrad_force_plan(0, -1)


if __name__ == '__main__':
    main()
