#!/usr/bin/env python3
"""Times the BatchNorm/activation kernels alone (no overlap) on W2L layer shapes."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wav2letter_pytorch_amd import _lib as L  # noqa: E402


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


def main():
    N, T = 32, 500
    st = L.stream_ptr()
    for Cn, pl, p in [(256, 5, 0.2), (512, 8, 0.2), (896, 28, 0.4), (1024, 0, 0.4)]:
        y = torch.randn(N, T, Cn, device='cuda').to(torch.bfloat16)
        scale, shift, mean, invstd = (torch.rand(Cn, device='cuda') + 0.5 for _ in range(4))
        mask = torch.empty(N * T * Cn // 8, dtype=torch.uint8, device='cuda')
        d = L.BnActDesc()
        d.N, d.T, d.C = N, T, Cn
        d.y, d.y_f32 = y.data_ptr(), 0
        d.scale, d.shift, d.mean, d.invstd = (t.data_ptr() for t in (scale, shift, mean, invstd))
        d.act, d.drop_p, d.seed, d.offset, d.mask = 1, p, 1, 1, mask.data_ptr()
        R = T + 2 * pl
        out = torch.empty(N, R, Cn, dtype=torch.bfloat16, device='cuda')
        g = torch.randn(N, R, Cn, device='cuda').to(torch.bfloat16)
        gs = L.GradSrc()
        gs.dxp, gs.f32, gs.pad_l, gs.pad_r, gs.pad_mode, gs.rows = g.data_ptr(), 0, pl, pl, 1, R
        nb = L.lib.w2l_bn_bwd_blocks(N, T, Cn)
        partial = torch.empty(nb, 2, Cn, device='cuda')
        sums = torch.empty(4, Cn, device='cuda')
        h = 28
        dy = torch.empty(h + N * (T + h), Cn, dtype=torch.bfloat16, device='cuda')
        t_f = timeit(lambda: L.check(L.lib.w2l_bn_act_fwd(C.byref(d), L.ptr(out), None, R, pl, pl, 1, st)))
        # the statistics finalize: its own launch over 125 per-tile rows (round 4) vs folded into the apply pass over 8 slot rows
        gam, bet, rm, rv = (torch.rand(Cn, device='cuda') + 0.5 for _ in range(4))
        part125, part8 = torch.rand(N * 4, 2, Cn, device='cuda') + 1, torch.rand(8, 2, Cn, device='cuda') + 1
        t_fz = timeit(lambda: L.check(L.lib.w2l_bn_finalize(L.ptr(part125), N * 4, Cn, N * T, L.ptr(gam), L.ptr(bet), 1e-3, 0.9, L.ptr(rm),
                                                            L.ptr(rv), L.ptr(mean), L.ptr(invstd), L.ptr(scale), L.ptr(shift), st)))
        f = L.BnFin()
        f.partial, f.rows, f.count = part8.data_ptr(), 8, N * T
        f.gamma, f.beta, f.eps, f.momentum = gam.data_ptr(), bet.data_ptr(), 1e-3, 0.9
        f.running_mean, f.running_var = rm.data_ptr(), rv.data_ptr()
        f.scale, f.shift, f.mean, f.invstd = (t.data_ptr() for t in (scale, shift, mean, invstd))
        t_ff = timeit(lambda: L.check(L.lib.w2l_bn_act_fwd_fin(C.byref(d), C.byref(f), None, L.ptr(out), None, 1.0, R, pl, pl, 1, st)))
        for t in (scale, shift, mean, invstd):                        # (the folded launch rewrote them from random sums)
            t.copy_(torch.rand(Cn, device='cuda') + 0.5)
        t_r = timeit(lambda: L.check(L.lib.w2l_bn_act_bwd_reduce(C.byref(d), C.byref(gs), None, L.ptr(partial), st)))
        t_z = timeit(lambda: L.check(L.lib.w2l_bn_bwd_finalize(L.ptr(partial), nb, Cn, 2, L.ptr(sums), st)))
        t_a = timeit(lambda: L.check(L.lib.w2l_bn_act_bwd_apply(C.byref(d), C.byref(gs), None, L.ptr(sums), L.ptr(dy), None, h,
                                                                None, None, 0, st)))
        NS = int(os.environ.get('SLOTS', '8'))
        slots = torch.zeros(NS, 2, Cn, device='cuda')
        t_r2 = timeit(lambda: L.check(L.lib.w2l_bn_act_bwd_reduce_slots(C.byref(d), C.byref(gs), L.ptr(slots), NS, st)))
        t_a2 = timeit(lambda: L.check(L.lib.w2l_bn_act_bwd_apply_slots(C.byref(d), C.byref(gs), L.ptr(slots), NS, L.ptr(sums), L.ptr(dy), h,
                                                                       None, st)))
        el = N * T * Cn
        print(f'C={Cn:5d}: bwd two-launch chain: reduce {t_r2:5.1f} us ({el * 4 / t_r2 / 1e6:5.2f} TB/s) + apply {t_a2:5.1f} us '
              f'({el * 6 / t_a2 / 1e6:5.2f} TB/s)   [three launches: {t_r:.1f} + {t_z:.1f} + {t_a:.1f}]')
        print(f'C={Cn:5d}: fwd finalize {t_fz:5.1f} + apply {t_f:5.1f} us | folded {t_ff:5.1f} us ({el * 4 / t_ff / 1e6:5.2f} TB/s)')
        print(f'C={Cn:5d}: fwd {t_f:7.1f} us ({el * 4 / t_f / 1e6:5.2f} TB/s)  reduce {t_r:7.1f} us ({el * 4 / t_r / 1e6:5.2f} TB/s)  '
              f'finalize {t_z:6.1f} us  apply {t_a:7.1f} us ({el * 6 / t_a / 1e6:5.2f} TB/s)')


def sgd_pack():
    """w2l_sgd_pack alone on the table's layer shapes: 24 B per parameter (read p, g, m; write p, m, two bf16 operands).  In the
    step it runs on the optimizer's stream under the next forward pass, beside MFMA-bound convolutions -- where a rocprofv3 trace
    shows 2.0 (serial-wgrad profile: beside the forward) to 3.5 TB/s (default profile) for the same kernel."""
    st = L.stream_ptr()
    print('w2l_sgd_pack alone (no other kernel on the chip):')
    total_b = total_t = 0.0
    for cin, cout, kw, mult in [(256, 256, 11, 3), (384, 384, 13, 2), (512, 512, 17, 2), (640, 640, 21, 2), (768, 768, 25, 2), (896, 896, 29, 2)]:
        p, g, m = (torch.randn(kw, cout, cin, device='cuda') for _ in range(3))
        fwd = torch.empty(kw, cout, cin, dtype=torch.bfloat16, device='cuda')
        dgr = torch.empty(kw, cin, cout, dtype=torch.bfloat16, device='cuda')
        t = timeit(lambda: L.check(L.lib.w2l_sgd_pack(L.ptr(p), L.ptr(g), L.ptr(m), 0, 1e-5, 0.9, 1e-5, 1, 0, cout, cin, kw, L.ptr(fwd), None,
                                                      L.ptr(dgr), None, None, None, 1.0, st)))
        b = cin * cout * kw * 24
        total_b += b * mult
        total_t += t * mult
        print(f'  {cin:4d} -> {cout:4d} k{kw:2d}: {t:7.1f} us  {b / t / 1e6:5.2f} TB/s   ({b / 1e6:6.1f} MB)')
    print(f'  these 13 layers: {total_t / 1e3:.3f} ms for {total_b / 1e9:.2f} GB = {total_b / total_t / 1e6:.2f} TB/s')


if __name__ == '__main__':
    main()
    sgd_pack()
