#!/usr/bin/env python3
"""Grouped weight-gradient launches (w2l_conv1d_wgrad_group) on rows of the Wav2Letter table: TFLOP/s of a group in every block
form beside the same layers launched one by one with their measured plans.
  python tools/bench_wgrad_group.py --groups "12,11,10;15,14,13" [--n 32] [--forms 17,21,1,5]"""
import argparse
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wav2letter_pytorch_amd import _lib as L  # noqa: E402
from tools.bench_conv import TABLE  # noqa: E402


def make_layer(N, T, cin, cout, kw, d):
    p = (kw - 1) * d
    rows = T + p
    x = torch.randn(N, rows, cin, device='cuda').to(torch.bfloat16)
    h = max(p, (T + 63) // 64 * 64 - T)
    per = T + h
    dy = torch.zeros(h + N * per, cout, dtype=torch.bfloat16, device='cuda')
    dy[h:].view(N, per, cout)[:, :T] = torch.randn(N, T, cout, device='cuda').to(torch.bfloat16)
    dw = torch.zeros(kw, cout, cin, device='cuda')
    return dict(x=x, dy=dy, dw=dw, h=h, per=per, rows=rows, cin=cin, cout=cout, kw=kw, d=d,
                flops=2.0 * N * T * cout * cin * kw)


def timeit(fn, reps):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--groups', default='12,11,10')
    ap.add_argument('--forms', default='17,21,1,5,9')
    ap.add_argument('--n', type=int, default=32)
    ap.add_argument('--t', type=int, default=500)
    ap.add_argument('--reps', type=int, default=20)
    args = ap.parse_args()
    N, T = args.n, args.t
    st = L.stream_ptr()
    layers = {}
    for grp in args.groups.split(';'):
        for r in grp.split(','):
            r = int(r)
            if r not in layers:
                cin, cout, kw, s, d = TABLE[r]
                assert s == 1
                layers[r] = make_layer(N, T, cin, cout, kw, d)
    # warm the clocks, then measure every layer's own plan
    any_l = next(iter(layers.values()))
    ws = torch.zeros(1 << 30, dtype=torch.uint8, device='cuda')

    def single(l):
        L.check(L.lib.w2l_conv1d_wgrad_ws(C.c_void_p(l['dy'].data_ptr() + l['h'] * l['cout'] * 2), l['per'] * l['cout'], L.ptr(l['x']),
                                          l['rows'] * l['cin'], N * l['rows'], L.ptr(l['dw']), N, l['cin'], l['cout'], T, l['kw'], 1,
                                          l['d'], 0, L.ptr(ws), ws.numel(), st))
    t0 = time.time()
    while time.time() - t0 < 3.0:
        for _ in range(20):
            single(any_l)
        torch.cuda.synchronize()
    own = {}
    for r, l in layers.items():
        L.check(L.lib.w2l_conv1d_wgrad_tune_x(C.c_void_p(l['dy'].data_ptr() + l['h'] * l['cout'] * 2), l['per'] * l['cout'], L.ptr(l['x']),
                                              l['rows'] * l['cin'], N * l['rows'], L.ptr(l['dw']), N, l['cin'], l['cout'], T, l['kw'], 1,
                                              l['d'], 3, L.ptr(ws), ws.numel(), 1, st))
        zero = bool(L.lib.w2l_wgrad_needs_zero_x(N, l['cin'], l['cout'], T, l['kw'], 1, l['d'], ws.numel()))

        def run(l=l, zero=zero):
            if zero:
                l['dw'].zero_()
            single(l)
        own[r] = timeit(run, args.reps)
        print(f'row {r}: {l["cin"]}->{l["cout"]} k{l["kw"]} d{l["d"]}: own plan {own[r]:.3f} ms = {l["flops"] / own[r] / 1e9:.0f} TFLOP/s'
              + (' (with its zero fill)' if zero else ''), flush=True)
    for grp in args.groups.split(';'):
        rows = [int(r) for r in grp.split(',')]
        ls = [layers[r] for r in rows]
        flops = sum(l['flops'] for l in ls)
        t_own = sum(own[r] for r in rows)
        arr = (L.WgradItem * len(ls))(*[L.WgradItem(l['dy'].data_ptr() + l['h'] * l['cout'] * 2, l['per'] * l['cout'], l['x'].data_ptr(),
                                                     l['rows'] * l['cin'], N * l['rows'], l['dw'].data_ptr(), l['cin'], l['cout'], l['kw'], 0)
                                        for l in ls])
        d = ls[0]['d']
        assert all(l['d'] == d for l in ls)
        res = []
        for form in [int(v) for v in args.forms.split(',')]:
            tiles = sum(L.lib.w2l_wgrad_group_tiles(l['cin'], l['cout'], l['kw'], form) for l in ls)
            slots = 256 if form & 4 else 512

            def run():
                L.check(L.lib.w2l_conv1d_wgrad_group(arr, len(ls), N, T, d, form, st), 'w2l_conv1d_wgrad_group')
            t = timeit(run, args.reps)
            res.append(f'form {form}: {tiles} tiles = {tiles / slots:.2f} rounds, {t:.3f} ms = {flops / t / 1e9:.0f} TF')
        print(f'group {rows}: one by one {t_own:.3f} ms = {flops / t_own / 1e9:.0f} TF | ' + ' | '.join(res), flush=True)


if __name__ == '__main__':
    main()
