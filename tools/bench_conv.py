#!/usr/bin/env python3
"""Per-layer timing of the conv kernels (forward igemm, dgrad igemm, wgrad) on the Wav2Letter
table shapes (N=32, T=1000).  HIP-event timing on the launch stream; prints TFLOP/s per layer."""
import argparse
import ctypes as C
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wav2letter_pytorch_amd import _lib as L  # noqa: E402

TABLE = ([(64, 256, 11, 2, 1)] + [(256, 256, 11, 1, 1)] * 3 + [(256, 384, 13, 1, 1)] + [(384, 384, 13, 1, 1)] * 2
         + [(384, 512, 17, 1, 1)] + [(512, 512, 17, 1, 1)] * 2 + [(512, 640, 21, 1, 1)] + [(640, 640, 21, 1, 1)] * 2
         + [(640, 768, 25, 1, 1)] + [(768, 768, 25, 1, 1)] * 2 + [(768, 896, 29, 1, 2)] + [(896, 896, 29, 1, 2)] * 2
         + [(896, 1024, 1, 1, 1), (1024, 64, 1, 1, 1)])


WARM_S = 0.0


def timeit(fn, reps):
    for _ in range(3):
        fn()
    if WARM_S > 0:                      # hold the launch for a while first: the chip settles at the clock it sustains
        import time
        t0 = time.time()
        while time.time() - t0 < WARM_S:
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--layers', default='all')
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--n', type=int, default=32)
    ap.add_argument('--t', type=int, default=1000)
    ap.add_argument('--sweep', action='store_true', help='time every igemm block-shape candidate per layer')
    ap.add_argument('--sweep-cfgs', default='', help='with --sweep: comma-separated configuration indices instead of the 26 PIPE=0 shapes '
                                                     '(index + 26 = the PIPE=1 loop of the same shape)')
    ap.add_argument('--warm-s', type=float, default=0.0, help='seconds of back-to-back launches before every timing')
    ap.add_argument('--deterministic', action='store_true', help='weight gradients through slabs + ticket (W2L_DETERMINISTIC=1 path)')
    ap.add_argument('--no-dealt', action='store_true', help='weight gradients without a workspace: no dealt stream-K plans (round 4\'s plan space)')
    ap.add_argument('--no-splitk', action='store_true', help='with --tune: measure without the split-K configurations')
    ap.add_argument('--orders', default=None, help='with --wgrad-plans: comma-separated plan orders to time instead of all')
    ap.add_argument('--wgrad-plans', action='store_true',
                    help='per layer: the best forced split count of every weight-gradient plan class (block order x tap groups, stream-K)')
    ap.add_argument('--fp8', action='store_true', help='also time the e4m3 weight-gradient kernel (w2l_conv1d_wgrad_fp8) on each layer')
    ap.add_argument('--tune', action='store_true', help='let the library measure and pick its configurations first')
    ap.add_argument('--tune-cache', default=None, help='with --tune: load the measured plans from this file if it exists (no measuring launches then) and save them to it afterwards')
    args = ap.parse_args()
    global WARM_S
    if args.tune_cache and os.path.exists(args.tune_cache):
        L.lib.w2l_tune_load(args.tune_cache.encode())
    WARM_S = args.warm_s
    N = args.n
    uniq = []
    for sh in TABLE:
        if sh not in uniq:
            uniq.append(sh)
    if args.layers != 'all':
        uniq = [uniq[int(i)] for i in args.layers.split(',')]
    st = L.stream_ptr()
    tot = {'fwd': [0, 0], 'dgrad': [0, 0], 'wgrad': [0, 0]}
    mult = {sh: TABLE.count(sh) for sh in uniq}
    print(f'{"Cin":>5} {"Cout":>5} {"K":>3} s d | {"fwd ms":>8} {"TF":>6} | {"dgrad ms":>8} {"TF":>6} | {"wgrad ms":>8} {"TF":>6}')
    for (cin, cout, kw, s, d) in uniq:
        T = args.t if s == 2 else args.t // 2
        p = (kw - 1) * d if s == 1 else 9
        pl, pr = p // 2, p - p // 2
        rows = T + pl + pr
        Tout = (rows - (kw - 1) * d - 1) // s + 1
        x = (torch.randn(N, rows, cin, device='cuda')).to(torch.bfloat16)
        w = (torch.randn(kw, cout, cin, device='cuda') * 0.05).to(torch.bfloat16)
        wd = (torch.randn(kw, cin, cout, device='cuda') * 0.05).to(torch.bfloat16)
        y = torch.empty(N, Tout, cout, dtype=torch.bfloat16, device='cuda')
        stats = torch.empty(L.lib.w2l_conv_stat_tiles(N, Tout), 2, cout, device='cuda')
        hb = (kw - 1) * d
        h = max(hb, (Tout + 63) // 64 * 64 - Tout)          # shared-halo layout
        per = Tout + h
        dy = torch.zeros(h + N * per, cout, dtype=torch.bfloat16, device='cuda')
        dy[h:].view(N, per, cout)[:, :Tout] = torch.randn(N, Tout, cout, device='cuda').to(torch.bfloat16)
        dx = torch.empty(N * per, cin, dtype=torch.bfloat16, device='cuda')
        dw = torch.zeros(kw, cout, cin, device='cuda')
        flops = 2.0 * N * Tout * cout * cin * kw

        ws = torch.zeros(min(1 << 30, int(L.lib.w2l_conv_splitk_workspace_bytes(N, max(cin, cout), max(N * per, Tout)))),
                         dtype=torch.uint8, device='cuda')

        def fwd():
            L.check(L.lib.w2l_conv1d_igemm_ws(L.ptr(x), rows * cin, N * rows, L.ptr(w), L.ptr(y), 0, 0, None, L.ptr(stats), N,
                                              cin, cout, Tout, kw, s, d, L.ptr(ws), ws.numel(), st))

        def dgrad():
            L.check(L.lib.w2l_conv1d_igemm_ws(C.c_void_p(dy.data_ptr() + (h - hb) * cout * 2), dy.shape[0] * cout,
                                              dy.shape[0] - (h - hb), L.ptr(wd), L.ptr(dx), 0, 0, None, None, 1, cout, cin,
                                              N * per, kw, 1, d, L.ptr(ws), ws.numel(), st))

        # the engine's default: a workspace sized for the dealt stream-K plans, classic splits keep their atomics (tune flag 1);
        # --deterministic: every split reduction through slabs; --no-dealt: round 4's plan space (no workspace)
        need = L.lib.w2l_wgrad_workspace_bytes(cin, cout, kw) if args.deterministic else L.lib.w2l_wgrad_dealt_workspace_bytes(cin, cout, kw)
        wws = torch.zeros(max(65536, min(1 << 30, int(need))), dtype=torch.uint8, device='cuda')
        wwsa = (None, 0) if (args.no_dealt and not args.deterministic) or need == 0 else (L.ptr(wws), wws.numel())
        wflags = 0 if args.deterministic else 1

        def wgrad():
            L.check(L.lib.w2l_conv1d_wgrad_ws(C.c_void_p(dy.data_ptr() + h * cout * 2), per * cout, L.ptr(x),
                                              rows * cin, N * rows, L.ptr(dw), N, cin, cout, Tout, kw, s, d, 0, *wwsa, st))

        if args.wgrad_plans:
            res = []
            # bit 0 block order, 2 two tap groups, 3 32x32x16 MFMA, 1 stream-K (6, 7: both), 4 three taps per block (AGPR accumulators)
            # + 32: dealt stream-K of that form (split = the range count: 512 / 256); classic splits are timed with atomics (+ 64)
            for order in ([int(v) for v in args.orders.split(',')] if args.orders else
                          (0, 1, 4, 5, 8, 9, 16, 17, 20, 21, 2, 3, 6, 7, 33, 37, 41, 49, 53)):
                best = (float('inf'), 0)
                if order & 32:
                    sps = (256,) if order & 4 else (512, 256)
                else:
                    sps = (1,) if order & 2 else (1, 2, 3, 4, 5, 6, 8, 10, 12, 16)
                for sp in sps:
                    L.lib.w2l_wgrad_force_plan(sp, order if order & 32 or args.deterministic else order | 64)
                    zero = bool(L.lib.w2l_wgrad_needs_zero_x(N, cin, cout, Tout, kw, s, d, wwsa[1]))
                    if order & 32 and zero:
                        continue                                   # no dealt form for this layer: the launch would be its fallback

                    def run():
                        if zero:
                            dw.zero_()
                        wgrad()
                    try:
                        t_ = timeit(run, args.reps)
                    except Exception:
                        continue
                    best = min(best, (t_, sp))
                res.append(f'order {order}: {flops / best[0] / 1e9:5.0f} TF (split {best[1]})')
            L.lib.w2l_wgrad_force_plan(0, -1)
            print('      ' + ' | '.join(res))
        if args.sweep:
            res = []
            for ci in ([int(v) for v in args.sweep_cfgs.split(',')] if args.sweep_cfgs else range(26)):
                L.lib.w2l_conv_force_tile_config(ci)
                try:
                    a = timeit(fwd, args.reps)
                except Exception:            # config not allowed with statistics / does not fit LDS
                    a = float('nan')
                try:
                    b = timeit(dgrad, args.reps) if s == 1 else float('nan')
                except Exception:
                    b = float('nan')
                res.append(f'{ci}:{flops / a / 1e9:5.0f}/{flops / b / 1e9:5.0f}')
            L.lib.w2l_conv_force_tile_config(-1)
            print('      ' + ' | '.join(res))
        if args.tune:
            wsa = (L.ptr(ws), ws.numel()) if not args.no_splitk else (None, 0)
            L.check(L.lib.w2l_conv1d_igemm_tune_ws(L.ptr(x), rows * cin, N * rows, L.ptr(w), L.ptr(y), 0, None, L.ptr(stats), N,
                                                   cin, cout, Tout, kw, s, d, 3, *wsa, st))
            if s == 1:
                L.check(L.lib.w2l_conv1d_igemm_tune_ws(C.c_void_p(dy.data_ptr() + (h - hb) * cout * 2), dy.shape[0] * cout,
                                                       dy.shape[0] - (h - hb), L.ptr(wd), L.ptr(dx), 0, None, None, 1, cout, cin,
                                                       N * per, kw, 1, d, 3, *wsa, st))
            L.check(L.lib.w2l_conv1d_wgrad_tune_x(C.c_void_p(dy.data_ptr() + h * cout * 2), per * cout, L.ptr(x), rows * cin,
                                                  N * rows, L.ptr(dw), N, cin, cout, Tout, kw, s, d, 3, *wwsa, wflags, st))
            dw.zero_()
        tw8 = float('nan')
        if args.fp8 and s == 1 and cin % 128 == 0 and cout % 128 == 0:
            # e4m3 copies in the layouts of the bf16 operands; the halo of dy must cover the 128-frame steps
            h8 = max(hb, (Tout + 127) // 128 * 128 - Tout)
            per8 = Tout + h8
            dyq = torch.zeros(h8 + N * per8, cout, dtype=torch.uint8, device='cuda')
            dyq[h8:].view(N, per8, cout)[:, :Tout] = (torch.randn(N, Tout, cout, device='cuda') * 16).to(torch.float8_e4m3fn).view(torch.uint8)
            xq = (x.float() * 16).to(torch.float8_e4m3fn).view(torch.uint8)

            def wgrad8():
                L.check(L.lib.w2l_conv1d_wgrad_fp8(C.c_void_p(dyq.data_ptr() + h8 * cout), per8 * cout, L.ptr(xq), rows * cin,
                                                   N * rows, L.ptr(dw), N, cin, cout, Tout, kw, d, 1.0, None, 0, st))
            if args.tune:
                L.check(L.lib.w2l_conv1d_wgrad_fp8_tune(C.c_void_p(dyq.data_ptr() + h8 * cout), per8 * cout, L.ptr(xq), rows * cin,
                                                        N * rows, L.ptr(dw), N, cin, cout, Tout, kw, d, 3, st))
            tw8 = timeit(wgrad8, args.reps)
            tot.setdefault('wgrad_fp8', [0, 0])
            tot['wgrad_fp8'][0] += mult[(cin, cout, kw, s, d)] * tw8
            tot['wgrad_fp8'][1] += mult[(cin, cout, kw, s, d)] * flops
        tf = timeit(fwd, args.reps)
        td = timeit(dgrad, args.reps) if s == 1 else float('nan')
        wzero = bool(L.lib.w2l_wgrad_needs_zero_x(N, cin, cout, Tout, kw, s, d, wwsa[1]))

        def wgrad_step():            # what the step pays for an atomic plan: the zero fill of dw as well
            dw.zero_()
            wgrad()
        tw = timeit(wgrad, args.reps)                        # the kernel alone (comparable with earlier rounds' tables)
        twz = timeit(wgrad_step, args.reps) if wzero else float('nan')
        m = mult[(cin, cout, kw, s, d)]
        for k, t in (('fwd', tf), ('dgrad', td), ('wgrad', tw)):
            if t == t:
                tot[k][0] += m * t
                tot[k][1] += m * flops
        print(f'{cin:5d} {cout:5d} {kw:3d} {s} {d} | {tf:8.3f} {flops / tf / 1e9:6.0f} | {td:8.3f} {flops / td / 1e9:6.0f} | '
              f'{tw:8.3f} {flops / tw / 1e9:6.0f}   x{m}' + (f' (atomic plan: {twz:.3f} ms with its zero fill)' if wzero else ' (stores: no fill)')
              + (f' | wgrad e4m3 {tw8:8.3f} {flops / tw8 / 1e9:6.0f}' if tw8 == tw8 else ''))
    if args.tune_cache:
        L.check(L.lib.w2l_tune_save(args.tune_cache.encode()), 'w2l_tune_save')
    for k, (t, f) in tot.items():
        if t:
            print(f'{k}: {t:.3f} ms/step-equivalent, {f / t / 1e9:.0f} TFLOP/s')


if __name__ == '__main__':
    main()
