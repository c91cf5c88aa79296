#!/usr/bin/env python3
"""In-process, interleaved A/B of step-engine switches on the bench workload: ONE model, ONE box, ONE process; the variants take
turns in blocks of steps (A B C A B C ...), so box-to-box and run-to-run drift (+-0.1 ms here) cancels and 0.03 ms shows.

  python tools/step_ab.py --variants "base;fold=FOLD_BN_FWD:0;spread=DEFER_SPREAD:start" [--model jasper10x5 --batch 16]
      [--defer 6] [--block 10] [--rounds 6]

A variant is name=FLAG:value[,FLAG:value...] over the module-level switches of wav2letter_pytorch_amd.engine that are read at
run time (FOLD_BN_FWD, STAT_SLOTS, DEFER_SPREAD, FOLD_BN_FINALIZE, FUSED_BN_REDUCE, DETERMINISTIC_WGRAD, DEALT_WGRAD) plus
GROUPS:<W2L_WGRAD_GROUPS value> and DEFER:<k> (optim.FusedSGD.defer_wgrad).  'base' = the defaults.  Every variant gets
--settle steps after a switch (plans are measured then) before its block is timed."""
import argparse
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--variants', required=True)
    ap.add_argument('--model', default='wav2letter', choices=['wav2letter', 'jasper10x5'])
    ap.add_argument('--dtype', default='bf16')
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--frames', type=int, default=1000)
    ap.add_argument('--defer', default='6')
    ap.add_argument('--block', type=int, default=10)
    ap.add_argument('--rounds', type=int, default=6)
    ap.add_argument('--settle', type=int, default=3)
    args = ap.parse_args()
    from wav2letter_pytorch_amd import Jasper, Wav2Letter, engine as E
    from wav2letter_pytorch_amd.defaults import synthetic_batch
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    if args.model == 'jasper10x5':
        model = Jasper(B.jasper10x5_cfg(args.dtype)).to(dev).train()
        model.check_nan = False
    else:
        model = Wav2Letter(B.w2l_cfg(20, precision=args.dtype)).to(dev).train()
    N, T = args.batch, args.frames
    x, il, tg, tl = synthetic_batch(N, T, seed=1234)
    x = x.to(dev)
    tg_d, tl_d = tg.to(dev), tl.to(dev)
    ol = model.compute_output_lengths(il).to(dev)
    lens_arg = il if args.model == 'jasper10x5' else None
    opt = model.configure_optimizers()[0][0]
    opt.overlap = True

    def step():
        opt.zero_grad(set_to_none=True)
        out, _ = model(x, lens_arg)
        loss = model.criterion(out.transpose(0, 1), tg_d, ol, tl_d)
        loss.backward()
        opt.step()
        return loss

    def fence():
        opt.join()
        torch.cuda.synchronize()

    defaults = {k: getattr(E, k) for k in ('FOLD_BN_FWD', 'STAT_SLOTS', 'DEFER_SPREAD', 'FOLD_BN_FINALIZE', 'FUSED_BN_REDUCE',
                                           'DETERMINISTIC_WGRAD', 'DEALT_WGRAD', 'WGRAD_AFTER_DGRAD', 'FAST_BN_BWD', 'WGROUP_MAX')}
    defaults['GROUPS'] = os.environ.get('W2L_WGRAD_GROUPS', E.WG.setting())
    defaults['DEFER'] = args.defer
    defaults['PROBE'] = ''            # what-if probes (wrong gradients, timing only): nowgrad = no weight-gradient launches (and no
    real_wgrad, real_step = E.StackEngine._wgrad, type(opt).step       # updates of those weights), nosgd = no optimizer step
    real_dgrad = E.StackEngine._dgrad

    def parse(v):
        if '=' not in v:
            return v, {}
        name, spec = v.split('=', 1)
        out = {}
        for item in spec.split(','):
            k, val = item.split(':', 1)
            out[k] = val
        return name, out

    variants = [parse(v) for v in args.variants.split(';')]

    def apply(over):
        fence()
        cfg = dict(defaults)
        cfg.update(over)
        for k, v in cfg.items():
            if k == 'GROUPS':                # explicit groups as 8.9.10|11.12.13 (',' and ';' separate flags and variants here)
                os.environ['W2L_WGRAD_GROUPS'] = str(v).replace('.', ',').replace('|', ';')
            elif k == 'PROBE':
                E.StackEngine._wgrad = (lambda self, *a, **kw: None) if 'nowgrad' in str(v) else real_wgrad
                type(opt).step = (lambda self, closure=None: None) if 'nosgd' in str(v) else real_step
                # a probe patches engine methods, which a recorded launch list neither sees nor is keyed on: probe variants run
                # the eager step (same device time: tools/replay_ab.py), everything else the replayed one
                from wav2letter_pytorch_amd import replay
                replay.ENABLED = not str(v)
            elif k == 'DEFER':
                opt.defer_wgrad(model, [int(t) for t in str(v).split('|')] if '|' in str(v) else int(v))
            else:
                cur = defaults[k]
                if k in ('FOLD_BN_FWD', 'FUSED_BN_REDUCE'):
                    v = str(v)
                elif isinstance(cur, bool):
                    v = str(v) in ('1', 'True', 'true')
                elif isinstance(cur, int):
                    v = int(v)
                setattr(E, k, v)

    for name, over in variants:              # every variant measures its plans once, outside the timing
        apply(over)
        for _ in range(max(args.settle, 4)):
            step()
    times = {name: [] for name, _ in variants}
    for rnd in range(args.rounds):
        for name, over in variants:
            apply(over)
            for _ in range(args.settle):
                step()
            fence()
            t0 = time.perf_counter()
            for _ in range(args.block):
                step()
            fence()
            times[name].append((time.perf_counter() - t0) / args.block * 1e3)
    base = statistics.mean(times[variants[0][0]])
    print(f'{args.model} N={N} T={T} {args.dtype}: {args.rounds} rounds x {args.block} steps per variant, interleaved, ms/step')
    for name, over in variants:
        v = times[name]
        print(f'  {name:14s} mean {statistics.mean(v):7.3f}  (min {min(v):7.3f}  max {max(v):7.3f}  sd {statistics.pstdev(v):.3f})  '
              f'{statistics.mean(v) - base:+.3f} vs {variants[0][0]}   {over}')


if __name__ == '__main__':
    main()
