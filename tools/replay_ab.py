#!/usr/bin/env python3
"""Replayed vs eager steps in ONE process, blocks of steps interleaved (same box, same clocks, same kernel plans):
ms per step (device time between two events around a block, host kept ahead) and host enqueue time per step.

  python tools/replay_ab.py [--model wav2letter|jasper10x5] [--batch N] [--mid-layers K] [--dtype bf16|fp8] [--blocks 6] [--steps 10]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--model', default='wav2letter')
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--frames', type=int, default=1000)
    ap.add_argument('--mid-layers', type=int, default=20)
    ap.add_argument('--dtype', default='bf16')
    ap.add_argument('--blocks', type=int, default=6)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--defer', type=int, default=4)
    args = ap.parse_args()
    from wav2letter_pytorch_amd import Jasper, Wav2Letter, replay
    from wav2letter_pytorch_amd.defaults import jasper10x5_model, synthetic_batch, wav2letter_model
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    if args.model == 'jasper10x5':
        model = Jasper(jasper10x5_model(precision=args.dtype)).to(dev).train()
        model.check_nan = False
    else:
        model = Wav2Letter(wav2letter_model(args.mid_layers, precision=args.dtype)).to(dev).train()
    x, il, tg, tl = synthetic_batch(args.batch, args.frames, seed=1234)
    x, tg, tl = x.to(dev), tg.to(dev), tl.to(dev)
    ol = model.compute_output_lengths(il).to(dev)
    lens = il if args.model == 'jasper10x5' else None
    opt = model.configure_optimizers()[0][0]
    opt.overlap = True
    n_units = len(model.engine().units)
    if args.defer and n_units >= 4:
        opt.defer_wgrad(model, min(args.defer, n_units))

    def step():
        opt.zero_grad(set_to_none=True)
        out, _ = model(x, lens)
        loss = model.criterion(out.transpose(0, 1), tg, ol, tl)
        loss.backward()
        opt.step()

    counts = []

    def block(on):
        replay.ENABLED = on
        for _ in range(3):
            step()
        opt.join()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        h0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        host = (time.perf_counter() - h0) / args.steps * 1e3
        opt.join()
        e.record()
        torch.cuda.synchronize()
        counts.append((on, replay.STATS['replayed_F']))
        return s.elapsed_time(e) / args.steps, host

    replay.ENABLED = True
    for _ in range(10):
        step()
    opt.join()
    torch.cuda.synchronize()
    res = {True: [], False: []}
    counts.clear()
    for b in range(args.blocks):
        for on in (True, False):
            res[on].append(block(on))
    replay.ENABLED = True
    for on in (True, False):
        ms = [r[0] for r in res[on]]
        host = [r[1] for r in res[on]]
        print('%-8s ms/step %s  (mean %.3f)   host enqueue ms/step %s (mean %.3f)' % (
            'replay' if on else 'eager', ' '.join('%.3f' % v for v in ms), sum(ms) / len(ms), ' '.join('%.2f' % v for v in host),
            sum(host) / len(host)))
    print('replayed forward passes after each block:', counts)
    print('replay state:', replay.report(model.engine()), replay.STATS)


if __name__ == '__main__':
    main()
