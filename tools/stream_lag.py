"""Which stream does the backward pass end on?  Events at the join of the main and the weight-gradient stream:
lag > 0 = the weight gradients finish that much after the main stream's last kernel (the step waits for the side stream),
lag < 0 = they were done earlier (the step is on the main stream's timeline).  usage: stream_lag.py [bf16|fp8] [batch] [model]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    which = sys.argv[3] if len(sys.argv) > 3 else 'wav2letter'
    import bench
    from wav2letter_pytorch_amd import Jasper, Wav2Letter, engine as E
    from wav2letter_pytorch_amd.defaults import synthetic_batch
    torch.manual_seed(0)
    if which == 'jasper10x5':
        model = Jasper(bench.jasper10x5_cfg(prec)).cuda().train()
        model.check_nan = False
    else:
        model = Wav2Letter(bench.w2l_cfg(20, precision=prec)).cuda().train()
    opt = model.configure_optimizers()[0][0]
    opt.overlap = True
    x, il, tg, tl = synthetic_batch(N, 1000, seed=1234)
    x, tg, tl = x.cuda(), tg.cuda(), tl.cuda()
    ol = model.compute_output_lengths(il).cuda()
    lens = il if which == 'jasper10x5' else None
    recs = []
    for it in range(16):
        if it == 8:
            E.JOIN_EVENTS = []
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        opt.zero_grad(set_to_none=True)
        a.record()
        out, _ = model(x, lens)
        loss = model.criterion(out.transpose(0, 1), tg, ol, tl)
        loss.backward()
        opt.step()
        b.record()
        recs.append((a, b))
    torch.cuda.synchronize()
    lags = [em.elapsed_time(es) for em, es in E.JOIN_EVENTS]
    steps = [a.elapsed_time(b) for a, b in recs[8:]]
    print(f'{which} {prec} N={N}: step {sum(steps) / len(steps):.2f} ms; weight-gradient stream ends '
          f'{sum(lags) / len(lags):+.3f} ms after the main stream (min {min(lags):+.3f}, max {max(lags):+.3f})')


if __name__ == '__main__':
    main()
