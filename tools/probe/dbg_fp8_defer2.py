"""How far apart do two IDENTICAL models end after three fp8 training steps?  (The question behind the tolerance of
tests/test_gpu_model.py::test_deferred_weight_gradients_fp8_and_jasper.)  Trials of two fresh models stepped in turn; env: DEFER=2|0
(model 0 defers its top units or not), PREC=fp8|bf16, WG / DG = 1|0 (e4m3 weight / data gradients), FOLD / FAST (BatchNorm paths),
CHURN=1 (random allocations between trials), SYNC=1 (device sync after every step), TRIALS.  Round 5: two PLAIN fp8 models are
> 5e-4 of scale apart in 10-30 % of the trials (always by one of a few values, e.g. 1.25e-2: one e4m3 step of one weight), never in
bf16 and never with bf16 data gradients; device syncs change nothing -- the split weight gradients' fp32 atomics give last-bit
differences, the e4m3 requantisation of the weights amplifies them."""
import os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
from oracle import w2l_oracle as O
from wav2letter_pytorch_amd import engine as E
from wav2letter_pytorch_amd.optim import FusedSGD
from test_gpu_model import build_w2l, scale_err
kw = dict(lr=0.05, momentum=0.9, nesterov=True, weight_decay=1e-3)
layers = [(128, 11, 2, 1, 0.0), (256, 13, 1, 1, 0.0), (128, 5, 1, 2, 0.0)]
sd = O.init_wav2letter_state(layers, seed=91)
E.FP8_DGRAD = os.environ.get('DG', '1'); E.FP8_WGRAD = os.environ.get('WG', '1')
PREC = os.environ.get('PREC', 'fp8'); DEFER = int(os.environ.get('DEFER', '2'))
E.FOLD_BN_FWD = os.environ.get('FOLD', '0'); E.FAST_BN_BWD = os.environ.get('FAST', '0') == '1'
x, il, tg, tl = O.synthetic_batch(4, 300, seed=92, s_lo=8, s_hi=25)
junk = []
bad = 0
for trial in range(int(os.environ.get('TRIALS', '30'))):
    if os.environ.get('CHURN'):                      # vary the allocator's state between trials
        junk = [torch.full((int(torch.randint(1, 64, (1,))) << 18,), float('nan') if os.environ.get('POISON') else 0.37, device='cuda') for _ in range(8)]
        del junk[::2]
    models = [build_w2l(layers, sd, PREC).train() for _ in range(2)]
    opts = []
    for i, m in enumerate(models):
        o = FusedSGD.from_sgd(torch.optim.SGD(m.parameters(), **kw)); o.overlap = True
        if i == 0 and DEFER: o.defer_wgrad(m, DEFER)
        opts.append(o)
    for it in range(3):
        for m, o in zip(models, opts):
            o.zero_grad(set_to_none=True)
            out, ol = m(x.cuda(), il)
            m.criterion(out.transpose(0, 1), tg, ol, tl).backward(); o.step()
            if os.environ.get('SYNC'): torch.cuda.synchronize()
    for o in opts: o.join()
    torch.cuda.synchronize()
    errs = {k: scale_err(pa.detach().cpu().numpy(), pb.detach().cpu().numpy()) for (k, pa), (_, pb) in zip(models[0].named_parameters(), models[1].named_parameters())}
    worst = max(errs, key=errs.get)
    if errs[worst] > 5e-4:
        bad += 1
        print('trial', trial, 'BAD', worst, '%.3e' % errs[worst], flush=True)
print('bad trials', bad)
