"""GPU-side phase times of a training step from HIP events on the main stream (forward + CTC, backward, optimizer launches on the\nmain stream, gap to the next forward): 4.75 / 8.9 / 0.07 / 0.007 ms on MI355X -- no host-induced gap at the step boundary."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from wav2letter_pytorch_amd import Wav2Letter
from wav2letter_pytorch_amd.defaults import wav2letter_model, synthetic_batch
torch.manual_seed(0)
model = Wav2Letter(wav2letter_model(20)).cuda().train()
opt = model.configure_optimizers()[0][0]
opt.overlap = os.environ.get('OV', '1') == '1'
x, il, tg, tl = synthetic_batch(32, 1000, seed=1234)
x = x.cuda(); tg = tg.cuda(); tl = tl.cuda(); ol = model.compute_output_lengths(il).cuda()
ev = lambda: torch.cuda.Event(enable_timing=True)
recs = []
for it in range(14):
    a, b, c, d = ev(), ev(), ev(), ev()
    opt.zero_grad(set_to_none=True)
    a.record()                       # forward start
    out, _ = model(x, None)
    loss = model.criterion(out.transpose(0, 1), tg, ol, tl)
    b.record()                       # forward + CTC enqueued
    loss.backward()
    c.record()                       # backward done (main stream incl. join)
    opt.step()
    d.record()                       # after optimizer launches on main
    recs.append((a, b, c, d))
torch.cuda.synchronize()
for i in range(6, 13):
    a, b, c, d = recs[i]
    na = recs[i + 1][0]
    print('fwd+ctc %.2f  bwd %.2f  opt(main) %.3f  d->next fwd start %.3f  step %.2f' % (
        a.elapsed_time(b), b.elapsed_time(c), c.elapsed_time(d), d.elapsed_time(na), a.elapsed_time(na)))
