import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from gpu_helpers import build_w2l
from oracle import w2l_oracle as O
from wav2letter_pytorch_amd import engine as E, replay
from wav2letter_pytorch_amd.optim import FusedSGD
E.FOLD_BN_FWD = '0'; E.FAST_BN_BWD = False
def main():
  layers = [(128, 11, 2, 1, 0.0), (128, 13, 1, 2, 0.0), (192, 5, 1, 1, 0.0)]
  sd = O.init_wav2letter_state(layers, seed=14)
  ma = build_w2l(layers, sd, 'bf16').train()
  mb = build_w2l(layers, sd, 'bf16').train()
  kw = dict(lr=0.05, momentum=0.9, nesterov=True, weight_decay=1e-3)
  oa = FusedSGD.from_sgd(torch.optim.SGD(ma.parameters(), **kw))
  oa.overlap = True
  oa.defer_wgrad(ma, 3)
  ob = torch.optim.SGD(mb.parameters(), **kw)
  x, il, tg, tl = O.synthetic_batch(2, 160, seed=11, s_lo=5, s_hi=15)
  for it in range(4):
      for m, o in ((ma, oa), (mb, ob)):
          o.zero_grad(set_to_none=True)
          out, ol = m(x.cuda(), il)
          m.criterion(out.transpose(0, 1), tg, ol, tl).backward()
          o.step()
      if it == 1:
          ea, eb = (m.eval()(x.cuda(), il)[0] for m in (ma, mb))
          ma.train(), mb.train()
  oa.join()
  torch.cuda.synchronize()
  print('params finite', all(torch.isfinite(p).all().item() for p in ma.parameters()), replay.STATS)
  for m in (ma, mb):
      m.zero_grad(set_to_none=True)
      out, ol = m(x.cuda(), il)
      print('out finite', torch.isfinite(out).all().item())
      m.criterion(out.transpose(0, 1), tg, ol, tl).backward()
  torch.cuda.synchronize()
  eng = ma.engine()
  for r in eng._deferred:
      print('deferred', r['conv'].name, 'dy finite', torch.isfinite(r['dy_hi'].float()).all().item(), 'src finite', torch.isfinite(r['src'].hi.float()).all().item(),
            r['dy_hi'].shape, r['halo'], r['Tout'])
  oa.join()
  torch.cuda.synchronize()
  for n, p in ma.named_parameters():
      print(n, None if p.grad is None else torch.isfinite(p.grad).all().item())
  print(replay.STATS, replay.report(eng))

for rep_ in range(2):
  print('=== pass', rep_)
  main()
