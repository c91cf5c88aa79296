#!/usr/bin/env python3
"""CTC kernel timing + accuracy vs torch CPU at the headline shape (N=32, T'=500, C=29, S in [80,160])."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wav2letter_pytorch_amd.ctc_loss import CTCLoss
g = torch.Generator().manual_seed(0)
N, T, Cn = 32, 500, 29
lp = torch.log_softmax(torch.randn(N, T, Cn, generator=g) * 2, -1)
tl = torch.randint(80, 161, (N,), generator=g, dtype=torch.int32)
tg = torch.randint(1, 29, (N, 160), generator=g, dtype=torch.int32)
il = torch.full((N,), T, dtype=torch.int32)
lr = lp.clone().requires_grad_(True)
ref = F.ctc_loss(lr.transpose(0, 1), tg, il, tl, blank=0, reduction='mean', zero_infinity=True); ref.backward()
ld = lp.cuda().requires_grad_(True)
crit = CTCLoss(0, 'mean', True)
tgd, ild, tld = tg.cuda(), il.cuda(), tl.cuda()
loss = crit(ld.transpose(0, 1), tgd, ild, tld); loss.backward()
print('loss', float(loss), float(ref), 'rel err', abs(float(loss) - float(ref)) / float(ref))
print('grad max abs err', float((ld.grad.cpu() - lr.grad).abs().max()), 'scale', float(lr.grad.abs().max()))
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3): crit(ld.transpose(0, 1), tgd, ild, tld)
s.record()
for _ in range(20): crit(ld.transpose(0, 1), tgd, ild, tld)
e.record(); torch.cuda.synchronize()
print('ctc fwd+grad: %.1f us' % (s.elapsed_time(e) / 20 * 1e3))
