"""diagnostic: first step / parameter at which the dropped-recording path departs from the eager run (48-channel Jasper fixture)"""
import ast, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from gpu_helpers import build_jasper
from oracle import w2l_oracle as O
from wav2letter_pytorch_amd import engine as E, replay
from wav2letter_pytorch_amd.optim import FusedSGD
E.FOLD_BN_FWD = '0'; E.FAST_BN_BWD = False; E.DETERMINISTIC_WGRAD = True
z = np.load(os.path.join(ROOT, 'tests/golden/jasper_dense.npz'), allow_pickle=True)
meta = ast.literal_eval(str(z['meta']))
sd = {k[3:]: torch.from_numpy(z[k].copy()) for k in z.files if k.startswith('p0/')}
batches = []
g = torch.Generator().manual_seed(3)
for b in range(3):
    x, il, tg, tl = O.synthetic_batch(4, 240, seed=60 + b, s_lo=5, s_hi=15)
    il = torch.randint(120, 241, (4,), generator=g, dtype=torch.int32); il[b] = 240
    for n in range(4): x[n, :, int(il[n]):] = 0
    tl = torch.minimum(tl, (il // 8).to(torch.int32)).clamp(min=1)
    batches.append((x.cuda(), il, tg.cuda(), tl.cuda()))

def run(on):
    replay.ENABLED = on
    torch.manual_seed(11)
    model = build_jasper(meta['blocks'], sd, 'bf16').cuda().train(); model.check_nan = False
    opt = FusedSGD.from_sgd(torch.optim.SGD(model.parameters(), lr=0.02, momentum=0.9, nesterov=True, weight_decay=1e-4)); opt.overlap = True
    snaps = []
    for i in range(7):
        x, il, tg, tl = batches[i % 3]
        opt.zero_grad(set_to_none=True)
        out, ol = model(x, il)
        loss = model.criterion(out.transpose(0, 1), tg, ol, tl)
        loss.backward()
        torch.cuda.synchronize()
        grads = {k: (p.grad.detach().float().cpu().numpy().copy() if p.grad is not None else None) for k, p in model.named_parameters()}
        opt.step(); opt.join(); torch.cuda.synchronize()
        snaps.append((float(loss), grads, {k: v.detach().cpu().numpy().copy() for k, v in model.named_parameters()},
                      {k: v.detach().cpu().numpy().copy() for k, v in model.named_buffers()}))
    return snaps
a, b = run(False), run(True)
for i, (sa, sb) in enumerate(zip(a, b)):
    gd = [k for k in sa[1] if sa[1][k] is not None and not np.array_equal(sa[1][k], sb[1][k])]
    pd = [k for k in sa[2] if not np.array_equal(sa[2][k], sb[2][k])]
    bd = [k for k in sa[3] if not np.array_equal(sa[3][k], sb[3][k])]
    print(i, 'loss', sa[0], sb[0], 'grads differ:', gd[:4], 'params differ:', pd[:4], 'buffers differ:', bd[:4])
print(replay.STATS)
