"""Which operand of the fp8 data gradient differs between two identical models?  Wraps engine._dgrad / _dy_e4m3 / _fp8_weights and
logs checksums per call; two plain models, one step each per trial, allocator churn between trials."""
import os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
from oracle import w2l_oracle as O
from wav2letter_pytorch_amd import engine as E
from wav2letter_pytorch_amd.optim import FusedSGD
from test_gpu_model import build_w2l
layers = [(128, 11, 2, 1, 0.0), (256, 13, 1, 1, 0.0), (128, 5, 1, 2, 0.0)]
sd = O.init_wav2letter_state(layers, seed=91)
E.FP8_DGRAD = '1'; E.FP8_WGRAD = os.environ.get('WG', '1'); E.FOLD_BN_FWD = '0'; E.FAST_BN_BWD = False
x, il, tg, tl = O.synthetic_batch(4, 300, seed=92, s_lo=8, s_hi=25)
LOG = []
def cs(t):
    return None if t is None else float(t.float().abs().double().sum())
real_dgrad = E.StackEngine._dgrad
def dgrad(self, conv, pk, dy_hi, dy_lo, halo, Tout, src, producer=None, amax=None):
    out = real_dgrad(self, conv, pk, dy_hi, dy_lo, halo, Tout, src, producer=producer, amax=amax)
    rec = {'name': conv.name, 'dy': cs(dy_hi), 'amax': None if amax is None else float(amax.max()), 'dx': cs(out[0]), 'rows': out[4]}
    hit = self._dyq.get(id(dy_hi))
    if hit is not None:
        rec['dyq'] = cs(hit[0]); rec['inv'] = float(hit[1])
    st = conv.weight.__dict__.get('_w2l_fp8')
    if st is not None:
        rec['wqd'] = cs(st['qd']); rec['scale'] = st['scale']
    # the rows the consumer reads
    N = src.N; per = out[4]; Tp = src.T + conv.pad_l + conv.pad_r
    rec['dx_valid'] = cs(out[0].view(N, per, -1)[:, :Tp])
    LOG.append(rec)
    return out
E.StackEngine._dgrad = dgrad
real_fw = E._fp8_weights
def fw(conv, pk, dgrad=False):
    q, scale = real_fw(conv, pk, dgrad)
    LOG.append({'name': conv.name + ('/qd' if dgrad else '/q'), 'scale': scale, 'q': cs(q), 'pk': cs(pk.dgr_hi if dgrad else pk.fwd_hi), 'ver': pk.version})
    return q, scale
E._fp8_weights = fw
real_wg = E.StackEngine._wgrad_now
def wg(self, conv, pk, dy_hi, dy_lo, halo, Tout, src, grads, fork=None, f8=None, sink=None):
    r = real_wg(self, conv, pk, dy_hi, dy_lo, halo, Tout, src, grads, fork=fork, f8=f8, sink=sink)
    torch.cuda.synchronize()
    g = grads.get(id(conv.weight)) if isinstance(grads, dict) else None
    LOG.append({'name': conv.name + '/dW', 'dy': cs(dy_hi), 'x': cs(src.hi), 'xq': cs(src.q), 'f8': f8 is not None, 'dw': cs(g) if g is not None else None})
    return r
E.StackEngine._wgrad_now = wg
kw = dict(lr=0.05, momentum=0.9, nesterov=True, weight_decay=1e-3)
bad = 0
for trial in range(int(os.environ.get('TRIALS', '40'))):
    junk = [torch.full((int(torch.randint(1, 64, (1,))) << 18,), 0.37, device='cuda') for _ in range(8)]
    del junk[::2]
    logs = []
    for i in range(2):
        m = build_w2l(layers, sd, 'fp8').train()
        o = FusedSGD.from_sgd(torch.optim.SGD(m.parameters(), **kw)); o.overlap = True
        LOG.clear()
        for it in range(int(os.environ.get('STEPS', '2'))):
            o.zero_grad(set_to_none=True)
            out, ol = m(x.cuda(), il)
            m.criterion(out.transpose(0, 1), tg, ol, tl).backward(); o.step()
            torch.cuda.synchronize()
        o.join(); torch.cuda.synchronize()
        logs.append([dict(r) for r in LOG])
    for a, b in zip(*logs):
        diff = {k: (a[k], b[k]) for k in a if a[k] != b[k]}
        if diff:
            bad += 1
            print('trial', trial, a['name'], diff, flush=True)
            break
print('bad trials', bad)
