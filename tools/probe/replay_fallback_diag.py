"""diagnostic: is the 48-channel Jasper fixture's eager step bit-reproducible from run to run? (tests/test_gpu_replay.py)"""
import ast, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from gpu_helpers import build_jasper
from oracle import w2l_oracle as O
from wav2letter_pytorch_amd import engine as E
import test_gpu_replay as T
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
make = lambda: build_jasper(meta['blocks'], sd, 'bf16')
runs = [T._run(make, batches, 8, False)[0], T._run(make, batches, 8, False)[0], T._run(make, batches, 8, True)[0], T._run(make, batches, 8, True)[0]]
for r in runs: print(['%.7f' % v for v in r])
