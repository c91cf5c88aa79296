"""diagnostic: does any kernel of the step read memory nobody wrote?  torch.empty filled with NaN (torch.utils.deterministic.
fill_uninitialized_memory) must not change a single loss."""
import ast, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from gpu_helpers import build_jasper, build_w2l
from oracle import w2l_oracle as O
from wav2letter_pytorch_amd import engine as E, replay
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
make48 = lambda: build_jasper(meta['blocks'], sd, 'bf16')
makew, bw = T._w2l_case()
for name, make, bt in (('jasper48', make48, batches), ('w2l', makew, bw)):
    for fill in (False, True):
        torch.use_deterministic_algorithms(fill, warn_only=True)
        torch.utils.deterministic.fill_uninitialized_memory = fill
        try:
            r = T._run(make, bt, 5, False)[0]
        finally:
            torch.use_deterministic_algorithms(False)
        print(name, 'nan-filled empty' if fill else 'plain', ['%.7f' % v for v in r], flush=True)
