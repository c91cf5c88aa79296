"""Round-5 probe (kept for the record): the implicit GEMM's split-K combine rewritten like the weight gradient's (16-byte buffer
stores / loads with sc1, no fences) returned 16 wrong dwords -- lanes 12..15 of each 16-lane group, first register of one
accumulator tile -- in ~5 % of the launches of the SIX-wave block shapes (384 threads) only, at random; never with 4 or 8 waves.
Not understood in the time there was; the implicit GEMM keeps its fenced combine (conv_igemm.hip).  This script compares every
split configuration with its unsplit twin and prints the error pattern."""
import ctypes as C, os, sys
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
from wav2letter_pytorch_amd import _lib as L
import test_gpu_kernels as TK
s, d, Kw, pl, pr = 1, 2, 5, 4, 4
N, Cin, Cout, T = 2, 256, 320, 300
x, w, b = TK.conv_inputs(N, Cin, Cout, Kw, T, pl, pr, 11)
xp = TK.to_ntc_padded(x, pl, pr, 1)
rows = xp.shape[1]
Tout = (rows - (Kw - 1) * d - 1) // s + 1
fh, _, _, _, coutp, cinp = TK.pack(L, w, False)
xh = xp.to(torch.bfloat16).cuda()
ws = torch.zeros(int(L.lib.w2l_conv_splitk_workspace_bytes(N, coutp, Tout)), dtype=torch.uint8, device='cuda')
def run(idx):
    y = torch.full((N, Tout, coutp), float('nan'), dtype=torch.bfloat16, device='cuda')
    L.lib.w2l_conv_force_tile_config(idx)
    rc = L.lib.w2l_conv1d_igemm_ws(L.ptr(xh), rows * cinp, N * rows, L.ptr(fh), L.ptr(y), 0, 0, None, None, N, cinp, coutp, Tout, Kw, s, d, L.ptr(ws), ws.numel(), L.stream_ptr())
    L.lib.w2l_conv_force_tile_config(-1)
    torch.cuda.synchronize()
    return rc, y
for idx in list(range(52, 120)):
    rc, y = run(idx)
    if rc: continue
    rc0, y0 = run(idx % 52)
    dlt = (y.float() - y0.float()).abs()
    bad = dlt > 0.05
    print(idx, 'base', idx % 52, 'bad frac %.4f' % bad.float().mean().item(), 'nan', int(torch.isnan(y.float()).sum()),
          'bad per utt', bad.float().mean((1, 2)).tolist(), 'bad cols(ch) first', bad.any(0).any(0).nonzero().flatten()[:6].tolist(), 'rows', bad.any(0).any(1).nonzero().flatten()[:6].tolist())
print('---- detail')
for idx in (61, 69, 71, 113, 95):
    for attempt in range(6):
        rc, y = run(idx)
        if rc: break
        rc0, y0 = run(idx % 52)
        dlt = (y.float() - y0.float())
        bad = dlt.abs() > 0.05
        if bad.any():
            n, t, c = bad.nonzero()[0].tolist()
            t0, c0 = t // 16 * 16, c // 16 * 16
            print(idx, 'attempt', attempt, 'first bad at', (n, t, c))
            torch.set_printoptions(linewidth=250, precision=3, sci_mode=False)
            print(dlt[n, t0:t0 + 16, c0:c0 + 16])
            break
