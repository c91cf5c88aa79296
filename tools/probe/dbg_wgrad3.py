import sys, os, ctypes as C
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import torch, torch.nn.functional as F
from wav2letter_pytorch_amd import _lib as L
torch.manual_seed(0)
for (Kw, d) in ((5, 2), (6, 1), (7, 2), (8, 1), (13, 1)):
    N, Cin, Cout, T = 3, 192, 320, 333
    pl = pr = 4
    x = torch.randn(N, Cin, T)
    xp = F.pad(x, (pl, pr)).transpose(1, 2).contiguous()
    rows = xp.shape[1]
    Tout = rows - (Kw - 1) * d
    dy = torch.randn(N, Cout, Tout)
    hb = (Kw - 1) * d
    ha = max(hb, (Tout + 63) // 64 * 64 - Tout)
    dyp = F.pad(dy, (hb, ha)).transpose(1, 2).contiguous()
    drows = dyp.shape[1]
    dyh, xh = dyp.to(torch.bfloat16).cuda(), xp.to(torch.bfloat16).cuda()
    ref = torch.nn.grad.conv1d_weight(xp.transpose(1, 2).to(torch.bfloat16).float(), (Cout, Cin, Kw), dy.to(torch.bfloat16).float(), dilation=d)
    for order in (0, 1, 4, 5, 8, 9, 16, 17, 20, 21, 16):
        dw = torch.zeros(Kw, Cout, Cin, device='cuda')
        L.lib.w2l_wgrad_force_plan(1, order)
        L.check(L.lib.w2l_conv1d_wgrad(C.c_void_p(dyh.data_ptr() + hb * Cout * 2), drows * Cout, L.ptr(xh), rows * Cin,
                                       N * rows, L.ptr(dw), N, Cin, Cout, Tout, Kw, 1, d, 0, L.stream_ptr()))
        torch.cuda.synchronize()
        got = dw.cpu().permute(1, 2, 0)
        bad = ~torch.isfinite(got)
        err = ((got - ref).abs() / ref.abs().max())
        err[bad] = 0
        print('Kw', Kw, 'd', d, 'order', order, 'nan', int(bad.sum()), 'per tap maxerr', ['%.1e' % float(err[:, :, k].max()) for k in range(Kw)])
        if bad.any():
            idx = bad.nonzero()
            print('   co range', int(idx[:, 0].min()), int(idx[:, 0].max()), 'ci range', int(idx[:, 1].min()), int(idx[:, 1].max()))
L.lib.w2l_wgrad_force_plan(0, -1)
