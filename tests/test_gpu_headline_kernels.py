"""The launch configurations the headline bench really uses, judged at KERNEL tolerance.

test_gpu_kernels.py checks every block shape / split-K plan tightly, but on a 256 -> 320, T = 300 toy problem; the
full-size model tests (test_gpu_fullsize.py) run the real shapes, but bf16 end to end over 21 layers, where only a loose
model-level bound holds.  Here each of the 14 distinct convolution shapes of the Wav2Letter table
(configuration/model/wav2letter.yaml:5-104 + the classifier, wav2letter.py:69) runs at the bench's own size -- N = 32,
T' = 500, K up to 25 984 -- through the same C-ABI calls the engine makes in bf16 mode, with the block shape / split-K /
block order MEASURED by the tuner exactly as in the bench's warm-up step:

* forward      w2l_conv1d_igemm_tune_ws + w2l_conv1d_igemm_ws        (bias, BatchNorm partial sums)
* data grad    the same kernel over the flat shared-halo dy sequence  (engine._dgrad)
* weight grad  w2l_conv1d_wgrad_tune_ws + w2l_conv1d_wgrad_ws         (fp32 atomics for split plans)

against a float64 evaluation of the same bf16-rounded operands (per-tap matrix products in torch float64 on the device:
14.7 TFLOP of float64 is minutes on the host cores, a second on the GPU; a slice of each result is re-derived on the CPU so
the reference itself is cross-checked).  Bounds: fp32 results within 2e-5 of the tensor scale (measured 2e-7 .. 3e-6: fp32
accumulation over K <= 25 984), bf16-stored results within half a bf16 ulp (at most 2^-8 of the value) of the float64 value plus 1e-4 of scale."""
import ctypes as C

import pytest
import torch

from test_gpu_kernels import pack

pytestmark = pytest.mark.gpu

N, TOUT = 32, 500


def _shapes():
    """(Cin, Cout, Kw, stride, dilation, has_bn) of the 14 distinct layers, first occurrence order"""
    from wav2letter_pytorch_amd.defaults import wav2letter_model
    cfg = wav2letter_model(20)
    cin, out, seen = cfg.input_size, [], set()
    for r in cfg.layers[:20]:
        key = (cin, r.output_size, r.kernel_size, r.stride, r.dilation, True)
        if key not in seen:
            seen.add(key)
            out.append(key)
        cin = r.output_size
    out.append((cin, len(cfg.labels), 1, 1, 1, False))
    return out


SHAPES = _shapes()


@pytest.fixture(scope='module')
def L():
    from wav2letter_pytorch_amd import _lib
    return _lib


def _err(got, ref):
    return float((got.double() - ref).abs().max() / ref.abs().max())


def _conv_ref64(x64, w64, tout, stride, dil):
    """y[n, t, co] = sum_k x[n, t*stride + k*dil, :] . w[co, :, k]   (x [N, rows, Cin], w [Cout, Cin, Kw], float64)"""
    y = None
    for k in range(w64.shape[2]):
        xs = x64[:, k * dil: k * dil + (tout - 1) * stride + 1: stride, :]
        t = xs @ w64[:, :, k].t()
        y = t if y is None else y + t
    return y


def test_the_table_has_14_distinct_shapes():
    assert len(SHAPES) == 14 and SHAPES[0] == (64, 256, 11, 2, 1, True) and SHAPES[-1] == (1024, 29, 1, 1, 1, False)
    assert (896, 896, 29, 1, 2, True) in SHAPES


@pytest.mark.parametrize('shape', SHAPES, ids=lambda s: '%dto%d_k%d_s%d_d%d' % s[:5])
def test_headline_layer_kernels_exact(L, shape):
    cin, cout, kw, stride, dil, has_bn = shape
    g = torch.Generator().manual_seed(cin * 31 + cout)
    rows = (TOUT - 1) * stride + (kw - 1) * dil + 1      # layer 0: 1000 frames + (4, 5) reflect pad = 1009 rows -> 500 outputs
    x = torch.randn(N, rows, cin, generator=g).clamp_(min=0).mul_(1.5).to(torch.bfloat16)      # post-clamp-like activations
    w = (torch.randn(cout, cin, kw, generator=g) / (cin * kw) ** 0.5).to(torch.bfloat16).float()
    bias = torch.randn(cout, generator=g)
    fh, _, dh, _, coutp, cinp = pack(L, w)
    assert cinp == cin                              # every table width is a multiple of 64
    xd = x.cuda()
    x64, w64 = xd.double(), w.cuda().double()
    st = L.stream_ptr()
    worst = {}
    bd = torch.zeros(coutp, device='cuda')
    bd[:cout] = bias.cuda()

    # ---------------------------------------------------------------- forward (+ bias, BatchNorm partial sums)
    ref = _conv_ref64(x64, w64, TOUT, stride, dil) + bias.cuda().double()
    sub = torch.nn.functional.conv1d(x[:2].float().transpose(1, 2).double(), w.double(), bias.double(), stride=stride,
                                     dilation=dil).transpose(1, 2)
    assert _err(ref[:2].cpu(), sub) < 1e-12         # the device-side float64 reference agrees with torch CPU float64
    tiles = L.lib.w2l_conv_stat_tiles(N, TOUT)
    stats = torch.zeros(tiles, 2, coutp, device='cuda') if has_bn else None
    ws = torch.zeros(int(L.lib.w2l_conv_splitk_workspace_bytes(N, coutp, TOUT)), dtype=torch.uint8, device='cuda')
    y16 = torch.empty(N, TOUT, coutp, dtype=torch.bfloat16, device='cuda')
    fwd = (L.ptr(xd), rows * cin, N * rows, L.ptr(fh))
    dims = (N, cin, coutp, TOUT, kw, stride, dil)
    L.check(L.lib.w2l_conv1d_igemm_tune_ws(*fwd, L.ptr(y16), 0, L.ptr(bd), L.ptr(stats), *dims, 2, L.ptr(ws), ws.numel(), st))
    for f32 in (1, 0):                              # the tuned plan, fp32 store (tight) and the bench's bf16 store
        y = torch.full((N, TOUT, coutp), float('nan'), dtype=torch.float32 if f32 else torch.bfloat16, device='cuda')
        if stats is not None:
            stats.zero_()
        L.check(L.lib.w2l_conv1d_igemm_ws(*fwd, L.ptr(y), f32, 0, L.ptr(bd), L.ptr(stats), *dims, L.ptr(ws), ws.numel(), st))
        torch.cuda.synchronize()
        got = y[:, :, :cout].double()
        scale = float(ref.abs().max())
        if f32:
            worst['fwd'] = _err(got, ref)
            assert worst['fwd'] < 2e-5, ('forward fp32 store', worst['fwd'])
        else:
            slack = (got - ref).abs() - (2.0 ** -8 * 1.01) * ref.abs() - 1e-4 * scale
            assert float(slack.max()) <= 0, ('forward bf16 store', float(slack.max()) / scale)
        if coutp > cout:
            assert not bool(y[:, :, cout:].float().abs().max() > 0)
        if stats is not None:
            s1, s2 = stats[:, 0, :cout].double().sum(0), stats[:, 1, :cout].double().sum(0)
            r1, r2 = ref.sum((0, 1)), (ref * ref).sum((0, 1))
            assert float((s1 - r1).abs().max()) <= 1e-4 * float(ref.abs().sum((0, 1)).max())
            assert float(((s2 - r2).abs() / r2).max()) < 1e-4
    assert not bool(ws[:65536].any())               # split-K tickets are back at zero
    del ref, y, y16

    # ---------------------------------------------------------------- weight gradient: dW[k] = sum_{n,t} dy (x) x[t*s + k*d]
    hb = (kw - 1) * dil
    halo = max(hb, (TOUT + 63) // 64 * 64 - TOUT)
    per = TOUT + halo
    dy = torch.zeros(halo + N * per, coutp, dtype=torch.bfloat16)
    dyv = (torch.randn(N, TOUT, cout, generator=g) * 0.05).to(torch.bfloat16)
    dy[halo:].view(N, per, coutp)[:, :TOUT, :cout] = dyv
    dyd = dy.cuda()
    dy64 = dyv.cuda().double()
    refw = torch.stack([torch.einsum('ntc,nti->ci', dy64, x64[:, k * dil: k * dil + (TOUT - 1) * stride + 1: stride, :])
                        for k in range(kw)])                                       # [Kw, Cout, Cin]
    dyp = C.c_void_p(dyd.data_ptr() + halo * coutp * 2)
    wargs = (dyp, per * coutp, L.ptr(xd), rows * cin, N * rows)
    wdims = (N, cin, coutp, TOUT, kw, stride, dil)
    scratch = torch.empty(kw, coutp, cin, device='cuda')
    L.check(L.lib.w2l_conv1d_wgrad_tune_ws(*wargs, L.ptr(scratch), *wdims, 2, None, 0, st))
    zero = bool(L.lib.w2l_wgrad_needs_zero_ws(N, cin, coutp, TOUT, kw, 0))
    dw = torch.zeros(kw, coutp, cin, device='cuda') if zero else torch.full((kw, coutp, cin), float('nan'), device='cuda')
    L.check(L.lib.w2l_conv1d_wgrad_ws(*wargs, L.ptr(dw), *wdims, 0, None, 0, st))
    torch.cuda.synchronize()
    worst['wgrad'] = _err(dw[:, :cout].double(), refw)
    assert worst['wgrad'] < 2e-5, ('weight gradient', worst['wgrad'])
    if coutp > cout:
        assert not bool(dw[:, cout:].abs().max() > 0)
    # the three-tap kernels with their accumulators in AGPRs (plan order bit 4; csrc/conv_wgrad3_dev.hip), whatever the measured
    # selection took above: the 4-wave and the 8-wave (six taps per block) form, unsplit and split, at this layer's real size --
    # their MFMAs are inline asm, so every hazard the compiler would have padded is this kernel's own business
    if stride == 1 and kw >= 3 and dil <= 4:
        for order, splits in ((16, 1), (17, 3), (20, 1), (21, 2)):
            L.lib.w2l_wgrad_force_plan(splits, order)
            try:
                for rep in range(2):
                    dw3 = torch.zeros(kw, coutp, cin, device='cuda') if splits > 1 else torch.full((kw, coutp, cin), float('nan'), device='cuda')
                    L.check(L.lib.w2l_conv1d_wgrad_ws(*wargs, L.ptr(dw3), *wdims, 0, None, 0, st))
                    torch.cuda.synchronize()
                    e3 = _err(dw3[:, :cout].double(), refw)
                    assert e3 < 2e-5, ('three-tap weight gradient', order, splits, rep, e3)
            finally:
                L.lib.w2l_wgrad_force_plan(0, -1)
    del refw, dw, scratch

    # ---------------------------------------------------------------- data gradient (stride-1 layers; layer 0 has none in training)
    if stride != 1:
        print(f'  {cin}->{cout} k{kw}: forward {worst["fwd"]:.1e}, weight gradient {worst["wgrad"]:.1e} of scale')
        return
    tp = TOUT + hb
    # dx[n, t', ci] = sum_k sum_co dy[n, t' - k*d, co] w[co, ci, k], t' in padded-input coordinates
    dyz = torch.zeros(N, tp + hb, cout, dtype=torch.float64, device='cuda')
    dyz[:, hb: hb + TOUT] = dy64
    refx = None
    for k in range(kw):
        t = dyz[:, hb - k * dil: hb - k * dil + tp, :] @ w64[:, :, k]
        refx = t if refx is None else refx + t
    flat_rows = N * per
    total = dyd.shape[0]
    row_off = halo - hb
    dsrc = (C.c_void_p(dyd.data_ptr() + row_off * coutp * 2), (total - row_off) * coutp, total - row_off, L.ptr(dh))
    ddims = (1, coutp, cin, flat_rows, kw, 1, dil)
    ws = torch.zeros(int(L.lib.w2l_conv_splitk_workspace_bytes(1, cin, flat_rows)), dtype=torch.uint8, device='cuda')
    dx16 = torch.empty(flat_rows, cin, dtype=torch.bfloat16, device='cuda')
    L.check(L.lib.w2l_conv1d_igemm_tune_ws(*dsrc, L.ptr(dx16), 0, None, None, *ddims, 2, L.ptr(ws), ws.numel(), st))
    scale = float(refx.abs().max())
    for f32 in (1, 0):
        dx = torch.full((flat_rows, cin), float('nan'), dtype=torch.float32 if f32 else torch.bfloat16, device='cuda')
        L.check(L.lib.w2l_conv1d_igemm_ws(*dsrc, L.ptr(dx), f32, 0, None, None, *ddims, L.ptr(ws), ws.numel(), st))
        torch.cuda.synchronize()
        got = dx.view(N, per, cin)[:, :tp].double()
        if f32:
            worst['dgrad'] = _err(got, refx)
            assert worst['dgrad'] < 2e-5, ('data gradient fp32 store', worst['dgrad'])
        else:
            slack = (got - refx).abs() - (2.0 ** -8 * 1.01) * refx.abs() - 1e-4 * scale
            assert float(slack.max()) <= 0, ('data gradient bf16 store', float(slack.max()) / scale)
    print(f'  {cin}->{cout} k{kw} d{dil}: forward {worst["fwd"]:.1e}, data gradient {worst["dgrad"]:.1e}, weight gradient '
          f'{worst["wgrad"]:.1e} of scale (fp32 stores)')
