"""The reference's nn.Modules called on their own (wav2letter.py:40-47, jasper.py:107-132, :379-419), eval-mode
gradients, operand-pack invalidation: each runs through a one-unit (open) step engine with autograd and is compared with
the reference-generated per-op fixture or with the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from gpu_helpers import build_w2l, scale_err

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), 'golden')


@pytest.mark.parametrize('tag', ['asym_s2', 'dil2', 'k1', 'even_k13'])
@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_conv1dblock_standalone_reference_fixture(tag, precision):
    """tests/golden/ops_conv1dblock.npz was produced by the reference's Conv1dBlock (asymmetric (4,5) reflect pad with
    stride 2; k29 dilation 2; 1x1; k13), non-trivial gamma / beta, a random upstream gradient: block(x), block.backward(gy)
    on the device vs the reference's y, dL/dx, every parameter gradient and the running statistics."""
    from wav2letter_pytorch_amd import Conv1dBlock
    z = np.load(os.path.join(GOLD, 'ops_conv1dblock.npz'))
    cin, cout, k, s, d, T = [int(v) for v in z[f'{tag}/cfg']]
    blk = Conv1dBlock(cin, cout, (k,), s, drop_out_prob=0.0, dilation=d)
    assert (blk.pad_l, blk.pad_r) == tuple(int(v) for v in z[f'{tag}/pad'])
    blk.load_state_dict({kk: torch.from_numpy(z[f'{tag}/p/{kk}'].copy()) for kk in
                         ('conv1.weight', 'conv1.bias', 'batch_norm.weight', 'batch_norm.bias')}, strict=False)
    blk = blk.cuda().train()
    blk.precision = precision
    x = torch.from_numpy(z[f'{tag}/x']).cuda().requires_grad_(True)
    y = blk(x)
    y.backward(torch.from_numpy(z[f'{tag}/gy']).cuda())
    tol = dict(fp32=1e-3, bf16=3e-2)[precision]
    assert y.shape == z[f'{tag}/y'].shape
    assert scale_err(y.detach().cpu().numpy(), z[f'{tag}/y']) < tol
    # gradients: a clamp gate decided differently within rounding of 0 / 20 moves single elements -> judged on the L2 norm
    gtol = dict(fp32=2e-3, bf16=1e-1)[precision]        # bf16: ~1 % of the gates flip (y within 8 bits of a bound)
    gx = x.grad.cpu().numpy()
    assert np.linalg.norm(gx - z[f'{tag}/gx']) < gtol * np.linalg.norm(z[f'{tag}/gx'])
    wscale = np.abs(z[f'{tag}/g/conv1.weight']).max()
    for kk, p in blk.named_parameters():
        ref = z[f'{tag}/g/{kk}']
        got = p.grad.cpu().numpy()
        assert got.shape == ref.shape, kk
        if kk == 'conv1.bias':            # identically zero under BatchNorm; the reference holds rounding noise
            assert np.abs(got - ref).max() < 1e-3 * wscale
        else:
            assert np.linalg.norm(got - ref) < gtol * np.linalg.norm(ref), kk
    assert scale_err(blk.batch_norm.running_mean.cpu().numpy(), z[f'{tag}/running_mean']) < tol
    assert scale_err(blk.batch_norm.running_var.cpu().numpy(), z[f'{tag}/running_var']) < tol
    assert int(blk.batch_norm.num_batches_tracked) == 1


def test_classifier_block_and_plain_conv_standalone():
    """the bn=False / activation_use=False block of wav2letter.py:69 (bias gradient = sum dy, no BatchNorm) and the bare
    Conv1d holder, against torch CPU ops"""
    import torch.nn.functional as F
    from wav2letter_pytorch_amd import Conv1dBlock
    torch.manual_seed(3)
    blk = Conv1dBlock(96, 29, (1,), 1, bn=False, activation_use=False).cuda().train()
    blk.precision = 'fp32'
    x = torch.randn(3, 96, 41)
    gy = torch.randn(3, 29, 41)
    xd = x.cuda().requires_grad_(True)
    y = blk(xd)
    y.backward(gy.cuda())
    w, b = blk.conv1.weight.detach().cpu().contiguous().requires_grad_(True), blk.conv1.bias.detach().cpu().requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    yr = F.conv1d(xr, w, b)
    yr.backward(gy)
    assert scale_err(y.detach().cpu().numpy(), yr.detach().numpy()) < 1e-3
    assert scale_err(xd.grad.cpu().numpy(), xr.grad.numpy()) < 1e-3
    assert scale_err(blk.conv1.weight.grad.cpu().numpy(), w.grad.numpy()) < 1e-3
    assert scale_err(blk.conv1.bias.grad.cpu().numpy(), b.grad.numpy()) < 1e-3
    conv = blk.conv1
    conv.precision = 'fp32'
    y2 = conv(x.cuda())
    assert scale_err(y2.detach().cpu().numpy(), yr.detach().numpy()) < 1e-3


def test_masked_conv1d_and_jasper_block_standalone():
    """MaskedConv1d(x, lens) and JasperBlock((x, lens)) on their own vs the oracle's restatement (jasper.py:107-132,
    :379-419): masked input, float length update, dense repeat-2 block with the residual 1x1 conv + BatchNorm"""
    from oracle import w2l_oracle as O
    from wav2letter_pytorch_amd.jasper import JasperBlock, MaskedConv1d
    torch.manual_seed(11)
    mc = MaskedConv1d(64, 96, 11, stride=2, padding=5, bias=True).cuda().train()
    mc.precision = 'fp32'
    x = torch.randn(3, 64, 90)
    lens = torch.tensor([90, 61, 33])
    y, l2 = mc(x.cuda(), lens)
    w, b = mc.conv.weight.detach().cpu().contiguous(), mc.conv.bias.detach().cpu()
    yr, lr = O.masked_conv1d(x, lens, w, b, stride=2, padding=5, dilation=1, groups=1)
    assert scale_err(y.detach().cpu().numpy(), yr.numpy()) < 1e-3
    assert torch.equal(l2.cpu(), lr) and l2.dtype == torch.float32          # true division (jasper.py:109-112)

    blk_cfg = dict(layer_size=96, kernel_size=11, stride=1, dilation=1, residual=True, repeat=2, separable=False)
    blk = JasperBlock(64, 96, repeat=2, kernel_size=11, residual=True, separable=False, conv_mask=True,
                      activation=torch.nn.ReLU()).cuda().train()
    blk.precision = 'fp32'
    with torch.no_grad():
        for m in blk.modules():
            if type(m).__name__ == 'BatchNorm1d':
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.3, 0.3)
    sd = {'b.' + k: v.detach().cpu().contiguous().clone() for k, v in blk.state_dict().items()}
    xd = x.cuda().requires_grad_(True)
    out, lo = blk((xd, lens))
    gy = torch.randn(out.shape)
    out.backward(gy.cuda())
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and 'running' not in k}
    work = dict(sd)
    work.update(params)
    xr = x.clone().requires_grad_(True)
    gates = []
    free, _ = O.jasper_block_forward(xr, lens, dict(work), 'b.', blk_cfg, training=True, update_stats=False, inter=gates)
    outr, lor = O.jasper_block_forward(xr, lens, work, 'b.', blk_cfg, training=True,
                                       gates=[g.detach() > 0 for g in gates])
    outr.backward(gy)
    assert torch.equal(lo.cpu(), lor.float())
    assert scale_err(out.detach().cpu().numpy(), outr.detach().numpy()) < 1e-3
    assert np.linalg.norm(xd.grad.cpu().numpy() - xr.grad.numpy()) < 2e-3 * np.linalg.norm(xr.grad.numpy())
    for k, p in blk.named_parameters():
        r = params['b.' + k].grad.numpy()
        assert np.linalg.norm(p.grad.cpu().numpy() - r) < 2e-3 * np.linalg.norm(r), k


def test_eval_mode_backward_uses_running_statistics():
    """loss.backward() through a model in eval(): BatchNorm normalises with its running statistics, which are constants,
    so dy = scale * g and the conv bias in front of it has the ordinary gradient sum(dy) -- checked against autograd on
    the oracle (parameter gradients for frozen-BN fine-tuning, and the spectrogram gradient)"""
    from oracle import w2l_oracle as O
    layers = [(64, 11, 2, 1, 0.0), (96, 13, 1, 1, 0.0), (64, 7, 1, 2, 0.0)]
    sd = O.init_wav2letter_state(layers, seed=17)
    g = torch.Generator().manual_seed(18)
    for k in sd:
        if 'running_mean' in k:
            sd[k] = torch.randn(sd[k].shape, generator=g) * 0.1
        elif 'running_var' in k:
            sd[k] = torch.rand(sd[k].shape, generator=g) * 0.5 + 0.05
        elif 'batch_norm.weight' in k:
            sd[k] = torch.rand(sd[k].shape, generator=g) + 0.5
    model = build_w2l(layers, sd, 'fp32').eval()
    x, il, tg, tl = O.synthetic_batch(3, 150, seed=19, s_lo=5, s_hi=20)
    xd = x.cuda().requires_grad_(True)
    out, ol = model(xd, il)
    loss = model.criterion(out.transpose(0, 1), tg, ol, tl)
    loss.backward()
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and 'running' not in k}
    work = {k: v.clone() for k, v in sd.items()}
    work.update(params)
    xr = x.clone().requires_grad_(True)
    lp, olr = O.wav2letter_forward(xr, work, layers, training=False, input_lengths=il)
    lr = O.ctc_criterion(lp, tg, olr, tl)
    lr.backward()
    assert scale_err(out.detach().cpu().numpy(), lp.detach().numpy()) < 1e-3
    assert abs(float(loss) - float(lr)) < 1e-4 * abs(float(lr))
    for k, p in model.named_parameters():
        r = params[k].grad.numpy()
        assert np.linalg.norm(p.grad.cpu().numpy() - r) < 3e-3 * max(np.linalg.norm(r), 1e-12), k
    assert float(model.conv1ds.conv1d_1.conv1.bias.grad.abs().max()) > 0       # NOT zero in eval mode
    assert np.linalg.norm(xd.grad.cpu().numpy() - xr.grad.numpy()) < 3e-3 * np.linalg.norm(xr.grad.numpy())
    for k, v in model.state_dict().items():                                      # eval mode leaves the buffers alone
        if 'running' in k or 'num_batches' in k:
            assert torch.equal(v.cpu(), sd[k]), k


def test_invalidate_packed_after_raw_weight_update():
    """weights changed through .data (old-style optimizers, EMA, broadcast) do not bump Parameter._version: the cached bf16
    operands stay until engine.invalidate_packed() is called, after which the forward follows the master weights again"""
    from oracle import w2l_oracle as O
    from wav2letter_pytorch_amd.engine import invalidate_packed
    layers = [(64, 11, 2, 1, 0.0)]
    model = build_w2l(layers, O.init_wav2letter_state(layers, seed=2), 'bf16').eval()
    x = torch.randn(2, 64, 80).cuda()
    with torch.no_grad():
        a, _ = model(x)
        model.conv1ds.conv1d_0.conv1.weight.data.mul_(1.5)
        stale, _ = model(x)
        assert torch.equal(stale, a)
        assert invalidate_packed(model) >= 1
        fresh, _ = model(x)
        assert not torch.equal(fresh, a)
        model.conv1ds.conv1d_0.conv1.weight.mul_(1.0 / 1.5)             # a version-bumping in-place op needs no call
        back, _ = model(x)
        assert scale_err(back.cpu().numpy(), a.cpu().numpy()) < 2e-2


@pytest.mark.parametrize('precision,tol', [('fp32', 1e-4), ('bf16', 2e-2)])
@pytest.mark.parametrize('k,stride,dil,c', [(33, 1, 1, 256), (11, 2, 1, 128), (13, 1, 2, 96)])
def test_depthwise_conv_standalone(precision, tol, k, stride, dil, c):
    """the depthwise half of Jasper's separable blocks called as a module of its own -- MaskedConv1d(C, C, k, groups=C)
    (jasper.py:96-105,319-330: masked_fill by length, conv, float length update) and the bare grouped Conv1d holder --
    forward, dL/dx and dL/dw against torch CPU ops"""
    import torch.nn.functional as F
    from wav2letter_pytorch_amd.jasper import MaskedConv1d, get_same_padding
    from wav2letter_pytorch_amd.layers import Conv1d
    torch.manual_seed(k)
    pad = get_same_padding(k, stride, dil)
    m = MaskedConv1d(c, c, k, stride=stride, padding=pad, dilation=dil, groups=c).cuda()
    m.conv.precision = precision
    n, t = 3, 301
    x = torch.randn(n, c, t)
    lens = torch.tensor([t, 200, 77])
    w = m.conv.weight.detach().cpu().clone()
    assert tuple(w.shape) == (c, 1, k)
    xd = x.cuda().requires_grad_(True)
    y, lens_out = m(xd, lens)
    gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(1))
    y.backward(gy.cuda())
    xin = x.clone().requires_grad_(True)
    xr = xin * (torch.arange(t)[None, None, :] < lens[:, None, None])          # masked_fill(t >= len, 0), jasper.py:116-119
    wr = w.clone().requires_grad_(True)
    yr = F.conv1d(xr, wr, None, stride=stride, padding=pad, dilation=dil, groups=c)
    yr.backward(gy)
    want_lens = (lens + 2 * pad - dil * (k - 1) - 1) / stride + 1           # true division: float lengths (jasper.py:109-112)
    assert torch.equal(lens_out.cpu().float(), want_lens.float())
    assert y.shape == yr.shape and scale_err(y.detach().cpu().numpy(), yr.detach().numpy()) < tol
    gx = xd.grad.cpu() * (torch.arange(t)[None, None, :] < lens[:, None, None])
    assert torch.equal(gx, xd.grad.cpu())                                    # no gradient flows into masked frames
    assert scale_err(gx.numpy(), xin.grad.numpy()) < tol
    assert m.conv.weight.grad.shape == (c, 1, k)
    assert scale_err(m.conv.weight.grad.cpu().numpy(), wr.grad.numpy()) < tol
    # the bare holder (no mask, with a bias)
    conv = Conv1d(c, c, k, stride=stride, padding=pad, dilation=dil, groups=c, bias=True).cuda()
    conv.precision = precision
    x2 = x.cuda().requires_grad_(True)
    y2 = conv(x2)
    y2.backward(gy.cuda())
    w2, b2 = conv.weight.detach().cpu().clone().requires_grad_(True), conv.bias.detach().cpu().clone().requires_grad_(True)
    x2r = x.clone().requires_grad_(True)
    F.conv1d(x2r, w2, b2, stride=stride, padding=pad, dilation=dil, groups=c).backward(gy)
    assert scale_err(x2.grad.cpu().numpy(), x2r.grad.numpy()) < tol
    assert scale_err(conv.weight.grad.cpu().numpy(), w2.grad.numpy()) < tol
    assert scale_err(conv.bias.grad.cpu().numpy(), b2.grad.numpy()) < max(tol, 1e-4)
    with pytest.raises(NotImplementedError):
        Conv1d(64, 64, 3, groups=4).cuda()(torch.randn(1, 64, 20).cuda())
