"""Shared helpers of the GPU parity tests: run one step on the device, replay its
activation-gate (and dropout) decisions through the CPU oracle, compare."""
import numpy as np
import torch

from oracle import w2l_oracle as O


def scale_err(got, ref):
    ref = np.asarray(ref, dtype=np.float64)
    return float(np.abs(np.asarray(got, dtype=np.float64) - ref).max() / max(np.abs(ref).max(), 1e-12))


def l2_cos(got, ref):
    """(relative L2 error, cosine) of two tensors, in float64"""
    g = np.asarray(got, dtype=np.float64).ravel()
    r = np.asarray(ref, dtype=np.float64).ravel()
    ng, nr = np.linalg.norm(g), np.linalg.norm(r)
    return float(np.linalg.norm(g - r) / max(nr, 1e-30)), float(np.dot(g, r) / max(ng * nr, 1e-30))


def build_w2l(layers, sd, precision, labels=None, dropout=False):
    from wav2letter_pytorch_amd import Wav2Letter
    from wav2letter_pytorch_amd.config import to_cfg
    labels = labels or O.ENGLISH_LOWERCASE
    cfg = to_cfg(dict(name='wav2letter', mid_layers=len(layers), input_size=64, labels=labels, precision=precision,
                      layers=[dict(output_size=c, kernel_size=k, stride=s, dilation=d, dropout=(p if dropout else 0.0))
                              for c, k, s, d, p in layers],
                      audio_conf=dict(window='hamming', window_stride=0.01, window_size=0.02, sample_rate=16000),
                      decoder=dict(_target_='decoder.GreedyDecoder', labels=labels),
                      optimizer=dict(_target_='torch.optim.SGD', lr=1e-5, momentum=0.9, nesterov=True, weight_decay=1e-5),
                      scheduler=dict(_target_='torch.optim.lr_scheduler.ExponentialLR', gamma=0.999)))
    model = Wav2Letter(cfg)
    model.load_state_dict({k: v.clone() for k, v in sd.items()})
    return model.cuda()


def device_step(model, x, il, tg, tl):
    """forward + CTC + backward through the public module surface (autograd path);
    returns out, out_lens, loss and the engine context (saved activations)."""
    model._debug_keep_ctx = True
    model.zero_grad(set_to_none=True)
    out, out_lens = model(x.cuda(), il)
    loss = model.criterion(out.transpose(0, 1), tg, out_lens, tl)
    loss.backward()
    torch.cuda.synchronize()
    return out.detach(), out_lens, loss.detach(), model._last_ctx


def device_gates(ectx, relu=False):
    """activation gradient gates the device used, recovered from its stored activations:
    clamp passes gradient iff 0 < a < 20 (exact ties at the bounds have measure zero), ReLU iff a > 0."""
    gates = []
    for i, uc in enumerate(ectx['units']):
        act = ectx['acts'][i + 1]
        a = act.hi[:, act.pad_l:act.pad_l + act.T, :act.C].float()
        if act.lo is not None:
            a = a + act.lo[:, act.pad_l:act.pad_l + act.T, :act.C].float()
        g = (a > 0) if relu else ((a > 0) & (a < 20))
        gates.append(g.transpose(1, 2).cpu())
    return gates


def device_dropout_masks(ectx, channels):
    masks = []
    for uc, c in zip(ectx['units'], channels):
        if uc.mask is None:
            masks.append(None)
            continue
        bits = uc.mask.cpu().numpy()
        n = ectx['x_shape'][0]
        m = np.unpackbits(bits[:, None], axis=1, bitorder='little').reshape(n, uc.Tout, -1)[:, :, :c]
        masks.append(torch.from_numpy(m.astype(np.float32)).transpose(1, 2))
    return masks


def check_gate_ties(ref_step, gates, layers, tie=2e-3, max_frac=2e-3):
    """every disagreement between the oracle's own clamp gate and the device's must sit within
    ``tie`` of a clamp bound in the oracle's activations, and be rare."""
    for a, g in zip(ref_step['activations'], gates):
        own = (a > 0) & (a < 20)
        dis = own != g
        if dis.any():
            near = (a.abs() < tie) | ((a - 20).abs() < tie)
            assert bool((dis & ~near).sum() == 0), 'activation gate differs away from a clamp bound'
            assert dis.float().mean() < max_frac


def compare_step(model, layers, sd, x, il, tg, tl, precision, drop=False, tie=None, max_frac=None, fp8_layers=()):
    """device step vs oracle step (gates/masks replayed); returns dict of errors.  ``tie`` / ``max_frac``: how far from
    a clamp bound (in the oracle's free-running activations) a differing gate decision may sit, and how many may differ.
    ``fp8_layers``: conv blocks the oracle evaluates under its e4m3 operand model (oracle.fp8_conv1d) -- the statement of
    what ``precision: fp8`` computes."""
    out, out_lens, loss, ectx = device_step(model, x, il, tg, tl)
    gates = device_gates(ectx)
    masks = device_dropout_masks(ectx, [l[0] for l in layers]) if drop else None
    sd_ref = {k: v.clone() for k, v in sd.items()}
    free = O.wav2letter_step(x, il, tg, tl, {k: v.clone() for k, v in sd.items()}, layers, drop_masks=masks)
    if tie != 'skip':      # (fp8 over 21 layers: the free-running activations are too far apart for the check to mean anything)
        check_gate_ties(free, gates, layers, tie=tie if tie is not None else (2e-3 if precision == 'fp32' else 0.25),
                        max_frac=max_frac if max_frac is not None else (2e-3 if precision == 'fp32' else 0.05))
    ref = O.wav2letter_step(x, il, tg, tl, sd_ref, layers, drop_masks=masks, gates=gates, fp8_layers=fp8_layers)
    compare_step.ref = ref
    errs = {'log_probs': scale_err(out.cpu().numpy(), ref['log_probs'].numpy()),
            'loss': abs(float(loss) - float(ref['loss'])) / max(1.0, abs(float(ref['loss'])))}
    head = f'conv1ds.conv1d_{len(layers)}.'
    compare_step.norms = {}                 # per parameter: (relative L2 error, cosine) of the gradient, beside the max-norm in errs
    for k, p in model.named_parameters():
        assert p.grad is not None and p.grad.shape == p.shape, k
        r = ref['grads'][k].numpy()
        compare_step.norms[k] = l2_cos(p.grad.cpu().numpy(), r)
        if k.endswith('conv1.bias') and not k.startswith(head):
            # conv bias under BatchNorm: the true gradient is 0; the reference holds fp32 rounding noise
            wscale = np.abs(ref['grads'][k.replace('bias', 'weight')].numpy()).max()
            errs[k] = float(np.abs(p.grad.cpu().numpy() - r).max() / max(wscale, 1e-12))
        else:
            errs[k] = scale_err(p.grad.cpu().numpy(), r)
    stats = {}
    msd = model.state_dict()
    for k, v in sd_ref.items():
        if 'running_' in k:
            stats[k] = scale_err(msd[k].cpu().numpy(), v.numpy())
        elif 'num_batches' in k:
            assert int(msd[k]) == int(v), k
    return errs, stats, out, out_lens, ref


def build_jasper(blocks, sd, precision, mid_layers=None):
    from wav2letter_pytorch_amd import Jasper
    from wav2letter_pytorch_amd.config import to_cfg
    labels = O.ENGLISH_LOWERCASE
    cfg = to_cfg(dict(name='jasper', mid_layers=mid_layers or len(blocks), jasper_blocks=blocks, input_size=64, labels=labels,
                      precision=precision, audio_conf=dict(window='hamming', window_stride=0.01, window_size=0.02, sample_rate=16000),
                      decoder=dict(_target_='decoder.GreedyDecoder', labels=labels)))
    model = Jasper(cfg)
    model.load_state_dict({k: v.clone() for k, v in sd.items()})
    return model.cuda()


def compare_jasper_step(model, blocks, sd, x, il, tg, tl, precision):
    """device Jasper step vs the oracle with the device's ReLU gates replayed"""
    out, out_lens, loss, ectx = device_step(model, x, il, tg, tl)
    gates = device_gates(ectx, relu=True)

    def run(gates_):
        params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and 'running_' not in k}
        work = {k: v.clone() for k, v in sd.items()}
        work.update(params)
        inter = []
        lp, ol = O.jasper_forward(x, il, work, blocks, training=True, gates=gates_, inter=inter)
        ls = O.ctc_criterion(lp, tg, ol, tl)
        ls.backward()
        return lp.detach(), ol, ls.detach(), {k: p.grad for k, p in params.items()}, work, [a.detach() for a in inter]

    lp0, _, _, _, _, inter0 = run(None)
    tie = 2e-3 if precision == 'fp32' else 0.25
    for a, g, uc in zip(inter0, gates, ectx['units']):
        dis = (a > 0) != g
        if uc.lens_out is not None:          # frames the next MaskedConv1d zeroes carry no gradient: not comparable
            t = torch.arange(a.shape[2])[None, None, :]
            dis = dis & (t < uc.lens_out.cpu().long()[:, None, None])
        if dis.any():
            assert bool((dis & ~(a.abs() < tie)).sum() == 0), 'ReLU gate differs away from 0'
            assert dis.float().mean() < (2e-3 if precision == 'fp32' else 0.05)
    lp, ol, ls, grads, work, _ = run(gates)
    errs = {'log_probs': scale_err(out.cpu().numpy(), lp.numpy()),
            'loss': abs(float(loss) - float(ls)) / max(1.0, abs(float(ls)))}
    assert torch.equal(out_lens.cpu(), ol.cpu()), (out_lens, ol)
    for k, p in model.named_parameters():
        assert p.grad is not None and p.grad.shape == p.shape, k
        errs[k] = scale_err(p.grad.cpu().numpy(), grads[k].numpy())
    stats = {}
    msd = model.state_dict()
    for k, v in work.items():
        if 'running_' in k:
            stats[k] = scale_err(msd[k].cpu().numpy(), v.numpy())
        elif 'num_batches' in k:
            assert int(msd[k]) == int(v), k
    return errs, stats, out, out_lens
