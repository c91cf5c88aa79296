"""The BASELINE.json configurations at their full size, on the device, against the CPU oracle.

test_gpu_model.py pins the kernels on small stacks and on the reference-generated fixtures; this file runs the
exact architectures the bench lines are quoted on, so the code paths those numbers come from (768/896/1024-wide
layers, K = 25 984, the 256/288-column block shapes, split-K plans, the 21-deep BatchNorm chain, the Jasper 10x5
residual blocks) meet the oracle in the driver-run suite:

* config 2 / 3  Wav2Letter, all 21 rows of configuration/model/wav2letter.yaml:5-104 + classifier (wav2letter.py:69):
  N=2 x T=1000 in the fp32 parity mode and in bf16, with the measured block-shape selection and with the cost model;
  N=32 x T=1000 bf16, dropout on -- the bench workload itself -- with the device's dropout masks and clamp gates
  replayed through the oracle.
* config 4      Jasper 10x5 (defaults.jasper10x5_model; jasper.py:198-255,289-298,439-451): N=2 fp32 parity, N=16 bf16
  properties + loss against the oracle's forward.

Tolerances: fp32 mode -- the north_star bounds (1e-3 of tensor scale for log-probs and gradients, 1e-4 CTC loss).
bf16 mode over 21 layers -- documented looser bounds, stated next to each assert."""
import numpy as np
import pytest
import torch

from gpu_helpers import (build_jasper, build_w2l, compare_step, device_dropout_masks, device_gates, device_step, l2_cos,
                         scale_err)

pytestmark = pytest.mark.gpu


def _w2l_table(dropout):
    from oracle import w2l_oracle as O
    return [l[:4] + ((l[4] if dropout else 0.0),) for l in O.W2L_LAYERS]


def _bf16_grad_bound(key, n):
    """bf16 mode, gradient error (max error over tensor scale) against the fp32 oracle on the 21-layer table, as measured on
    MI355X: 0.08-0.15 at N=32 (classifier 0.01), 0.16-0.27 at N=2 where BatchNorm's statistics rest on 1000 frames --
    the rounding of every backward stage and the clamp gates that fall the other way add up; bounds with ~30 % margin"""
    return 0.2 if n >= 32 else 0.35


# per-tensor gradient bounds at N=32, bf16 vs the fp32 oracle.  Measured on MI355X: L2 0.090-0.116, cosine 0.9933-0.9960, nearly
# the same for all 20 layers -- the deviation enters at the TOP (the bf16 forward's ~3e-2 of the log-prob scale changes
# softmax - posterior, and the last BatchNorm's backward subtracts the large per-channel mean of that gradient, which
# amplifies the difference) and travels down unchanged; the classifier's own gradient is at 0.006.  (Carrying the
# classifier's data gradient in fp32 with a hi + lo split of softmax - posterior was measured: no change -- the source is
# the forward deviation, not a backward rounding.)
# Round 6: the figures move with the dropout REALISATION (Philox seed and offsets, i.e. with whatever ran before in the process):
# L2 0.083-0.108 / cosine 0.9942-0.9966 for the test on its own, 0.113-0.137 / 0.9907-0.9935 behind the rest of the suite.  The test
# now pins seed and offset; the bounds keep room for both populations.
BF16_N32_GRAD_L2, BF16_N32_GRAD_COS = 0.15, 0.9885


def _report(title, errs):
    rows = sorted(((v, k) for k, v in errs.items() if k.endswith('conv1.weight')), key=lambda r: r[1])
    print(f'{title}: log-probs {errs.get("log_probs", float("nan")):.2e} loss {errs.get("loss", float("nan")):.2e}; '
          'weight-gradient errors by layer: '
          + ' '.join(f'{int(k.split("conv1d_")[1].split(".")[0])}:{v:.3f}' for v, k in sorted(rows, key=lambda r: int(r[1].split('conv1d_')[1].split('.')[0]))))


def _worst(errs):
    return max((v, k) for k, v in errs.items() if k not in ('log_probs', 'loss'))


@pytest.mark.parametrize('autotune', [True, False])
def test_w2l_full_table_fp32(autotune, monkeypatch):
    """all 21 conv rows + classifier, N=2 x T=1000, fp32 parity mode vs the oracle step: log-probs / every gradient within
    1e-3 of scale, loss within 1e-4, BatchNorm running statistics; once with measured kernel selection, once with the
    library's cost model (W2L_AUTOTUNE=0)"""
    from oracle import w2l_oracle as O
    from wav2letter_pytorch_amd import engine as E
    monkeypatch.setattr(E, 'AUTOTUNE', autotune)
    layers = _w2l_table(False)
    assert len(layers) == 20 and layers[16][:4] == (896, 29, 1, 2)
    sd = O.init_wav2letter_state(layers, seed=0)
    model = build_w2l(layers, sd, 'fp32').train()
    x, il, tg, tl = O.synthetic_batch(2, 1000, seed=1234)
    errs, stats, out, out_lens, ref = compare_step(model, layers, sd, x, il, tg, tl, 'fp32')
    assert out.shape == (2, 500, 29) and [int(v) for v in out_lens] == [500, 500]
    print(f'w2l full table fp32 (autotune {autotune}): log-probs {errs["log_probs"]:.2e} loss {errs["loss"]:.2e} '
          f'worst gradient {_worst(errs)[0]:.2e} ({_worst(errs)[1]}) running stats {max(stats.values()):.2e}')
    assert errs['log_probs'] < 1e-3, errs['log_probs']
    assert errs['loss'] < 1e-4, errs['loss']
    assert _worst(errs)[0] < 1e-3, _worst(errs)
    assert max(stats.values()) < 1e-3
    assert len(errs) == 2 + 20 * 4 + 2          # every parameter of the 153 M-parameter table was compared


@pytest.mark.parametrize('autotune', [True, False])
def test_w2l_full_table_bf16(autotune, monkeypatch):
    """the production arithmetic (bf16 operands and activations, fp32 accumulate / statistics / CTC) on the same step.
    Bounds over the 21-layer chain at N=2: log-probs 1e-1 of scale, loss 2e-2; gradients: _bf16_grad_bound."""
    from oracle import w2l_oracle as O
    from wav2letter_pytorch_amd import engine as E
    monkeypatch.setattr(E, 'AUTOTUNE', autotune)
    layers = _w2l_table(False)
    sd = O.init_wav2letter_state(layers, seed=0)
    model = build_w2l(layers, sd, 'bf16').train()
    x, il, tg, tl = O.synthetic_batch(2, 1000, seed=1234)
    # bf16 activations drift from the fp32 oracle's by up to 8 significant bits PER LAYER: 20 layers down a clamp gate may
    # differ where the oracle's own activation is as far as 1.0 from a bound, on up to a quarter of the elements (the
    # gate decisions themselves are pinned by the fp32-mode test above; here they are replayed)
    errs, stats, out, out_lens, ref = compare_step(model, layers, sd, x, il, tg, tl, 'bf16', tie=1.0, max_frac=0.25)
    _report('w2l full table bf16 N=2', errs)
    assert errs['log_probs'] < 1e-1, errs['log_probs']          # measured 6.3e-2 at N=2 (3.0e-2 at N=32)
    assert errs['loss'] < 2e-2, errs['loss']                    # measured 5e-4
    for k, v in errs.items():
        if k not in ('log_probs', 'loss'):
            assert v < _bf16_grad_bound(k, 2), (k, v)
    assert max(stats.values()) < 2e-2


def test_w2l_full_table_N32_bench_workload_bf16():
    """BASELINE config 2 exactly as bench.py runs it: 21-layer table, N=32 x T=1000 x 64 mel, bf16, yaml dropout ON.
    (a) properties: finite normalised log-probs, every gradient finite and non-zero, the forward is bit-reproducible
    given the same dropout offsets; (b) parity: the device's recorded dropout masks and clamp gates replayed through the
    fp32 oracle step at the same N=32 -- loss within 2e-2, log-probs within 5e-2 of scale, gradients as in the N=2 test."""
    from oracle import w2l_oracle as O
    from wav2letter_pytorch_amd import engine as E
    layers = _w2l_table(True)
    sd = O.init_wav2letter_state(layers, seed=0)
    model = build_w2l(layers, sd, 'bf16', dropout=True).train()
    x, il, tg, tl = O.synthetic_batch(32, 1000, seed=1234)
    torch.manual_seed(20261005)         # (the masks are a function of torch's seed and the engine's draw counter: pinned, so that the
    E._dropout_calls = 0                # comparison below does not depend on which tests ran before this one)
    start = E._dropout_calls
    out, out_lens, loss, ectx = device_step(model, x, il, tg, tl)
    assert out.shape == (32, 500, 29) and torch.isfinite(out).all()
    assert float((out.exp().sum(-1) - 1).abs().max()) < 1e-4
    assert np.isfinite(float(loss)) and float(loss) > 0
    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    for k, g in grads.items():
        assert torch.isfinite(g).all(), k
        if not (k.endswith('conv1.bias') and not k.startswith('conv1ds.conv1d_20.')):
            assert float(g.abs().max()) > 0, k
    masks = device_dropout_masks(ectx, [l[0] for l in layers])
    for m, l in zip(masks, layers):
        assert abs(float(m.mean()) - (1 - l[4])) < 5e-3          # keep rate of the yaml's p
    gates = device_gates(ectx)
    # same Philox offsets again -> the same masks and a bit-identical forward (running statistics do not enter a
    # training-mode forward)
    E._dropout_calls = start
    with torch.no_grad():
        out2, _ = model(x.cuda(), il)
    assert torch.equal(out2, out)
    del ectx, out2
    torch.cuda.empty_cache()
    ref = O.wav2letter_step(x, il, tg, tl, {k: v.clone() for k, v in sd.items()}, layers, drop_masks=masks, gates=gates)
    assert abs(float(loss) - float(ref['loss'])) < 2e-2 * abs(float(ref['loss'])), (float(loss), float(ref['loss']))
    assert scale_err(out.cpu().numpy(), ref['log_probs'].numpy()) < 5e-2      # 21 bf16 layers + dropout's 1/(1-p) gains
    errs, norms = {}, {}
    for k, g in grads.items():
        if k.endswith('conv1.bias') and not k.startswith('conv1ds.conv1d_20.'):
            continue                                              # identically zero under BatchNorm
        errs[k] = scale_err(g.cpu().numpy(), ref['grads'][k].numpy())
        norms[k] = l2_cos(g.cpu().numpy(), ref['grads'][k].numpy())
    _report('w2l full table bf16 N=32 dropout on', errs)
    print('  weight-gradient L2 error / cosine by layer: '
          + ' '.join(f'{i}:{norms[f"conv1ds.conv1d_{i}.conv1.weight"][0]:.3f}/{norms[f"conv1ds.conv1d_{i}.conv1.weight"][1]:.4f}'
                     for i in range(21)))
    for k, v in errs.items():
        assert v < _bf16_grad_bound(k, 32), (k, v)
    # the max-norm above is set by the single worst element; the training signal as a whole is bounded in the L2 norm and in
    # direction: a regression that doubles one layer's error fails here
    for k, (l2, cos) in norms.items():
        assert l2 <= BF16_N32_GRAD_L2 and cos >= BF16_N32_GRAD_COS, (k, l2, cos)


def test_w2l_full_table_bf16_vs_operand_model():
    """WHERE the bf16 mode's distance from the fp32 oracle comes from.  The 21-layer table at N=8 x T=1000, dropout on, one
    device step; its masks and clamp gates replayed through (a) the fp32 oracle and (b) the oracle's bf16 OPERAND MODEL
    (oracle.bf16_conv1d + conv1d_block_forward(stats_before_rounding): bf16 operands, fp32 accumulation, BatchNorm statistics
    from the accumulators, every stored tensor -- y, activation, dy, dx -- rounded to bf16: the places where the engine
    rounds, and nothing else of it; the order of the fp32 sums is the one thing it does not share with the kernels).
    Asserted:
      * the device's stored activations are BIT-IDENTICAL to the model's for the first layer (>= 99.9 % of the elements; measured
        100.00 %), >= 99 % / 96 % for the next two (99.87-99.88 / 98.8-99.0): the kernels round where the model says and nowhere else.
        From there the two part ways one bf16 ulp at a time -- an element whose accumulator differs in the last fp32 bits rounds
        the other way, and every such flip seeds more in the next layer (95 %, 87 %, ... ~70 % = the zeros);
      * so three evaluations that agree on the ARITHMETIC still differ in the gradients: per tensor, device vs model
        0.045-0.062 in relative L2 (cosine 0.998-0.999), model vs fp32 -- CPU only, no device involved -- 0.072-0.10, device vs
        fp32 0.075-0.138.  The device must be closer to the model than the model is to fp32, and of the model's own distance from
        fp32 (<= 1.5 x + 0.005; the ratio scatters 0.96-1.27 with the dropout realisation): the distance the fp32 comparisons
        of this file see is what storing bf16 tensors costs, the device adds nothing to it."""
    from oracle import w2l_oracle as O
    layers = _w2l_table(True)
    sd = O.init_wav2letter_state(layers, seed=0)
    model = build_w2l(layers, sd, 'bf16', dropout=True).train()
    x, il, tg, tl = O.synthetic_batch(8, 1000, seed=77)
    from wav2letter_pytorch_amd import engine as E
    E._dropout_calls = 7000                      # the same Philox offsets, i.e. the same masks, whatever ran before this test
    out, out_lens, loss, ectx = device_step(model, x, il, tg, tl)
    masks = device_dropout_masks(ectx, [l[0] for l in layers])
    gates = device_gates(ectx)
    acts = []
    for i in range(4):
        a = ectx['acts'][i + 1]
        acts.append(a.hi[:, a.pad_l:a.pad_l + a.T, :a.C].float().transpose(1, 2).cpu())
    del ectx
    got = {k: p.grad.detach().cpu().numpy() for k, p in model.named_parameters()}
    ref = {name: O.wav2letter_step(x, il, tg, tl, {k: v.clone() for k, v in sd.items()}, layers, drop_masks=masks, gates=gates,
                                   bf16_model=flag) for name, flag in (('fp32', False), ('bf16_model', True))}
    same = [float((a == r.to(torch.bfloat16).float()).float().mean()) for a, r in zip(acts, ref['bf16_model']['activations'])]
    print('bf16 full table N=8: stored activations bit-identical to the operand model, layers 0-3: '
          + ' '.join(f'{v:.4f}' for v in same))
    assert same[0] >= 0.999 and same[1] >= 0.99 and same[2] >= 0.96, same
    keys = [k for k in got if not (k.endswith('conv1.bias') and not k.startswith('conv1ds.conv1d_20.'))]
    vs = {name: {k: l2_cos(got[k], r['grads'][k].numpy()) for k in keys} for name, r in ref.items()}
    arith = {k: l2_cos(ref['bf16_model']['grads'][k].numpy(), ref['fp32']['grads'][k].numpy()) for k in keys}

    def row(d):
        return ' '.join(f'{i}:{d[f"conv1ds.conv1d_{i}.conv1.weight"][0]:.4f}/{d[f"conv1ds.conv1d_{i}.conv1.weight"][1]:.5f}'
                        for i in range(21))
    for title, d in (('device vs the bf16 operand model', vs['bf16_model']), ('device vs fp32', vs['fp32']),
                     ('the bf16 operand model vs fp32 (CPU only)', arith)):
        print(f'bf16 full table N=8, {title}: weight-gradient L2 / cosine by layer: {row(d)}')
        print('   worst over all tensors: L2 %.4f  cosine %.5f' % (max(v[0] for v in d.values()), min(v[1] for v in d.values())))
    for name, r, lp_bound in (('fp32', ref['fp32'], 5e-2), ('bf16_model', ref['bf16_model'], 3e-2)):
        e_loss = abs(float(loss) - float(r['loss'])) / abs(float(r['loss']))
        e_lp = scale_err(out.cpu().numpy(), r['log_probs'].numpy())
        print(f'  vs {name}: loss {e_loss:.2e} log-probs {e_lp:.2e}')
        assert e_loss < 1e-3 and e_lp < lp_bound, (name, e_loss, e_lp)
    for k in keys:
        (l2m, cosm), (l2f, cosf), (l2a, cosa) = vs['bf16_model'][k], vs['fp32'][k], arith[k]
        # three noisy evaluations of one function: the ratios scatter with the dropout realisation (device-vs-model / (c)
        # 0.57-0.86, device-vs-fp32 / (c) 0.96-1.27 over the realisations measured)
        assert l2m <= l2a + 1e-3 and cosm >= cosa - 1e-4, (k, l2m, l2a, cosm, cosa)
        assert l2f <= 1.5 * l2a + 5e-3, (k, l2f, l2a)


def _jasper10x5():
    from wav2letter_pytorch_amd import Jasper
    from wav2letter_pytorch_amd.defaults import jasper10x5_model
    cfg = jasper10x5_model()
    blocks = [dict(b) for b in cfg.jasper_blocks]
    torch.manual_seed(7)
    sd = {k: v.detach().clone() for k, v in Jasper(cfg).state_dict().items()}
    return blocks, sd


def _oracle_jasper_blocks(x, il, tg, tl, sd, blocks):
    """the oracle's Jasper step kept block by block: inputs, lengths, outputs and -- from ONE backward through the whole
    network -- the gradient of the loss wrt every block output, every parameter and the spectrogram"""
    from oracle import w2l_oracle as O
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and 'running' not in k}
    work = {k: v.clone() for k, v in sd.items()}
    work.update(params)
    xin = x.clone().requires_grad_(True)
    cur, lens = xin, il
    rec = []
    for i, blk in enumerate(blocks):
        out, lens_out = O.jasper_block_forward(cur, lens, work, f'jasper_encoder.{i}.', blk, training=True)
        out.retain_grad()
        rec.append(dict(x=cur, lens=lens, out=out, lens_out=lens_out))
        cur, lens = out, lens_out
    y = torch.nn.functional.conv1d(cur, work['final_layer.0.weight'], work['final_layer.0.bias']).transpose(2, 1)
    lp = torch.log_softmax(y, dim=-1)
    ol = lens.to(torch.int64)
    loss = O.ctc_criterion(lp, tg, ol, tl)
    loss.backward()
    return rec, lp.detach(), ol, loss.detach(), {k: p.grad for k, p in params.items()}, xin.grad


def test_jasper10x5_blockwise_fp32():
    """Jasper 10x5 (13 blocks, 54 convs, 322 M parameters; jasper.py:198-255,289-298,439-451), N=2 x T=1000, one ragged
    utterance.  A randomly initialised 53-unit stack with batch statistics amplifies ANY perturbation by ~1.15x per unit
    (tools/debug_jasper10x5.py: the fp32 mode's 6e-6 after unit 0 is 1.4e-2 after unit 52 -- two correct fp32 evaluations
    drift apart the same way), so the 1e-3 kernel parity is taken block by block, teacher-forced: every block runs on the
    device from the ORACLE's block input and is compared with the oracle's block output (1e-3 of scale), and its backward,
    driven by the oracle's upstream gradient, with the oracle's input / parameter gradients (2e-3 in the L2 norm: a ReLU
    gate decided within rounding of 0 moves single elements).  All channel widths (256-1024), kernel sizes (11-29, 1),
    the stride-2 prologue, the dilated block and the residual 1x1 convs at their real sizes."""
    from oracle import w2l_oracle as O
    blocks, sd = _jasper10x5()
    assert len(blocks) == 13 and sum(v.numel() for k, v in sd.items() if v.dtype.is_floating_point and 'running' not in k) > 3.2e8
    model = build_jasper(blocks, sd, 'fp32').train()
    x, il, tg, tl = O.synthetic_batch(2, 1000, seed=99)
    il[1] = 801
    x[1, :, 801:] = 0
    rec, lp, ol, loss, pgrads, xgrad = _oracle_jasper_blocks(x, il, tg, tl, sd, blocks)
    assert [int(v) for v in ol] == [500, 401]                                   # 801 -> 401
    # gradients are judged on the L2 norm: the oracle's own ReLU gates are used (not replayed), and a fraction f of gates
    # decided differently within rounding of 0 moves the norm by ~sqrt(f) -- f = 1e-5 (activations agree to 1e-5) is 3e-3
    GTOL, worst_in, worst_p = 2e-2, 0.0, 0.0
    for i, (blk, r) in enumerate(zip(model.jasper_encoder, rec)):
        blk.precision = 'fp32'
        xd = r['x'].detach().cuda().requires_grad_(True)
        out, lens_out = blk((xd, r['lens']))
        ref = r['out'].detach()
        assert torch.equal(lens_out.cpu().float(), r['lens_out'].float()), i
        assert scale_err(out.detach().cpu().numpy(), ref.numpy()) < 1e-3, i
        out.backward(r['out'].grad.cuda())
        gin = xgrad if i == 0 else rec[i - 1]['out'].grad
        rel = float(np.linalg.norm(xd.grad.cpu().numpy() - gin.numpy()) / np.linalg.norm(gin.numpy()))
        worst_in = max(worst_in, rel)
        assert rel < GTOL, (i, rel)
        for k, p in blk.named_parameters():
            g = pgrads[f'jasper_encoder.{i}.{k}'].numpy()
            rel = float(np.linalg.norm(p.grad.cpu().numpy() - g) / max(np.linalg.norm(g), 1e-20))
            worst_p = max(worst_p, rel)
            assert rel < GTOL, (i, k, rel)
        blk.zero_grad(set_to_none=True)
        blk.__dict__.pop('_solo_engine', None)
    print(f'jasper10x5 blockwise fp32: worst L2 error of an input gradient {worst_in:.2e}, of a parameter gradient {worst_p:.2e}')


def _bf16r(t):
    return t.to(torch.bfloat16).float()


def _operand_model_conv(mode):
    """the dense-convolution stand-in of the oracle's Jasper blocks for ``precision: bf16`` / ``fp8``: which convolutions
    take e4m3 operands follows engine._conv_forward (stride 1, 128 | C_in; ReLU activations carry scale 8)"""
    from oracle import w2l_oracle as O

    def conv(x, w, b, stride=1, padding=0, dilation=1):
        if mode == 'fp8' and stride == 1 and w.shape[1] % 128 == 0:
            return O.fp8_conv1d(x, w, b, stride=stride, padding=padding, dilation=dilation, act_scale=8.0, round_dx=True)
        return O.bf16_conv1d(x, w, b, stride=stride, padding=padding, dilation=dilation, round_out=False)

    return conv


@pytest.mark.parametrize('mode', ['bf16', 'fp8'])
def test_jasper10x5_blockwise_operand_model(mode, monkeypatch):
    """Jasper 10x5 (BASELINE configs 4 / 5) in its PRODUCTION arithmetic, block by block, teacher-forced, against a model of
    that arithmetic -- the oracle's blocks with every dense convolution replaced by the bf16 / e4m3 operand model
    (oracle.bf16_conv1d / fp8_conv1d: operands rounded where the engine rounds them, fp32 accumulation, BatchNorm statistics
    from the accumulators, y / dy / dx stored as bf16 tensors).  Each of the 13 blocks runs on the device as a stand-alone
    JasperBlock in ``precision: <mode>`` from the fp32 oracle's block input (rounded to bf16: what the previous block's
    BatchNorm kernel would have stored) and backward from the fp32 oracle's upstream gradient; the model does the same on
    the CPU with the device's ReLU gates replayed.  fp8: forward, data and weight gradients of every qualifying convolution
    on e4m3 operands (the first convolutions of a block included: an open fp8 engine quantises its input like the network
    does).  Bounds: block output within 1e-2 (bf16) / 6e-2 (fp8: five chained quantisers) of scale, input and parameter
    gradients within 0.03 / 0.10 in the L2 norm and cosine >= 0.999 / 0.995; in fp8 mode every unit of a five-repeat block is
    held to the model on its own as well (1e-2 / 0.03 / 0.999)."""
    from oracle import w2l_oracle as O
    from wav2letter_pytorch_amd import engine as E
    blocks, sd = _jasper10x5()
    model = build_jasper(blocks, sd, mode).train()
    x, il, tg, tl = O.synthetic_batch(2, 1000, seed=99)
    il[1] = 801
    x[1, :, 801:] = 0
    rec, lp, ol, loss, pgrads, xgrad = _oracle_jasper_blocks(x, il, tg, tl, sd, blocks)
    monkeypatch.setattr(E, 'FP8_DGRAD', '1')
    monkeypatch.setattr(E, 'FP8_WGRAD', '1')
    conv = _operand_model_conv(mode)
    # fp8, a block of FIVE chained e4m3 units: measured 0.032-0.041 / 0.059 / 0.9983 (one-unit blocks: 0.003-0.005 / 0.005 /
    # 1.0000); the per-unit bound below is the tight one (1e-2 / 0.03 / 0.999)
    out_tol, l2_tol, cos_tol = (1e-2, 0.03, 0.999) if mode == 'bf16' else (6e-2, 0.10, 0.995)
    worst = dict(out=0.0, l2=0.0, cos=1.0)
    report, failed, unit_rows = [], [], []
    for i, (blk, r) in enumerate(zip(model.jasper_encoder, rec)):
        blk.precision = mode
        blk._debug_keep_ctx = True
        xin = _bf16r(r['x'].detach())
        gout = r['out'].grad
        xd = xin.cuda().requires_grad_(True)
        out, lens_out = blk((xd, r['lens']))
        out.backward(gout.cuda())
        ectx = blk._last_ctx
        gates = device_gates(ectx, relu=True)
        if mode == 'fp8':          # every stride-1 convolution with 128 | C_in of the block ran on e4m3 operands
            want_q = sum(1 for u in ectx['units'] for c in (u.unit.main, u.unit.res)
                         if c is not None and c.stride == 1 and c.cin % 128 == 0)
            have_q = sum(1 for u in ectx['units'] for c, a in ((u.unit.main, ectx['acts'][u.unit.src]),
                                                               (u.unit.res, ectx['acts'][u.unit.res_src] if u.unit.res else None))
                         if c is not None and a is not None and a.q is not None and c.stride == 1)
            assert want_q == have_q and (want_q > 0 or i == 0), (i, want_q, have_q)
        del ectx
        blk._last_ctx = None
        prefix = f'jasper_encoder.{i}.'
        params = {k: v.clone().requires_grad_(True) for k, v in sd.items()
                  if k.startswith(prefix) and v.dtype.is_floating_point and 'running' not in k}
        work = {k: v.clone() for k, v in sd.items() if k.startswith(prefix)}
        work.update(params)
        xm = xin.clone().requires_grad_(True)
        inter = []
        mo, mlens = O.jasper_block_forward(xm, r['lens'], work, prefix, blocks[i], training=True, gates=list(gates), conv=conv,
                                           stats_before_rounding=True, inter=inter)
        for a in inter:
            a.retain_grad()
        mo.backward(gout)
        assert torch.equal(lens_out.cpu().float(), mlens.float()), i
        e_out = scale_err(out.detach().cpu().numpy(), mo.detach().numpy())
        l2, cos = l2_cos(xd.grad.cpu().numpy(), xm.grad.numpy())
        worst['out'] = max(worst['out'], e_out)
        report.append((i, e_out, l2, cos))
        failed += [(i, 'out', e_out)] if e_out >= out_tol else []
        rows = [('x', l2, cos)]
        for k, p in blk.named_parameters():
            rows.append((k, *l2_cos(p.grad.cpu().numpy(), params[prefix + k].grad.numpy())))
        for name, l2, cos in rows:
            worst['l2'], worst['cos'] = max(worst['l2'], l2), min(worst['cos'], cos)
            failed += [(i, name, l2, cos)] if not (l2 < l2_tol and cos > cos_tol) else []
        blk.zero_grad(set_to_none=True)
        if mode == 'fp8' and len(inter) > 1:
            # five chained e4m3 units drift apart like any chain of quantisers (test_gpu_fp8.py): hold every UNIT to the model
            # as well -- units 0 .. repeat-2 of the block (the last one carries the residual branch), each as a one-unit
            # engine from the model's own unit input (rounded to bf16) and the model's gradient wrt the unit's output
            from wav2letter_pytorch_amd.engine import StackEngine, UnitSpec
            from wav2letter_pytorch_amd.layers import run_stack
            us = blk.units(0, 1, 'blk', mask_last_output=False)
            for ur in range(len(inter) - 1):
                uin = xin if ur == 0 else _bf16r(inter[ur - 1].detach())
                gu = inter[ur].grad
                u = us[ur]
                eng = StackEngine([UnitSpec(main=u.main, src=0, act=u.act, update_lens=u.update_lens, mask_out=False)], None, 0,
                                  fp8=True)
                ud = uin.cuda().requires_grad_(True)
                uo, _, uctx = run_stack(eng, ud, r['lens'], True, keep_ctx=True)
                uo.backward(gu.cuda())
                ugate = device_gates(uctx, relu=True)
                assert uctx['acts'][0].q is not None
                del uctx
                mini = {'u.mconv.0.conv.weight': sd[f'{prefix}mconv.{4 * ur}.conv.weight'].clone().requires_grad_(True)}
                for kk in ('weight', 'bias', 'running_mean', 'running_var'):
                    mini[f'u.mconv.1.{kk}'] = sd[f'{prefix}mconv.{4 * ur + 1}.{kk}'].clone()
                mini['u.mconv.1.weight'].requires_grad_(True)
                mini['u.mconv.1.bias'].requires_grad_(True)
                one = dict(kernel_size=mini['u.mconv.0.conv.weight'].shape[2], stride=1, dilation=blocks[i].get('dilation', 1),
                           repeat=1, residual=False, separable=False)
                um = uin.clone().requires_grad_(True)
                umo, _ = O.jasper_block_forward(um, r['lens'], mini, 'u.', one, training=True, gates=list(ugate), conv=conv,
                                                stats_before_rounding=True)
                umo.backward(gu)
                e_u = scale_err(uo.detach().cpu().numpy(), umo.detach().numpy())
                gx = l2_cos(ud.grad.cpu().numpy(), um.grad.numpy())
                gw = l2_cos(u.main.weight.grad.cpu().numpy(), mini['u.mconv.0.conv.weight'].grad.numpy())
                unit_rows.append((i, ur, e_u, gx, gw))
                failed += [(i, ur, 'unit', e_u, gx, gw)] if not (e_u < 1e-2 and gx[0] < 0.03 and gx[1] > 0.999
                                                                 and gw[0] < 0.03 and gw[1] > 0.999) else []
                blk.zero_grad(set_to_none=True)
        blk.__dict__.pop('_solo_engine', None)
    print(f'jasper10x5 blockwise {mode} vs its operand model: worst block output {worst["out"]:.2e} of scale, worst gradient '
          f'L2 {worst["l2"]:.3f}, lowest cosine {worst["cos"]:.5f}; per block (output | dx L2/cos): '
          + ' '.join(f'{i}:{e:.3f}|{l2:.3f}/{c:.4f}' for i, e, l2, c in report))
    if unit_rows:
        print('   unit by unit (one-unit engines, teacher-forced from the model): worst output %.2e, worst dx L2 %.4f / cosine %.5f, '
              'worst dw L2 %.4f / cosine %.5f over %d units' % (max(r_[2] for r_ in unit_rows), max(r_[3][0] for r_ in unit_rows),
                                                               min(r_[3][1] for r_ in unit_rows), max(r_[4][0] for r_ in unit_rows),
                                                               min(r_[4][1] for r_ in unit_rows), len(unit_rows)))
    assert not failed, failed


def test_jasper10x5_whole_network_fp32():
    """the same step through Jasper.forward -> CTC -> backward as ONE engine (tuned kernels, split-K plans, residual
    fan-out): lengths bit-equal, loss within 1e-3, log-probs within the network's own sensitivity -- the oracle's response
    to a 1e-5 perturbation of the spectrogram, which this stack amplifies ~1000x -- and every parameter gradient close to
    the oracle's in direction (cosine > 0.98 in the top blocks, > 0.9 throughout)."""
    from oracle import w2l_oracle as O
    blocks, sd = _jasper10x5()
    model = build_jasper(blocks, sd, 'fp32').train()
    x, il, tg, tl = O.synthetic_batch(2, 1000, seed=99)
    il[1] = 801
    x[1, :, 801:] = 0
    out, out_lens, loss, ectx = device_step(model, x, il, tg, tl)
    del ectx
    rec, lp, ol, loss_ref, pgrads, _ = _oracle_jasper_blocks(x, il, tg, tl, sd, blocks)
    assert out.shape == (2, 500, 29) and torch.equal(out_lens.cpu(), ol)
    assert abs(float(loss) - float(loss_ref)) < 1e-3 * abs(float(loss_ref))
    with torch.no_grad():
        g = torch.Generator().manual_seed(5)
        x2 = x + 1e-5 * torch.randn(x.shape, generator=g) * (x != 0)
        lp2, _ = O.jasper_forward(x2, il, {k: v.clone() for k, v in sd.items()}, blocks, training=True)
    sens = scale_err(lp2.numpy(), lp.numpy())
    err = scale_err(out.cpu().numpy(), lp.numpy())
    print(f'jasper10x5 fp32: log-prob err {err:.2e}, sensitivity to 1e-5 input noise {sens:.2e}')
    assert err < max(1e-3, 5 * sens), (err, sens)
    by_block = {}
    for k, p in model.named_parameters():
        gd, gr = p.grad.cpu().numpy().ravel().astype(np.float64), pgrads[k].numpy().ravel().astype(np.float64)
        assert np.isfinite(gd).all(), k
        cos = float(np.dot(gd, gr) / max(np.linalg.norm(gd) * np.linalg.norm(gr), 1e-30))
        b = 13 if k.startswith('final_layer') else int(k.split('.')[1])
        by_block[b] = min(by_block.get(b, 1.0), cos)
    print('jasper10x5 fp32: worst gradient cosine per block ' + ' '.join(f'{b}:{c:.3f}' for b, c in sorted(by_block.items())))
    for b, c in by_block.items():                   # the perturbation grows on the way up AND on the way back down
        assert c > (0.98 if b >= 11 else 0.9), (b, c)          # measured 0.96-0.97 in blocks 0-10, 0.99+ above


def test_jasper10x5_N16_bench_workload_bf16():
    """BASELINE config 4 as `bench.py --model jasper10x5` runs it (N=16 x T=1000, bf16): properties of the step; the loss
    against the oracle's forward at the same N=16 (2e-2); the first units' activations against the oracle's (5e-2 of scale
    -- further down the bf16 rounding is amplified like any other perturbation, see test_jasper10x5_blockwise_fp32)"""
    from oracle import w2l_oracle as O
    blocks, sd = _jasper10x5()
    model = build_jasper(blocks, sd, 'bf16').train()
    x, il, tg, tl = O.synthetic_batch(16, 1000, seed=1234)
    il[3] = 777
    x[3, :, 777:] = 0
    out, out_lens, loss, ectx = device_step(model, x, il, tg, tl)
    assert out.shape == (16, 500, 29) and torch.isfinite(out).all()
    assert float((out.exp().sum(-1) - 1).abs().max()) < 1e-4
    assert int(out_lens[3]) == 389 and int(out_lens[0]) == 500
    for k, p in model.named_parameters():
        assert torch.isfinite(p.grad).all(), k
        assert float(p.grad.abs().max()) > 0, k
    inter = []
    with torch.no_grad():
        lp, ol = O.jasper_forward(x, il, {k: v.clone() for k, v in sd.items()}, blocks, training=True, inter=inter)
        ls = O.ctc_criterion(lp, tg, ol, tl)
    assert torch.equal(out_lens.cpu(), ol.cpu())
    assert abs(float(loss) - float(ls)) < 2e-2 * abs(float(ls))
    for i in range(6):
        act, uc = ectx['acts'][i + 1], ectx['units'][i]
        a = act.hi[:, act.pad_l:act.pad_l + act.T, :act.C].float().transpose(1, 2).cpu()
        ref = inter[i]
        if uc.lens_out is not None:
            ref = ref * (torch.arange(ref.shape[2])[None, None, :] < uc.lens_out.cpu().long()[:, None, None])
        assert scale_err(a.numpy(), ref.numpy()) < 5e-2, i
