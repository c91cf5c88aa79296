"""End-to-end parity of the HIP step engine on a real MI355X.

Method: the same seeded parameters and batch go through (a) the public module surface
(Wav2Letter.forward -> CTCLoss -> loss.backward()) on the device and (b) the CPU oracle
(oracle/w2l_oracle.py, pinned to the reference by tests/golden).  clamp(0,20)'s gradient is
discontinuous, so the oracle replays the device's gate decisions and the test separately
asserts that every decision that differs from the oracle's own is a genuine tie (activation
within 2e-3 of a bound) -- see oracle conv1d_block_forward(gate=...).

Tolerances (north_star): fp32 mode (split-bf16 MFMA, fp32 accumulate): log-probs and every
gradient 1e-3 of the tensor's scale, CTC loss 1e-4 relative, greedy indices bit-exact on
identical probabilities.  bf16 production mode vs the same fp32 oracle: 3e-2 (log-probs),
2e-2 (loss), 8e-2 (gradients) -- bf16 keeps 8 significant bits."""
import ast
import os

import numpy as np
import pytest
import torch

from gpu_helpers import build_w2l, compare_step, scale_err

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), 'golden')
TOL = {'fp32': dict(lp=1e-3, loss=1e-4, grad=1e-3, stat=1e-3), 'bf16': dict(lp=3e-2, loss=2e-2, grad=8e-2, stat=2e-2)}


def load(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=True)


def check_fixture_grads(model, z, tol=1e-2):
    """device gradients against the reference's own (fixture keys g/*), directly: no oracle in between, no gate replay --
    hence the L2 norm (a ReLU / clamp gate that falls the other way within rounding moves single elements)"""
    worst = 0.0
    for k, p in model.named_parameters():
        ref_g, got_g = z['g/' + k], p.grad.cpu().numpy()
        n = float(np.linalg.norm(ref_g))
        if n < 1e-6 * max(float(np.linalg.norm(z['p0/' + k])), 1e-30) or n < 1e-7:
            continue                    # identically-zero gradients (conv bias under BatchNorm): covered by compare_step
        rel = float(np.linalg.norm(got_g - ref_g)) / n
        worst = max(worst, rel)
        assert rel < tol, (k, rel)
    return worst


def check(errs, stats, precision, grad_tol=None):
    t = dict(TOL[precision])
    if grad_tol is not None:
        t['grad'] = grad_tol
    assert errs['log_probs'] < t['lp'], errs['log_probs']
    assert errs['loss'] < t['loss'], errs['loss']
    worst = max((v, k) for k, v in errs.items() if k not in ('log_probs', 'loss'))
    assert worst[0] < t['grad'], worst
    if stats:
        ws = max((v, k) for k, v in stats.items())
        assert ws[0] < t['stat'], ws


# model-level agreement of the greedy indices with the reference's argmax matrix in fp32 (split-bf16) mode, per fixture: the
# floor is what was MEASURED on MI355X (round 6), 1.0 = every frame identical; below 1.0 only where two labels of a frame are
# within the 1e-3 logit tolerance of each other and the device's rounding falls the other way
ARGMAX_AGREEMENT_FLOOR = {'w2l_ml1': 1.0, 'w2l_ml3': 1.0, 'w2l_mix5': 1.0}


@pytest.mark.parametrize('case', ['w2l_ml1', 'w2l_ml3', 'w2l_mix5'])
@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_w2l_golden(case, precision):
    """inputs and parameters of the reference-generated fixtures; forward results are compared with
    the fixture itself (reference outputs), gradients with the gate-replayed oracle"""
    from oracle import w2l_oracle as O
    z = load(case + '.npz')
    meta = ast.literal_eval(str(z['meta']))
    layers = [(l['output_size'], l['kernel_size'], l['stride'], l['dilation'], 0.0) for l in meta['layers']]
    sd = {k[3:]: torch.from_numpy(z[k].copy()) for k in z.files if k.startswith('p0/')}
    model = build_w2l(layers, sd, precision).train()
    x = torch.from_numpy(z['x'])
    il, tg, tl = (torch.from_numpy(z[k]) for k in ('in_lens', 'targets', 'target_lens'))
    errs, stats, out, out_lens, ref = compare_step(model, layers, sd, x, il, tg, tl, precision)
    check(errs, stats, precision)
    t = TOL[precision]
    np.testing.assert_array_equal(out_lens.numpy(), z['out_lens'])
    assert scale_err(out.cpu().numpy(), z['log_probs']) < t['lp']
    assert abs(float(ref['loss']) - float(z['loss'])) < 1e-4 * max(1.0, abs(float(z['loss'])))
    sdm = model.state_dict()
    for k in z.files:
        if k.startswith('p1/') and 'running' in k:
            assert scale_err(sdm[k[3:]].cpu().numpy(), z[k]) < t['stat'], k
    if precision == 'fp32':
        # the reference's own gradients (fixture keys g/*), compared with the device directly -- no oracle in between, no
        # gate replay, hence the L2 norm: a clamp gate that falls the other way within rounding moves single elements
        head = f'conv1ds.conv1d_{len(layers)}.'
        wscale = {k: float(np.abs(z['g/' + k]).max()) for k, _ in model.named_parameters()}
        for k, p in model.named_parameters():
            ref_g, got_g = z['g/' + k], p.grad.cpu().numpy()
            if k.endswith('conv1.bias') and not k.startswith(head):       # ~0 under BatchNorm (reference: rounding noise)
                assert np.abs(got_g - ref_g).max() < 1e-3 * wscale[k.replace('bias', 'weight')], k
            else:
                assert np.linalg.norm(got_g - ref_g) < 1e-2 * np.linalg.norm(ref_g), (k, np.linalg.norm(got_g - ref_g) / np.linalg.norm(ref_g))
        texts = [str(s) for s in z['texts']]
        m = model.add_string_metrics(out, out_lens, texts, 'train')
        from wav2letter_pytorch_amd.decoder import argmax_indices
        idx = argmax_indices(out).cpu().numpy()
        agree = (idx == z['argmax']).mean()
        valid = np.arange(idx.shape[1])[None, :] < out_lens.numpy()[:, None]
        agree_valid = (idx == z['argmax'])[valid].mean()
        # the measured agreement with the reference's argmax matrix, reported (pytest -s, and gpurun_out/argmax_agreement.txt
        # when that directory exists), not only bounded: north_star asks for bit-exact greedy indices
        line = (f'[argmax agreement] {case} fp32 mode: {agree * 100:.4f} % of all {idx.size} frames, {agree_valid * 100:.4f} % of '
                f'the {int(valid.sum())} valid frames ({int((idx != z["argmax"]).sum())} frames differ)')
        print(line)
        rec = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
        if os.path.isdir(rec):
            with open(os.path.join(rec, 'argmax_agreement.txt'), 'a') as f:
                f.write(line + '\n')
        assert agree >= ARGMAX_AGREEMENT_FLOOR[case], line
        # decoded strings: every utterance whose argmax path equals the reference's must decode to the reference's string,
        # and at least one such utterance must exist (a near-tie between two labels may flip single frames elsewhere)
        decoded = model.ctc_decoder.decode(out, out_lens)
        same = [bool((idx[n, :int(out_lens[n])] == z['argmax'][n, :int(out_lens[n])]).all()) for n in range(len(texts))]
        assert any(same)
        for n, ok in enumerate(same):
            if ok:
                assert decoded[n] == str(z['decoded'][n]), n
        if all(same):
            assert abs(m['train_cer'] - float(z['cer'])) < 1e-12 and abs(m['train_wer'] - float(z['wer'])) < 1e-12
        # the metric arithmetic itself, unconditionally: the reference's decoded strings through this package's CER / WER
        dec = model.ctc_decoder
        ce, cr = map(sum, zip(*(dec.cer_ratio(t, str(h)) for t, h in zip(texts, z['decoded']))))
        we, wr = map(sum, zip(*(dec.wer_ratio(t, str(h)) for t, h in zip(texts, z['decoded']))))
        assert abs(ce / cr - float(z['cer'])) < 1e-12 and abs(we / wr - float(z['wer'])) < 1e-12
        model.eval()                  # eval mode: running statistics
        with torch.no_grad():
            oe, _ = model(x.cuda(), il)
        assert scale_err(oe.cpu().numpy(), z['out_eval']) < 2e-3


def test_greedy_decoder_golden_bit_exact():
    from wav2letter_pytorch_amd.decoder import GreedyDecoder
    z = load('greedy_cases.npz')
    dec = GreedyDecoder(list(z['labels']), blank_index=0)
    probs = torch.from_numpy(z['probs']).cuda()
    strings, offsets = dec.decode(probs, torch.from_numpy(z['sizes']), return_offsets=True)
    assert strings == list(z['strings'])
    for o, ref in zip(offsets, z['offsets']):
        assert o[0].tolist() == list(ref)
    small = GreedyDecoder(['_', 'A', 'B', ' '], blank_index=0).decode(       # unit_tests/decoder_test.py:40-42
        torch.FloatTensor([[0.8, 0.2, 0, 0], [0.6, 0.4, 0, 0]]).unsqueeze(0), sizes=None)
    assert small == ['']
    for (a, b), c, w in zip(z['pairs'], z['cer'], z['wer']):
        assert dec.cer_ratio(str(a), str(b)) == tuple(c)
        assert dec.wer_ratio(str(a), str(b)) == tuple(w)


def _synthetic_case(layers, N, T, precision, seed, ragged=False, drop=False):
    from oracle import w2l_oracle as O
    sd = O.init_wav2letter_state(layers, seed=seed)
    model = build_w2l(layers, sd, precision, dropout=drop).train()
    x, il, tg, tl = O.synthetic_batch(N, T, seed=seed + 1, s_lo=max(2, T // 12), s_hi=max(3, T // 6), ragged=ragged)
    errs, stats, *_ = compare_step(model, layers, sd, x, il, tg, tl, precision, drop=drop)
    check(errs, stats, precision)


def test_w2l_real_widths_fp32():
    """true channel widths of the yaml table (256/384): rows 0, 1, 4 + classifier, ragged lengths"""
    from oracle import w2l_oracle as O
    _synthetic_case([O.W2L_LAYERS[i][:4] + (0.0,) for i in (0, 1, 4)], N=2, T=300, precision='fp32', seed=5, ragged=True)


def test_w2l_dilated_wide_fp32():
    from oracle import w2l_oracle as O
    layers = [O.W2L_LAYERS[0][:4] + (0.0,), (384, 13, 1, 1, 0.0), (512, 29, 1, 2, 0.0), (640, 1, 1, 1, 0.0)]
    _synthetic_case(layers, N=2, T=260, precision='fp32', seed=6)


def test_w2l_dilated_wide_bf16():
    from oracle import w2l_oracle as O
    layers = [O.W2L_LAYERS[0][:4] + (0.0,), (384, 13, 1, 1, 0.0), (512, 29, 1, 2, 0.0), (640, 1, 1, 1, 0.0)]
    _synthetic_case(layers, N=2, T=260, precision='bf16', seed=6)


def test_dropout_replay_fp32():
    """p > 0: keep-rate ~ 1-p; the device's recorded mask, replayed through the oracle, reproduces
    forward and backward (dropout sits BEFORE the clamp, wav2letter.py:44-46)"""
    layers = [(128, 11, 2, 1, 0.3), (128, 11, 1, 1, 0.25)]
    from oracle import w2l_oracle as O
    from gpu_helpers import device_dropout_masks, device_step
    sd = O.init_wav2letter_state(layers, seed=9)
    model = build_w2l(layers, sd, 'fp32', dropout=True).train()
    x, il, tg, tl = O.synthetic_batch(3, 200, seed=10, s_lo=10, s_hi=30)
    _, _, _, ectx = device_step(model, x, il, tg, tl)
    for m, (_, _, _, _, p) in zip(device_dropout_masks(ectx, [128, 128]), layers):
        assert abs(float(m.mean()) - (1 - p)) < 0.01
    model2 = build_w2l(layers, sd, 'fp32', dropout=True).train()
    errs, stats, *_ = compare_step(model2, layers, sd, x, il, tg, tl, 'fp32', drop=True)
    check(errs, stats, 'fp32')


def test_headline_shapes_properties_bf16():
    """BASELINE config-2 shapes (N=32, T=1000) on a truncated stack (rows 0-1 + the dilated row 16 +
    classifier; the full 21-layer table is exercised by bench.py): size-independent properties --
    finite outputs, log-probs normalised, loss > 0, gradient of every parameter finite and non-zero,
    batch-order equivariance of the log-probs under per-utterance permutation is NOT expected (BatchNorm
    couples utterances) so instead: repeating the step on identical inputs reproduces the forward bit for bit."""
    from oracle import w2l_oracle as O
    layers = [O.W2L_LAYERS[0][:4] + (0.0,), O.W2L_LAYERS[1][:4] + (0.0,), (256, 29, 1, 2, 0.0)]
    sd = O.init_wav2letter_state(layers, seed=3)
    model = build_w2l(layers, sd, 'bf16').train()
    x, il, tg, tl = O.synthetic_batch(32, 1000, seed=1234)
    outs = []
    for _ in range(2):
        model.zero_grad(set_to_none=True)
        out, ol = model(x.cuda(), il)
        loss = model.criterion(out.transpose(0, 1), tg, ol, tl)
        loss.backward()
        outs.append(out.detach().clone())
    torch.cuda.synchronize()
    assert out.shape == (32, 500, 29) and torch.isfinite(out).all()
    assert (out.exp().sum(-1) - 1).abs().max() < 1e-4
    assert float(loss) > 0 and np.isfinite(float(loss))
    assert torch.equal(outs[0], outs[1])
    for k, p in model.named_parameters():
        assert torch.isfinite(p.grad).all(), k
        if not (k.endswith('conv1.bias') and not k.startswith('conv1ds.conv1d_3.')):
            assert float(p.grad.abs().max()) > 0, k


@pytest.mark.parametrize('overlap,recycle', [(False, False), (True, False), (False, True), (True, True)])
def test_fused_sgd_matches_torch_sgd(overlap, recycle):
    """FusedSGD (w2l_sgd_pack for conv weights) == torch.optim.SGD(nesterov, momentum, weight decay) over 3 steps,
    and the bf16 operands it emits are the ones the next forward uses (pack cache coherent)."""
    from oracle import w2l_oracle as O
    from wav2letter_pytorch_amd.optim import FusedSGD
    layers = [(128, 11, 2, 1, 0.0), (128, 13, 1, 2, 0.0)]
    sd = O.init_wav2letter_state(layers, seed=4)
    ma = build_w2l(layers, sd, 'bf16').train()
    mb = build_w2l(layers, sd, 'bf16').train()
    kw = dict(lr=0.05, momentum=0.9, nesterov=True, weight_decay=1e-3)
    oa = FusedSGD.from_sgd(torch.optim.SGD(ma.parameters(), **kw))
    oa.overlap = overlap             # True: updates on the optimizer's side stream, ordered by per-layer events
    oa.recycle_grads = recycle       # True (W2L_RECYCLE_GRADS=1): the kernel zeroes the consumed gradient, the engine reuses it as dW
    ob = torch.optim.SGD(mb.parameters(), **kw)
    x, il, tg, tl = O.synthetic_batch(2, 160, seed=11, s_lo=5, s_hi=15)
    for it in range(3):
        for m, o in ((ma, oa), (mb, ob)):
            o.zero_grad(set_to_none=True)
            out, ol = m(x.cuda(), il)
            m.criterion(out.transpose(0, 1), tg, ol, tl).backward()
            o.step()
        if it == 2:
            oa.join()                    # parameters are read outside the step engine below
            for (k, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()):
                assert scale_err(pa.detach().cpu().numpy(), pb.detach().cpu().numpy()) < 2e-5, (it, k)
    wa = ma.conv1ds.conv1d_1.conv1.weight
    pk = wa._w2l_pack[False]
    assert pk.version == wa._version
    ref = wa.detach().permute(2, 0, 1).to(torch.bfloat16)
    assert torch.equal(pk.fwd_hi, ref)
    assert torch.equal(pk.dgr_hi, wa.detach().flip(2).permute(2, 1, 0).to(torch.bfloat16).contiguous())
    assert isinstance(ma.configure_optimizers()[0][0], FusedSGD)


def _bit_reproducible_engine(monkeypatch):
    """two models that must end up with the SAME parameters to 5e-5 need the step's bit-reproducible kernels: the folded forward
    finalize and the two-launch BatchNorm-backward chain sum with fp32 atomics (run-to-run noise in the last bits, which four
    bf16 steps amplify to ~1e-4)"""
    from wav2letter_pytorch_amd import engine as E
    monkeypatch.setattr(E, 'FOLD_BN_FWD', '0')
    monkeypatch.setattr(E, 'FAST_BN_BWD', False)


@pytest.mark.parametrize('k', [1, 3, (-3, -1)])
def test_deferred_weight_gradients_match_plain_sgd(k, monkeypatch):
    """optim.FusedSGD.defer_wgrad: the top k units' weight gradients are computed at the start of the NEXT forward pass and
    consumed by the fused update directly.  Four steps (with an eval-mode forward and a gradient-only backward in between)
    must leave the same parameters as torch.optim.SGD on an identical model; the deferred weights never see a .grad."""
    from oracle import w2l_oracle as O
    from wav2letter_pytorch_amd.optim import FusedSGD
    _bit_reproducible_engine(monkeypatch)
    layers = [(128, 11, 2, 1, 0.0), (128, 13, 1, 2, 0.0), (192, 5, 1, 1, 0.0)]
    sd = O.init_wav2letter_state(layers, seed=14)
    ma = build_w2l(layers, sd, 'bf16').train()
    mb = build_w2l(layers, sd, 'bf16').train()
    kw = dict(lr=0.05, momentum=0.9, nesterov=True, weight_decay=1e-3)
    oa = FusedSGD.from_sgd(torch.optim.SGD(ma.parameters(), **kw))
    oa.overlap = True
    oa.defer_wgrad(ma, k)
    ob = torch.optim.SGD(mb.parameters(), **kw)
    x, il, tg, tl = O.synthetic_batch(2, 160, seed=11, s_lo=5, s_hi=15)
    body = [b.conv1.weight for b in list(ma.conv1ds.children())[:-1]]
    top = body[-k:] if isinstance(k, int) else [body[i] for i in k]
    k = len(top)
    for it in range(4):
        for m, o in ((ma, oa), (mb, ob)):
            o.zero_grad(set_to_none=True)
            out, ol = m(x.cuda(), il)
            m.criterion(out.transpose(0, 1), tg, ol, tl).backward()
            if m is ma:
                assert all(w.grad is None for w in top) and len(ma.engine()._deferred) == k
            o.step()
        if it == 1:                      # an eval forward between two steps sees the updated weights on both models
            ea, eb = (m.eval()(x.cuda(), il)[0] for m in (ma, mb))
            assert not ma.engine()._deferred
            assert scale_err(ea.detach().cpu().numpy(), eb.detach().cpu().numpy()) < 2e-2
            ma.train(), mb.train()
    oa.join()
    for (name, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()):
        assert scale_err(pa.detach().cpu().numpy(), pb.detach().cpu().numpy()) < 5e-5, name
    # the operand packs the next forward will use are those of the updated weights
    for w in top:
        pk = w._w2l_pack[False]
        assert pk.version == w._version and torch.equal(pk.fwd_hi, w.detach().permute(2, 0, 1).to(torch.bfloat16))
    # a backward pass that is NOT followed by an optimizer step: the held-back gradients land in .grad at the next flush
    for m in (ma, mb):
        m.zero_grad(set_to_none=True)
        out, ol = m(x.cuda(), il)
        m.criterion(out.transpose(0, 1), tg, ol, tl).backward()
    assert all(w.grad is None for w in top)
    oa.join()
    torch.cuda.synchronize()
    for (name, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()):
        assert pa.grad is not None, name
        assert scale_err(pa.grad.detach().cpu().numpy(), pb.grad.detach().cpu().numpy()) < 1e-2 or float(pb.grad.abs().max()) < 1e-10, name
    oa.defer_wgrad(ma, 0)
    assert ma.engine().defer_wgrad == 0


@pytest.mark.parametrize('spread', ['even', 'la:1', 'la:2'])
def test_deferred_weight_gradients_spread_over_the_forward(spread, monkeypatch):
    """W2L_DEFER_SPREAD other than the default 'start': the held-back weight gradients are launched in front of later units of the
    forward pass (flush_deferred(pos=ui): 'even' = spread over the units in front of the first deferred layer, 'la:K' = K units ahead
    of the layer whose update it carries).  Five steps with the top three of six units deferred must end on the parameters of the
    same run with everything launched at the start -- every update still lands before its layer's forward convolution."""
    from oracle import w2l_oracle as O
    from wav2letter_pytorch_amd import engine as E
    from wav2letter_pytorch_amd.optim import FusedSGD
    _bit_reproducible_engine(monkeypatch)
    monkeypatch.setattr(E, 'DETERMINISTIC_WGRAD', True)
    layers = [(128, 11, 2, 1, 0.0), (128, 13, 1, 2, 0.0), (128, 5, 1, 1, 0.0), (192, 5, 1, 1, 0.0), (128, 7, 1, 1, 0.0),
              (128, 3, 1, 1, 0.0)]
    sd = O.init_wav2letter_state(layers, seed=15)
    x, il, tg, tl = O.synthetic_batch(2, 160, seed=12, s_lo=5, s_hi=15)
    res = {}
    for mode in ('start', spread):
        monkeypatch.setattr(E, 'DEFER_SPREAD', mode)
        m = build_w2l(layers, sd, 'bf16').train()
        o = FusedSGD.from_sgd(torch.optim.SGD(m.parameters(), lr=0.05, momentum=0.9, nesterov=True, weight_decay=1e-3))
        o.overlap = True
        o.defer_wgrad(m, 3)
        launched_at = []
        real = E.StackEngine.flush_deferred

        def spy(self, pos=None, _real=real):
            before = len(self._deferred)
            _real(self, pos)
            if before != len(self._deferred):
                launched_at.append(pos)

        monkeypatch.setattr(E.StackEngine, 'flush_deferred', spy)
        for it in range(5):
            o.zero_grad(set_to_none=True)
            out, ol = m(x.cuda(), il)
            m.criterion(out.transpose(0, 1), tg, ol, tl).backward()
            assert len(m.engine()._deferred) == 3
            o.step()
        o.join()
        monkeypatch.setattr(E.StackEngine, 'flush_deferred', real)
        torch.cuda.synchronize()
        res[mode] = ({k: v.detach().cpu().numpy().copy() for k, v in m.named_parameters()}, launched_at)
    assert set(p for p in res['start'][1] if p is not None) == {0}
    assert any(p not in (0, None) for p in res[spread][1]), res[spread][1]          # something really was launched mid-forward
    for k, v in res['start'][0].items():
        assert np.array_equal(v, res[spread][0][k]), k


def test_deferred_weight_gradients_skipped_step_and_double_backward(monkeypatch):
    """the two ways a held-back gradient could leak (round-4 advisor finding): (1) backward, NO step (a non-finite-loss guard),
    optimizer.zero_grad(), next batch: the skipped batch's top-layer gradients must be gone -- zero_grad drops them --, not
    resurface in the next step; (2) forward, forward, backward, backward, step: the first backward defers, the second must not
    (its gradient goes to .grad) and step() adds the held one to it: ONE momentum update with the summed gradient.  Both
    against torch.optim.SGD on an identical model without deferral."""
    from oracle import w2l_oracle as O
    from wav2letter_pytorch_amd.optim import FusedSGD
    _bit_reproducible_engine(monkeypatch)
    layers = [(128, 11, 2, 1, 0.0), (128, 13, 1, 2, 0.0), (192, 5, 1, 1, 0.0)]
    sd = O.init_wav2letter_state(layers, seed=15)
    ma = build_w2l(layers, sd, 'bf16').train()
    mb = build_w2l(layers, sd, 'bf16').train()
    kw = dict(lr=0.05, momentum=0.9, nesterov=True, weight_decay=1e-3)
    oa = FusedSGD.from_sgd(torch.optim.SGD(ma.parameters(), **kw))
    oa.overlap = True
    oa.defer_wgrad(ma, 2)
    ob = torch.optim.SGD(mb.parameters(), **kw)
    xa, il, tg, tl = O.synthetic_batch(2, 160, seed=21, s_lo=5, s_hi=15)
    xb = O.synthetic_batch(2, 160, seed=22, s_lo=5, s_hi=15)[0]
    top = [b.conv1.weight for b in list(ma.conv1ds.children())[:-1]][-2:]

    def fb(m, x):
        out, ol = m(x.cuda(), il)
        m.criterion(out.transpose(0, 1), tg, ol, tl).backward()

    def same(tol=5e-5):
        oa.join()
        for (name, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()):
            assert scale_err(pa.detach().cpu().numpy(), pb.detach().cpu().numpy()) < tol, name

    for m, o in ((ma, oa), (mb, ob)):          # one ordinary step so that momentum buffers exist
        o.zero_grad(set_to_none=True)
        fb(m, xa)
        o.step()
    # (1) a skipped step
    for m, o in ((ma, oa), (mb, ob)):
        o.zero_grad(set_to_none=True)
        fb(m, xb)                              # ... the loss is "not finite": no step
        if m is ma:
            assert len(ma.engine()._deferred) == 2 and all(w.grad is None for w in top)
        o.zero_grad(set_to_none=True)
        if m is ma:
            assert not ma.engine()._deferred   # dropped with the other gradients
        fb(m, xa)
        o.step()
    same()
    # (2) two backward passes before one step
    for m, o in ((ma, oa), (mb, ob)):
        o.zero_grad(set_to_none=True)
        o1, l1 = m(xa.cuda(), il)
        o2, l2 = m(xb.cuda(), il)
        m.criterion(o1.transpose(0, 1), tg, l1, tl).backward()
        if m is ma:
            assert len(ma.engine()._deferred) == 2 and all(w.grad is None for w in top)
        m.criterion(o2.transpose(0, 1), tg, l2, tl).backward()
        if m is ma:                            # the second gradient of a weight whose first is held back is computed at once
            assert len(ma.engine()._deferred) == 2 and all(w.grad is not None for w in top)
        o.step()
        if m is ma:
            assert not ma.engine()._deferred   # ... and step() added the held ones to .grad: one update
    same(1e-4)


@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_jasper_dense_golden(precision):
    """Jasper with dense (non-separable) blocks, repeat 2, residual 1x1 conv + BN, dilation 2, stride-2 first block,
    length masking with ragged (odd) lengths: reference-generated fixture jasper_dense.npz"""
    from gpu_helpers import build_jasper, compare_jasper_step
    z = load('jasper_dense.npz')
    meta = ast.literal_eval(str(z['meta']))
    sd = {k[3:]: torch.from_numpy(z[k].copy()) for k in z.files if k.startswith('p0/')}
    model = build_jasper(meta['blocks'], sd, precision).train()
    x = torch.from_numpy(z['x'])
    il, tg, tl = (torch.from_numpy(z[k]) for k in ('in_lens', 'targets', 'target_lens'))
    errs, stats, out, out_lens = compare_jasper_step(model, meta['blocks'], sd, x, il, tg, tl, precision)
    check(errs, stats, precision)
    t = TOL[precision]
    np.testing.assert_array_equal(out_lens.numpy(), z['out_lens'])
    assert scale_err(out.cpu().numpy(), z['log_probs']) < t['lp']
    if precision == 'fp32':
        check_fixture_grads(model, z)
        model.eval()                 # eval: running stats AND softmax instead of log_softmax (jasper.py:470-473)
        with torch.no_grad():
            oe, _ = model(x.cuda(), il)
        assert scale_err(oe.cpu().numpy(), z['out_eval']) < 2e-3
        assert abs(float(oe.sum(-1).mean()) - 1.0) < 1e-4


@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_jasper_conv_mask_false_golden(precision):
    """``conv_mask: False`` blocks (plain nn.Conv1d: no length masking, lengths passed through; jasper.py:288-298,393-397,446)
    between masked ones: reference-generated fixture jasper_nomask.npz -- state-dict keys without ``.conv``, an un-masked
    block reading a masked block's un-masked output, a masked block behind it, an un-masked block in front of the head"""
    from gpu_helpers import build_jasper, compare_jasper_step
    z = load('jasper_nomask.npz')
    meta = ast.literal_eval(str(z['meta']))
    sd = {k[3:]: torch.from_numpy(z[k].copy()) for k in z.files if k.startswith('p0/')}
    assert 'jasper_encoder.1.mconv.0.weight' in sd and 'jasper_encoder.2.mconv.0.conv.weight' in sd
    model = build_jasper(meta['blocks'], sd, precision).train()
    assert set(model.state_dict()) == set(sd)
    x = torch.from_numpy(z['x'])
    il, tg, tl = (torch.from_numpy(z[k]) for k in ('in_lens', 'targets', 'target_lens'))
    errs, stats, out, out_lens = compare_jasper_step(model, meta['blocks'], sd, x, il, tg, tl, precision)
    check(errs, stats, precision)
    np.testing.assert_array_equal(out_lens.numpy(), z['out_lens'])
    assert scale_err(out.cpu().numpy(), z['log_probs']) < TOL[precision]['lp']
    if precision == 'fp32':
        check_fixture_grads(model, z)
        model.eval()
        with torch.no_grad():
            oe, _ = model(x.cuda(), il)
        assert scale_err(oe.cpu().numpy(), z['out_eval']) < 2e-3
    with pytest.raises(AttributeError):          # jasper.py:458 reads mconv[0].conv.stride: a bare Conv1d has no .conv
        model.scaling_factor


@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_jasper_separable_golden(precision):
    """the shipped jasper.yaml form: depthwise (k32->33 / k38->39, stride 2 first) + pointwise convs, residual,
    masking with an odd ragged length (float length arithmetic): reference-generated fixture jasper_sep2.npz"""
    from gpu_helpers import build_jasper, compare_jasper_step
    z = load('jasper_sep2.npz')
    meta = ast.literal_eval(str(z['meta']))
    sd = {k[3:]: torch.from_numpy(z[k].copy()) for k in z.files if k.startswith('p0/')}
    model = build_jasper(meta['blocks'], sd, precision).train()
    x = torch.from_numpy(z['x'])
    il, tg, tl = (torch.from_numpy(z[k]) for k in ('in_lens', 'targets', 'target_lens'))
    errs, stats, out, out_lens = compare_jasper_step(model, meta['blocks'], sd, x, il, tg, tl, precision)
    check(errs, stats, precision)
    np.testing.assert_array_equal(out_lens.numpy(), z['out_lens'])
    assert scale_err(out.cpu().numpy(), z['log_probs']) < TOL[precision]['lp']
    if precision == 'fp32':
        check_fixture_grads(model, z)


@pytest.mark.parametrize('defer', ['0', '2'])
def test_trainer_fit_loop_and_checkpoint(tmp_path, defer, monkeypatch):
    """training_step / validation_step / configure_optimizers through the minimal Trainer (train.py:34-37 flow):
    loss decreases on a fixed batch, metrics carry the reference's log keys, the checkpoint reloads.  ``defer`` = 2: the
    weight gradients of both conv units are held back for the next forward pass (W2L_DEFER_WGRAD): validation and the
    checkpoint after each epoch must see every update of the epoch (Trainer joins the optimizer, which flushes), and the
    two runs must end on the same parameters."""
    monkeypatch.setenv('W2L_DEFER_WGRAD', defer)
    from oracle import w2l_oracle as O
    from wav2letter_pytorch_amd.trainer import Trainer
    layers = [(128, 11, 2, 1, 0.0), (128, 11, 1, 1, 0.0)]
    sd = O.init_wav2letter_state(layers, seed=21)
    model = build_w2l(layers, sd, 'bf16')
    model._cfg.optimizer.lr = 0.05
    x, il, tg, tl = O.synthetic_batch(4, 200, seed=22, s_lo=5, s_hi=12)
    texts = tuple(''.join(O.ENGLISH_LOWERCASE[int(i)] for i in tg[n, :int(tl[n])]) for n in range(4))
    batch = (x, il, tg, tl, ('a', 'b', 'c', 'd'), texts)
    tr = Trainer(default_root_dir=str(tmp_path), max_epochs=2, log_every_n_steps=1)
    tr.fit(model, [batch] * 6, [batch])
    losses = [l['train_loss'] for _, l in tr.logged]
    assert losses[-1] < losses[0]
    keys = set(tr.logged[-1][1])
    assert {'train_loss', 'learning_rate', 'train_cer', 'train_wer', 'train_len_ratio'} <= keys
    assert {'val_loss', 'val_cer', 'val_wer', 'val_len_ratio'} <= set(model._logged)
    ck = sorted(os.listdir(tmp_path))
    assert len(ck) == 2 and ck[0].endswith('.ckpt')
    state = torch.load(os.path.join(tmp_path, ck[-1]))['state_dict']
    m2 = build_w2l(layers, sd, 'bf16')
    m2.load_state_dict(state)
    for (k, a), (_, b) in zip(model.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a.cpu(), b.cpu()), k
    assert model.engine().defer_wgrad == int(defer) and not model.engine()._deferred
    final = {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items() if v.dtype.is_floating_point}
    other = _TRAINER_RUNS.setdefault('final', final)
    for k, v in final.items():                    # (second parametrisation: same end state as the first)
        assert scale_err(v, other[k]) < 1e-4, k


_TRAINER_RUNS = {}


@pytest.mark.parametrize('mode', ['async', 'sync'])
def test_lightning_base_class_branch(mode):
    """base_asr_models takes `ptl.LightningModule` as its base class when pytorch_lightning is importable (reference
    base_asr_models.py:15-17).  The image has no Lightning, so a subprocess injects a minimal one (tests/lightning_shim_worker.py:
    LightningModule = nn.Module + log_dict / optimizers / hooks, Trainer.fit = automatic optimisation in Lightning's hook
    order, as train.py:34-37 calls it) BEFORE importing the package: the class hierarchy, tensor-valued log_dict (the
    synchronous order logs the loss TENSOR, as the reference does), `self.optimizers()`, a plain optimizer step over
    un-deferred gradients (every p.grad present between backward() and step()), on_train_batch_end / on_train_epoch_end
    resolving the asynchronous metrics, validation_step in eval mode."""
    import json
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    out = subprocess.run([sys.executable, os.path.join(here, 'lightning_shim_worker.py'), mode], capture_output=True, text=True,
                         timeout=280)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    assert len(d['train_loss']) == 10 and d['train_loss'][-1] < d['train_loss'][0] and d['n_val'] == 2
    assert d['train_keys'] == ['learning_rate', 'train_cer', 'train_len_ratio', 'train_loss', 'train_wer']
    assert d['val_keys'] == ['val_cer', 'val_len_ratio', 'val_loss', 'val_wer']
    assert all(d['grads_seen']) and d['moved'] >= 6 and d['pending'] == 0
    assert d['tensor_valued'] == (mode == 'sync')
    assert d['lr'][0] == 0.05 and abs(d['lr'][-1] - 0.05 * 0.999) < 1e-12        # ExponentialLR stepped once per epoch


@pytest.mark.parametrize('kind', ['w2l', 'jasper'])
def test_training_step_async_metrics_equal_synchronous(kind, monkeypatch):
    """training_step only ENQUEUES the string metrics (argmax kernel + asynchronous copy + event) and on_train_batch_end logs
    them behind backward() / optimizer.step(): the values logged for step i must be exactly -- bit for bit -- what the
    reference's order (score inside training_step: async_metrics = False) logs for the same step, for every step of a short
    run: train_loss, learning_rate, train_cer, train_wer, train_len_ratio.  Two identical models, identical batches (three
    different ones in rotation, the last with ragged lengths), FusedSGD with deferred weight gradients as the trainer runs it."""
    from gpu_helpers import build_jasper
    from oracle import w2l_oracle as O
    from wav2letter_pytorch_amd import engine as E
    # (bit-for-bit equality of two runs needs the step's bit-reproducible kernels: no fp32 atomics anywhere)
    _bit_reproducible_engine(monkeypatch)
    monkeypatch.setattr(E, 'DETERMINISTIC_WGRAD', True)
    if kind == 'w2l':
        layers = [(128, 11, 2, 1, 0.0), (128, 11, 1, 1, 0.0)]
        sd = O.init_wav2letter_state(layers, seed=91)
        make = lambda: build_w2l(layers, sd, 'bf16')         # noqa: E731
    else:
        z = load('jasper_dense.npz')
        meta = ast.literal_eval(str(z['meta']))
        sdj = {k[3:]: torch.from_numpy(z[k].copy()) for k in z.files if k.startswith('p0/')}
        make = lambda: build_jasper(meta['blocks'], sdj, 'bf16')        # noqa: E731
    batches = []
    for b, (n, t) in enumerate(((4, 200), (3, 260), (4, 200))):
        x, il, tg, tl = O.synthetic_batch(n, t, seed=92 + b, s_lo=5, s_hi=12)
        if b == 2:
            il = torch.tensor([200, 150, 180, 120], dtype=torch.int32)
            tl = torch.minimum(tl, torch.tensor(20, dtype=torch.int32))
        texts = tuple(''.join(O.ENGLISH_LOWERCASE[int(i)] for i in tg[k, :int(tl[k])]) for k in range(n))
        batches.append((x, il, tg, tl, tuple('f%d' % k for k in range(n)), texts))
    logs = {}
    for mode in (True, False):
        torch.manual_seed(5)
        model = make().cuda().train()
        model.check_nan = False
        model.async_metrics = mode
        from wav2letter_pytorch_amd.optim import FusedSGD
        opt = FusedSGD.from_sgd(torch.optim.SGD(model.parameters(), lr=0.02, momentum=0.9, nesterov=True, weight_decay=1e-5))
        model._optimizers = opt
        opt.overlap = True
        seen = []
        for i in range(6):
            batch = batches[i % 3]
            opt.zero_grad(set_to_none=True)
            loss = model.training_step(batch, i)
            if mode:
                assert len(model._pending_metrics) >= 1            # nothing was scored inside training_step
            loss.backward()
            opt.step()
            model.on_train_batch_end(loss, batch, i)
            if not mode:
                seen.append(dict(model._logged))
            else:
                model.resolve_metrics(wait_all=True)
                seen.append(dict(model._logged))
        opt.join()
        logs[mode] = seen
    keys = {'train_loss', 'learning_rate', 'train_cer', 'train_wer', 'train_len_ratio'}
    for a, b in zip(logs[True], logs[False]):
        assert set(a) >= keys and set(b) >= keys
        for k in keys:
            assert a[k] == b[k], (k, a[k], b[k])
    # and without the explicit resolve: the hook alone never leaves more than METRICS_MAX_PENDING batches outstanding, an epoch end none
    model.async_metrics = True
    for i in range(4):
        opt.zero_grad(set_to_none=True)
        loss = model.training_step(batches[i % 3], i)
        loss.backward()
        opt.step()
        model.on_train_batch_end(loss, batches[i % 3], i)
        assert len(model._pending_metrics) <= model.METRICS_MAX_PENDING
    model.on_train_epoch_end()
    assert not model._pending_metrics
    opt.join()




def test_deferred_weight_gradients_fp8_and_jasper(monkeypatch):
    """the deferred mode on the other two engine paths: (a) ``precision: fp8`` with e4m3 weight gradients -- the held-back
    launch keeps dy's e4m3 copy and its device-side scale alive until the next forward pass; (b) a Jasper stack whose top
    units carry a residual 1x1 branch (both convolutions of the last unit are deferred).  Same parameters after three steps as
    the plain FusedSGD step on an identical model.  (The BatchNorm sums take their bit-reproducible kernels here: with the
    atomic ones, one run in a few dozen amplified a last-bit difference through the e4m3 copies to 3e-3 of scale.)"""
    from gpu_helpers import build_jasper
    _bit_reproducible_engine(monkeypatch)
    from oracle import w2l_oracle as O
    from wav2letter_pytorch_amd import engine as E
    from wav2letter_pytorch_amd.optim import FusedSGD
    kw = dict(lr=0.05, momentum=0.9, nesterov=True, weight_decay=1e-3)

    def run(models, batch, tol=1e-4, chaotic=False):
        x, il, tg, tl = batch
        opts = []
        for i, m in enumerate(models):
            o = FusedSGD.from_sgd(torch.optim.SGD(m.parameters(), **kw))
            o.overlap = True
            if i == 0:
                o.defer_wgrad(m, 2)
            opts.append(o)

        def step():
            for m, o in zip(models, opts):
                o.zero_grad(set_to_none=True)
                out, ol = m(x.cuda(), il)
                m.criterion(out.transpose(0, 1), tg, ol, tl).backward()
                o.step()

        def worst():
            for o in opts:
                o.join()
            assert models[0].engine().defer_wgrad == 2
            errs = {k: scale_err(pa.detach().cpu().numpy(), pb.detach().cpu().numpy())
                    for (k, pa), (_, pb) in zip(models[0].named_parameters(), models[1].named_parameters())}
            k = max(errs, key=errs.get)
            return k, errs[k]

        step()
        if chaotic:
            # fp8: split weight gradients add their pieces with fp32 atomics, and from the SECOND forward on a last-bit difference
            # in a weight can move its e4m3 copy by a whole step -- two identical PLAIN models then end 1e-2 of scale apart after
            # three steps in one run out of three (tools/probe/dbg_fp8_defer2.py).  So the exact comparison is made where no
            # such forward has happened yet: one step each, then ONE MORE FORWARD of the deferred model -- at whose start its
            # held-back gradients (dy's e4m3 copy and scale kept alive since the backward) are launched and applied
            assert len(models[0].engine()._deferred) == 2
            models[0](x.cuda(), il)
            assert not models[0].engine()._deferred
            k, e = worst()
            assert e < tol, (k, e)
        step()
        step()
        k, e = worst()
        assert e < (0.25 if chaotic else tol), (k, e)          # (chaotic: a guard against garbage only)
        assert all(torch.isfinite(p).all() for m in models for p in m.parameters())

    layers = [(128, 11, 2, 1, 0.0), (256, 13, 1, 1, 0.0), (128, 5, 1, 2, 0.0)]
    sd = O.init_wav2letter_state(layers, seed=91)
    E.FP8_DGRAD = E.FP8_WGRAD = '1'
    try:
        run([build_w2l(layers, sd, 'fp8').train() for _ in range(2)], O.synthetic_batch(4, 300, seed=92, s_lo=8, s_hi=25), tol=2e-5,
            chaotic=True)
    finally:
        E.FP8_DGRAD = E.FP8_WGRAD = 'auto'
    z = load('jasper_dense.npz')
    meta = ast.literal_eval(str(z['meta']))
    jsd = {k[3:]: torch.from_numpy(z[k].copy()) for k in z.files if k.startswith('p0/')}
    il, tg, tl = (torch.from_numpy(z[k]) for k in ('in_lens', 'targets', 'target_lens'))
    run([build_jasper(meta['blocks'], jsd, 'bf16').train() for _ in range(2)], (torch.from_numpy(z['x']), il, tg, tl))


@pytest.mark.parametrize('case', ['w2l_ml3', 'w2l_mix5', 'jasper_dense', 'jasper_sep2'])
def test_input_gradient_golden(case):
    """d loss / d spectrogram (cold path: zero-stuffed strided data gradient + reflect fold) vs the fixture's
    input_grad; compared on the L2 norm because activation-gate ties perturb single elements"""
    z = load(case + '.npz')
    meta = ast.literal_eval(str(z['meta']))
    sd = {k[3:]: torch.from_numpy(z[k].copy()) for k in z.files if k.startswith('p0/')}
    if case.startswith('jasper'):
        from gpu_helpers import build_jasper
        model = build_jasper(meta['blocks'], sd, 'fp32').train()
    else:
        layers = [(l['output_size'], l['kernel_size'], l['stride'], l['dilation'], 0.0) for l in meta['layers']]
        model = build_w2l(layers, sd, 'fp32').train()
    x = torch.from_numpy(z['x']).cuda().requires_grad_(True)
    il, tg, tl = (torch.from_numpy(z[k]) for k in ('in_lens', 'targets', 'target_lens'))
    out, ol = model(x, il)
    model.criterion(out.transpose(0, 1), tg, ol, tl).backward()
    ref = z['input_grad']
    got = x.grad.cpu().numpy()
    assert got.shape == ref.shape
    rel = np.linalg.norm(got - ref) / np.linalg.norm(ref)
    assert rel < 2e-2, rel
    assert np.mean(np.abs(got - ref) > 1e-3 * np.abs(ref).max()) < 0.05


@pytest.mark.parametrize('N,T', [(1, 57), (5, 129), (2, 64)])
def test_w2l_edge_shapes_fp32(N, T):
    """batch of one, odd / tiny utterance lengths (Tout < one 64-row wgrad step, partial 128-row tiles), ragged"""
    layers = [(64, 11, 2, 1, 0.0), (96, 13, 1, 1, 0.0), (64, 7, 1, 2, 0.0)]
    _synthetic_case(layers, N=N, T=T, precision='fp32', seed=30 + N, ragged=N > 1)


def test_reflect_pad_longer_than_sequence_raises():
    """ReflectionPad1d needs pad < T (torch raises in the reference too): T'=10 after stride 2, pad 12"""
    from oracle import w2l_oracle as O
    from wav2letter_pytorch_amd._lib import W2LError
    layers = [(64, 11, 2, 1, 0.0), (64, 13, 1, 2, 0.0)]
    model = build_w2l(layers, O.init_wav2letter_state(layers, seed=1), 'fp32').train()
    with pytest.raises(W2LError):
        model(torch.randn(2, 64, 20).cuda(), torch.tensor([20, 20]))


def test_infeasible_and_empty_targets_end_to_end():
    """zero_infinity through the whole step: an utterance whose target cannot be aligned contributes 0 loss and
    0 gradient; an empty target is legal (base_asr_models.py:23)"""
    from oracle import w2l_oracle as O
    layers = [(64, 11, 2, 1, 0.0), (64, 11, 1, 1, 0.0)]
    sd = O.init_wav2letter_state(layers, seed=40)
    model = build_w2l(layers, sd, 'fp32').train()
    x, il, tg, tl = O.synthetic_batch(3, 80, seed=41, s_lo=5, s_hi=10)
    tg = torch.cat([tg, torch.randint(1, 29, (3, 60), dtype=torch.int32)], 1)
    tl = torch.tensor([70, 0, int(tl[2])], dtype=torch.int32)          # 70 labels cannot fit 40 frames; empty target
    tg[1] = 0
    errs, stats, out, out_lens, ref = compare_step(model, layers, sd, x, il, tg, tl, 'fp32')
    check(errs, stats, 'fp32')
    assert float(ref['loss']) > 0


@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_jasper_long_utterance_T16000(precision):
    """BASELINE config 5's sequence length (T = 16 000 input frames -> T' = 8 000): the conv kernels tile time, BatchNorm
    reduces over 2 x 8 000 frames, and the CTC kernel streams log-probs from HBM (8 000 x 29 fp32 no longer fits LDS).
    Parameters of the reference-generated Jasper fixture, a new ragged batch, checked against the oracle."""
    from gpu_helpers import build_jasper, compare_jasper_step
    from oracle import w2l_oracle as O
    z = load('jasper_dense.npz')
    meta = ast.literal_eval(str(z['meta']))
    sd = {k[3:]: torch.from_numpy(z[k].copy()) for k in z.files if k.startswith('p0/')}
    model = build_jasper(meta['blocks'], sd, precision).train()
    x, il, tg, tl = O.synthetic_batch(2, 16000, seed=77, s_lo=900, s_hi=1500)
    il[1] = 12345
    x[1, :, 12345:] = 0
    errs, stats, out, out_lens = compare_jasper_step(model, meta['blocks'], sd, x, il, tg, tl, precision)
    # log-probs and loss at the usual bounds.  Gradients: at T' = 8 000 alpha/beta reach -3e4 nats, where one fp32 ulp is
    # 2e-3 of a posterior; torch's own fp32 CPU CTC gradient (what the reference runs, and what the oracle replays) is 4e-2
    # of scale away from its float64 evaluation at this length, so two correct fp32 pipelines agree to a few 1e-2 only.
    check(errs, stats, precision, grad_tol=5e-2 if precision == 'fp32' else 1.2e-1)
    assert out.shape == (2, 8000, 29) and [int(v) for v in out_lens] == [8000, 6173]


def test_w2l_long_utterance_T16000_fp32():
    """same length through the Wav2Letter stack (reflect padding, stride 2, dilation 2)"""
    from oracle import w2l_oracle as O
    layers = [(64, 11, 2, 1, 0.0), (128, 13, 1, 1, 0.0), (64, 29, 1, 2, 0.0)]
    sd = O.init_wav2letter_state(layers, seed=31)
    model = build_w2l(layers, sd, 'fp32').train()
    x, il, tg, tl = O.synthetic_batch(2, 16000, seed=78, s_lo=1000, s_hi=2000)
    errs, stats, out, out_lens, ref = compare_step(model, layers, sd, x, il, tg, tl, 'fp32')
    check(errs, stats, 'fp32', grad_tol=5e-2)           # see test_jasper_long_utterance_T16000
    assert out.shape == (2, 8000, 29)


def test_deterministic_mode_is_bit_reproducible(monkeypatch):
    """W2L_DETERMINISTIC: weight-gradient split reductions summed in a fixed order (slabs + ticket) instead of fp32 atomics:
    two runs of the same step give bit-identical gradients, and they agree with the default (atomic) path"""
    from oracle import w2l_oracle as O
    from wav2letter_pytorch_amd import engine as E
    layers = [(128, 11, 2, 1, 0.0), (256, 13, 1, 1, 0.0), (128, 29, 1, 2, 0.0)]
    sd = O.init_wav2letter_state(layers, seed=41)
    x, il, tg, tl = O.synthetic_batch(8, 400, seed=42, s_lo=10, s_hi=30)

    def grads(det):
        monkeypatch.setattr(E, 'DETERMINISTIC_WGRAD', det)
        model = build_w2l(layers, sd, 'bf16').train()
        out, ol = model(x.cuda(), il)
        model.criterion(out.transpose(0, 1), tg, ol, tl).backward()
        torch.cuda.synchronize()
        return [p.grad.detach().clone() for p in model.parameters()]

    # the default-mode step runs FIRST: its measured plans (classic splits that add atomically although a workspace is at
    # hand, plan bit 6) are in the plan table when the deterministic steps run -- they must not bring the atomics back
    c, a, b = grads(False), grads(True), grads(True)
    for ga, gb, gc in zip(a, b, c):
        assert torch.equal(ga, gb)
        # (the default path also sums the BatchNorm statistics / backward sums with fp32 atomics -- the folded finalize, the
        # two-launch backward chain --: a last-bit difference there moves bf16 activations by an ulp, gradients by ~1e-3)
        assert scale_err(ga.cpu().numpy(), gc.cpu().numpy()) < 1e-2


@pytest.mark.parametrize('fused,fold', [(True, False), (False, True), (True, True), (False, False)])
def test_optional_bn_backward_paths(fused, fold, monkeypatch):
    """the forms of the BatchNorm backward -- the reduction formed in the data gradient's epilogue (W2L_FUSED_BN_REDUCE:
    w2l_conv1d_dgrad_bnreduce_ws; 'auto' = on below 12 288 frames, so forced on AND off here) and the finalize folded into the
    dy kernel (W2L_FOLD_BN_FINALIZE=1, w2l_bn_act_bwd_apply_fin) -- all give the step of the three-launch form: Wav2Letter stack with dropout, reflect
    padding, stride 2, dilation 2 (fp32 mode falls back to the separate reduction: bf16 here), and the Jasper fixture with
    residual branches, masks and two gradient sources per block input"""
    from gpu_helpers import build_jasper, compare_jasper_step
    from oracle import w2l_oracle as O
    from wav2letter_pytorch_amd import engine as E
    monkeypatch.setattr(E, 'FUSED_BN_REDUCE', fused)
    monkeypatch.setattr(E, 'FOLD_BN_FINALIZE', fold)
    layers = [(128, 11, 2, 1, 0.2), (192, 13, 1, 1, 0.3), (128, 29, 1, 2, 0.2)]
    sd = O.init_wav2letter_state(layers, seed=81)
    model = build_w2l(layers, sd, 'bf16', dropout=True).train()
    x, il, tg, tl = O.synthetic_batch(5, 333, seed=82, s_lo=10, s_hi=40)
    errs, stats, *_ = compare_step(model, layers, sd, x, il, tg, tl, 'bf16', drop=True)
    check(errs, stats, 'bf16')
    z = load('jasper_dense.npz')
    meta = ast.literal_eval(str(z['meta']))
    sdj = {k[3:]: torch.from_numpy(z[k].copy()) for k in z.files if k.startswith('p0/')}
    for precision in ('bf16', 'fp32'):
        mj = build_jasper(meta['blocks'], sdj, precision).train()
        xj = torch.from_numpy(z['x'])
        ilj, tgj, tlj = (torch.from_numpy(z[k]) for k in ('in_lens', 'targets', 'target_lens'))
        errs, stats, out, out_lens = compare_jasper_step(mj, meta['blocks'], sdj, xj, ilj, tgj, tlj, precision)
        check(errs, stats, precision)


@pytest.mark.parametrize('order', [2, 3])
def test_stream_k_weight_gradients_end_to_end(order):
    """every weight gradient of a step through the stream-K plan of conv_wgrad_kernel (forced; the tuner picks it per
    shape): parity with the oracle as for the split-K plans.  The force hook is thread-local, so backward runs on this thread."""
    from oracle import w2l_oracle as O
    from wav2letter_pytorch_amd import _lib as L
    from wav2letter_pytorch_amd import engine as E
    layers = [(128, 11, 2, 1, 0.0), (256, 13, 1, 1, 0.0), (320, 29, 1, 2, 0.0), (128, 5, 1, 1, 0.0)]
    sd = O.init_wav2letter_state(layers, seed=71)
    x, il, tg, tl = O.synthetic_batch(6, 700, seed=72, s_lo=20, s_hi=60)
    old = E.AUTOTUNE
    E.AUTOTUNE = False
    L.lib.w2l_wgrad_force_plan(0, order)
    try:
        with torch.autograd.set_multithreading_enabled(False):
            model = build_w2l(layers, sd, 'bf16').train()
            errs, stats, *_ = compare_step(model, layers, sd, x, il, tg, tl, 'bf16')
    finally:
        L.lib.w2l_wgrad_force_plan(0, -1)
        E.AUTOTUNE = old
    check(errs, stats, 'bf16')


def test_graphed_train_step():
    """the step captured into a hipGraph (graph.GraphedTrainStep): replays train like the eager step, and dropout masks
    change from replay to replay (device-side Philox offset)"""
    from oracle import w2l_oracle as O
    from wav2letter_pytorch_amd.graph import GraphedTrainStep
    layers = [(128, 11, 2, 1, 0.0), (128, 13, 1, 2, 0.0)]
    sd = O.init_wav2letter_state(layers, seed=51)
    x, il, tg, tl = O.synthetic_batch(4, 200, seed=52, s_lo=5, s_hi=15)

    def make(dropout):
        lay = [(c, k, s, d, 0.3 if dropout else 0.0) for c, k, s, d, _ in layers]
        m = build_w2l(lay, sd, 'bf16', dropout=dropout).train()
        m._cfg.optimizer.lr = 0.02
        return m, m.configure_optimizers()[0][0]

    # (a) no dropout: 3 eager warm-up steps + 4 replays (capturing records, it does not execute) == 7 eager steps
    mg, og = make(False)
    step = GraphedTrainStep(mg, og, x, il, tg, tl, warmup=3)
    losses = [float(step()) for _ in range(4)]
    me, oe = make(False)
    ol = me.compute_output_lengths(il)
    for _ in range(7):
        oe.zero_grad(set_to_none=True)
        out, _ = me(x.cuda(), None)
        le = me.criterion(out.transpose(0, 1), tg, ol, tl)
        le.backward()
        oe.step()
    assert abs(losses[-1] - float(le)) < 2e-2 * abs(float(le)), (losses, float(le))
    assert losses[-1] < losses[0]
    for (k, pa), (_, pb) in zip(mg.named_parameters(), me.named_parameters()):
        assert scale_err(pa.detach().cpu().numpy(), pb.detach().cpu().numpy()) < 2e-2, k
    assert int(step.counter) == 7 * step.n_units
    # a new batch of the same shape goes through the static buffers
    x2, _, tg2, tl2 = O.synthetic_batch(4, 200, seed=53, s_lo=5, s_hi=15)
    if tg2.shape == tg.shape:
        l2 = float(step(x2, tg2, tl2))
        assert np.isfinite(l2) and abs(l2 - losses[-1]) > 1e-6
    # (b) dropout on, lr 0: the parameters never move, so the loss only changes through the masks
    md, od = make(True)
    for g in od.param_groups:
        g['lr'] = 0.0
        g['weight_decay'] = 0.0
    sd_ = GraphedTrainStep(md, od, x, il, tg, tl, warmup=2)
    vals = [float(sd_()) for _ in range(3)]
    assert len({round(v, 6) for v in vals}) == 3, vals


def test_data_parallel_two_ranks_on_one_gpu(tmp_path):
    """Two ranks (gloo, both on cuda:0) run one data-parallel step through the production path; checked against the
    same two batches run one after the other in this process: identical replicas after the broadcast, every gradient
    equal to the mean of the two per-batch gradients, identical parameters after the optimizer step."""
    import socket
    import subprocess
    import sys
    from oracle import w2l_oracle as O
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    here = os.path.dirname(os.path.abspath(__file__))
    base = str(tmp_path / 'dp')
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK='0', WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(here, 'dp_gpu_worker.py'), base], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0
    z = [np.load(base + f'.rank{r}.npz') for r in range(2)]
    keys = [k[3:] for k in z[0].files if k.startswith('p0/')]
    layers = [(128, 11, 2, 1, 0.0), (256, 13, 1, 1, 0.0), (128, 29, 1, 2, 0.0)]
    sd = O.init_wav2letter_state(layers, seed=60)               # rank 0's state is what the broadcast distributes
    single = []
    for r in range(2):
        m = build_w2l(layers, sd, 'fp32').train()
        x, il, tg, tl = O.synthetic_batch(4, 240, seed=70 + r, s_lo=5, s_hi=20)
        out, ol = m(x.cuda(), il)
        m.criterion(out.transpose(0, 1), tg, ol, tl).backward()
        single.append({k: v.grad.detach().cpu().numpy() for k, v in m.named_parameters()})
    for k in keys:
        np.testing.assert_array_equal(z[0]['p0/' + k], z[1]['p0/' + k])            # broadcast
        np.testing.assert_array_equal(z[0]['g/' + k], z[1]['g/' + k])              # same averaged gradient everywhere
        want = 0.5 * (single[0][k] + single[1][k])
        assert scale_err(z[0]['g/' + k], want) < 2e-3 or np.abs(want).max() < 1e-12, k
        np.testing.assert_array_equal(z[0]['p1/' + k], z[1]['p1/' + k])            # replicas stay identical
        assert not np.array_equal(z[0]['p1/' + k], z[0]['p0/' + k]) or 'conv1.bias' in k, k


def test_data_parallel_deferred_wgrad_two_ranks(tmp_path):
    """the deferred weight gradients under data parallelism: their all-reduce is started behind the kernel at the start of
    the next forward pass and the fused update waits for it.  Two gloo ranks on cuda:0, three steps, once with the top two
    units held back and once plain: replicas stay bit-identical across the ranks, and the two modes end on the same
    parameters."""
    import socket
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    res = {}
    for defer in (2, 0):
        s = socket.socket()
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
        s.close()
        base = str(tmp_path / f'dpd{defer}')
        procs = []
        for r in range(2):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK='0', WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                       W2L_AUTOTUNE='0')
            procs.append(subprocess.Popen([sys.executable, os.path.join(here, 'dp_gpu_worker.py'), base, str(defer)], env=env))
        for p in procs:
            assert p.wait(timeout=300) == 0
        res[defer] = [np.load(base + f'.rank{r}.npz') for r in range(2)]
    assert list(res[2][0]['held']) == [2, 2, 2] and list(res[0][0]['held']) == [0, 0, 0]
    for k in [k for k in res[2][0].files if k.startswith('p/')]:
        np.testing.assert_array_equal(res[2][0][k], res[2][1][k])
        assert scale_err(res[2][0][k], res[0][0][k]) < 5e-5, k


def test_data_parallel_grouped_wgrad_two_ranks(tmp_path):
    """grouped weight gradients (w2l_conv1d_wgrad_group: ONE side-stream launch for two layers) under data parallelism: the
    members' gradients are handed to the reducer on the stream that wrote them (its ordering event must cover the group
    kernel).  Two gloo ranks on cuda:0, one step, once with backward-order layers 2 and 3 forced into a group
    (W2L_WGRAD_GROUPS='2,3', no autotune so the group launch is what runs) and once one by one: replicas bit-identical across
    the ranks, both modes on the same parameters."""
    import socket
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    res = {}
    for groups in ('2,3', '0'):
        s = socket.socket()
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
        s.close()
        base = str(tmp_path / f'dpg{groups[0]}')
        procs = []
        for r in range(2):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK='0', WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                       W2L_AUTOTUNE='0', W2L_WGRAD_GROUPS=groups, W2L_TEST_GROUP_STACK='1', W2L_TEST_STEPS='1')
            procs.append(subprocess.Popen([sys.executable, os.path.join(here, 'dp_gpu_worker.py'), base, '0'], env=env))
        for p in procs:
            assert p.wait(timeout=300) == 0
        res[groups] = [np.load(base + f'.rank{r}.npz') for r in range(2)]
    assert len(res['2,3'][0]['grouped']) == 1 and len(res['0'][0]['grouped']) == 0
    for k in [k for k in res['0'][0].files if k.startswith('p/')]:
        np.testing.assert_array_equal(res['2,3'][0][k], res['2,3'][1][k])
        # (ONE step: the group launch sums each tile in one block, the single launches split theirs and add atomically -- a
        # last-bit difference that further bf16 steps would amplify; a gradient read before it was written would be off by O(1))
        assert scale_err(res['2,3'][0][k], res['0'][0][k]) < 2e-3, k


def test_bench_two_rank_command_line(tmp_path):
    """the driver's multi-GPU command line (python -m torch.distributed.run ... bench.py --gpus 2) rehearsed on one GPU with
    the gloo backend: both ranks must reach the end (every collective matched on every rank) and rank 0 prints one JSON line"""
    import json
    import socket
    import subprocess
    import sys
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, W2L_DIST_BACKEND='gloo')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
           '--mid-layers', '2', '--batch', '2', '--frames', '200', '--no-cpu-baseline']
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['config']['global_batch'] == 4 and d['value'] > 0 and d['roofline']['achieved'] > 0


def test_bench_self_launch_two_ranks(tmp_path):
    """plain `python bench.py --gpus 2` (no torchrun, no rendezvous variables): bench.py starts the two ranks itself
    (launch.spawn_ranks) and forwards rank 0's single JSON line with the size of the process group it really ran in;
    rehearsed on one GPU with gloo.  A mismatch between --gpus and an outer launch is refused."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env['W2L_DIST_BACKEND'] = 'gloo'
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--mid-layers', '2',
           '--batch', '2', '--frames', '200', '--no-cpu-baseline', '--defer-wgrad', '1', '--collective-ab']
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['defer_wgrad']['units'] == '1'
    assert d['n_gpus'] == 2 and d['rccl_world'] == 2 and d['backend'] == 'gloo' and d['config']['global_batch'] == 4
    assert len(d['rank_ms_per_step']) == 2 and all(v > 0 for v in d['rank_ms_per_step'])
    assert d['exposed_comm_ms'] is not None and abs(d['per_gpu_value'] * 2 - d['value']) < 1.0
    assert len(d['exposed_comm_ms_by_rank']) == 2 and d['exposed_comm_ms'] == max(d['exposed_comm_ms_by_rank'])
    # what the first multi-GPU curve needs to explain itself: the payload of the gradient collectives per step (fp32: four
    # bytes per parameter + the per-channel pool), what a bandwidth-bound ring would take, where the bytes are reduced, and --
    # whenever more than a tenth of the step is exposed communication, as over gloo here -- a bf16-transport leg beside it
    cb = d['comm_bytes_on_wire']
    assert cb['transport'] == 'fp32' and cb['per_step'] >= d['grad_bytes']['total'] * 0.99, cb
    assert d['expected_ring_ms'] > 0 and d['trainer_ms_per_step'] > 0 and d['trainer_loop']['async_metrics']['ms_per_step'] > 0
    leg = d['bf16_transport_leg']           # (decided on the torch.distributed leg's figures; --collective-ab may have replaced them)
    if leg is not None:
        assert leg['ms_per_step'] > 0 and leg['comm_bytes_on_wire_per_step'] < 0.75 * cb['per_step'], leg
    # the in-run A/B of the two collective paths: both ran (the native one on one-rank rehearsal communicators here -- RCCL
    # refuses two ranks of one communicator on one device), the line carries both timings and names the path it kept, and
    # the timed region ran on that path
    ab = d['collective_paths_ms']
    native, via_torch = 'w2l_rccl_* (C ABI)', 'torch.distributed'
    assert ab[via_torch] > 0 and ab[native] is not None and ab[native] > 0, ab
    assert ab['kept'] == (native if ab[native] < ab[via_torch] else via_torch) and 'rehearsal' in ab['note']
    assert d['collectives_via'] == ab['kept']
    # rank 0 tuned, rank 1 loaded its plans: one plan table on both ranks
    tp = d['tune_plans']
    assert tp['identical'] and len(tp['sha16_by_rank']) == 2 and len(set(tp['sha16_by_rank'])) == 1
    bad = subprocess.run(cmd, env=dict(env, RANK='0', WORLD_SIZE='3', MASTER_ADDR='127.0.0.1', MASTER_PORT='1'), cwd=root,
                         capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and 'must agree' in bad.stderr


@pytest.mark.parametrize('amsgrad', [False, True])
def test_fused_novograd_matches_torch_ops(amsgrad):
    """Novograd with the fused conv-weight path (w2l_novograd_pack) == the same optimizer restricted to torch ops (itself
    pinned to the reference's novograd.py by tests/golden/novograd_cases.npz), on real engine gradients; and the bf16
    operands it emits are the ones the next forward uses."""
    from oracle import w2l_oracle as O
    from wav2letter_pytorch_amd.novograd import Novograd
    layers = [(128, 11, 2, 1, 0.0), (128, 13, 1, 2, 0.0)]
    sd = O.init_wav2letter_state(layers, seed=5)
    ma = build_w2l(layers, sd, 'bf16').train()
    mb = build_w2l(layers, sd, 'bf16').train()
    kw = dict(lr=0.02, betas=(0.95, 0.5), weight_decay=1e-3, grad_averaging=True, amsgrad=amsgrad)
    oa, ob = Novograd(ma.parameters(), **kw), Novograd(mb.parameters(), **kw)
    ob.fused = False
    x, il, tg, tl = O.synthetic_batch(2, 160, seed=12, s_lo=5, s_hi=15)
    for it in range(3):
        for m, o in ((ma, oa), (mb, ob)):
            o.zero_grad(set_to_none=True)
            out, ol = m(x.cuda(), il)
            m.criterion(out.transpose(0, 1), tg, ol, tl).backward()
            o.step()
    for (k, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()):
        assert scale_err(pa.detach().cpu().numpy(), pb.detach().cpu().numpy()) < 3e-5, k
    wa = ma.conv1ds.conv1d_1.conv1.weight
    st = oa.state[wa]
    vb = float(ob.state[mb.conv1ds.conv1d_1.conv1.weight]['exp_avg_sq'])
    assert abs(float(st['exp_avg_sq']) - vb) <= 1e-5 * vb
    pk = wa._w2l_pack[False]
    assert pk.version == wa._version
    assert torch.equal(pk.fwd_hi, wa.detach().permute(2, 0, 1).to(torch.bfloat16))
    assert torch.equal(pk.dgr_hi, wa.detach().flip(2).permute(2, 1, 0).to(torch.bfloat16).contiguous())
    assert '_scratch' not in next(iter(oa.state_dict()['state'].values()))


def test_random_stacks_smoke(capsys):
    """a few cases of the randomised sweeps (tools/fuzz_model.py, tools/fuzz_jasper.py) as a regression check"""
    import importlib.util
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for name, argv in (('fuzz_model', ['x', '6', '7', 'fp32']), ('fuzz_jasper', ['x', '4', '7', 'fp32'])):
        spec = importlib.util.spec_from_file_location(name, os.path.join(root, 'tools', name + '.py'))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        old = sys.argv
        sys.argv = argv
        try:
            mod.main()
        finally:
            sys.argv = old
        out = capsys.readouterr().out
        assert out.count(' ok ') == int(argv[1]) and 'FAIL' not in out and 'EXCEPTION' not in out, out
