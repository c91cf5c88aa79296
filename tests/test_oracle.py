"""Pin the CPU oracle against golden vectors generated from the reference
(tests/golden/make_golden.py) and cross-check torch CPU ops against the
independent numpy restatements.  CPU only."""
import ast
import math
import os

import numpy as np
import pytest
import torch

from oracle import w2l_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def load(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=True)


def meta_of(z):
    return ast.literal_eval(str(z['meta']))


def sd_from(z, prefix='p0/'):
    return {k[len(prefix):]: torch.from_numpy(z[k].copy()) for k in z.files if k.startswith(prefix)}


@pytest.mark.parametrize('case', ['w2l_ml1', 'w2l_ml3', 'w2l_mix5'])
def test_w2l_end_to_end_matches_reference(case):
    z = load(case + '.npz')
    meta = meta_of(z)
    layers = [(l['output_size'], l['kernel_size'], l['stride'], l['dilation'], 0.0) for l in meta['layers']]
    sd = sd_from(z)
    res = O.wav2letter_step(torch.from_numpy(z['x']), torch.from_numpy(z['in_lens']), torch.from_numpy(z['targets']),
                            torch.from_numpy(z['target_lens']), sd, layers, want_input_grad=True)
    np.testing.assert_allclose(res['log_probs'].numpy(), z['log_probs'], rtol=1e-5, atol=1e-5)
    assert abs(float(res['loss']) - float(z['loss'])) < 1e-4 * max(1.0, abs(float(z['loss'])))
    np.testing.assert_array_equal(res['out_lens'].numpy(), z['out_lens'])
    np.testing.assert_allclose(res['input_grad'].numpy(), z['input_grad'], rtol=1e-4, atol=1e-6)
    for k, g in res['grads'].items():
        ref = z['g/' + k]
        scale = max(np.abs(ref).max(), 1e-6)
        assert np.abs(g.numpy() - ref).max() <= 2e-4 * scale + 2e-5, k
    for k in z.files:
        if k.startswith('p1/'):
            np.testing.assert_allclose(sd[k[3:]].numpy(), z[k], rtol=1e-5, atol=1e-6, err_msg=k)
    # greedy decode + metrics
    idx = O.argmax_lowest(res['log_probs'].numpy())
    np.testing.assert_array_equal(idx, z['argmax'])
    strings = O.greedy_strings(idx, z['out_lens'], O.ENGLISH_LOWERCASE)
    assert strings == list(z['decoded'])
    cs = [O.cer_ratio(e, p) for e, p in zip(z['texts'], strings)]
    ws = [O.wer_ratio(e, p) for e, p in zip(z['texts'], strings)]
    assert abs(sum(c[0] for c in cs) / sum(c[1] for c in cs) - float(z['cer'])) < 1e-12
    assert abs(sum(w[0] for w in ws) / sum(w[1] for w in ws) - float(z['wer'])) < 1e-12
    # eval mode (running stats)
    lp_eval, _ = O.wav2letter_forward(torch.from_numpy(z['x']), sd, layers, training=False)
    np.testing.assert_allclose(lp_eval.numpy(), z['out_eval'], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('case', ['jasper_sep2', 'jasper_dense', 'jasper_nomask'])
def test_jasper_end_to_end_matches_reference(case):
    z = load(case + '.npz')
    meta = meta_of(z)
    sd = sd_from(z)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()
              if v.dtype.is_floating_point and 'running_' not in k}
    work = dict(sd)
    work.update(params)
    x = torch.from_numpy(z['x']).requires_grad_(True)
    lp, out_lens = O.jasper_forward(x, torch.from_numpy(z['in_lens']), work, meta['blocks'], training=True)
    np.testing.assert_allclose(lp.detach().numpy(), z['log_probs'], rtol=1e-5, atol=1e-5)
    np.testing.assert_array_equal(out_lens.numpy(), z['out_lens'])
    loss = O.ctc_criterion(lp, torch.from_numpy(z['targets']), out_lens, torch.from_numpy(z['target_lens']))
    assert abs(float(loss) - float(z['loss'])) < 1e-4 * max(1.0, abs(float(z['loss'])))
    loss.backward()
    for k, p in params.items():
        ref = z['g/' + k]
        assert np.abs(p.grad.numpy() - ref).max() <= 2e-4 * max(np.abs(ref).max(), 1e-6) + 2e-5, k
    for k in z.files:
        if k.startswith('p1/'):
            np.testing.assert_allclose(work[k[3:]].numpy(), z[k], rtol=1e-5, atol=1e-6, err_msg=k)
    out_eval, _ = O.jasper_forward(torch.from_numpy(z['x']), torch.from_numpy(z['in_lens']), work, meta['blocks'],
                                   training=False)
    np.testing.assert_allclose(out_eval.detach().numpy(), z['out_eval'], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('case', ['jasper_sep2', 'jasper_dense', 'jasper_nomask'])
def test_jasper_step_matches_reference(case):
    """O.jasper_step (bench.py's CPU baseline for the Jasper workload) against the reference-generated fixtures: loss,
    log-probs, output lengths, every parameter gradient, the BatchNorm buffers after the step"""
    z = load(case + '.npz')
    meta = meta_of(z)
    sd = sd_from(z)
    r = O.jasper_step(torch.from_numpy(z['x']), torch.from_numpy(z['in_lens']), torch.from_numpy(z['targets']),
                      torch.from_numpy(z['target_lens']), sd, meta['blocks'])
    np.testing.assert_allclose(r['log_probs'].numpy(), z['log_probs'], rtol=1e-5, atol=1e-5)
    np.testing.assert_array_equal(r['out_lens'].numpy(), z['out_lens'])
    assert abs(float(r['loss']) - float(z['loss'])) < 1e-4 * max(1.0, abs(float(z['loss'])))
    for k, g in r['grads'].items():
        ref = z['g/' + k]
        assert np.abs(g.numpy() - ref).max() <= 2e-4 * max(np.abs(ref).max(), 1e-6) + 2e-5, k
    for k in z.files:
        if k.startswith('p1/') and 'running_' in k:
            np.testing.assert_allclose(sd[k[3:]].numpy(), z[k], rtol=1e-5, atol=1e-6, err_msg=k)


def test_conv1dblock_ops_numpy_restatement():
    z = load('ops_conv1dblock.npz')
    for tag in ['asym_s2', 'dil2', 'k1', 'even_k13']:
        cin, cout, k, s, d, T = [int(v) for v in z[f'{tag}/cfg']]
        pl, pr = O.conv1d_block_padding(cin, k, s, d)
        assert (pl, pr) == tuple(int(v) for v in z[f'{tag}/pad'])
        x = z[f'{tag}/x']
        w, b = z[f'{tag}/p/conv1.weight'], z[f'{tag}/p/conv1.bias']
        y = O.np_conv1d(O.np_reflect_pad(x, pl, pr), w, b, s, d)
        ybn, m, v, vu = O.np_batch_norm_train(y, z[f'{tag}/p/batch_norm.weight'], z[f'{tag}/p/batch_norm.bias'], 1e-3)
        out = np.clip(ybn, 0, 20)
        np.testing.assert_allclose(out, z[f'{tag}/y'], rtol=2e-4, atol=2e-4)
        np.testing.assert_allclose(0.1 * 0 + 0.9 * m, z[f'{tag}/running_mean'], rtol=1e-4, atol=1e-5)   # momentum .9
        np.testing.assert_allclose(0.1 * 1 + 0.9 * vu, z[f'{tag}/running_var'], rtol=1e-4, atol=1e-5)
        # and the torch-composed oracle block
        sd = {'conv1.weight': torch.from_numpy(w), 'conv1.bias': torch.from_numpy(b),
              'batch_norm.weight': torch.from_numpy(z[f'{tag}/p/batch_norm.weight']),
              'batch_norm.bias': torch.from_numpy(z[f'{tag}/p/batch_norm.bias']),
              'batch_norm.running_mean': torch.zeros(cout), 'batch_norm.running_var': torch.ones(cout)}
        yt = O.conv1d_block_forward(torch.from_numpy(x), sd, '', stride=s, dilation=d, bn=True, activation=True,
                                    training=True)
        np.testing.assert_allclose(yt.numpy(), z[f'{tag}/y'], rtol=1e-5, atol=1e-5)
    np.testing.assert_array_equal(z['clamp/grad'], [0, 1, 1, 1, 0])      # closed interval


def test_ctc_numpy_matches_reference_criterion():
    z = load('ctc_cases.npz')
    loss, nll, grad = O.np_ctc_mean(z['log_probs'], z['targets'], z['in_lens'], z['target_lens'])
    assert abs(loss - float(z['loss'])) < 1e-4
    np.testing.assert_allclose(nll, z['nll'], rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(grad, z['grad'], rtol=1e-4, atol=1e-6)
    assert nll[3] == 0 and nll[5] == 0          # zero_infinity cases
    assert np.all(grad[1, 33:] == 0)            # t >= input length


def test_greedy_known_answers():
    z = load('greedy_cases.npz')
    labels = list(z['labels'])
    idx = O.argmax_lowest(z['probs'])
    np.testing.assert_array_equal(idx, z['argmax'])
    assert O.greedy_strings(idx, z['sizes'], labels) == list(z['strings'])
    # unit_tests/decoder_test.py:40-42
    small = O.greedy_strings(O.argmax_lowest(np.array([[[0.8, 0.2, 0, 0], [0.6, 0.4, 0, 0]]])), None,
                             ['_', 'A', 'B', ' '])
    assert small == [''] == list(z['small'])
    for (a, b), c, w in zip(z['pairs'], z['cer'], z['wer']):
        assert O.cer_ratio(str(a), str(b)) == tuple(c)
        assert O.wer_ratio(str(a), str(b)) == tuple(w)


def test_padding_rule_full_table():
    cin = 64
    pads = []
    for (c, k, s, d, _) in O.W2L_LAYERS:
        pads.append(sum(O.conv1d_block_padding(cin, k, s, d)))
        cin = c
    assert pads == [9, 10, 10, 10, 12, 12, 12, 16, 16, 16, 20, 20, 20, 24, 24, 24, 56, 56, 56, 0]


# ---- feature front-end / augmentation oracle vs the reference-generated fixture (features.npz) ----
def test_features_oracle_matches_reference_fixture():
    import random
    from oracle import features_oracle as FO
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'features.npz'), allow_pickle=True)
    conf = dict(window='hamming', window_stride=0.01, window_size=0.02, sample_rate=16000)
    specs = []
    for i in range(int(z['n_cases'])):
        s = FO.extract(z[f'audio{i}'], z[f'noise{i}'], conf)
        assert s.shape == z[f'spect{i}'].shape
        assert np.abs(s - z[f'spect{i}']).max() < 5e-6
        specs.append(z[f'spect{i}'])
    x, il, tg, tl = FO.collate(specs, [list(t) for t in z['col_targets']])
    np.testing.assert_array_equal(x, z['col_inputs'])
    np.testing.assert_array_equal(il, z['col_il'])
    np.testing.assert_array_equal(tg, z['col_tg'])
    np.testing.assert_array_equal(tl, z['col_tl'])
    ax = z['aug_x']
    np.testing.assert_array_equal(FO.apply_rects(ax, FO.spec_augment_rects(ax.shape, random.Random(11), 2, 2, 15, 50)), z['specaug'])
    np.testing.assert_array_equal(FO.apply_rects(ax, FO.spec_cutout_rects(ax.shape, random.Random(12), 5, 60, 25)), z['speccut'])
    xs = z['aug_xs']
    np.testing.assert_array_equal(FO.apply_rects(xs, FO.spec_augment_rects(xs.shape, random.Random(13), 1, 2, 15, 50)),
                                  z['specaug_short'])


def test_product_mel_filterbank_equals_oracle_restatement():
    from oracle import features_oracle as FO
    from wav2letter_pytorch_amd.data.mel import mel_filterbank
    for sr, n_fft, n_mels in ((16000, 512, 64), (8000, 256, 40), (22050, 1024, 80)):
        a, b = mel_filterbank(sr, n_fft, n_mels, 0.0, sr / 2), FO.mel_filterbank(sr, n_fft, n_mels, 0.0, sr / 2)
        assert a.shape == (n_mels, n_fft // 2 + 1)
        assert np.abs(a - b).max() <= 1e-9
        assert (a >= 0).all() and (a.sum(1) > 0).all()


def test_mel_filterbank_published_known_answers():
    """librosa is absent from this image, so the mel matrix cannot be compared with a librosa output; what IS published:
    (1) the example in the docstring of ``librosa.filters.mel`` -- ``librosa.filters.mel(sr=22050, n_fft=2048)`` (128 bands)
    prints ``[[0.   , 0.016, ..., 0.   , 0.   ], ...`` -- i.e. M[0, 0] = 0 and M[0, 1] = 0.016 to three decimals;
    (2) the Slaney scale's defining constants: linear below 1 kHz at 200/3 Hz per mel (1000 Hz = 15 mel), logarithmic above
    with log(6.4)/27 per mel, so 6400 Hz = 42 mel (Auditory Toolbox, the ``htk=False`` default);
    (3) Slaney area normalisation: every filter integrates to 1 over frequency (sum * bin width, up to the sampling of the
    triangle on the FFT grid).  Checked for the oracle's restatement and for the product's."""
    from oracle import features_oracle as FO
    from wav2letter_pytorch_amd.data import mel as PM
    for build in (FO.mel_filterbank, PM.mel_filterbank):
        M = np.asarray(build(22050, 2048, 128, 0.0, 22050 / 2), dtype=np.float64)
        assert M.shape == (128, 1025)
        assert M[0, 0] == 0.0 and abs(M[0, 1] - 0.016) < 5e-4, M[0, :3]
        assert M[-1, -1] == 0.0 and (M[:, 0] == 0).all()
        # (3): wide filters (>= 8 bins) integrate to 1 within the triangle's sampling error
        df = 22050 / 2048
        for row in M:
            if (row > 0).sum() >= 8:
                assert abs(row.sum() * df - 1.0) < 0.02, row.sum() * df
    for h2m, m2h in ((FO._hz_to_mel_slaney, FO._mel_to_hz_slaney), (PM.hz_to_mel, PM.mel_to_hz)):
        assert abs(float(h2m(1000.0)) - 15.0) < 1e-9 and abs(float(h2m(6400.0)) - 42.0) < 1e-9
        assert abs(float(h2m(500.0)) - 7.5) < 1e-9
        assert abs(float(m2h(42.0)) - 6400.0) < 1e-6 and abs(float(m2h(3.0)) - 200.0) < 1e-9


def test_fp8_operand_model_known_answers():
    """oracle.e4m3_values / pow2_scale: the OCP e4m3 grid (1 + 3 mantissa bits, saturation at 448, subnormal step 2^-9),
    round-to-nearest-even, and the power-of-two scales of the product's fp8 mode"""
    import torch
    from oracle import w2l_oracle as O
    grid = torch.tensor([0.0, 2.0 ** -9, 0.0625, 1.0, 1.125, 1.875, 240.0, 448.0])
    assert torch.equal(O.e4m3_values(grid, 1.0), grid)                      # representable values come back unchanged
    assert torch.equal(O.e4m3_values(-grid, 1.0), -grid)
    v = torch.tensor([1.0625, 1.1875, 1.3125, 460.0, 1e4, -1e4, 2.0 ** -11])
    # ties go to the even mantissa: 1.0625 -> 1.0, 1.1875 -> 1.25, 1.3125 -> 1.25; beyond 448 saturates; half the smallest
    # subnormal (2^-10) ties to 0, 2^-11 is below it
    assert O.e4m3_values(v, 1.0).tolist() == [1.0, 1.25, 1.25, 448.0, 448.0, -448.0, 0.0]
    assert float(O.e4m3_values(torch.tensor([20.0]), 16.0)) == 20.0        # clamp's upper bound x 16 = 320 is on the grid
    for amax in (1e-4, 0.013, 1.0, 3.7, 447.0, 449.0):
        for top in (448.0, 224.0):
            s = O.pow2_scale(amax, top)
            assert amax * s <= top < 2 * amax * s and math.log2(s) == round(math.log2(s))


def test_bf16_operand_model_known_answers():
    """oracle.bf16_conv1d / conv1d_block_forward(stats_before_rounding): the bf16 operand model of the product's default
    mode rounds exactly where it says -- operands, the stored output, dy and dx -- and takes BatchNorm's batch statistics from
    the accumulators while normalising their bf16 rounding"""
    import torch
    import torch.nn.functional as F
    from oracle import w2l_oracle as O
    torch.manual_seed(0)
    x, w, b = torch.randn(2, 8, 40), torch.randn(6, 8, 3) * 0.3, torch.randn(6)
    xb, wb = x.bfloat16().float(), w.bfloat16().float()
    x1, w1, b1 = (t.clone().requires_grad_(True) for t in (x, w, b))
    y = O.bf16_conv1d(x1, w1, b1, stride=1, dilation=2)
    assert torch.equal(y, F.conv1d(xb, wb, b, dilation=2).bfloat16().float())
    g = torch.randn_like(y)
    y.backward(g)
    gb = g.bfloat16().float()
    assert torch.equal(x1.grad, torch.nn.grad.conv1d_input(x.shape, wb, gb, dilation=2).bfloat16().float())
    assert torch.equal(w1.grad, torch.nn.grad.conv1d_weight(xb, w.shape, gb, dilation=2))
    assert torch.equal(b1.grad, gb.sum((0, 2)))
    assert torch.equal(O.bf16_head_conv1d(x, w, b, dilation=2), F.conv1d(xb, wb, b, dilation=2))      # logits stay fp32
    # one block: statistics from the accumulators, the normalised tensor is their rounding
    sd = {'c.conv1.weight': w, 'c.conv1.bias': b, 'c.batch_norm.weight': torch.rand(6) + 0.5, 'c.batch_norm.bias': torch.randn(6),
          'c.batch_norm.running_mean': torch.zeros(6), 'c.batch_norm.running_var': torch.ones(6)}
    got = O.conv1d_block_forward(x, {k: v.clone() for k, v in sd.items()}, 'c.', stride=1, dilation=1, bn=True, activation=False, training=True,
                                 conv=O.bf16_head_conv1d, stats_before_rounding=True)
    xp = F.pad(x, (1, 1), mode='reflect')
    acc = F.conv1d(xp.bfloat16().float(), wb, b)
    mean, var = acc.mean((0, 2)), acc.var((0, 2), unbiased=False)
    want = ((acc.bfloat16().float() - mean[None, :, None]) * torch.rsqrt(var + 1e-3)[None, :, None]
            * sd['c.batch_norm.weight'][None, :, None] + sd['c.batch_norm.bias'][None, :, None])
    assert torch.equal(got, want)
    # the running statistics move like nn.BatchNorm1d(momentum=0.9)'s
    ref_sd, mod_sd = ({k: v.clone() for k, v in sd.items()} for _ in range(2))      # F.batch_norm updates its buffers in place
    O.conv1d_block_forward(xb, ref_sd, 'c.', stride=1, dilation=1, bn=True, activation=False, training=True,
                           conv=lambda x_, w_, b_, **kw: F.conv1d(x_, wb, b_, **kw))
    O.conv1d_block_forward(x, mod_sd, 'c.', stride=1, dilation=1, bn=True, activation=False, training=True,
                           conv=O.bf16_head_conv1d, stats_before_rounding=True)
    for k in ('c.batch_norm.running_mean', 'c.batch_norm.running_var'):
        assert torch.allclose(ref_sd[k], mod_sd[k], rtol=1e-5, atol=1e-6), k


def test_bf16_model_deviation_is_the_arithmetic():
    """The bf16 operand model against the fp32 evaluation of the same small network, both on the CPU with the same clamp
    gates: the loss agrees to 1e-3 but the gradients differ by percents in the L2 norm although nothing but bf16 storage
    separates the two -- what tests/test_gpu_fullsize.py::test_w2l_full_table_bf16_vs_operand_model then measures the device
    against."""
    import torch
    from oracle import w2l_oracle as O
    layers = [(128, 11, 2, 1, 0.0), (128, 13, 1, 1, 0.0), (256, 5, 1, 2, 0.0)]
    sd = O.init_wav2letter_state(layers, seed=1)
    x, il, tg, tl = O.synthetic_batch(2, 120, seed=1, s_lo=5, s_hi=15)
    a = O.wav2letter_step(x, il, tg, tl, {k: v.clone() for k, v in sd.items()}, layers)
    gates = [(v > 0) & (v < 20) for v in a['activations']]
    a = O.wav2letter_step(x, il, tg, tl, {k: v.clone() for k, v in sd.items()}, layers, gates=gates)
    b = O.wav2letter_step(x, il, tg, tl, {k: v.clone() for k, v in sd.items()}, layers, gates=gates, bf16_model=True)
    assert abs(float(a['loss']) - float(b['loss'])) < 1e-3 * float(a['loss'])
    rel = {k: float((a['grads'][k] - b['grads'][k]).norm() / a['grads'][k].norm().clamp_min(1e-20))
           for k in a['grads'] if k.endswith('conv1.weight')}
    assert all(2e-3 < rel[f'conv1ds.conv1d_{i}.conv1.weight'] < 0.1 for i in range(3)), rel     # ~1e-2 after three layers
    assert rel['conv1ds.conv1d_3.conv1.weight'] < 5e-3, rel                                     # the classifier: 1e-3
    for k in ('running_mean', 'running_var'):                 # the model's buffers follow the fp32 path's
        ka = f'conv1ds.conv1d_1.batch_norm.{k}'
        sa, sb = {kk: v.clone() for kk, v in sd.items()}, {kk: v.clone() for kk, v in sd.items()}
        O.wav2letter_step(x, il, tg, tl, sa, layers, gates=gates)
        O.wav2letter_step(x, il, tg, tl, sb, layers, gates=gates, bf16_model=True)
        assert torch.allclose(sa[ka], sb[ka], rtol=2e-2, atol=2e-3), ka


def test_fp8_model_deviation_is_the_arithmetic():
    """The e4m3 operand model against the fp32 evaluation of the SAME small network, both on the CPU in float32 with the
    same clamp gates: gradients differ by tens of percent in the L2 norm although nothing but operand rounding separates
    them -- the fp8 tests therefore hold the device to the e4m3 model, and quote the distance to fp32 as a property of the
    arithmetic.  With the model's gradients left in bf16 operands (FP8_MODEL off) only the forward quantisation remains."""
    import torch
    from oracle import w2l_oracle as O
    layers = [(128, 11, 2, 1, 0.0), (128, 13, 1, 1, 0.0), (256, 5, 1, 2, 0.0)]
    sd = O.init_wav2letter_state(layers, seed=1)
    x, il, tg, tl = O.synthetic_batch(2, 120, seed=1, s_lo=5, s_hi=15)
    a = O.wav2letter_step(x, il, tg, tl, {k: v.clone() for k, v in sd.items()}, layers)
    gates = [(v > 0) & (v < 20) for v in a['activations']]
    a = O.wav2letter_step(x, il, tg, tl, {k: v.clone() for k, v in sd.items()}, layers, gates=gates)
    b = O.wav2letter_step(x, il, tg, tl, {k: v.clone() for k, v in sd.items()}, layers, gates=gates, fp8_layers=(1, 2))
    assert abs(float(a['loss']) - float(b['loss'])) < 1e-2 * float(a['loss'])
    rel = {k: float((a['grads'][k] - b['grads'][k]).norm() / a['grads'][k].norm().clamp_min(1e-20))
           for k in a['grads'] if k.endswith('conv1.weight')}
    assert 0.02 < rel['conv1ds.conv1d_1.conv1.weight'] < 0.6, rel        # percent-level to tens of percent: operand rounding
    assert rel['conv1ds.conv1d_3.conv1.weight'] < 0.05, rel              # the classifier sits above every e4m3 layer's backward
    # layers outside fp8_layers are untouched: with no e4m3 layer the two evaluations are identical
    c = O.wav2letter_step(x, il, tg, tl, {k: v.clone() for k, v in sd.items()}, layers, gates=gates, fp8_layers=())
    assert all(torch.equal(a['grads'][k], c['grads'][k]) for k in a['grads'])
