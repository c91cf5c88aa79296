"""Generate golden vectors by importing the REFERENCE (read-only at /root/reference).

Run in the build container only:  python -B tests/golden/make_golden.py
Outputs small .npz fixtures next to this file.  Only data (inputs, parameters by
state-dict key, outputs) is written -- never reference source.  The GPU box has
no /root/reference; tests read the .npz files only.

Stub recipe: pytorch_lightning / hydra / librosa / soundfile /
Levenshtein are absent in the image, so minimal stand-ins are registered in
sys.modules before the import; ``data`` is pre-registered as a bare package so
that only data/label_sets.py is loaded (data/__init__ pulls data_loader, which
needs a scipy function removed from modern scipy).
"""
import importlib
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True


def _install_stubs():
    ptl = types.ModuleType('pytorch_lightning')

    class LightningModule(nn.Module):
        def log_dict(self, d, *a, **k):
            self._last_logs = dict(d)

        def optimizers(self):
            return self._optim

    ptl.LightningModule = LightningModule
    sys.modules['pytorch_lightning'] = ptl

    hydra = types.ModuleType('hydra')
    hutils = types.ModuleType('hydra.utils')

    def instantiate(cfg, **kw):
        cfg = dict(cfg)
        target = cfg.pop('_target_')
        mod, name = target.rsplit('.', 1)
        cfg.update(kw)
        return getattr(importlib.import_module(mod), name)(**cfg)

    hutils.instantiate = instantiate
    hydra.utils = hutils
    sys.modules['hydra'] = hydra
    sys.modules['hydra.utils'] = hutils
    for name in ('librosa', 'soundfile'):
        sys.modules[name] = types.ModuleType(name)
    lev = types.ModuleType('Levenshtein')

    def distance(a, b):
        prev = list(range(len(b) + 1))
        for i, ca in enumerate(a, 1):
            cur = [i]
            for j, cb in enumerate(b, 1):
                cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
            prev = cur
        return prev[-1]

    lev.distance = distance
    sys.modules['Levenshtein'] = lev
    data = types.ModuleType('data')
    data.__path__ = [os.path.join(REF, 'data')]
    sys.modules['data'] = data
    sys.path.insert(0, REF)
    importlib.import_module('data.label_sets')


class Cfg(dict):
    """dict with attribute access + .get; lists stay lists (slicing works)."""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError:
            raise AttributeError(k)
        return v

    def __setattr__(self, k, v):
        self[k] = v


def to_cfg(o):
    if isinstance(o, dict):
        return Cfg({k: to_cfg(v) for k, v in o.items()})
    if isinstance(o, list):
        return [to_cfg(v) for v in o]
    return o


W2L_TABLE = ([(256, 11, 2, 1)] + [(256, 11, 1, 1)] * 3 + [(384, 13, 1, 1)] * 3 + [(512, 17, 1, 1)] * 3
             + [(640, 21, 1, 1)] * 3 + [(768, 25, 1, 1)] * 3 + [(896, 29, 1, 2)] * 3 + [(1024, 1, 1, 1)])


def model_cfg(name, labels, **extra):
    base = dict(name=name, input_size=64, labels=labels,
                audio_conf=dict(window='hamming', window_stride=0.01, window_size=0.02, sample_rate=16000),
                decoder=dict(_target_='decoder.GreedyDecoder', labels=labels),
                optimizer=dict(_target_='torch.optim.SGD', lr=1e-5, momentum=0.9, nesterov=True, weight_decay=1e-5),
                scheduler=dict(_target_='torch.optim.lr_scheduler.ExponentialLR', gamma=0.999))
    base.update(extra)
    return to_cfg(base)


def np_sd(sd):
    return {k: v.detach().cpu().numpy().copy() for k, v in sd.items()}


def batch(N, T, seed, ragged, scaling=2, smax_cap=None):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, 64, T, generator=g)
    if ragged:
        in_lens = torch.randint(max(T // 2, 8), T + 1, (N,), generator=g, dtype=torch.int32)
        in_lens[0] = T
    else:
        in_lens = torch.full((N,), T, dtype=torch.int32)
    hi = max(2, (T // scaling) // 3)
    tl = torch.randint(1, hi + 1, (N,), generator=g, dtype=torch.int32)
    tg = torch.randint(1, 29, (N, int(tl.max())), generator=g, dtype=torch.int32)
    for n in range(N):
        tg[n, int(tl[n]):] = 0
        x[n, :, int(in_lens[n]):] = 0          # _collator zero-pads (data_loader.py:153-156)
    return x, in_lens, tg, tl


def run_model_case(model, x, in_lens, tg, tl, labels, fname, extra_meta):
    model.train()
    sd0 = np_sd(model.state_dict())
    xin = x.clone().requires_grad_(True)
    out, out_lens = model(xin, in_lens)
    loss = model.criterion(out.transpose(0, 1), tg, out_lens, tl)
    loss.backward()
    grads = {k: p.grad.detach().numpy() for k, p in model.named_parameters()}
    sd1 = np_sd(model.state_dict())
    texts = []
    for n in range(x.shape[0]):
        texts.append(''.join(labels[int(i)] for i in tg[n, :int(tl[n])]))
    decoded = model.ctc_decoder.decode(out.detach(), out_lens)
    metrics = model.add_string_metrics(out.detach(), out_lens, texts, 'train')
    _, argmax = torch.max(out.detach(), 2)
    model.eval()
    with torch.no_grad():
        out_eval, _ = model(x, in_lens)
    # training_step through the stubbed Lightning surface (base_asr_models.py:78-85)
    model.train()
    save = dict(
        x=x.numpy(), in_lens=in_lens.numpy(), targets=tg.numpy(), target_lens=tl.numpy(),
        log_probs=out.detach().numpy(), out_lens=np.asarray(out_lens), loss=np.float64(loss.item()),
        input_grad=xin.grad.numpy(), argmax=argmax.numpy(), out_eval=out_eval.numpy(),
        decoded=np.array(decoded), texts=np.array(texts),
        cer=np.float64(metrics['train_cer']), wer=np.float64(metrics['train_wer']),
        len_ratio=np.float64(metrics['train_len_ratio']),
        meta=np.array(repr(dict(torch=torch.__version__, **extra_meta))),
    )
    for k, v in sd0.items():
        save['p0/' + k] = v
    for k, v in sd1.items():
        if 'running_' in k or 'num_batches' in k:
            save['p1/' + k] = v
    for k, v in grads.items():
        save['g/' + k] = v
    np.savez_compressed(os.path.join(HERE, fname), **save)
    print(fname, 'loss', loss.item(), 'decoded0', repr(decoded[0][:30]))


def main():
    _install_stubs()
    from wav2letter import Wav2Letter, Conv1dBlock
    from jasper import Jasper
    import decoder as ref_decoder
    from data import label_sets
    labels = label_sets.labels_map['english_lowercase']

    # ---- end-to-end tiny Wav2Letter (dropout forced to 0 for parity) ----
    def w2l_layers(rows, chans=None):
        out = []
        for i, r in enumerate(rows):
            c, k, s, d = W2L_TABLE[r]
            if chans is not None:
                c = chans[i]
            out.append(dict(output_size=c, kernel_size=k, stride=s, dilation=d, dropout=0.0))
        return out

    cases = [
        ('w2l_ml1', w2l_layers([0]), 1, 3, 100, False, 11),
        ('w2l_ml3', w2l_layers([0, 1, 4], chans=[96, 96, 128]), 3, 3, 137, True, 12),
        # kernel/dilation variety of the full table at reduced width: k11 s2, k13, k17, k29 d2, k1
        ('w2l_mix5', w2l_layers([0, 4, 7, 16, 19], chans=[64, 96, 64, 128, 160]), 5, 2, 180, True, 13),
    ]
    for name, layers, ml, N, T, ragged, seed in cases:
        torch.manual_seed(seed)
        cfg = model_cfg('wav2letter', labels, mid_layers=ml, layers=layers)
        model = Wav2Letter(cfg)
        x, il, tg, tl = batch(N, T, seed + 100, ragged)
        run_model_case(model, x, il, tg, tl, labels, f'{name}.npz',
                       dict(case=name, layers=layers, mid_layers=ml, seed=seed))

    # ---- end-to-end tiny Jasper ----
    jcases = [
        ('jasper_sep2', [dict(layer_size=64, kernel_size=32, stride=2, residual=False, separable=True),
                         dict(layer_size=96, kernel_size=38, stride=1, residual=True, separable=True)], 2, 3, 161, 21),
        ('jasper_dense', [dict(layer_size=48, kernel_size=11, stride=2, residual=False, separable=False),
                          dict(layer_size=64, kernel_size=13, stride=1, residual=True, separable=False, repeat=2),
                          dict(layer_size=64, kernel_size=29, stride=1, dilation=2, residual=True, separable=False,
                               repeat=2)], 3, 2, 150, 22),
    ]
    for name, blocks, ml, N, T, seed in jcases:
        torch.manual_seed(seed)
        cfg = model_cfg('jasper', labels, mid_layers=ml, jasper_blocks=blocks)
        model = Jasper(cfg)
        x, il, tg, tl = batch(N, T, seed + 100, True)
        il[1] = 2 * (T // 4) + 1          # an odd ragged length exercises the float length update (jasper.py:109-112)
        x[1, :, int(il[1]):] = 0
        run_model_case(model, x, il, tg, tl, labels, f'{name}.npz', dict(case=name, blocks=blocks, mid_layers=ml, seed=seed))

    # ---- per-op: Conv1dBlock variants (wav2letter.py:12-47) ----
    ops = {}
    torch.manual_seed(31)
    for tag, (cin, cout, k, s, d, T) in dict(asym_s2=(64, 32, 11, 2, 1, 61), dil2=(32, 48, 29, 1, 2, 70),
                                             k1=(48, 40, 1, 1, 1, 33), even_k13=(32, 32, 13, 1, 1, 50)).items():
        blk = Conv1dBlock(cin, cout, (k,), s, drop_out_prob=0.0, dilation=d)
        blk.train()
        with torch.no_grad():
            blk.batch_norm.weight.uniform_(0.5, 1.5)
            blk.batch_norm.bias.uniform_(-0.5, 0.5)
        x = (torch.randn(3, cin, T) * 2).requires_grad_(True)
        y = blk(x)
        gy = torch.randn_like(y)
        y.backward(gy)
        ops[f'{tag}/x'] = x.detach().numpy()
        ops[f'{tag}/y'] = y.detach().numpy()
        ops[f'{tag}/gy'] = gy.numpy()
        ops[f'{tag}/gx'] = x.grad.numpy()
        ops[f'{tag}/cfg'] = np.array([cin, cout, k, s, d, T])
        ops[f'{tag}/pad'] = np.array(blk.paddingAdded.padding if hasattr(blk.paddingAdded, 'padding') else (0, 0))
        for kname, p in blk.named_parameters():
            ops[f'{tag}/p/{kname}'] = p.detach().numpy()
            ops[f'{tag}/g/{kname}'] = p.grad.numpy()
        ops[f'{tag}/running_mean'] = blk.batch_norm.running_mean.numpy()
        ops[f'{tag}/running_var'] = blk.batch_norm.running_var.numpy()
    # clamp edge gradient (closed interval) -- wav2letter.py:46
    e = torch.tensor([-1.0, 0.0, 5.0, 20.0, 21.0], requires_grad=True)
    torch.clamp(e, min=0, max=20).sum().backward()
    ops['clamp/in'] = e.detach().numpy()
    ops['clamp/grad'] = e.grad.numpy()
    np.savez_compressed(os.path.join(HERE, 'ops_conv1dblock.npz'), **ops)
    print('ops_conv1dblock.npz', {k: v.tolist() for k, v in ops.items() if k.endswith('/pad')}, ops['clamp/grad'])

    # ---- CTC through the reference's criterion object (base_asr_models.py:23) ----
    torch.manual_seed(41)
    cfg = model_cfg('wav2letter', labels, mid_layers=1, layers=w2l_layers([0]))
    crit = Wav2Letter(cfg).criterion
    ctc = {}
    T, N, C = 40, 6, 29
    lp = torch.log_softmax(torch.randn(N, T, C) * 2, dim=-1).requires_grad_(True)
    in_l = torch.tensor([40, 33, 40, 12, 40, 7], dtype=torch.int32)
    tg = torch.zeros(N, 12, dtype=torch.int32)
    tl = torch.tensor([10, 8, 0, 12, 5, 6], dtype=torch.int32)      # n=2 empty target; n=3 infeasible (12 > 12 frames w/ repeats)
    g = torch.Generator().manual_seed(5)
    for n in range(N):
        tg[n, :int(tl[n])] = torch.randint(1, C, (int(tl[n]),), generator=g, dtype=torch.int32)
    tg[0, :10] = torch.tensor([3, 3, 3, 7, 7, 1, 2, 2, 9, 9], dtype=torch.int32)       # repeated labels
    tg[3, :12] = torch.tensor([4] * 12, dtype=torch.int32)                                # needs 23 frames, has 12 -> inf
    tg[5, :6] = torch.tensor([5, 5, 6, 6, 5, 5], dtype=torch.int32)                       # needs 9 frames, has 7 -> inf
    loss = crit(lp.transpose(0, 1), tg, in_l, tl)
    loss.backward()
    per = torch.nn.functional.ctc_loss(lp.detach().transpose(0, 1), tg, in_l, tl, blank=0, reduction='none',
                                       zero_infinity=True)
    ctc.update(log_probs=lp.detach().numpy(), in_lens=in_l.numpy(), targets=tg.numpy(), target_lens=tl.numpy(),
               loss=np.float64(loss.item()), grad=lp.grad.numpy(), nll=per.numpy())
    np.savez_compressed(os.path.join(HERE, 'ctc_cases.npz'), **ctc)
    print('ctc_cases.npz loss', loss.item(), 'nll', per.numpy())

    # ---- greedy decoder known answers (unit_tests/decoder_test.py:40-42, decoder.py:305-311) ----
    dec = ref_decoder.GreedyDecoder(labels, blank_index=0)
    torch.manual_seed(51)
    probs = torch.softmax(torch.randn(4, 60, 29) * 3, dim=-1)
    probs[0, 10:14] = probs[0, 10:11]            # repeated frames
    probs[1, :, 0] += 0.2                        # blank-heavy
    probs[2, 5, :] = 1.0 / 29                    # exact tie -> lowest index
    sizes = torch.tensor([60, 41, 60, 1])
    strings, offsets = dec.decode(probs, sizes, return_offsets=True)
    _, am = torch.max(probs, 2)
    small = ref_decoder.GreedyDecoder(['_', 'A', 'B', ' '], blank_index=0).decode(
        torch.FloatTensor([[0.8, 0.2, 0, 0], [0.6, 0.4, 0, 0]]).unsqueeze(0), sizes=None)
    pairs = [('the cat sat', 'the cat sat'), ('the cat sat', 'the bat sat on'), ('hello world', ''),
             ('a b c', 'abc'), ("it's", 'its')]
    cer = [dec.cer_ratio(a, b) for a, b in pairs]
    wer = [dec.wer_ratio(a, b) for a, b in pairs]
    np.savez_compressed(os.path.join(HERE, 'greedy_cases.npz'), probs=probs.numpy(), sizes=sizes.numpy(),
                        strings=np.array(strings), argmax=am.numpy(),
                        offsets=np.array([o[0].numpy().tolist() for o in offsets], dtype=object),
                        small=np.array(small), pairs=np.array(pairs), cer=np.array(cer), wer=np.array(wer),
                        labels=np.array(labels))
    print('greedy_cases.npz', strings, small)


def novograd_fixture():
    """4 steps of the reference Novograd (novograd.py:52-114) on two small tensors, both amsgrad settings"""
    _install_stubs()
    from novograd import Novograd
    out = {}
    for tag, kw in dict(plain=dict(lr=0.01, betas=(0.95, 0.5), weight_decay=1e-3, grad_averaging=True),
                        ams=dict(lr=0.02, betas=(0.9, 0.25), weight_decay=0.0, grad_averaging=False, amsgrad=True)).items():
        g = torch.Generator().manual_seed(7)
        ps = [torch.nn.Parameter(torch.randn(4, 3, generator=g)), torch.nn.Parameter(torch.randn(5, generator=g))]
        out[f'{tag}/p0_0'], out[f'{tag}/p0_1'] = ps[0].detach().numpy().copy(), ps[1].detach().numpy().copy()
        opt = Novograd(ps, **kw)
        grads = []
        for it in range(4):
            gs = [torch.randn(4, 3, generator=g) * (it + 1), torch.randn(5, generator=g)]
            grads.append([x.numpy().copy() for x in gs])
            for q, x in zip(ps, gs):
                q.grad = x.clone()
            opt.step()
        out[f'{tag}/grads0'] = np.stack([x[0] for x in grads])
        out[f'{tag}/grads1'] = np.stack([x[1] for x in grads])
        out[f'{tag}/p4_0'], out[f'{tag}/p4_1'] = ps[0].detach().numpy().copy(), ps[1].detach().numpy().copy()
        out[f'{tag}/v_0'] = opt.state[ps[0]]['exp_avg_sq'].numpy().copy()
    np.savez_compressed(os.path.join(HERE, 'novograd_cases.npz'), **out)
    print('novograd_cases.npz', out['plain/p4_1'])


def features_fixture():
    """Feature front-end, collate and augmentation vectors from the reference's own data/data_loader.py and
    data/augmentations.py.  data_loader.py needs three things this image lacks: librosa (absent), soundfile (absent; not
    used by the functions called here) and the scipy.signal window aliases removed from modern scipy (module-level dict,
    data_loader.py:18; aliased to scipy.signal.windows.* -- never called).  librosa.filters.mel is given the oracle's
    restatement of librosa's published algorithm (oracle/features_oracle.py), so the fixture pins everything EXCEPT the
    mel matrix itself, which is saved alongside."""
    import random
    import scipy.signal
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import features_oracle as FO
    _install_stubs()
    for w in ('hamming', 'hann', 'blackman', 'bartlett'):
        if not hasattr(scipy.signal, w):
            setattr(scipy.signal, w, getattr(scipy.signal.windows, w))
    librosa = sys.modules['librosa']
    filt = types.ModuleType('librosa.filters')
    filt.mel = lambda sr, n_fft, n_mels, fmin, fmax: FO.mel_filterbank(sr, n_fft, n_mels, fmin, fmax)
    librosa.filters = filt
    sys.modules['librosa.filters'] = filt
    sys.modules.pop('data', None)
    data = types.ModuleType('data')
    data.__path__ = [os.path.join(REF, 'data')]
    sys.modules['data'] = data
    dl = importlib.import_module('data.data_loader')
    aug = importlib.import_module('data.augmentations')
    conf = Cfg(window='hamming', window_stride=0.01, window_size=0.02, sample_rate=16000)
    ext = dl.SpectrogramExtractor(conf, mel_spec=64)
    out = {'fb': ext.fb[0].numpy().copy(), 'window': ext.window.numpy().copy(), 'n_fft': np.array(ext.n_fft)}
    g = np.random.default_rng(5)
    specs = []
    for i, L in enumerate((4000, 16000, 1237, 9613)):
        t = np.arange(L) / 16000.0
        audio = (0.3 * np.sin(2 * np.pi * (200 + 150 * i) * t * (1 + 0.5 * t)) + 0.05 * g.standard_normal(L)).astype(np.float32)
        torch.manual_seed(100 + i)
        noise = torch.randn(audio.shape).numpy().copy()
        torch.manual_seed(100 + i)
        spect = ext.extract(audio)
        spect = spect.numpy() if torch.is_tensor(spect) else np.asarray(spect)
        out[f'audio{i}'], out[f'noise{i}'], out[f'spect{i}'] = audio, noise, spect.astype(np.float32)
        specs.append(spect.astype(np.float32))
    out['n_cases'] = np.array(4)
    # _collator (data_loader.py:149-158)
    targets = [[3, 5, 7], [1], [2, 2, 9, 28, 4], [6, 6]]
    batch_ = [(specs[i], targets[i], 'f%d.wav' % i, 't%d' % i) for i in range(4)]
    inputs, il, tg, tl, paths, texts = dl._collator(batch_)
    out['col_inputs'], out['col_il'], out['col_tg'], out['col_tl'] = inputs.numpy(), il.numpy(), tg.numpy(), tl.numpy()
    out['col_targets'] = np.array(targets, dtype=object)
    # SpecAugment / SpecCutout (augmentations.py) with a seeded random.Random
    gx = torch.Generator().manual_seed(3)
    x = torch.randn(3, 64, 211, generator=gx)
    out['aug_x'] = x.numpy().copy()
    out['specaug'] = aug.SpecAugment(freq_masks=2, time_masks=2, freq_width=15, time_width=50, rng=random.Random(11))(x).numpy()
    out['speccut'] = aug.SpecCutout(rect_masks=5, rect_time=60, rect_freq=25, rng=random.Random(12))(x).numpy()
    xs = torch.randn(2, 64, 37, generator=gx)          # shorter than time_width: negative lefts, Python slice semantics
    out['aug_xs'] = xs.numpy().copy()
    out['specaug_short'] = aug.SpecAugment(freq_masks=1, time_masks=2, freq_width=15, time_width=50, rng=random.Random(13))(xs).numpy()
    np.savez_compressed(os.path.join(HERE, 'features.npz'), **out)
    print('features.npz', [out[f'spect{i}'].shape for i in range(4)], float(np.abs(out['spect1']).max()))


def jasper_nomask_fixture():
    """Jasper blocks with ``conv_mask: False`` (jasper.py:446 -> plain nn.Conv1d, :288-298; un-masked loop :393-397)
    between masked ones: masked stride-2 prologue, an UN-masked residual block with two repeats (sees the prologue's
    un-masked output, passes the lengths through), a masked residual block behind it, and an un-masked last block.
    State-dict keys of the plain convs carry no ``.conv``.  (An un-masked STRIDED block leaves the lengths un-halved and the
    reference's CTC call then fails -- input_lengths > T' -- so the strided block stays masked.)"""
    _install_stubs()
    from jasper import Jasper
    from data import label_sets
    labels = label_sets.labels_map['english_lowercase']
    blocks = [dict(layer_size=64, kernel_size=11, stride=2, residual=False, separable=False),
              dict(layer_size=64, kernel_size=13, stride=1, residual=True, separable=False, repeat=2, conv_mask=False),
              dict(layer_size=128, kernel_size=5, stride=1, residual=True, separable=False, repeat=2),
              dict(layer_size=128, kernel_size=7, stride=1, residual=True, separable=False, conv_mask=False)]
    torch.manual_seed(23)
    model = Jasper(model_cfg('jasper', labels, mid_layers=4, jasper_blocks=blocks))
    x, il, tg, tl = batch(3, 150, 123, True, smax_cap=None)
    il[1] = 61                                   # shorter than T' = 75: the masked block really masks
    x[1, :, 61:] = 0
    tl = torch.minimum(tl, torch.tensor([20, 20, 20], dtype=torch.int32))
    for n in range(3):
        tg[n, int(tl[n]):] = 0
    run_model_case(model, x, il, tg, tl, labels, 'jasper_nomask.npz', dict(case='jasper_nomask', blocks=blocks, mid_layers=4, seed=23))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'jasper_nomask':
        jasper_nomask_fixture()
    elif len(sys.argv) > 1 and sys.argv[1] == 'novograd':
        novograd_fixture()
    elif len(sys.argv) > 1 and sys.argv[1] == 'features':
        features_fixture()
    else:
        main()
        novograd_fixture()
