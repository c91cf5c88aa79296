"""Built-in configuration: the hyper-parameters of the reference's config tree (configuration/config.yaml:1-28,
configuration/model/wav2letter.yaml:1-104, configuration/model/jasper.yaml:1-78, configuration/audio/standard_16k.yaml,
configuration/optimizer/exp_lr_optimizer.yaml:2-10) as Python data, so that the package trains without a checkout of
the reference next to it.  ``train.py --config-dir`` loads a YAML tree instead (config.load_config)."""
from __future__ import annotations

from .config import to_cfg
from .data import label_sets

# (output_size, kernel_size, stride, dilation, dropout) -- configuration/model/wav2letter.yaml:5-104
W2L_LAYERS = ([(256, 11, 2, 1, 0.2)] + [(256, 11, 1, 1, 0.2)] * 3 + [(384, 13, 1, 1, 0.2)] * 3 + [(512, 17, 1, 1, 0.2)] * 3
              + [(640, 21, 1, 1, 0.3)] * 3 + [(768, 25, 1, 1, 0.3)] * 3 + [(896, 29, 1, 2, 0.4)] * 3 + [(1024, 1, 1, 1, 0.4)])
# (layer_size, kernel_size, stride, residual, separable) -- configuration/model/jasper.yaml:4-78
JASPER_BLOCKS = ([(256, 32, 2, False, True)] + [(256, 32, 1, True, True)] * 3 + [(256, 38, 1, True, True)] * 3
                 + [(512, 50, 1, True, True)] * 3 + [(512, 62, 1, True, True)] * 3 + [(512, 74, 1, True, True)]
                 + [(1024, 1, 1, False, False)])

AUDIO_CONF = dict(window='hamming', window_stride=0.01, window_size=0.02, sample_rate=16000)
OPTIMIZER = dict(_target_='torch.optim.SGD', lr=1e-5, momentum=0.9, nesterov=True, weight_decay=1e-5)
SCHEDULER = dict(_target_='torch.optim.lr_scheduler.ExponentialLR', gamma=0.999)


def _common(labels):
    if isinstance(labels, str):
        labels = list(label_sets.labels_map[labels])
    return dict(input_size=64, labels=labels, audio_conf=dict(AUDIO_CONF),
                decoder=dict(_target_='decoder.GreedyDecoder', labels=labels), optimizer=dict(OPTIMIZER),
                scheduler=dict(SCHEDULER))


def wav2letter_model(mid_layers: int = 1, dropout: bool = True, labels='english_lowercase', precision: str = 'bf16'):
    """cfg.model of the Wav2Letter stack; ``mid_layers`` defaults to the yaml's 1, 20 = the whole table"""
    layers = [dict(output_size=c, kernel_size=k, stride=s, dilation=d, dropout=(p if dropout else 0.0))
              for c, k, s, d, p in W2L_LAYERS]
    return to_cfg(dict(name='wav2letter', mid_layers=mid_layers, layers=layers, precision=precision, **_common(labels)))


def jasper_model(mid_layers: int = 1, labels='english_lowercase', precision: str = 'bf16'):
    blocks = [dict(layer_size=c, kernel_size=k, stride=s, residual=r, separable=sep) for c, k, s, r, sep in JASPER_BLOCKS]
    return to_cfg(dict(name='jasper', mid_layers=mid_layers, jasper_blocks=blocks, precision=precision, **_common(labels)))


def jasper10x5_model(labels='english_lowercase', precision: str = 'bf16'):
    """Jasper 10x5 (BASELINE config 4) through the reference's own jasper_blocks keys: a stride-2 prologue, 10 dense
    residual blocks of 5 repeats, a dilated block and a 1x1 block -- 13 blocks, 322 M parameters."""
    blocks = [dict(layer_size=256, kernel_size=11, stride=2, residual=False, separable=False, repeat=1)]
    for c, k in ((256, 11), (384, 13), (512, 17), (640, 21), (768, 25)):
        blocks += [dict(layer_size=c, kernel_size=k, stride=1, residual=True, separable=False, repeat=5)] * 2
    blocks += [dict(layer_size=896, kernel_size=29, stride=1, dilation=2, residual=False, separable=False, repeat=1),
               dict(layer_size=1024, kernel_size=1, stride=1, residual=False, separable=False, repeat=1)]
    return to_cfg(dict(name='jasper', mid_layers=len(blocks), jasper_blocks=blocks, precision=precision, **_common(labels)))


def root_config(model: str = 'wav2letter', **model_kw):
    """The whole tree of configuration/config.yaml: data / model / trainer"""
    m = {'wav2letter': wav2letter_model, 'jasper': jasper_model, 'jasper10x5': jasper10x5_model}[model](**model_kw)
    return to_cfg(dict(data=dict(train_manifest='???', val_manifest='???', batch_size=4, mel_spec=m['input_size'],
                                 audio_conf=dict(m['audio_conf'])),
                       model=m, trainer=dict(default_root_dir='.', max_epochs=5, max_steps=None, gpus=0)))


def synthetic_batch(N: int, T: int, n_mel: int = 64, seed: int = 1234, s_lo: int = 80, s_hi: int = 160, scaling: int = 2,
                    n_labels: int = 29, ragged: bool = False):
    """Synthetic training batch in _collator's layout: N(0,1) spectrograms [N, n_mel, T] at full length,
    int32 targets U{1..n_labels-1} zero-padded to the longest, target lengths U{s_lo..s_hi} capped so that every CTC
    alignment is feasible (T' >= 2 S).  ``ragged`` (SURVEY 8d's second run): input lengths U{T/2..T} with the longest
    utterance at T, spectrograms zero beyond their length as data_loader.py:149-158 pads them."""
    import torch
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, n_mel, T, generator=g)
    in_lens = torch.full((N,), T, dtype=torch.int32)
    if ragged:
        in_lens = torch.randint(T // 2, T + 1, (N,), generator=g, dtype=torch.int32)
        in_lens[int(torch.argmax(in_lens))] = T
        for n in range(N):
            x[n, :, int(in_lens[n]):] = 0.0
    tl = torch.randint(s_lo, s_hi + 1, (N,), generator=g, dtype=torch.int32)
    tl = torch.minimum(tl, (in_lens // scaling // 2).to(torch.int32)).clamp(min=1)
    tg = torch.randint(1, n_labels, (N, int(tl.max())), generator=g, dtype=torch.int32)
    for n in range(N):
        tg[n, int(tl[n]):] = 0
    return x, in_lens, tg, tl
