"""Command-line entry of the training flow, the counterpart of the reference's ``train.py:28-41`` for a machine without
Hydra / Lightning:

    python -m wav2letter_pytorch_amd.train [--config-dir /path/to/configuration] data.train_manifest=train.csv \\
           data.val_manifest=val.csv [model=jasper] [model.mid_layers=20] [trainer.max_epochs=1] ...

Same override syntax and config keys as ``python train.py ...``; without ``--config-dir`` the built-in copy of the
hyper-parameters (defaults.py) is used.  Manifests, labels, feature extraction and batching are data/data_loader.py's;
the fit loop is trainer.Trainer.  Under ``python -m torch.distributed.run --nproc-per-node N`` every rank trains on its
own GPU and gradients are averaged with RCCL (distributed.GradReducer)."""
from __future__ import annotations

import os
import sys

import torch

from . import Jasper, Wav2Letter
from .config import _yaml_load, load_config, to_cfg
from .data import label_sets
from .data.data_loader import BatchAudioDataLoader, SpectrogramDataset
from .defaults import root_config
from .trainer import Trainer

name_to_model = {'jasper': Jasper, 'wav2letter': Wav2Letter}


def get_data_loaders(labels, cfg):
    """train.py:21-26"""
    train = SpectrogramDataset(cfg.train_manifest, cfg.audio_conf, labels, mel_spec=cfg.mel_spec)
    val = SpectrogramDataset(cfg.val_manifest, cfg.audio_conf, labels, mel_spec=cfg.mel_spec)
    return BatchAudioDataLoader(train, batch_size=cfg.batch_size), BatchAudioDataLoader(val, batch_size=cfg.batch_size)


def build_config(argv):
    config_dir = None
    rest = []
    it = iter(argv)
    for a in it:
        if a == '--config-dir':
            config_dir = next(it)
        elif a.startswith('--config-dir='):
            config_dir = a.split('=', 1)[1]
        else:
            rest.append(a)
    overrides = [a for a in rest if '=' in a]
    if config_dir is not None:
        return load_config(config_dir, overrides)
    group = 'wav2letter'
    plain = []
    for ov in overrides:
        k, v = ov.split('=', 1)
        if k == 'model':
            group = v
        else:
            plain.append((k, v))
    cfg = root_config(group)
    for k, v in plain:
        cur = cfg
        parts = k.split('.')
        for p in parts[:-1]:
            cur = cur.setdefault(p, to_cfg({}))
        cur[parts[-1]] = to_cfg(_yaml_load(v))
    return cfg


def main(argv=None):
    cfg = build_config(sys.argv[1:] if argv is None else argv)
    for key in ('train_manifest', 'val_manifest'):
        if cfg.data.get(key) in (None, '???'):
            raise SystemExit(f'data.{key} is required (e.g. data.{key}=/path/to/manifest.csv)')
    if type(cfg.model.labels) is str:
        cfg.model.labels = list(label_sets.labels_map[cfg.model.labels])
    if isinstance(cfg.model.get('decoder'), dict):
        cfg.model.decoder.labels = cfg.model.labels
    world = int(os.environ.get('WORLD_SIZE', '1'))
    torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
    train_loader, val_loader = get_data_loaders(cfg.model.labels, cfg.data)
    model = name_to_model[cfg.model.name](cfg.model)
    if world > 1:
        from .distributed import GradReducer, broadcast_parameters, init_process_group_from_env
        init_process_group_from_env()
        model = model.cuda()
        broadcast_parameters(model)
        model.grad_reducer = GradReducer()
    trainer = Trainer(**{k: v for k, v in cfg.trainer.items()})
    trainer.fit(model, train_loader, val_loader)
    return trainer, model


if __name__ == '__main__':
    main()
