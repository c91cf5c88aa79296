"""Command-line entry of the training flow, the counterpart of the reference's ``train.py:28-41`` for a machine without
Hydra / Lightning:

    python -m wav2letter_pytorch_amd.train [--config-dir /path/to/configuration] data.train_manifest=train.csv \\
           data.val_manifest=val.csv [model=jasper] [model.mid_layers=20] [trainer.max_epochs=1] ...

Same override syntax and config keys as ``python train.py ...``; without ``--config-dir`` the built-in copy of the
hyper-parameters (defaults.py) is used.  Manifests, labels, feature extraction and batching are data/data_loader.py's;
the fit loop is trainer.Trainer.  ``trainer.gpus=N`` (the reference's Lightning flag, README.md:40) starts N ranks of this
command line (launch.py), one per GPU; ``python -m torch.distributed.run --nproc-per-node N`` works too.  Every rank trains
on its own shard of the manifest and gradients are averaged with RCCL (distributed.GradReducer)."""
from __future__ import annotations

import os
import sys

from .config import _yaml_load, load_config, to_cfg
from .launch import spawn_ranks, under_launcher

# Nothing above maps libw2l_hip.so (or imports torch): with trainer.gpus=N this process only starts the ranks, and a launch
# parent must not have touched the GPU (launch.py; bench.py follows the same rule).  The model / data / trainer modules are
# imported where a rank first needs them.


class _Models(dict):
    """``name_to_model`` of the reference's train.py:15-19.  The model classes pull in torch and the HIP library, which a
    launch parent must not map, so the table fills itself the first time ANY read touches it -- lookups, ``in``, ``get``,
    iteration, ``len`` -- and behaves like the reference's plain dict from then on."""

    def _fill(self):
        if not dict.__len__(self):
            from . import Jasper, Wav2Letter
            dict.update(self, jasper=Jasper, wav2letter=Wav2Letter)
        return self

    def __missing__(self, key):
        if dict.__contains__(self._fill(), key):
            return dict.__getitem__(self, key)
        raise KeyError(key)

    def __contains__(self, key):
        return dict.__contains__(self._fill(), key)

    def __iter__(self):
        return dict.__iter__(self._fill())

    def __len__(self):
        return dict.__len__(self._fill())

    def get(self, key, default=None):
        return dict.get(self._fill(), key, default)

    def keys(self):
        return dict.keys(self._fill())

    def values(self):
        return dict.values(self._fill())

    def items(self):
        return dict.items(self._fill())

    def __repr__(self):
        return dict.__repr__(self._fill())


name_to_model = _Models()


def get_data_loaders(labels, cfg, rank: int = 0, world: int = 1):
    """train.py:21-26.  With more than one rank each loader walks its own shard of the manifest: a DistributedSampler
    over the raw items (indices rank, rank + world, ...; the tail padded so every rank takes the same number of steps --
    a rank that ran out of batches early would leave the others waiting in a collective), which is what Lightning's DDP
    injects into the reference's loaders."""
    from torch.utils.data.distributed import DistributedSampler
    from .data.data_loader import BatchAudioDataLoader, SpectrogramDataset
    loaders = []
    for manifest in (cfg.train_manifest, cfg.val_manifest):
        ds = SpectrogramDataset(manifest, cfg.audio_conf, labels, mel_spec=cfg.mel_spec)
        kw = {}
        if world > 1:
            kw['sampler'] = DistributedSampler(range(len(ds)), num_replicas=world, rank=rank, shuffle=False)
        loaders.append(BatchAudioDataLoader(ds, batch_size=cfg.batch_size, **kw))
    return loaders[0], loaders[1]


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
    from .defaults import root_config
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


def _requested_gpus(cfg) -> int:
    """trainer.gpus of the reference's config tree (configuration/config.yaml: passed to pytorch_lightning.Trainer):
    an int is a device count, a list names devices"""
    g = cfg.trainer.get('gpus', 0)
    if isinstance(g, (list, tuple)):
        return len(g)
    try:
        return int(g or 0)
    except (TypeError, ValueError):
        return 0


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    cfg = build_config(argv)
    for key in ('train_manifest', 'val_manifest'):
        if cfg.data.get(key) in (None, '???'):
            raise SystemExit(f'data.{key} is required (e.g. data.{key}=/path/to/manifest.csv)')
    n_gpus = _requested_gpus(cfg)
    if n_gpus > 1 and not under_launcher():
        # Trainer(gpus=N) of the reference = N DDP processes.  This parent has not touched the GPU: it only starts the ranks.
        rc = spawn_ranks(n_gpus, [sys.executable, '-m', 'wav2letter_pytorch_amd.train'] + argv)
        if rc:
            raise SystemExit(rc)
        return None, None
    import torch
    from .data import label_sets
    from .trainer import Trainer
    if type(cfg.model.labels) is str:
        cfg.model.labels = list(label_sets.labels_map[cfg.model.labels])
    if isinstance(cfg.model.get('decoder'), dict):
        cfg.model.decoder.labels = cfg.model.labels
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = 0 if os.environ.get('W2L_DIST_BACKEND') == 'gloo' else int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local)
    train_loader, val_loader = get_data_loaders(cfg.model.labels, cfg.data, rank, world)
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
