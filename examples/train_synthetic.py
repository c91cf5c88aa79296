#!/usr/bin/env python3
"""End-to-end plumbing check of the drop-in surface without audio dependencies (reference flow:
train.py:28-41 with a Hydra-style config tree, labels resolved from label_sets, the model chosen by
``cfg.model.name``, a Trainer driving training_step / validation_step):

    python examples/train_synthetic.py /path/to/reference/configuration model.mid_layers=3 trainer.max_epochs=1

The data loader is a synthetic stand-in for SpectrogramDataset + _collator (data/data_loader.py:149-158):
same 6-tuple batch layout, random spectrograms and transcripts."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wav2letter_pytorch_amd import Jasper, Wav2Letter  # noqa: E402
from wav2letter_pytorch_amd.config import load_config  # noqa: E402
from wav2letter_pytorch_amd.data import label_sets  # noqa: E402
from wav2letter_pytorch_amd.trainer import Trainer  # noqa: E402

name_to_model = {"jasper": Jasper, "wav2letter": Wav2Letter}


def synthetic_loader(labels, n_batches, batch_size, n_mel, seed):
    g = torch.Generator().manual_seed(seed)
    batches = []
    for _ in range(n_batches):
        lens = torch.randint(300, 501, (batch_size,), generator=g, dtype=torch.int32)
        tmax = int(lens.max())
        x = torch.randn(batch_size, n_mel, tmax, generator=g)
        tl = torch.randint(10, 40, (batch_size,), generator=g, dtype=torch.int32)
        tg = torch.zeros(batch_size, int(tl.max()), dtype=torch.int32)
        texts = []
        for n in range(batch_size):
            x[n, :, int(lens[n]):] = 0
            ids = torch.randint(1, len(labels), (int(tl[n]),), generator=g)
            tg[n, :int(tl[n])] = ids.to(torch.int32)
            texts.append(''.join(labels[int(i)] for i in ids))
        batches.append((x, lens, tg, tl, tuple(f'synthetic_{n}.wav' for n in range(batch_size)), tuple(texts)))
    return batches


def main():
    cfg_dir = sys.argv[1]
    overrides = [a for a in sys.argv[2:] if '=' in a]
    overrides += ['data.train_manifest=synthetic', 'data.val_manifest=synthetic']
    cfg = load_config(cfg_dir, overrides)
    if type(cfg.model.labels) is str:
        cfg.model.labels = label_sets.labels_map[cfg.model.labels]
        cfg.model.decoder.labels = cfg.model.labels
    train = synthetic_loader(cfg.model.labels, 8, cfg.data.batch_size, cfg.model.input_size, 0)
    val = synthetic_loader(cfg.model.labels, 2, cfg.data.batch_size, cfg.model.input_size, 1)
    model = name_to_model[cfg.model.name](cfg.model)
    trainer = Trainer(**{k: v for k, v in cfg.trainer.items()})
    trainer.fit(model, train, val)


if __name__ == '__main__':
    main()
