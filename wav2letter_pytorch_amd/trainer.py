"""Minimal fit loop standing in for ``pytorch_lightning.Trainer(**cfg.trainer).fit(model, train, val)``
(reference train.py:34-37) when Lightning is not installed: epochs / max_steps, optimizer + per-epoch
scheduler from ``configure_optimizers`` (base_asr_models.py:73-76), ``training_step`` /
``validation_step`` with their ``log_dict`` metrics, one checkpoint (reference state-dict keys) per
epoch under ``default_root_dir``.  Single process; data-parallel runs attach a
``distributed.GradReducer`` to the model (one process per GPU) before calling ``fit``."""
from __future__ import annotations

import os
import time
from typing import Optional

import torch


class Trainer:
    def __init__(self, default_root_dir: str = '.', max_epochs: int = 5, max_steps: Optional[int] = None, gpus=0,
                 log_every_n_steps: int = 50, enable_checkpointing: bool = True, **unused):
        self.default_root_dir = default_root_dir
        self.max_epochs = max_epochs
        self.max_steps = max_steps
        self.log_every_n_steps = log_every_n_steps
        self.enable_checkpointing = enable_checkpointing
        self.global_step = 0
        self.logged = []

    def fit(self, model, train_dataloader, val_dataloader=None):
        if not torch.cuda.is_available():
            raise RuntimeError('wav2letter_pytorch_amd trains on MI355X only (no CPU path); trainer.gpus is implied')
        model = model.cuda()
        optimizers, schedulers = model.configure_optimizers()
        opt = optimizers[0]
        model._optimizers = opt
        if hasattr(opt, 'overlap'):          # optim.FusedSGD: weight updates stream under the next forward pass
            opt.overlap = True
        join = getattr(opt, 'join', lambda: None)
        done = False
        for epoch in range(self.max_epochs):
            model.train()
            t0 = time.time()
            for i, batch in enumerate(train_dataloader):
                opt.zero_grad(set_to_none=True)
                loss = model.training_step(batch, i)
                loss.backward()
                opt.step()
                self.global_step += 1
                if self.global_step % self.log_every_n_steps == 0 or self.global_step == 1:
                    logs = dict(getattr(model, '_logged', {}))
                    self.logged.append((self.global_step, logs))
                    print(f'epoch {epoch} step {self.global_step} ' + ' '.join(f'{k}={v:.4g}' for k, v in logs.items()))
                if self.max_steps is not None and self.global_step >= self.max_steps:
                    done = True
                    break
            for sch in schedulers:
                sch.step()
            join()                               # parameters are read below (validation, checkpoint)
            if val_dataloader is not None:
                model.eval()
                with torch.no_grad():
                    for i, batch in enumerate(val_dataloader):
                        model.validation_step(batch, i)
                logs = {k: v for k, v in getattr(model, '_logged', {}).items() if k.startswith('val')}
                print(f'epoch {epoch} validation ' + ' '.join(f'{k}={v:.4g}' for k, v in logs.items()))
            if self.enable_checkpointing:
                os.makedirs(self.default_root_dir, exist_ok=True)
                path = os.path.join(self.default_root_dir, f'epoch={epoch}-step={self.global_step}.ckpt')
                torch.save({'state_dict': {k: v.detach().cpu().contiguous() for k, v in model.state_dict().items()},
                            'epoch': epoch, 'global_step': self.global_step}, path)
            print(f'epoch {epoch} done in {time.time() - t0:.1f}s')
            if done:
                break
        join()
        return model
