"""Minimal fit loop standing in for ``pytorch_lightning.Trainer(**cfg.trainer).fit(model, train, val)``
(reference train.py:34-37) when Lightning is not installed: epochs / max_steps, optimizer + per-epoch
scheduler from ``configure_optimizers`` (base_asr_models.py:73-76), ``training_step`` /
``validation_step`` with their ``log_dict`` metrics, one checkpoint per epoch under ``default_root_dir``.

Data parallel (one process per GPU, launch.py / torchrun): every rank runs this loop on its own shard of the
manifest (train.py attaches a DistributedSampler, as Lightning's DDP does), gradients are averaged by the
``distributed.GradReducer`` attached to the model; only rank 0 prints and writes checkpoints.

Checkpoints carry Lightning's keys: ``state_dict`` (reference parameter names), ``epoch``, ``global_step``,
``optimizer_states`` and ``lr_schedulers`` (lists, one entry per optimizer / scheduler), so an interrupted run
resumes with its momentum buffers and learning rate: ``Trainer(resume_from_checkpoint=path)`` or
``fit(..., ckpt_path=path)``."""
from __future__ import annotations

import os
import time
from typing import Optional

import torch
import torch.distributed as dist


class _EpochMean:
    """Batch-size-weighted mean of the logged scalars of one validation epoch (Lightning's on_epoch reduction)."""

    def __init__(self):
        self.sums, self.weight = {}, {}

    def add(self, logs: dict, n: int):
        for k, v in logs.items():
            self.sums[k] = self.sums.get(k, 0.0) + float(v) * n
            self.weight[k] = self.weight.get(k, 0) + n

    def result(self) -> dict:
        return {k: self.sums[k] / self.weight[k] for k in self.sums if self.weight[k]}


class Trainer:
    def __init__(self, default_root_dir: str = '.', max_epochs: int = 5, max_steps: Optional[int] = None, gpus=0,
                 log_every_n_steps: int = 50, enable_checkpointing: bool = True, resume_from_checkpoint: Optional[str] = None,
                 **unused):
        self.default_root_dir = default_root_dir
        self.max_epochs = max_epochs
        self.max_steps = max_steps
        self.log_every_n_steps = log_every_n_steps
        self.enable_checkpointing = enable_checkpointing
        self.resume_from_checkpoint = resume_from_checkpoint
        self.global_step = 0
        self.current_epoch = 0
        self.logged = []
        self.val_logged = []                 # one dict of epoch-mean validation metrics per epoch

    @property
    def global_rank(self) -> int:
        return dist.get_rank() if dist.is_initialized() else 0

    @property
    def is_global_zero(self) -> bool:
        return self.global_rank == 0

    def _say(self, msg: str):
        if self.is_global_zero:
            print(msg, flush=True)

    # ------------------------------------------------------------------ checkpoints
    def save_checkpoint(self, path: str, model, optimizers, schedulers, epoch: int):
        state = {k: v.detach().cpu().contiguous() for k, v in model.state_dict().items()}
        torch.save({'state_dict': state, 'epoch': epoch, 'global_step': self.global_step,
                    'optimizer_states': [o.state_dict() for o in optimizers],      # FusedSGD.state_dict joins its stream
                    'lr_schedulers': [s.state_dict() for s in schedulers]}, path)

    def _restore(self, path: str, model, optimizers, schedulers) -> int:
        ck = torch.load(path, map_location='cpu')
        model.load_state_dict(ck['state_dict'])
        from .engine import invalidate_packed
        invalidate_packed(model)
        for o, st in zip(optimizers, ck.get('optimizer_states', [])):
            o.load_state_dict(st)
        for s, st in zip(schedulers, ck.get('lr_schedulers', [])):
            s.load_state_dict(st)
        self.global_step = int(ck.get('global_step', 0))
        return int(ck.get('epoch', -1)) + 1          # the checkpoint is written at the END of its epoch

    # ------------------------------------------------------------------ loop
    def fit(self, model, train_dataloader, val_dataloader=None, ckpt_path: Optional[str] = None):
        if not torch.cuda.is_available():
            raise RuntimeError('wav2letter_pytorch_amd trains on MI355X only (no CPU path); trainer.gpus is implied')
        model = model.cuda()
        optimizers, schedulers = model.configure_optimizers()
        opt = optimizers[0]
        model._optimizers = opt
        join = getattr(opt, 'join', lambda: None)
        first_epoch = 0
        ckpt_path = ckpt_path or self.resume_from_checkpoint
        if ckpt_path:
            first_epoch = self._restore(ckpt_path, model, optimizers, schedulers)
            self._say(f'resumed from {ckpt_path}: epoch {first_epoch}, step {self.global_step}')
        if hasattr(opt, 'overlap'):          # optim.FusedSGD: weight updates stream under the next forward pass
            opt.overlap = True
            # ... and the top units' weight gradients run beside it (W2L_DEFER_WGRAD=k, 0 = off; default: 4 of a deep stack).
            # While a gradient is held back ``p.grad`` of that weight is None at step() ("Deferred weight
            # gradients"): set W2L_DEFER_WGRAD=0 for anything that reads gradients between backward() and step().
            # (after the restore: load_state_dict drops the step engine, which is what counts the units)
            n_units = len(model.engine().units) if hasattr(model, 'engine') else 0
            k = int(os.environ.get('W2L_DEFER_WGRAD', min(4, n_units // 4)))
            if k and hasattr(opt, 'defer_wgrad'):
                opt.defer_wgrad(model, k)
        from . import engine as _E, replay as _replay
        self._say('wav2letter_pytorch_amd: '
                  + ('bit-reproducible step (W2L_DETERMINISTIC=1)' if _E.DETERMINISTIC_WGRAD else
                     'default step: fp32 atomics in split reductions, not bit-reproducible from run to run (W2L_DETERMINISTIC=1: +0.6 %)')
                  + ('; warm step shapes are replayed from recorded launch lists (W2L_REPLAY=0: eager)' if _replay.ENABLED else
                     '; eager step (W2L_REPLAY=0)')
                  + ('; string metrics scored behind backward()' if getattr(model, 'async_metrics', False) else ''))
        done = self.max_steps is not None and self.global_step >= self.max_steps
        for epoch in range(first_epoch, self.max_epochs):
            if done:
                break
            self.current_epoch = epoch
            sampler = getattr(train_dataloader, 'sampler', None)
            if hasattr(sampler, 'set_epoch'):
                sampler.set_epoch(epoch)
            model.train()
            t0 = time.time()
            for i, batch in enumerate(train_dataloader):
                opt.zero_grad(set_to_none=True)
                loss = model.training_step(batch, i)
                loss.backward()
                opt.step()
                # Lightning's hook order: the batch's string metrics (greedy decode, CER / WER) are scored HERE, with the
                # backward pass and the update already enqueued -- training_step itself never waits for the GPU
                model.on_train_batch_end(loss, batch, i)
                self.global_step += 1
                if self.global_step % self.log_every_n_steps == 0 or self.global_step == 1:
                    if hasattr(model, 'resolve_metrics'):
                        model.resolve_metrics(wait_all=True)         # a logging point reports THIS step (one sync per log line)
                    logs = dict(getattr(model, '_logged', {}))
                    self.logged.append((self.global_step, logs))
                    self._say(f'epoch {epoch} step {self.global_step} ' + ' '.join(f'{k}={v:.4g}' for k, v in logs.items()))
                if self.max_steps is not None and self.global_step >= self.max_steps:
                    done = True
                    break
            model.on_train_epoch_end()
            for sch in schedulers:
                sch.step()
            join()                               # parameters are read below (validation, checkpoint)
            if val_dataloader is not None:
                model.eval()
                mean = _EpochMean()
                with torch.no_grad():
                    for i, batch in enumerate(val_dataloader):
                        for k in [k for k in getattr(model, '_logged', {}) if k.startswith('val')]:
                            del model._logged[k]
                        model.validation_step(batch, i)
                        mean.add({k: v for k, v in getattr(model, '_logged', {}).items() if k.startswith('val')},
                                 len(batch[0]))
                logs = mean.result()
                if hasattr(model, '_logged'):
                    model._logged.update(logs)   # what a callback / the caller reads after the epoch: the epoch means
                self.val_logged.append(logs)
                self._say(f'epoch {epoch} validation ' + ' '.join(f'{k}={v:.4g}' for k, v in logs.items()))
            if self.enable_checkpointing and self.is_global_zero:
                os.makedirs(self.default_root_dir, exist_ok=True)
                path = os.path.join(self.default_root_dir, f'epoch={epoch}-step={self.global_step}.ckpt')
                self.save_checkpoint(path, model, optimizers, schedulers, epoch)
            eng = getattr(model, '_engine_cache', None)
            if eng is not None and getattr(eng[1], 'fp8', False):
                clipped = eng[1].fp8_saturated()          # one sync per epoch
                if clipped:
                    self._say(f'epoch {epoch}: {clipped} activation elements saturated the e4m3 range in fp8 mode (fixed '
                              'per-tensor activation scales, engine.FP8_ACT_SCALE): their forward operands were clipped')
            self._say(f'epoch {epoch} done in {time.time() - t0:.1f}s')
        join()
        return model
