"""ConvCTCASR: the LightningModule surface of the reference (base_asr_models.py:16-94) over
the HIP step engine.  Same constructor, attributes, methods, log keys and config keys; the bodies
are this package's own: one shared step for training and validation, batch-level metric totals."""
from __future__ import annotations

import os
import random
from typing import Callable, Dict, Sequence

import torch
import torch.nn as nn

from .config import instantiate
from .ctc_loss import CTCLoss

try:  # PyTorch-Lightning is optional: absent in the build image
    import pytorch_lightning as ptl  # type: ignore
    _Base = ptl.LightningModule
except ImportError:  # pragma: no cover - exercised in this image
    class _Base(nn.Module):
        """The slice of LightningModule that ConvCTCASR uses (log_dict / optimizers); the
        minimal fit loop in trainer.py drives it."""

        def __init__(self):
            super().__init__()
            self._logged = {}
            self._optimizers = None

        def log_dict(self, d, *args, **kwargs):
            self._logged.update({k: (float(v.detach()) if torch.is_tensor(v) else float(v)) for k, v in d.items()})

        # the two hooks of Lightning's loop that ConvCTCASR uses (trainer.Trainer calls them at the same points)
        def on_train_batch_end(self, outputs=None, batch=None, batch_idx=0, *args):
            pass

        def on_train_epoch_end(self, *args):
            pass

        def optimizers(self):
            return self._optimizers


EXAMPLE_BATCH, EXAMPLE_FRAMES = 4, (100, 200)         # base_asr_models.py:27-31

_SCORING = []


def _scoring_pool():
    """the scoring threads of the process: greedy collapse + CER / WER of the batches training_step enqueued"""
    if not _SCORING:
        from concurrent.futures import ThreadPoolExecutor
        # (jobs are independent and are harvested in submission order; the C call in each holds no interpreter lock)
        _SCORING.append(ThreadPoolExecutor(max_workers=max(1, min(4, (os.cpu_count() or 2) // 2)), thread_name_prefix='w2l-metrics'))
    return _SCORING[0]


def _nothing():
    return None


def _drop_engine_after_load(module, incompatible_keys):
    """load_state_dict post-hook (a module-level function: a lambda here would make every model unpicklable)"""
    module.invalidate_engine()


class EngineSlot(tuple):
    """(key, StackEngine) as cached in a module's ``__dict__``.  An engine holds HIP streams and raw-pointer specs of THIS
    module's tensors: a copy of the module (``copy.deepcopy``, ``torch.save`` of the whole module, pickling for a worker)
    gets an empty slot instead and rebuilds its own engine on first use."""

    def __deepcopy__(self, memo):
        return None

    def __reduce__(self):
        return (_nothing, ())


def feature_size(cfg, audio_conf) -> int:
    """rows of the input spectrogram: ``cfg.input_size`` when the config names one, else the one-sided STFT bins of the
    analysis window, 1 + n_fft / 2 with n_fft = sample_rate * window_size (wav2letter.py:53-57, jasper.py:426-430)"""
    if cfg.input_size:
        return cfg.input_size
    return int(1 + audio_conf['sample_rate'] * audio_conf['window_size'] / 2)


class ConvCTCASR(_Base):
    def __init__(self, cfg):
        super().__init__()
        self._cfg = cfg
        self.audio_conf = cfg.audio_conf
        self.labels = cfg.labels
        self.ctc_decoder = instantiate(cfg.decoder)
        self.criterion = CTCLoss(blank=0, reduction='mean', zero_infinity=True)      # base_asr_models.py:23
        self.print_decoded_prob = cfg.get('print_decoded_prob', 0)
        self.example_input_array = self.create_example_input_array()
        # load_state_dict(assign=True) swaps Parameter OBJECTS under the engine's specs: rebuild after any load
        self.register_load_state_dict_post_hook(_drop_engine_after_load)

    def create_example_input_array(self):
        """(spectrograms [4, input_size, 200] ~ U[0,1), lengths [4] ~ U{100..199}) -- Lightning's model summary input;
        the length draw comes first, as in the reference, so a seeded construction yields the same example"""
        lo, hi = EXAMPLE_FRAMES
        lengths = torch.randint(lo, hi, (EXAMPLE_BATCH,))
        return torch.rand(EXAMPLE_BATCH, self._cfg.input_size, hi), lengths

    def compute_output_lengths(self, input_lengths):
        """floor(input_lengths / scaling_factor) (base_asr_models.py:33-39)"""
        return input_lengths // self.scaling_factor

    @property
    def scaling_factor(self):
        raise NotImplementedError()

    def forward(self, inputs, input_lengths):
        raise NotImplementedError()

    # ------------------------------------------------------------------ engine cache
    def _apply(self, fn, *args, **kwargs):
        # .to() / .cuda() / .float() replace parameter data and buffer tensors: the engine's specs must be rebuilt
        self.invalidate_engine()
        return super()._apply(fn, *args, **kwargs)

    def invalidate_engine(self):
        """drop the cached StackEngine (call after replacing a Parameter / buffer OBJECT by hand; ``module.to()`` and
        friends do it themselves, in-place updates -- optimizers, load_state_dict -- never need it)"""
        hit = self.__dict__.pop('_engine_cache', None)
        eng = hit[1] if hit is not None else None
        if eng is not None and getattr(eng, '_deferred', None):
            # weight gradients held back for the next forward pass (optim.FusedSGD.defer_wgrad) live in the engine: a stepped
            # batch's updates are applied before the engine goes (the optimizer only holds engines weakly)
            eng.flush_deferred()
            eng.join_side()

    def _cached_engine(self, build: Callable[[], 'object']):
        """The StackEngine of this module tree, built once and kept until the module is moved / cast (``_apply``) or
        ``invalidate_engine()`` is called (the per-call guard is just precision + device: walking the module tree on every
        forward cost Jasper 10x5 0.7 ms of host time per step).  The per-step switches (weight-gradient overlap, graph-mode
        dropout counter, data-parallel reducer) are re-read on every call, so attaching a ``grad_reducer`` after the first
        forward takes effect."""
        key = (getattr(self, 'precision', None), next(self.parameters()).device)
        hit = self.__dict__.get('_engine_cache')
        if hit is None or hit[0] != key:
            hit = EngineSlot((key, build()))
            self.__dict__['_engine_cache'] = hit
        eng = hit[1]
        eng.overlap_wgrad = getattr(self, '_overlap_wgrad', True)
        eng.dropout_counter = getattr(self, '_dropout_counter', None)               # graph.GraphedTrainStep
        reducer = getattr(self, 'grad_reducer', None)                               # set by data-parallel drivers
        eng.grad_ready = reducer.on_grad if reducer is not None else None
        eng.flat_ready = getattr(reducer, 'on_flat', None)
        eng.backward_done = reducer.finish if reducer is not None else None
        # deferred weight gradients (optim.FusedSGD.defer_wgrad): the top k units' dW + update run beside the NEXT forward
        opt = getattr(self, '_deferred_opt', None)
        eng.defer_wgrad = getattr(self, '_defer_wgrad', 0) if opt is not None else 0
        eng.deferred = opt
        eng.grad_reduce_start = reducer.start if (reducer is not None and reducer.active) else None
        if opt is not None:
            opt._register_engine(eng)
        return eng

    # ------------------------------------------------------------------ metrics
    # The reference decodes and scores EVERY training batch on the host before it returns the loss (base_asr_models.py:83):
    # argmax, 16 000 .item() calls, 64 Levenshtein runs -- with the GPU idle, because backward() is enqueued after it.  Here
    # the step only ENQUEUES what the metrics need -- the argmax kernel, one asynchronous copy of the int32 index matrix (and
    # of the loss scalar) into pinned host memory, an event -- and returns; the strings are scored when the event has fired,
    # which is after backward() and the optimizer step have been enqueued: a worker thread waits for the event and makes ONE C
    # call (decoder.GreedyDecoder.score_batch, no interpreter lock held), on_train_batch_end logs what has arrived.  The logged
    # values are the same numbers, bit for bit; they reach log_dict one hook (or, while the scorer is busy, a few) later.  ``async_metrics = False`` (or W2L_SYNC_METRICS=1) restores the reference's order: score, log, then return.
    async_metrics = os.environ.get('W2L_SYNC_METRICS', '0') != '1'
    METRICS_MAX_PENDING = 4          # batches whose metrics may be outstanding before the enqueuing thread waits for the oldest

    def _score_indices(self, idx_host, sizes, texts, prefix) -> Dict[str, float]:
        """host part of add_string_metrics: argmax indices (on the host) -> the three logged ratios"""
        dec = self.ctc_decoder
        if sizes is not None and torch.is_tensor(sizes):
            sizes = sizes.to(torch.int32)
        if hasattr(dec, 'score_batch'):
            hyps, (char_err, char_ref, word_err, word_ref) = dec.score_batch(idx_host, sizes, texts)
        else:                            # any other decoder object with the reference's interface
            hyps = [h[0] for h in dec.convert_to_strings(idx_host, sizes, remove_repetitions=True)]
            char_err, char_ref = map(sum, zip(*(dec.cer_ratio(ref, hyp) for ref, hyp in zip(texts, hyps))))
            word_err, word_ref = map(sum, zip(*(dec.wer_ratio(ref, hyp) for ref, hyp in zip(texts, hyps))))
        if random.random() < self.print_decoded_prob:
            print(f'reference: {texts[0]}')
            print(f'decoded  : {hyps[0]}')
        return {f'{prefix}_cer': char_err / char_ref, f'{prefix}_wer': word_err / word_ref,
                f'{prefix}_len_ratio': sum(len(h) for h in hyps) / sum(len(t) for t in texts)}

    def add_string_metrics(self, out, output_lengths, texts, prefix) -> Dict[str, float]:
        """greedy decode, then batch-level CER / WER (edit-distance totals over reference-length totals) and the
        decoded / reference length ratio, under the reference's keys (base_asr_models.py:53-69).  SYNCHRONOUS (one
        device-to-host copy of the index matrix): the training loop goes through enqueue_string_metrics instead."""
        from .decoder import argmax_indices
        if len(out.shape) == 2:
            out = out.unsqueeze(0)
        return self._score_indices(argmax_indices(out).cpu(), output_lengths, texts, prefix)

    def enqueue_string_metrics(self, out, output_lengths, texts, prefix, loss=None, extra=None):
        """the device half of add_string_metrics, without a host synchronisation: argmax kernel + asynchronous copies into
        pinned memory + an event on the current stream.  The record is scored and logged by resolve_metrics()."""
        from .decoder import argmax_indices
        idx = argmax_indices(out)
        idx_host = torch.empty(idx.shape, dtype=torch.int32, pin_memory=True)
        idx_host.copy_(idx, non_blocking=True)
        loss_host = None
        if loss is not None:
            loss_host = torch.empty(1, dtype=torch.float32, pin_memory=True)
            loss_host.copy_(loss.detach().reshape(1), non_blocking=True)
        sizes = output_lengths
        if torch.is_tensor(sizes) and sizes.is_cuda:
            host = torch.empty(sizes.shape, dtype=sizes.dtype, pin_memory=True)
            host.copy_(sizes, non_blocking=True)
            sizes = host
        ev = torch.cuda.Event()
        ev.record()
        # scored on the package's worker thread: it waits for the event (the copies have landed), then makes ONE C call that holds
        # no interpreter lock (decoder.GreedyDecoder.score_batch) -- the thread that enqueues the step never waits for either
        job = _scoring_pool().submit(self._score_job, ev, idx_host, sizes, tuple(texts), prefix)
        pend = self.__dict__.setdefault('_pending_metrics', [])
        pend.append((job, loss_host, prefix, dict(extra or {}), idx))
        return ev

    def _score_job(self, ev, idx_host, sizes, texts, prefix):
        ev.synchronize()
        return self._score_indices(idx_host, sizes, texts, prefix)

    def resolve_metrics(self, wait_all: bool = True) -> int:
        """log the enqueued batches, oldest first, as their scores arrive from the worker thread.  ``wait_all``: block until
        every one is in (logging points, epoch ends, validation); otherwise take what is ready and wait only while more than
        METRICS_MAX_PENDING batches are outstanding (the host never runs further ahead than that).  Returns the number of
        batches logged."""
        pend = self.__dict__.get('_pending_metrics')
        done = 0
        while pend:
            job = pend[0][0]
            if not wait_all and len(pend) <= self.METRICS_MAX_PENDING and not job.done():
                break
            metrics = job.result()                       # (re-raises whatever the scoring raised)
            _, loss_host, prefix, extra, _idx = pend.pop(0)
            logs = {}
            if loss_host is not None:
                logs[f'{prefix}_loss'] = float(loss_host[0])
            logs.update(extra)
            logs.update(metrics)
            self.log_dict(logs)
            done += 1
        return done

    # Lightning calls these around every training batch / epoch (our trainer.Trainer does the same): the metrics of the
    # batch just enqueued are logged here, behind backward() and optimizer.step()
    def on_train_batch_end(self, outputs=None, batch=None, batch_idx=0, *args):
        self.resolve_metrics(wait_all=False)

    def on_train_epoch_end(self, *args):
        self.resolve_metrics(wait_all=True)

    # ------------------------------------------------------------------ Lightning hooks
    def configure_optimizers(self):
        optimizer = instantiate(self._cfg.optimizer, params=self.parameters())
        if type(optimizer) is torch.optim.SGD and next(self.parameters()).is_cuda:
            from .optim import FusedSGD            # same update rule and state dict; conv weights update in one fused pass
            optimizer = FusedSGD.from_sgd(optimizer)
        scheduler = instantiate(self._cfg.scheduler, optimizer=optimizer)
        return [optimizer], [scheduler]

    def _device_batch(self, inputs):
        """the spectrograms on the model's device without stalling the host: a host tensor that is not page-locked is staged
        through one of two pinned buffers owned by the module (an H2D copy from pageable memory makes the HIP runtime wait
        for the stream -- i.e. for everything the host has run ahead by); a DataLoader with pin_memory=True skips the staging"""
        dev = next(self.parameters()).device
        if inputs.device == dev:
            return inputs
        if inputs.device.type == 'cpu' and dev.type == 'cuda' and not inputs.is_pinned():
            ring = self.__dict__.setdefault('_stage_ring', [None, None, 0])
            slot = ring[2] & 1
            ring[2] += 1
            ent = ring[slot]
            if ent is None or ent[0].numel() < inputs.numel() or ent[0].dtype != inputs.dtype:
                ent = ring[slot] = [torch.empty(inputs.numel(), dtype=inputs.dtype, pin_memory=True), None]
            if ent[1] is not None:
                ent[1].synchronize()               # the copy issued from this buffer two batches ago (long done)
            stage = ent[0][: inputs.numel()].view(inputs.shape)
            stage.copy_(inputs)
            out, ent[1] = self._upload(stage, dev)
            return out
        if inputs.device.type == 'cpu' and dev.type == 'cuda':
            return self._upload(inputs, dev)[0]
        return inputs.to(dev, non_blocking=True)

    def _upload(self, pinned, dev):
        """page-locked host tensor -> device on a COPY STREAM of the module's own: the host enqueues step i + 1 while the GPU
        still runs step i, so the 8 MB of the next batch cross PCIe under step i's kernels instead of in front of step i + 1's
        first one (0.16 ms of an idle chip per step on the caller's stream); the caller's stream waits for the copy's event"""
        st = self.__dict__.get('_copy_stream')
        if st is None or st.device != dev:
            st = self.__dict__['_copy_stream'] = torch.cuda.Stream(device=dev)
        main = torch.cuda.current_stream(dev)
        with torch.cuda.stream(st):
            out = pinned.to(dev, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(st)
        main.wait_event(ev)
        out.record_stream(main)                # (allocated on the copy stream, consumed on the caller's)
        return out, ev

    def _device_ints(self, dev, *tensors):
        """small host integer tensors of a batch (targets, lengths) -> int32 device tensors through ONE pinned buffer and one
        asynchronous copy (three pageable copies would be three stream synchronisations)"""
        if dev.type != 'cuda' or any(t.is_cuda for t in tensors):
            return tuple(t.to(device=dev, dtype=torch.int32) for t in tensors)
        flat = [t.reshape(-1).to(torch.int32) for t in tensors]
        total = sum(f.numel() for f in flat)
        host = torch.empty(total, dtype=torch.int32, pin_memory=True)
        torch.cat(flat, out=host)
        devbuf = host.to(dev, non_blocking=True)
        outs, off = [], 0
        for t, f in zip(tensors, flat):
            outs.append(devbuf[off: off + f.numel()].view(t.shape))
            off += f.numel()
        self.__dict__['_ints_keep'] = host         # alive until the next batch's copy has been enqueued behind it
        return tuple(outs)

    def _step(self, batch, prefix: str, extra: Dict[str, float]):
        """forward -> CTC -> string metrics -> log_dict; the body of training_step and validation_step
        (base_asr_models.py:78-94).  batch = _collator's 6-tuple (data_loader.py:149-158).  Nothing in here waits for the
        GPU when ``async_metrics`` is on and the step is a training step: the metrics are logged by on_train_batch_end."""
        spect, spect_lens, targets, target_lens, _paths, texts = batch
        x = self._device_batch(spect)
        out, out_lens = self.forward(x, spect_lens)
        if x.is_cuda and torch.is_tensor(out_lens) and torch.is_tensor(targets) and torch.is_tensor(target_lens):
            tg_d, ol_d, tl_d = self._device_ints(x.device, targets, out_lens, target_lens)
        else:
            tg_d, ol_d, tl_d = targets, out_lens, target_lens
        loss = self.criterion(out.transpose(0, 1), tg_d, ol_d, tl_d)
        if self.async_metrics and x.is_cuda:
            self.enqueue_string_metrics(out, out_lens, texts, prefix, loss=loss, extra=extra)
            if prefix != 'train':                  # validation: nothing to overlap the scoring with, and the epoch mean
                self.resolve_metrics(wait_all=True)     # is formed batch by batch
        else:
            self.log_dict({f'{prefix}_loss': loss, **extra, **self.add_string_metrics(out, out_lens, texts, prefix)})
        return loss

    def training_step(self, batch, batch_idx):
        return self._step(batch, 'train', {'learning_rate': self.optimizers().param_groups[0]['lr']})

    def validation_step(self, batch, batch_idx):
        return self._step(batch, 'val', {})
