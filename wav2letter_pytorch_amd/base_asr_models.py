"""ConvCTCASR: the LightningModule surface of the reference (base_asr_models.py:16-94) over
the HIP step engine.  Same constructor, attributes, methods, log keys and config keys; the bodies
are this package's own: one shared step for training and validation, batch-level metric totals."""
from __future__ import annotations

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

        def optimizers(self):
            return self._optimizers


EXAMPLE_BATCH, EXAMPLE_FRAMES = 4, (100, 200)         # base_asr_models.py:27-31


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
    def add_string_metrics(self, out, output_lengths, texts, prefix) -> Dict[str, float]:
        """greedy decode, then batch-level CER / WER (edit-distance totals over reference-length totals) and the
        decoded / reference length ratio, under the reference's keys (base_asr_models.py:53-69)"""
        hyps: Sequence[str] = self.ctc_decoder.decode(out, output_lengths)
        if random.random() < self.print_decoded_prob:
            print(f'reference: {texts[0]}')
            print(f'decoded  : {hyps[0]}')
        dec = self.ctc_decoder
        char_err, char_ref = map(sum, zip(*(dec.cer_ratio(ref, hyp) for ref, hyp in zip(texts, hyps))))
        word_err, word_ref = map(sum, zip(*(dec.wer_ratio(ref, hyp) for ref, hyp in zip(texts, hyps))))
        return {f'{prefix}_cer': char_err / char_ref, f'{prefix}_wer': word_err / word_ref,
                f'{prefix}_len_ratio': sum(len(h) for h in hyps) / sum(len(t) for t in texts)}

    # ------------------------------------------------------------------ Lightning hooks
    def configure_optimizers(self):
        optimizer = instantiate(self._cfg.optimizer, params=self.parameters())
        if type(optimizer) is torch.optim.SGD and next(self.parameters()).is_cuda:
            from .optim import FusedSGD            # same update rule and state dict; conv weights update in one fused pass
            optimizer = FusedSGD.from_sgd(optimizer)
        scheduler = instantiate(self._cfg.scheduler, optimizer=optimizer)
        return [optimizer], [scheduler]

    def _device_batch(self, inputs):
        dev = next(self.parameters()).device
        return inputs.to(dev, non_blocking=True) if inputs.device != dev else inputs

    def _step(self, batch, prefix: str, extra: Dict[str, float]):
        """forward -> CTC -> string metrics -> log_dict; the body of training_step and validation_step
        (base_asr_models.py:78-94).  batch = _collator's 6-tuple (data_loader.py:149-158)."""
        spect, spect_lens, targets, target_lens, _paths, texts = batch
        out, out_lens = self.forward(self._device_batch(spect), spect_lens)
        loss = self.criterion(out.transpose(0, 1), targets, out_lens, target_lens)
        self.log_dict({f'{prefix}_loss': loss, **extra, **self.add_string_metrics(out, out_lens, texts, prefix)})
        return loss

    def training_step(self, batch, batch_idx):
        return self._step(batch, 'train', {'learning_rate': self.optimizers().param_groups[0]['lr']})

    def validation_step(self, batch, batch_idx):
        return self._step(batch, 'val', {})
