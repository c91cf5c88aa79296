"""ConvCTCASR: the LightningModule surface of the reference (base_asr_models.py:16-94) over
the HIP step engine.  Same constructor, attributes, methods, log keys and config keys."""
from __future__ import annotations

import random

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

    def create_example_input_array(self):
        batch_size = 4
        min_length, max_length = 100, 200
        lengths = torch.randint(min_length, max_length, (4,))
        return (torch.rand(batch_size, self._cfg.input_size, max_length), lengths)

    def compute_output_lengths(self, input_lengths):
        """floor(input_lengths / scaling_factor) (base_asr_models.py:33-39)"""
        return input_lengths // self.scaling_factor

    @property
    def scaling_factor(self):
        raise NotImplementedError()

    def forward(self, inputs, input_lengths):
        raise NotImplementedError()

    def add_string_metrics(self, out, output_lengths, texts, prefix):
        decoded_texts = self.ctc_decoder.decode(out, output_lengths)
        if random.random() < self.print_decoded_prob:
            print(f'reference: {texts[0]}')
            print(f'decoded  : {decoded_texts[0]}')
        wer_sum, cer_sum, wer_denom_sum, cer_denom_sum = 0, 0, 0, 0
        for expected, predicted in zip(texts, decoded_texts):
            cer_value, cer_denom = self.ctc_decoder.cer_ratio(expected, predicted)
            wer_value, wer_denom = self.ctc_decoder.wer_ratio(expected, predicted)
            cer_sum += cer_value
            cer_denom_sum += cer_denom
            wer_sum += wer_value
            wer_denom_sum += wer_denom
        cer = cer_sum / cer_denom_sum
        wer = wer_sum / wer_denom_sum
        lengths_ratio = sum(map(len, decoded_texts)) / sum(map(len, texts))
        return {prefix + '_cer': cer, prefix + '_wer': wer, prefix + '_len_ratio': lengths_ratio}

    # PyTorch Lightning methods
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

    def training_step(self, batch, batch_idx):
        inputs, input_lengths, targets, target_lengths, file_paths, texts = batch
        out, output_lengths = self.forward(self._device_batch(inputs), input_lengths)
        loss = self.criterion(out.transpose(0, 1), targets, output_lengths, target_lengths)
        logs = {'train_loss': loss, 'learning_rate': self.optimizers().param_groups[0]['lr']}
        logs.update(self.add_string_metrics(out, output_lengths, texts, 'train'))
        self.log_dict(logs)
        return loss

    def validation_step(self, batch, batch_idx):
        inputs, input_lengths, targets, target_lengths, file_paths, texts = batch
        out, output_lengths = self.forward(self._device_batch(inputs), input_lengths)
        loss = self.criterion(out.transpose(0, 1), targets, output_lengths, target_lengths)
        logs = {'val_loss': loss}
        logs.update(self.add_string_metrics(out, output_lengths, texts, 'val'))
        self.log_dict(logs)
        return loss
