"""Data side of the training step with the reference's interface (data/data_loader.py): ``load_audio``,
``SpectrogramExtractor``, ``SpectrogramDataset``, ``_collator``, ``BatchAudioDataLoader``.

MI355X-first difference: features are not computed one utterance at a time on the host.  The loader reads raw audio,
pads it into one [N, L_max] buffer, moves it to the GPU once and runs the whole batch through three HIP launches
(w2l_logmel, w2l_feature_normalize: csrc/features.hip), which write the batch directly in the right-zero-padded
[N, n_mels, T_max] layout ``_collator`` (data_loader.py:149-158) produces.  ``SpectrogramExtractor.extract`` and
``SpectrogramDataset.__getitem__`` keep the reference's per-utterance semantics (same kernels, N = 1).
"""
from __future__ import annotations

import json
import math
import os
import wave
from typing import List, Optional, Sequence

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

from .._lib import check, lib, ptr, stream_ptr
from .mel import mel_filterbank

_TORCH_WINDOWS = {'hann': torch.hann_window, 'hamming': torch.hamming_window, 'blackman': torch.blackman_window,
                  'bartlett': torch.bartlett_window, 'none': None}


def load_audio(path, duration=-1, offset=0):
    """float32 samples in [-1, 1) (data_loader.py:20-31).  Uses soundfile when it is installed (any format it reads);
    otherwise PCM / float WAV files through the standard library."""
    try:
        import soundfile as sf
    except ImportError:
        sf = None
    if sf is not None:
        with sf.SoundFile(path, 'r') as f:
            sr = f.samplerate
            if offset > 0:
                f.seek(int(offset * sr))
            samples = f.read(int(duration * sr), dtype='float32') if duration > 0 else f.read(dtype='float32')
        return samples.transpose()
    samples, sr = _read_wav(path)
    start = int(offset * sr) if offset > 0 else 0
    stop = start + int(duration * sr) if duration > 0 else None
    return samples[start:stop].transpose()


def _read_wav(path):
    try:
        with wave.open(path, 'rb') as w:
            sr, nch, width, n = w.getframerate(), w.getnchannels(), w.getsampwidth(), w.getnframes()
            raw = w.readframes(n)
        if width == 2:
            data = np.frombuffer(raw, dtype='<i2').astype(np.float32) / 32768.0
        elif width == 4:
            data = np.frombuffer(raw, dtype='<i4').astype(np.float32) / 2147483648.0
        elif width == 1:
            data = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
        else:
            raise ValueError(f'{path}: unsupported sample width {width}')
    except wave.Error:
        from scipy.io import wavfile                      # IEEE-float WAV
        sr, data = wavfile.read(path)
        nch = 1 if data.ndim == 1 else data.shape[1]
        data = data.astype(np.float32).reshape(-1)
    if nch > 1:
        data = data.reshape(-1, nch)
    return data, sr


def _sample_rate(path) -> int:
    try:
        import soundfile as sf
        return sf.info(path).samplerate
    except ImportError:
        return _read_wav(path)[1]


class SpectrogramExtractor(torch.nn.Module):
    """Log-mel features (data_loader.py:33-88): dither 1e-5, pre-emphasis 0.97, STFT (n_fft = next power of two of the
    window, hop = window_stride, centred, reflect), power, mel, log1p(. + 2^-24), per-feature mean / (unbiased std + 1e-5)
    over time.  Buffers ``fb`` [1, n_mels, n_fft/2+1] and ``window`` carry the reference's names."""
    dithering = 1e-5
    preemph = 0.97
    epsilon = 1e-5
    log_zero_guard_value = 2 ** -24

    def __init__(self, audio_conf, mel_spec=64, use_cuda=False, device=None):
        super().__init__()
        sr = audio_conf['sample_rate']
        self.win_length = int(sr * audio_conf['window_size'])
        self.hop_length = int(sr * audio_conf['window_stride'])
        self.n_fft = 2 ** math.ceil(math.log2(self.win_length))
        if mel_spec is None:
            raise ValueError('mel_spec is required (the reference builds a mel filterbank unconditionally, data_loader.py:39-43)')
        self.n_mels = int(mel_spec)
        fb = torch.from_numpy(mel_filterbank(sr, self.n_fft, self.n_mels, 0.0, sr / 2)).unsqueeze(0)
        self.register_buffer('fb', fb)
        fn = _TORCH_WINDOWS.get(audio_conf['window'], None)
        window = fn(self.win_length, periodic=False).float() if fn else torch.ones(self.win_length)
        self.register_buffer('window', window)
        self.register_buffer('_fbT', fb[0].t().contiguous(), persistent=False)
        nz = fb[0] != 0                                   # run of non-zero bins per filter: [first, one past last)
        first = torch.where(nz.any(1), nz.float().argmax(1), torch.zeros(self.n_mels, dtype=torch.long))
        last = torch.where(nz.any(1), fb.shape[2] - nz.flip(1).float().argmax(1), torch.zeros(self.n_mels, dtype=torch.long))
        self.register_buffer('_fb_range', torch.stack([first, last], 1).to(torch.int32).contiguous(), persistent=False)
        if device is None:
            if not torch.cuda.is_available():
                raise RuntimeError('SpectrogramExtractor runs on MI355X only (HIP kernels, no CPU path)')
            device = torch.device('cuda', torch.cuda.current_device())
        self.to(device)

    def n_frames(self, n_samples: int) -> int:
        return 1 + n_samples // self.hop_length

    # ------------------------------------------------------------------ batched device path
    def _launch(self, audio: torch.Tensor, lens: torch.Tensor, noise: Optional[torch.Tensor], take_log: bool):
        n, lmax = audio.shape
        tmax = self.n_frames(int(lmax))
        out = torch.empty(n, tmax, self.n_mels, dtype=torch.float32, device=audio.device)
        check(lib.w2l_logmel(ptr(audio), ptr(lens), ptr(noise), self.dithering, self.preemph, n, lmax, ptr(self.window),
                             self.win_length, self.n_fft, self.hop_length, ptr(self._fbT), ptr(self._fb_range), self.n_mels,
                             int(take_log),
                             float(self.log_zero_guard_value), ptr(out), tmax, stream_ptr()), 'w2l_logmel')
        return out, tmax

    def _stage(self, signals: Sequence, noise):
        dev = self.fb.device
        if dev.type != 'cuda':
            raise RuntimeError('SpectrogramExtractor runs on MI355X only (HIP kernels, no CPU path)')
        arrs = [np.asarray(s.detach().cpu() if torch.is_tensor(s) else s, dtype=np.float32).reshape(-1) for s in signals]
        lens = np.array([a.shape[0] for a in arrs], dtype=np.int32)
        if lens.min() <= self.n_fft // 2:
            raise ValueError(f'an utterance of {int(lens.min())} samples is shorter than the STFT reflect padding '
                             f'({self.n_fft // 2}); torch.stft rejects it in the reference too')
        host = torch.zeros(len(arrs), int(lens.max()), dtype=torch.float32).pin_memory()
        for i, a in enumerate(arrs):
            host[i, :a.shape[0]] = torch.from_numpy(a)
        audio = host.to(dev, non_blocking=True)
        lens_d = torch.from_numpy(lens).to(dev, non_blocking=True)
        if noise is None:
            noise_d = torch.randn(audio.shape, dtype=torch.float32, device=dev) if self.dithering > 0 else None
        elif noise is False:
            noise_d = None
        else:
            noise_d = torch.zeros_like(audio)
            for i, z in enumerate(noise):
                z = torch.as_tensor(np.asarray(z, dtype=np.float32))
                noise_d[i, :z.shape[0]] = z.to(dev)
        return audio, lens_d, noise_d, lens

    def extract_batch(self, signals: Sequence, noise=None):
        """signals: N 1-D float arrays (any lengths).  noise: None = draw the dither on the device, False = no dither, or N
        arrays of N(0,1) draws (parity tests inject the reference's).  Returns (inputs fp32 [N, n_mels, T_max] on the
        device, zero beyond each utterance's frames; input_lengths IntTensor [N] on the host) -- _collator's layout."""
        audio, lens_d, noise_d, lens = self._stage(signals, noise)
        logmel, tmax = self._launch(audio, lens_d, noise_d, True)
        n = audio.shape[0]
        mean = torch.empty(n, self.n_mels, dtype=torch.float32, device=audio.device)
        std = torch.empty_like(mean)
        out = torch.empty(n, self.n_mels, tmax, dtype=torch.float32, device=audio.device)
        check(lib.w2l_feature_normalize(ptr(logmel), ptr(lens_d), self.hop_length, n, tmax, self.n_mels, float(self.epsilon),
                                        ptr(mean), ptr(std), ptr(out), stream_ptr()), 'w2l_feature_normalize')
        return out, torch.from_numpy(1 + lens // self.hop_length).to(torch.int32)

    # ------------------------------------------------------------------ the reference's per-utterance methods
    def _get_spect(self, audio, noise=None):
        """mel power spectrogram [1, n_mels, T] (data_loader.py:64-72)"""
        a, lens_d, noise_d, _ = self._stage([audio], None if noise is None else [noise])
        out, _ = self._launch(a, lens_d, noise_d, False)
        return out.transpose(1, 2).contiguous()

    def extract(self, signal, noise=None):
        """normalised log-mel features [n_mels, T] (data_loader.py:75-88), on the extractor's device"""
        out, _ = self.extract_batch([signal], None if noise is None else [noise])
        return out[0]


class SpectrogramDataset(Dataset):
    """Manifest dataset (data_loader.py:90-147): a .csv (first column = index; columns audio_filepath, text[, offset,
    duration]) or JSON lines.  ``dataset[i]`` -> (spect [n_mels, T], target ids, path, transcript) as in the reference;
    ``dataset.raw(i)`` -> the same with the raw samples instead of features (what BatchAudioDataLoader batches)."""

    def __init__(self, manifest_filepath, audio_conf, labels, mel_spec=None, use_cuda=False):
        super().__init__()
        rows = self._read_manifest(manifest_filepath)
        self.rows = rows
        self.size = len(rows)
        self.window_stride = audio_conf['window_stride']
        self.window_size = audio_conf['window_size']
        self.sample_rate = audio_conf['sample_rate']
        self.use_cuda = use_cuda
        self.mel_spec = mel_spec
        self.labels_map = dict([(labels[i], i) for i in range(len(labels))])
        self.validate_sample_rate()
        self.extractor = SpectrogramExtractor(audio_conf, mel_spec, use_cuda)

    @staticmethod
    def _read_manifest(path) -> List[dict]:
        if path.endswith('.csv'):
            import pandas as pd
            rows = pd.read_csv(path, index_col=0).to_dict('records')
        else:
            with open(path) as f:
                rows = [json.loads(line) for line in f if line.strip()]
        for r in rows:
            r.setdefault('offset', 0)
            r.setdefault('duration', -1)
        return rows

    def _target(self, transcript):
        # filter(None, ...) drops unknown characters AND label 0, the blank (data_loader.py:127)
        return list(filter(None, [self.labels_map.get(c) for c in list(transcript)]))

    def raw(self, index):
        r = self.rows[index]
        audio = load_audio(r['audio_filepath'], r['duration'], r['offset'])
        return audio, self._target(r['text']), r['audio_filepath'], r['text']

    def __getitem__(self, index):
        audio, target, path, text = self.raw(index)
        return self.extractor.extract(audio), target, path, text

    def parse_audio(self, audio_path, duration, offset):
        return self.extractor.extract(load_audio(audio_path, duration, offset))

    def validate_sample_rate(self):
        path = self.rows[0]['audio_filepath']
        sr = _sample_rate(path)
        assert sr == self.sample_rate, 'Expected sample rate %d but found %d in first file' % (self.sample_rate, sr)

    def __len__(self):
        return self.size

    def data_channels(self):
        return self.mel_spec or int(1 + (int(self.sample_rate * self.window_size) / 2))


def _pad_targets(targets):
    target_lengths = torch.IntTensor([len(t) for t in targets])
    longest = int(target_lengths.max()) if len(targets) else 0
    tg = torch.zeros(len(targets), longest, dtype=torch.int32)
    for i, t in enumerate(targets):
        if len(t):
            tg[i, :len(t)] = torch.as_tensor(list(t), dtype=torch.int32)
    return tg, target_lengths


def _collator(batch):
    """(spect, target, path, text) items -> (inputs [N, F, T_max] right-zero-padded, input_lengths, targets [N, S_max]
    zero-padded int32, target_lengths, paths, texts) -- data_loader.py:149-158.  Spectrograms may live on the device."""
    inputs, targets, file_paths, texts = zip(*batch)
    inputs = [torch.as_tensor(x) for x in inputs]
    input_lengths = torch.IntTensor([x.shape[1] for x in inputs])
    longest = int(input_lengths.max())
    out = torch.zeros(len(inputs), inputs[0].shape[0], longest, dtype=torch.float32, device=inputs[0].device)
    for i, x in enumerate(inputs):
        out[i, :, :x.shape[1]] = x
    tg, target_lengths = _pad_targets(targets)
    return out, input_lengths, tg, target_lengths, file_paths, texts


class _RawItems(Dataset):
    def __init__(self, ds: SpectrogramDataset):
        self.ds = ds

    def __len__(self):
        return len(self.ds)

    def __getitem__(self, i):
        return self.ds.raw(i)


class BatchAudioDataLoader(DataLoader):
    """DataLoader yielding the reference's 6-tuple batches (data_loader.py:160-163).  For a SpectrogramDataset the
    features of the whole batch are computed on the GPU in one pass from the raw audio."""

    def __init__(self, dataset, *args, **kwargs):
        self._spect_ds = dataset if isinstance(dataset, SpectrogramDataset) else None
        if self._spect_ds is not None:
            if kwargs.get('num_workers', 0):
                raise ValueError('GPU feature extraction runs in the loader process: num_workers must be 0')
            super().__init__(_RawItems(dataset), *args, **kwargs)
            self.collate_fn = self._device_collate
        else:
            super().__init__(dataset, *args, **kwargs)
            self.collate_fn = _collator

    def _device_collate(self, batch):
        audio, targets, file_paths, texts = zip(*batch)
        inputs, input_lengths = self._spect_ds.extractor.extract_batch(audio)
        tg, target_lengths = _pad_targets(targets)
        return inputs, input_lengths, tg, target_lengths, file_paths, texts
