"""CPU oracle for the feature front-end and the augmentations -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product path (wav2letter_pytorch_amd/data) never does.

Restates, over torch CPU ops (where the reference's arithmetic lives) and numpy:
  * SpectrogramExtractor._get_spect / extract          data/data_loader.py:33-88
  * _collator                                          data/data_loader.py:149-158
  * SpecAugment / SpecCutout                           data/augmentations.py:11-99
  * librosa.filters.mel (Slaney scale, Slaney area normalisation) -- the reference calls
    ``librosa.filters.mel(sr, n_fft=..., n_mels=..., fmin=0, fmax=sr/2)`` (data_loader.py:39-43).
    librosa is a third-party dependency that is ABSENT from this image and unpinned in the
    reference's requirements.txt, so the filterbank below restates librosa's published algorithm
    (librosa/filters.py ``mel``, ``mel_frequencies``, ``hz_to_mel``; htk=False, norm='slaney').
    PARITY OF THE MEL MATRIX IS UNPINNED: no librosa output is available here to check it against.
    Everything downstream of the matrix is pinned by tests/golden/features.npz, which was generated
    by running the reference's own data_loader.py with this matrix plugged in for the missing
    librosa call (tests/golden/make_golden.py).
"""
from __future__ import annotations

import math

import numpy as np
import torch


# --------------------------------------------------------------------------- mel filterbank
def _hz_to_mel_slaney(f: float) -> float:
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = math.log(6.4) / 27.0
    if f >= min_log_hz:
        return min_log_mel + math.log(f / min_log_hz) / logstep
    return f / f_sp


def _mel_to_hz_slaney(m: float) -> float:
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = math.log(6.4) / 27.0
    if m >= min_log_mel:
        return min_log_hz * math.exp(logstep * (m - min_log_mel))
    return f_sp * m


def mel_filterbank(sr: float, n_fft: int, n_mels: int, fmin: float = 0.0, fmax: float | None = None) -> np.ndarray:
    """[n_mels, 1 + n_fft//2] float32 triangular filters, plain loops (librosa.filters.mel semantics)."""
    if fmax is None:
        fmax = sr / 2.0
    n_bins = 1 + n_fft // 2
    fft_f = [k * (sr / 2.0) / (n_bins - 1) for k in range(n_bins)]
    m_lo, m_hi = _hz_to_mel_slaney(fmin), _hz_to_mel_slaney(fmax)
    mel_f = [_mel_to_hz_slaney(m_lo + (m_hi - m_lo) * i / (n_mels + 1)) for i in range(n_mels + 2)]
    w = np.zeros((n_mels, n_bins), dtype=np.float64)
    for i in range(n_mels):
        lo, ce, hi = mel_f[i], mel_f[i + 1], mel_f[i + 2]
        enorm = 2.0 / (hi - lo)
        for k, f in enumerate(fft_f):
            up = (f - lo) / (ce - lo)
            down = (hi - f) / (hi - ce)
            w[i, k] = max(0.0, min(up, down)) * enorm
    return w.astype(np.float32)


# --------------------------------------------------------------------------- features
WINDOWS = {'hann': torch.hann_window, 'hamming': torch.hamming_window, 'blackman': torch.blackman_window,
           'bartlett': torch.bartlett_window}


def stft_params(audio_conf):
    """(win_length, hop, n_fft) -- data_loader.py:36-38"""
    win = int(audio_conf['sample_rate'] * audio_conf['window_size'])
    hop = int(audio_conf['sample_rate'] * audio_conf['window_stride'])
    n_fft = 2 ** math.ceil(math.log2(win))
    return win, hop, n_fft


def mel_power(audio: np.ndarray, noise: np.ndarray | None, audio_conf, n_mels: int = 64, dither: float = 1e-5,
              preemph: float = 0.97) -> torch.Tensor:
    """data_loader.py:64-72: dither, pre-emphasis, STFT (center, reflect), |.|^2, mel.  ``noise`` is the N(0,1)
    draw the reference takes from torch.randn (None = no dither).  Returns [n_mels, frames] fp32."""
    win, hop, n_fft = stft_params(audio_conf)
    x = torch.as_tensor(np.asarray(audio, dtype=np.float32))
    if noise is not None:
        x = x + torch.as_tensor(np.asarray(noise, dtype=np.float32)) * dither
    x = torch.cat((x[0].unsqueeze(0), x[1:] - preemph * x[:-1]), dim=0)
    window = WINDOWS[audio_conf['window']](win, periodic=False).float()
    spec = torch.stft(x, n_fft=n_fft, hop_length=hop, win_length=win, center=True, window=window, return_complex=True)
    mag = torch.sqrt(spec.real.pow(2) + spec.imag.pow(2))
    power = mag.pow(2)
    fb = torch.from_numpy(mel_filterbank(audio_conf['sample_rate'], n_fft, n_mels, 0.0, audio_conf['sample_rate'] / 2))
    return torch.matmul(fb, power)


def extract(audio, noise, audio_conf, n_mels: int = 64, eps: float = 1e-5, guard: float = 2.0 ** -24) -> np.ndarray:
    """data_loader.py:75-88: log1p(mel + 2^-24), then per-feature normalisation over time (unbiased std + 1e-5)."""
    spect = torch.log1p(mel_power(audio, noise, audio_conf, n_mels) + guard)
    mean = spect.mean(dim=1, keepdim=True)
    std = spect.std(dim=1, keepdim=True) + eps
    return ((spect - mean) / std).numpy()


def collate(specs, targets):
    """_collator (data_loader.py:149-158): right zero-pad to the longest; int32 lengths."""
    il = np.array([s.shape[1] for s in specs], dtype=np.int32)
    tl = np.array([len(t) for t in targets], dtype=np.int32)
    x = np.zeros((len(specs), specs[0].shape[0], int(il.max())), dtype=np.float32)
    tg = np.zeros((len(specs), int(tl.max()) if len(tl) else 0), dtype=np.int32)
    for i, (s, t) in enumerate(zip(specs, targets)):
        x[i, :, :s.shape[1]] = s
        tg[i, :len(t)] = t
    return x, il, tg, tl


# --------------------------------------------------------------------------- augmentations
def spec_augment_rects(shape, rng, freq_masks=1, time_masks=1, freq_width=15, time_width=50):
    """The rectangles SpecAugment.forward zeroes, in its draw order (augmentations.py:39-58):
    list of (n, f0, f1, t0, t1), half-open."""
    N, F, T = shape
    rects = []
    for n in range(N):
        for _ in range(freq_masks):
            left = int(rng.uniform(0, F - freq_width))
            w = int(rng.uniform(0, freq_width))
            rects.append((n, left, left + w, 0, T))
        for _ in range(time_masks):
            left = int(rng.uniform(0, T - time_width))
            w = int(rng.uniform(0, time_width))
            rects.append((n, 0, F, left, left + w))
    return rects


def spec_cutout_rects(shape, rng, rect_masks=5, rect_time=60, rect_freq=25):
    """SpecCutout.forward (augmentations.py:79-99).  Quirk kept: the width along the frequency axis is drawn from
    rect_time and the width along time from rect_freq (``w_x = uniform(0, rect_time)``, ``w_y = uniform(0, rect_freq)``)."""
    N, F, T = shape
    rects = []
    for n in range(N):
        for _ in range(rect_masks):
            rx = int(rng.uniform(0, F - rect_freq))
            ry = int(rng.uniform(0, T - rect_time))
            wx = int(rng.uniform(0, rect_time))
            wy = int(rng.uniform(0, rect_freq))
            rects.append((n, rx, rx + wx, ry, ry + wy))
    return rects


def apply_rects(x: np.ndarray, rects) -> np.ndarray:
    out = np.array(x, copy=True)
    for (n, f0, f1, t0, t1) in rects:
        out[n, f0:f1, t0:t1] = 0          # Python slice semantics, as the reference's mask[idx, a:b, c:d] = 1
    return out
