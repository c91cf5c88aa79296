"""Mel filterbank for the feature front-end.  The reference takes it from
``librosa.filters.mel(sr, n_fft=n_fft, n_mels=n_mels, fmin=0, fmax=sr/2)`` (data/data_loader.py:39-43); librosa is
not a dependency of this package, so the matrix is built here from librosa's published definition: Slaney mel scale
(linear below 1 kHz, logarithmic above: 27 steps per factor 6.4), triangular filters between n_mels + 2 equally spaced
mel points, each scaled by 2 / (its bandwidth in Hz) ("slaney" area normalisation).  Init-time host code (numpy)."""
import numpy as np

_F_SP = 200.0 / 3
_MIN_LOG_HZ = 1000.0
_MIN_LOG_MEL = _MIN_LOG_HZ / _F_SP
_LOGSTEP = np.log(6.4) / 27.0


def hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    return np.where(f >= _MIN_LOG_HZ, _MIN_LOG_MEL + np.log(np.maximum(f, 1e-30) / _MIN_LOG_HZ) / _LOGSTEP, f / _F_SP)


def mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    return np.where(m >= _MIN_LOG_MEL, _MIN_LOG_HZ * np.exp(_LOGSTEP * (m - _MIN_LOG_MEL)), _F_SP * m)


def mel_filterbank(sample_rate, n_fft, n_mels, fmin=0.0, fmax=None):
    """float32 [n_mels, 1 + n_fft // 2]"""
    fmax = sample_rate / 2.0 if fmax is None else fmax
    bins = np.linspace(0.0, sample_rate / 2.0, 1 + n_fft // 2)
    pts = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    width = np.diff(pts)
    ramps = pts[:, None] - bins[None, :]
    rising = -ramps[:-2] / width[:-1, None]
    falling = ramps[2:] / width[1:, None]
    tri = np.maximum(0.0, np.minimum(rising, falling))
    tri *= (2.0 / (pts[2:] - pts[:-2]))[:, None]
    return tri.astype(np.float32)
