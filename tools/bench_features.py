#!/usr/bin/env python3
"""Timing of the feature front-end on the BASELINE shape (32 x 10 s of 16 kHz audio -> [32, 64, 1001]):
HIP-event time of the three launches with the audio already resident, vs algorithmic bytes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wav2letter_pytorch_amd._lib import check, lib, ptr, stream_ptr  # noqa: E402
from wav2letter_pytorch_amd.data.data_loader import SpectrogramExtractor  # noqa: E402

ext = SpectrogramExtractor(dict(window='hamming', window_stride=0.01, window_size=0.02, sample_rate=16000), 64)
N, L = 32, 160000
audio = 0.1 * torch.randn(N, L, device='cuda')
noise = torch.randn(N, L, device='cuda')
lens = torch.full((N,), L, dtype=torch.int32, device='cuda')
T = 1 + L // 160
mean = torch.empty(N, 64, device='cuda'); std = torch.empty_like(mean)
out = torch.empty(N, 64, T, device='cuda')


def run():
    lm, _ = ext._launch(audio, lens, noise, True)
    check(lib.w2l_feature_normalize(ptr(lm), ptr(lens), 160, N, T, 64, 1e-5, ptr(mean), ptr(std), ptr(out), stream_ptr()))


for _ in range(3):
    run()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20):
    run()
e.record()
torch.cuda.synchronize()
ms = s.elapsed_time(e) / 20
alg = N * L * 4 * 2 + 3 * N * T * 64 * 4 + N * T * 64 * 4      # audio + noise read; logmel write + 2 reads; output write
print(f'features: {ms * 1e3:.1f} us per batch of {N} x {L / 16000:.0f} s = {N * T / ms * 1e3 / 1e6:.1f} M frames/s; '
      f'algorithmic {alg / 1e6:.1f} MB -> {alg / ms / 1e9 * 1e3:.0f} GB/s')
