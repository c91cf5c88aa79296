#!/usr/bin/env python3
"""Randomised end-to-end parity sweep (not part of the test suite: ~1 min on the GPU): random Wav2Letter stacks (channel
widths that are not multiples of 64, kernel sizes 1-33, dilations 1-3, stride 1/2 first layer), batch sizes and lengths
through the public module surface vs the CPU oracle in the fp32 parity mode, with the test suite's tolerances."""
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from gpu_helpers import build_w2l, compare_step  # noqa: E402
from oracle import w2l_oracle as O  # noqa: E402


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    prec = sys.argv[3] if len(sys.argv) > 3 else 'fp32'          # bf16: production kernels incl. tuner / split-K choices
    tol = dict(fp32=(1e-3, 1e-4, 1e-3, 1e-3), bf16=(3e-2, 2e-2, 8e-2, 2e-2))[prec]
    worst = 0.0
    for case in range(n_cases):
        nl = rng.randint(1, 4)
        layers = []
        for i in range(nl):
            c = rng.choice([24, 48, 64, 96, 128, 160, 200, 256, 320])
            k = rng.choice([1, 3, 5, 7, 11, 13, 17, 29, 33])
            d = rng.choice([1, 1, 2, 3]) if k > 1 else 1
            s = rng.choice([1, 2]) if i == 0 else 1
            if s == 2:
                d = 1
            layers.append((c, k, s, d, 0.0))
        N = rng.choice([1, 2, 3, 5, 8])
        need = max((k - 1) * d for c, k, s, d, _ in layers) + 2
        T = rng.randint(max(40, 2 * need + 4), 420)
        sd = O.init_wav2letter_state(layers, seed=1000 + case)
        x, il, tg, tl = O.synthetic_batch(N, T, seed=2000 + case, s_lo=1, s_hi=max(2, T // 8))
        try:
            model = build_w2l(layers, sd, prec).train()
            errs, stats, out, out_lens, ref = compare_step(model, layers, sd, x, il, tg, tl, prec)
        except Exception as e:                      # noqa: BLE001
            print(f'case {case}: layers={layers} N={N} T={T}: EXCEPTION {type(e).__name__}: {e}')
            continue
        g = max(v for k_, v in errs.items() if k_ not in ('log_probs', 'loss'))
        st = max(stats.values()) if stats else 0.0
        bad = errs['log_probs'] >= tol[0] or errs['loss'] >= tol[1] or g >= tol[2] or st >= tol[3]
        worst = max(worst, errs['log_probs'], g)
        print(f'case {case:2d}: {"FAIL" if bad else "ok  "} layers={layers} N={N} T={T} lp={errs["log_probs"]:.1e} '
              f'loss={errs["loss"]:.1e} grad={g:.1e} stat={st:.1e}', flush=True)
    print('worst', worst)


if __name__ == '__main__':
    main()
