#!/usr/bin/env python3
"""Randomised end-to-end parity sweep for Jasper (not part of the test suite): random block tables (dense and separable,
residual branches, repeats, dilation, stride-2 prologue, kernel sizes the reference bumps to odd), ragged lengths, vs the CPU
oracle in the fp32 parity mode."""
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from gpu_helpers import build_jasper, compare_jasper_step  # noqa: E402
from oracle import w2l_oracle as O  # noqa: E402


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 25
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    prec = sys.argv[3] if len(sys.argv) > 3 else 'fp32'
    tol = dict(fp32=(1e-3, 1e-4, 1e-3, 1e-3), bf16=(3e-2, 2e-2, 8e-2, 2e-2))[prec]
    from wav2letter_pytorch_amd import Jasper
    from wav2letter_pytorch_amd.config import to_cfg
    for case in range(n_cases):
        nb = rng.randint(1, 3)
        blocks = []
        for b in range(nb):
            blk = dict(layer_size=rng.choice([32, 48, 64, 96, 128]), kernel_size=rng.choice([3, 5, 11, 12, 13, 28]),
                       stride=2 if (b == 0 and rng.random() < 0.6) else 1, residual=b > 0 and rng.random() < 0.7,
                       separable=rng.random() < 0.5, repeat=rng.choice([1, 1, 2, 3]))
            if blk['stride'] == 1 and rng.random() < 0.3:
                blk['dilation'] = 2
            blocks.append(blk)
        N = rng.choice([1, 2, 3, 4])
        T = rng.randint(90, 300)
        torch.manual_seed(3000 + case)
        labels = O.ENGLISH_LOWERCASE
        cfg = to_cfg(dict(name='jasper', mid_layers=nb, jasper_blocks=blocks, input_size=64, labels=labels, precision=prec,
                          audio_conf=dict(window='hamming', window_stride=0.01, window_size=0.02, sample_rate=16000),
                          decoder=dict(_target_='decoder.GreedyDecoder', labels=labels)))
        sd = {k: v.detach().clone() for k, v in Jasper(cfg).state_dict().items()}
        scale = 2 if blocks[0]['stride'] == 2 else 1
        x, il, tg, tl = O.synthetic_batch(N, T, seed=4000 + case, s_lo=1, s_hi=max(2, T // (8 * scale)), scaling=scale)
        for n in range(1, N):
            il[n] = rng.randint(T // 2, T)
            x[n, :, int(il[n]):] = 0
            tl[n] = min(int(tl[n]), max(1, int(il[n]) // (4 * scale)))
            tg[n, int(tl[n]):] = 0
        try:
            model = build_jasper(blocks, sd, prec).train()
            errs, stats, out, out_lens = compare_jasper_step(model, blocks, sd, x, il, tg, tl, prec)
        except Exception as e:                      # noqa: BLE001
            print(f'case {case}: blocks={blocks} N={N} T={T}: EXCEPTION {type(e).__name__}: {str(e)[:200]}')
            continue
        g = max(v for k_, v in errs.items() if k_ not in ('log_probs', 'loss'))
        st = max(stats.values()) if stats else 0.0
        bad = errs['log_probs'] >= tol[0] or errs['loss'] >= tol[1] or g >= tol[2] or st >= tol[3]
        print(f'case {case:2d}: {"FAIL" if bad else "ok  "} N={N} T={T} lp={errs["log_probs"]:.1e} loss={errs["loss"]:.1e} '
              f'grad={g:.1e} stat={st:.1e} blocks={[(b["layer_size"], b["kernel_size"], b["stride"], b.get("dilation", 1), b["repeat"], int(b["residual"]), int(b["separable"])) for b in blocks]}',
              flush=True)


if __name__ == '__main__':
    main()
