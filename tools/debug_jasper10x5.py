#!/usr/bin/env python3
"""Per-unit comparison of the Jasper 10x5 forward (device, fp32 parity mode or bf16) with the CPU oracle: prints, for
every post-ReLU activation in execution order, the error relative to the oracle's activation scale -- the first unit
whose error jumps is where a divergence starts.  usage: debug_jasper10x5.py [N] [T] [fp32|bf16] [ragged 0/1]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from gpu_helpers import build_jasper, device_step  # noqa: E402
from oracle import w2l_oracle as O  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    prec = sys.argv[3] if len(sys.argv) > 3 else 'fp32'
    ragged = (sys.argv[4] if len(sys.argv) > 4 else '1') == '1'
    from wav2letter_pytorch_amd import Jasper
    from wav2letter_pytorch_amd.defaults import jasper10x5_model
    cfg = jasper10x5_model()
    blocks = [dict(b) for b in cfg.jasper_blocks]
    torch.manual_seed(7)
    sd = {k: v.detach().clone() for k, v in Jasper(cfg).state_dict().items()}
    model = build_jasper(blocks, sd, prec).train()
    x, il, tg, tl = O.synthetic_batch(N, T, seed=99)
    if ragged and N > 1:
        il[1] = 801 * T // 1000
        x[1, :, int(il[1]):] = 0
    out, out_lens, loss, ectx = device_step(model, x, il, tg, tl)
    inter = []
    with torch.no_grad():
        lp, ol = O.jasper_forward(x, il, {k: v.clone() for k, v in sd.items()}, blocks, training=True, inter=inter)
    print('out_lens', out_lens.tolist(), ol.tolist())
    for i, (uc, a_ref) in enumerate(zip(ectx['units'], inter)):
        act = ectx['acts'][i + 1]
        a = act.hi[:, act.pad_l:act.pad_l + act.T, :act.C].float()
        if act.lo is not None:
            a = a + act.lo[:, act.pad_l:act.pad_l + act.T, :act.C].float()
        a = a.transpose(1, 2).cpu()
        if uc.lens_out is not None:      # the device masks where the NEXT conv would: compare on the valid frames only
            t = torch.arange(a.shape[2])[None, None, :]
            m = (t < uc.lens_out.cpu().long()[:, None, None])
            a_ref = a_ref * m
        err = float((a - a_ref).abs().max() / a_ref.abs().max())
        print(f'unit {i:2d} {uc.unit.main.name:16s} C={act.C:4d} T={act.T} res={uc.unit.res is not None} err={err:.2e} '
              f'scale={float(a_ref.abs().max()):.3g}', flush=True)
    e = float((out.cpu() - lp).abs().max() / lp.abs().max())
    print(f'log-probs err {e:.3e}; loss {float(loss):.5f} vs {float(O.ctc_criterion(lp, tg, ol, tl)):.5f}')


if __name__ == '__main__':
    main()
