"""nn.CTCLoss replacement backed by the alpha-beta HIP kernels (base_asr_models.py:23,81,90)."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib
from ._lib import check, lib, ptr, stream_ptr


class _CTCFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, log_probs, targets, input_lengths, target_lengths, blank, zero_infinity):
        # log_probs arrives as [T, N, C] (the reference passes out.transpose(0,1)); kernels are batch-major
        _lib.require_device(log_probs)
        lp = log_probs.detach().transpose(0, 1)
        if not lp.is_contiguous() or lp.dtype != torch.float32:
            lp = lp.contiguous().float()
        n, t, c = lp.shape
        dev = lp.device
        if targets.dim() != 2:
            raise NotImplementedError('CTCLoss: targets must be 2-D [N, S_max] (padded), as the collator produces')
        tg = targets.to(device=dev, dtype=torch.int32).contiguous()
        il = torch.as_tensor(input_lengths).to(device=dev, dtype=torch.int32).contiguous()
        tl = torch.as_tensor(target_lengths).to(device=dev, dtype=torch.int32).contiguous()
        smax = tg.shape[1]
        ws = torch.empty(max(int(lib.w2l_ctc_workspace_bytes(n, t, smax)), 4), dtype=torch.uint8, device=dev)
        nll = torch.empty(n, dtype=torch.float32, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        grad = torch.empty(n, t, c, dtype=torch.float32, device=dev)
        check(lib.w2l_ctc_loss(ptr(lp), ptr(tg), ptr(il), ptr(tl), n, t, c, smax, int(blank), int(zero_infinity),
                               ptr(nll), ptr(loss), ptr(grad), ptr(ws), stream_ptr()), 'w2l_ctc_loss')
        ctx.grad = grad
        ctx.nll = nll
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        grad = ctx.grad * g
        return grad.transpose(0, 1), None, None, None, None, None


class CTCLoss(nn.Module):
    """CTCLoss(blank=0, reduction='mean', zero_infinity=True) -- the only configuration the
    reference instantiates.  Input layout and semantics follow torch.nn.CTCLoss:
    log_probs [T, N, C], targets [N, S_max] padded, lengths [N]."""

    def __init__(self, blank=0, reduction='mean', zero_infinity=False):
        super().__init__()
        if reduction != 'mean':
            raise NotImplementedError("only reduction='mean' is implemented (base_asr_models.py:23)")
        self.blank = blank
        self.reduction = reduction
        self.zero_infinity = zero_infinity

    def forward(self, log_probs, targets, input_lengths, target_lengths):
        return _CTCFn.apply(log_probs, targets, input_lengths, target_lengths, self.blank, self.zero_infinity)
