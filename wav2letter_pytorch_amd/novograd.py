"""Novograd (reference: novograd.py:11-114, after NVIDIA's Jasper optimizers): Adam-like first moment with a
PER-TENSOR second moment -- one scalar per parameter tensor, the running average of ||g||^2.

    v   = ||g||^2                      on the first step, else  beta2 * v + (1 - beta2) * ||g||^2
    g'  = g / (sqrt(v) + eps) + weight_decay * p          (amsgrad: v is replaced by its running max)
    g' *= (1 - beta1)                  if grad_averaging
    m   = beta1 * m + g' ;   p -= lr * m

Device-resident: the reference's ``if exp_avg_sq == 0`` (novograd.py:92) is a host sync on every tensor and
step; here the first-step case is selected on the device.  Conv weights in the step engine's tap-major layout take a
fused path (w2l_novograd_pack: norm, second-moment update, parameter update and the bf16 operand pack of the next
forward in three launches); everything else runs the same rule as torch ops."""
from __future__ import annotations

import torch
from torch.optim import Optimizer


def _is_tap_major(t: torch.Tensor) -> bool:
    if t.dim() != 3:
        return False
    co, ci, kw = t.shape
    return t.stride() == (ci, 1, co * ci)


class Novograd(Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.95, 0), eps=1e-8, weight_decay=0, grad_averaging=False,
                 amsgrad=False):
        if not 0.0 <= lr:
            raise ValueError("Invalid learning rate: {}".format(lr))
        if not 0.0 <= eps:
            raise ValueError("Invalid epsilon value: {}".format(eps))
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError("Invalid beta parameter at index 0: {}".format(betas[0]))
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError("Invalid beta parameter at index 1: {}".format(betas[1]))
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, grad_averaging=grad_averaging,
                        amsgrad=amsgrad)
        super(Novograd, self).__init__(params, defaults)

    fused = True            # use w2l_novograd_pack for eligible conv weights (False: torch ops only)

    def _fused_step(self, p, grad, state, group) -> bool:
        if not (p.is_cuda and p.dtype == torch.float32 and _is_tap_major(p) and grad.stride() == p.stride()
                and grad.dtype == torch.float32 and p.shape[0] % 64 == 0 and p.shape[1] % 64 == 0
                and state['exp_avg'].stride() == p.stride()):
            return False
        from . import engine as E
        from ._lib import check, lib, ptr, stream_ptr
        cout, cin, kw = p.shape
        dev = p.device
        cache = getattr(p, '_w2l_pack', None)
        if cache is None:
            cache = E._Volatile()
            p._w2l_pack = cache
        precise = any(k for k in cache)
        old = cache.get(precise)
        if old is not None and old.fwd_hi.device == dev and old.coutp == cout and old.cinp == cin:
            fwd_hi, fwd_lo, dgr_hi, dgr_lo = old.fwd_hi, old.fwd_lo, old.dgr_hi, old.dgr_lo
        else:
            fwd_hi = torch.empty(kw, cout, cin, dtype=torch.bfloat16, device=dev)
            dgr_hi = torch.empty(kw, cin, cout, dtype=torch.bfloat16, device=dev)
            fwd_lo = torch.empty_like(fwd_hi) if precise else None
            dgr_lo = torch.empty_like(dgr_hi) if precise else None
        scratch = state.get('_scratch')
        if scratch is None or scratch.device != dev:
            scratch = torch.empty(1025, dtype=torch.float32, device=dev)
            state['_scratch'] = scratch
        beta1, beta2 = group['betas']
        check(lib.w2l_novograd_pack(ptr(p), ptr(grad), ptr(state['exp_avg']), ptr(state['exp_avg_sq']),
                                    ptr(state['max_exp_avg_sq']) if group['amsgrad'] else None, ptr(scratch), scratch.numel(),
                                    float(group['lr']), float(beta1), float(beta2), float(group['eps']),
                                    float(group['weight_decay']), int(bool(group['grad_averaging'])), cout, cin, kw,
                                    ptr(fwd_hi), ptr(fwd_lo), ptr(dgr_hi), ptr(dgr_lo), stream_ptr()), 'w2l_novograd_pack')
        torch.autograd.graph.increment_version(p)                    # p changed through its raw pointer
        cache.clear()
        cache[precise] = E._PackedW(p._version, fwd_hi, fwd_lo, dgr_hi, dgr_lo, cin, cout, p.data_ptr())
        return True

    def state_dict(self):
        sd = super().state_dict()
        for st in sd['state'].values():
            st.pop('_scratch', None)                                 # workspace, not optimizer state
        return sd

    def __setstate__(self, state):
        super(Novograd, self).__setstate__(state)
        for group in self.param_groups:
            group.setdefault('amsgrad', False)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            beta1, beta2 = group['betas']
            for p in group['params']:
                if p.grad is None:
                    continue
                grad = p.grad
                if grad.is_sparse:
                    raise RuntimeError('Sparse gradients are not supported.')
                state = self.state[p]
                if len(state) == 0:
                    state['step'] = 0
                    state['exp_avg'] = torch.zeros_like(p)           # (preserves the tap-major strides of conv weights)
                    state['exp_avg_sq'] = torch.zeros([], device=p.device)
                    if group['amsgrad']:
                        state['max_exp_avg_sq'] = torch.zeros([], device=p.device)
                state['step'] += 1
                if self.fused and self._fused_step(p, grad, state, group):
                    continue
                v = state['exp_avg_sq']
                norm = grad.float().pow(2).sum()
                v.copy_(torch.where(v == 0, norm, beta2 * v + (1 - beta2) * norm))
                if group['amsgrad']:
                    torch.maximum(state['max_exp_avg_sq'], v, out=state['max_exp_avg_sq'])
                    denom = state['max_exp_avg_sq'].sqrt() + group['eps']
                else:
                    denom = v.sqrt() + group['eps']
                g = grad / denom
                if group['weight_decay'] != 0:
                    g = g.add(p, alpha=group['weight_decay'])
                if group['grad_averaging']:
                    g = g * (1 - beta1)
                state['exp_avg'].mul_(beta1).add_(g)
                p.add_(state['exp_avg'], alpha=-group['lr'])
        return loss
