"""Novograd (reference: novograd.py:11-114, after NVIDIA's Jasper optimizers): Adam-like first moment with a
PER-TENSOR second moment -- one scalar per parameter tensor, the running average of ||g||^2.

    v   = ||g||^2                      on the first step, else  beta2 * v + (1 - beta2) * ||g||^2
    g'  = g / (sqrt(v) + eps) + weight_decay * p          (amsgrad: v is replaced by its running max)
    g' *= (1 - beta1)                  if grad_averaging
    m   = beta1 * m + g' ;   p -= lr * m

Device-resident: the reference's ``if exp_avg_sq == 0`` (novograd.py:92) is a host sync on every tensor and
step; here the first-step case is selected on the device."""
from __future__ import annotations

import torch
from torch.optim import Optimizer


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
                    state['exp_avg'] = torch.zeros_like(p)
                    state['exp_avg_sq'] = torch.zeros([], device=p.device)
                    if group['amsgrad']:
                        state['max_exp_avg_sq'] = torch.zeros([], device=p.device)
                state['step'] += 1
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
