"""Worker of test_gpu_model.py::test_data_parallel_two_ranks_on_one_gpu: one rank of a 2-rank data-parallel step.
Both ranks share cuda:0 (the test box has one GPU), so the collective backend is gloo -- RCCL refuses two ranks on one
device -- but everything else is the production path: broadcast_parameters, the step engine handing gradients to
GradReducer as they are produced, the pooled per-channel gradients, the fused SGD step after the averaging."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main_steps(out_path, defer, steps=int(os.environ.get('W2L_TEST_STEPS', '3'))):
    """``steps`` data-parallel training steps in bf16 mode with the top ``defer`` units' weight gradients held back for the
    next forward pass (optim.FusedSGD.defer_wgrad; 0 = the plain step): final parameters of every rank"""
    import torch.distributed as dist
    from gpu_helpers import build_w2l
    from oracle import w2l_oracle as O
    from wav2letter_pytorch_amd.distributed import GradReducer, broadcast_parameters, init_process_group_from_env
    torch.cuda.set_device(0)
    rank, world = init_process_group_from_env(backend='gloo')
    layers = [(128, 11, 2, 1, 0.0), (256, 13, 1, 1, 0.0), (128, 29, 1, 2, 0.0)]
    if os.environ.get('W2L_TEST_GROUP_STACK'):        # two consecutive stride-1, dilation-1 layers: a group of the backward pass
        layers = [(128, 11, 2, 1, 0.0), (256, 13, 1, 1, 0.0), (256, 13, 1, 1, 0.0), (128, 29, 1, 2, 0.0)]
    sd = O.init_wav2letter_state(layers, seed=60 + rank)
    model = build_w2l(layers, sd, 'bf16').train()
    broadcast_parameters(model)
    model.grad_reducer = GradReducer()
    model._cfg.optimizer.lr = 0.05
    opt = model.configure_optimizers()[0][0]
    opt.overlap = True
    opt.defer_wgrad(model, defer)
    x, il, tg, tl = O.synthetic_batch(4, 240, seed=70 + rank, s_lo=5, s_hi=20)
    held = []
    for _ in range(steps):
        opt.zero_grad(set_to_none=True)
        out, ol = model(x.cuda(), il)
        model.criterion(out.transpose(0, 1), tg, ol, tl).backward()
        held.append(len(model.engine()._deferred))
        opt.step()
    opt.join()
    torch.cuda.synchronize()
    grouped = sorted(set(v[0] for v in getattr(model.engine(), '_wg_seen', {}).values()))
    np.savez(out_path + f'.rank{rank}.npz', held=np.array(held), grouped=np.array(grouped),
             **{'p/' + k: v.detach().cpu().numpy() for k, v in model.named_parameters()})
    dist.barrier()
    dist.destroy_process_group()


def main():
    out_path = sys.argv[1]
    if len(sys.argv) > 2:
        return main_steps(out_path, int(sys.argv[2]))
    import torch.distributed as dist
    from gpu_helpers import build_w2l
    from oracle import w2l_oracle as O
    from wav2letter_pytorch_amd.distributed import GradReducer, broadcast_parameters, init_process_group_from_env
    torch.cuda.set_device(0)
    rank, world = init_process_group_from_env(backend='gloo')
    layers = [(128, 11, 2, 1, 0.0), (256, 13, 1, 1, 0.0), (128, 29, 1, 2, 0.0)]
    sd = O.init_wav2letter_state(layers, seed=60 + rank)          # different on purpose: the broadcast must fix it
    model = build_w2l(layers, sd, 'fp32').train()       # split-bf16 mode: no tuner-dependent bf16 roundings
    broadcast_parameters(model)
    model.grad_reducer = GradReducer()
    model._cfg.optimizer.lr = 0.05
    opt = model.configure_optimizers()[0][0]
    x, il, tg, tl = O.synthetic_batch(4, 240, seed=70 + rank, s_lo=5, s_hi=20)
    p0 = {k: v.detach().cpu().numpy().copy() for k, v in model.named_parameters()}
    out, ol = model(x.cuda(), il)
    loss = model.criterion(out.transpose(0, 1), tg, ol, tl)
    loss.backward()
    grads = {k: v.grad.detach().cpu().numpy().copy() for k, v in model.named_parameters()}
    opt.step()
    if hasattr(opt, 'join'):
        opt.join()
    torch.cuda.synchronize()
    p1 = {k: v.detach().cpu().numpy().copy() for k, v in model.named_parameters()}
    np.savez(out_path + f'.rank{rank}.npz', loss=float(loss), **{'p0/' + k: v for k, v in p0.items()},
             **{'g/' + k: v for k, v in grads.items()}, **{'p1/' + k: v for k, v in p1.items()})
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
