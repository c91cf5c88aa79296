"""Worker of test_gpu_features.py::test_native_rccl_helpers_one_rank: the C ABI's RCCL helpers (w2l_rccl_*) on a 1-rank
communicator -- RCCL refuses two ranks on one device and the test box has one GPU -- and a whole training step whose
gradients travel through them (GradReducer(native=True)).  Own process: it creates a process group."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    import torch.distributed as dist
    from gpu_helpers import build_w2l
    from oracle import w2l_oracle as O
    from wav2letter_pytorch_amd._lib import W2LError, lib
    from wav2letter_pytorch_amd.distributed import GradReducer, NativeComm, init_process_group_from_env
    torch.cuda.set_device(0)
    assert lib.w2l_rccl_available() == 1
    # one RCCL per process: the helpers must bind the copy torch itself loaded, not open a second instance
    used = os.path.realpath(lib.w2l_rccl_library().decode())
    mapped = {os.path.realpath(line.split()[-1]) for line in open('/proc/self/maps') if 'librccl' in line}
    assert used in mapped and len(mapped) == 1, (used, mapped)
    assert used.startswith(os.path.realpath(os.path.dirname(torch.__file__))), used
    # ---- the helpers on their own
    uid = NativeComm.unique_id()
    assert len(uid) == 128 and uid != NativeComm.unique_id()
    comm = NativeComm(0, 1, uid)
    import ctypes as C
    w = C.c_int(0)
    assert lib.w2l_rccl_world(comm._comm, C.byref(w)) == 0 and w.value == 1
    g = torch.Generator(device='cuda').manual_seed(1)
    for dtype in (torch.float32, torch.bfloat16):
        t = torch.randn(3 << 20, device='cuda', generator=g).to(dtype)
        ref = t.clone()
        comm.all_reduce(t, average=True)
        comm.all_reduce(t, average=False)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        comm.all_reduce(t, average=True, stream=side)
        comm.broadcast(t, root=0, stream=side)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        assert torch.equal(t, ref), dtype
    for bad in (lambda: comm.all_reduce(torch.zeros(4, device='cuda', dtype=torch.float16)),
                lambda: comm.all_reduce(torch.zeros(4, 4, device='cuda').t()),
                lambda: NativeComm(0, 1, b'short')):
        try:
            bad()
        except (TypeError, ValueError):
            pass
        else:
            raise AssertionError('bad argument accepted')
    try:
        NativeComm(3, 2, uid)
    except W2LError as e:
        assert 'outside world' in str(e)
    else:
        raise AssertionError('rank outside the world accepted')
    comm.close()
    comm.close()                                   # idempotent

    # ---- one training step, gradients through the native communicator vs no reducer at all
    rank, world = init_process_group_from_env(force=True)
    assert (rank, world) == (0, 1)
    layers = [(128, 11, 2, 1, 0.0), (256, 13, 1, 1, 0.0), (128, 29, 1, 2, 0.0)]
    sd = O.init_wav2letter_state(layers, seed=60)
    x, il, tg, tl = O.synthetic_batch(4, 240, seed=70, s_lo=5, s_hi=20)
    grads = []
    for native in (None, True, False):
        model = build_w2l(layers, sd, 'fp32').train()
        if native is not None:
            model.grad_reducer = GradReducer(force=True, native=native, small_bytes=1 << 12)
            assert model.grad_reducer.active and (model.grad_reducer._comm is not None) == native
        out, ol = model(x.cuda(), il)
        model.criterion(out.transpose(0, 1), tg, ol, tl).backward()
        torch.cuda.synchronize()
        grads.append({k: v.grad.detach().cpu().numpy().copy() for k, v in model.named_parameters()})
    for k in grads[0]:               # average over one rank: the gradient itself (split-K atomics: equal to fp32 rounding)
        ref = grads[0][k]
        for other in grads[1:]:
            assert np.linalg.norm(other[k] - ref) <= 1e-5 * max(np.linalg.norm(ref), 1e-12), k
    # ---- bf16 transport (W2L_DP_BF16=1) through the native communicator: the averaged gradient is the bf16 rounding
    model = build_w2l(layers, sd, 'fp32').train()
    red = GradReducer(force=True, native=True, small_bytes=1 << 12)
    red.bf16 = True
    model.grad_reducer = red
    out, ol = model(x.cuda(), il)
    model.criterion(out.transpose(0, 1), tg, ol, tl).backward()
    torch.cuda.synchronize()
    rounded = 0
    for k, v in model.named_parameters():
        ref = torch.from_numpy(grads[0][k])
        got = v.grad.detach().cpu()
        # conv weights travel alone, the per-channel gradients as one pooled message: either way a message of >= small_bytes
        # went as bf16, so a gradient is the fp32 value or its bf16 rounding (2^-9 relative)
        assert float((got - ref).norm()) <= 4e-3 * max(float(ref.norm()), 1e-12), k
        if ref.numel() * 4 >= (1 << 12):
            want = ref.to(torch.bfloat16).float()
            assert float((got - want).norm()) <= 1e-3 * max(float(want.norm()), 1e-12), k
            rounded += int(not torch.equal(got, ref))
    assert rounded >= 3
    dist.destroy_process_group()
    print('NATIVE_RCCL_OK')


if __name__ == '__main__':
    main()
