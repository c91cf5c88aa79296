"""Side streams of the step, chosen by measurement.

A training step keeps up to four streams busy at once: the main stream (forward, data gradients, BatchNorm), the
weight-gradient stream, the stream of the fused SGD updates, and -- data parallel -- the stream of the gradient
collectives.  HIP maps a process's streams onto a few hardware queues (GPU_MAX_HW_QUEUES) and those onto the command
processor's pipes in the order the streams first launch something; two streams that end up on one queue or pipe run
their kernels strictly one after the other (a dispatch larger than the chip holds its queue until its last workgroup is
placed).  Which stream collides with which depends on every stream creation in the process, torch's and RCCL's included:
with the RCCL communicator created before the first step (what the parameter broadcast of a multi-rank run does) the
weight-gradient stream landed beside the main stream and the step took 20.0 instead of 14.0 ms.

So a side stream is not "the next pool stream" but the first candidate that ``w2l_stream_probe`` shows running BESIDE
the main stream and beside every side stream chosen before it.  Off with W2L_STREAM_PROBE=0 (plain pool streams).
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Tuple

import torch

PROBE = os.environ.get('W2L_STREAM_PROBE', '1') != '0'
TRIES = 12                 # candidates drawn from torch's stream pool (32 per device) before settling for the best seen
SERIAL_FRAC = 0.35         # a stamp kernel that starts later than this fraction of the fill kernel's run was queued behind it
ROUNDS, SPIN_US = 6, 40    # the fill kernel: 6 rounds of 2 blocks per CU x 40 us ~ 0.25 ms

_chosen: Dict[int, Dict[str, 'torch.cuda.Stream']] = {}       # device index -> role -> stream
_main: Dict[int, 'torch.cuda.Stream'] = {}
report: List[Tuple[int, str, int, float]] = []                # (device, role, candidates tried, worst overlap fraction)


def overlap_fraction(a: 'torch.cuda.Stream', b: 'torch.cuda.Stream', dev: torch.device) -> float:
    """where in the life of a chip-filling kernel on ``a`` a kernel launched right after it on ``b`` started:
    ~0 = side by side, ~(ROUNDS-1)/ROUNDS or more = queued behind it.  SYNCHRONISES the device (warm-up only)."""
    import ctypes as C
    from ._lib import check, lib
    with torch.cuda.device(dev):
        stamps = torch.zeros(3, dtype=torch.int64, device=dev)
        torch.cuda.synchronize(dev)
        check(lib.w2l_stream_probe(C.c_void_p(a.cuda_stream), C.c_void_p(b.cuda_stream), C.c_void_p(stamps.data_ptr()),
                                   ROUNDS, SPIN_US), 'w2l_stream_probe')
        torch.cuda.synchronize(dev)
        t0, t1, tb = (int(v) for v in stamps.tolist())
    if t1 <= t0:
        return 0.0
    return (tb - t0) / (t1 - t0)


def serialise(a: 'torch.cuda.Stream', b: 'torch.cuda.Stream', dev: torch.device, reps: int = 2) -> float:
    """the worst overlap fraction of the pair over both directions and ``reps`` repetitions: the relation is neither
    symmetric nor perfectly steady (tools/stream_map.py: main -> s 0.87 in one pass and 0.04 in the next while s -> main
    stayed at 0.85), and one bad direction is enough to lose the overlap"""
    worst = 0.0
    for _ in range(reps):
        for x, y in ((a, b), (b, a)):
            worst = max(worst, overlap_fraction(x, y, dev))
            if worst >= SERIAL_FRAC:
                return worst
    return worst


def concurrent_stream(dev: torch.device, role: str, main: Optional['torch.cuda.Stream'] = None) -> 'torch.cuda.Stream':
    """the process-wide stream of ``role`` ('wgrad', 'sgd', 'collectives') on ``dev``: one per role for the life of the
    process, concurrent (by measurement) with the main stream and with the roles chosen before it.  ``main``: the stream
    the step itself runs on (first caller decides; default: the current stream)."""
    dev = torch.device(dev)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    dev = torch.device('cuda', idx)
    reg = _chosen.setdefault(idx, {})
    st = reg.get(role)
    if st is not None:
        return st
    if main is not None:
        _main.setdefault(idx, main)
    main = _main.setdefault(idx, torch.cuda.current_stream(dev))
    if not PROBE or torch.cuda.is_current_stream_capturing():
        st = torch.cuda.Stream(device=dev)
        reg[role] = st
        return st
    against = [main] + [s for s in reg.values()]
    best = None
    tried = 0
    for tried in range(1, TRIES + 1):
        cand = torch.cuda.Stream(device=dev)
        worst = 0.0
        for s in against:
            worst = max(worst, serialise(s, cand, dev))
            if worst >= SERIAL_FRAC:
                break
        if best is None or worst < best[0]:
            best = (worst, cand)
        if worst < SERIAL_FRAC:
            break
    reg[role] = best[1]
    report.append((idx, role, tried, round(best[0], 3)))
    if best[0] >= SERIAL_FRAC:
        import warnings
        warnings.warn(f'wav2letter_pytorch_amd.streams: none of {tried} candidate streams for role {role!r} on cuda:{idx} runs '
                      f'beside the {len(against)} busy stream(s) (best overlap fraction {best[0]:.2f}): kernels of that role '
                      f'will queue behind another stream and the step loses that overlap (more busy streams than the command '
                      f'processor runs side by side? see tools/stream_map.py)')
    return best[1]


def chosen(dev=None) -> Dict[str, 'torch.cuda.Stream']:
    idx = torch.cuda.current_device() if dev is None else torch.device(dev).index
    return dict(_chosen.get(idx, {}))
