"""One-node data-parallel launcher: one process per GPU, torchrun's environment contract.

The reference delegates this to Lightning (``Trainer(gpus=N)``, README.md:40, train.py:34-37), which re-executes the
training script once per device.  Here the parent process -- which must not have touched the GPU -- starts N children of the
SAME command line with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, waits for them and returns the
worst exit code.  Children are ordinary subprocesses (never ``exec`` from a process that has initialised HIP).  This
module imports nothing but the standard library so that ``bench.py`` / ``train.py`` can use it before the package (and
with it libw2l_hip.so and the HIP runtime) is loaded."""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import time
from typing import List, Optional, Sequence


def under_launcher() -> bool:
    """True inside a rank started by torchrun / spawn_ranks (the rendezvous variables are present)"""
    return 'WORLD_SIZE' in os.environ and 'RANK' in os.environ


def free_port() -> int:
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def visible_gpu_count() -> Optional[int]:
    """GPUs this process's children can open, WITHOUT any HIP / HSA call (a launch parent must not initialise the runtime:
    torch.cuda.device_count() falls back to hipGetDeviceCount -- which opens KFD -- whenever amdsmi discovery fails).
    Counted from the KFD topology in sysfs (nodes with a gfx target and SIMDs are GPUs; CPUs report 0), narrowed by
    HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES when one is set.  None = unknown (no sysfs here): the
    caller then lets each rank find out for itself after set_device."""
    for var in ('HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES'):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(',') if x.strip() != ''])
    root = '/sys/class/kfd/kfd/topology/nodes'
    try:
        nodes = os.listdir(root)
    except OSError:
        return None
    n = 0
    for node in nodes:
        try:
            props = dict(line.split()[:2] for line in open(os.path.join(root, node, 'properties')) if len(line.split()) >= 2)
        except OSError:
            continue
        if int(props.get('simd_count', '0')) > 0 and int(props.get('gfx_target_version', '0')) > 0:
            n += 1
    return n


def rank_env(rank: int, world: int, port: int, base: Optional[dict] = None) -> dict:
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC: RCCL's intra-node transport on this driver
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // world)))
    return env


def spawn_ranks(world: int, argv: Sequence[str], timeout: Optional[float] = None, poll_s: float = 0.2) -> int:
    """Run ``argv`` once per rank; rank 0 inherits stdout (its single JSON line / its logs reach the caller unchanged),
    every rank inherits stderr.  If a rank fails the others are terminated (a rank waiting in a collective for a dead peer
    would hang until RCCL's timeout).  Returns 0 iff every rank exited 0."""
    port = free_port()
    procs: List[subprocess.Popen] = []
    for r in range(world):
        out = None if r == 0 else subprocess.DEVNULL
        procs.append(subprocess.Popen(list(argv), env=rank_env(r, world, port), stdout=out))
    t0 = time.time()
    rc = 0
    alive = set(range(world))
    try:
        while alive:
            for r in sorted(alive):
                code = procs[r].poll()
                if code is None:
                    continue
                alive.discard(r)
                if code != 0:
                    rc = rc or code
                    print(f'[launch] rank {r} exited with code {code}; stopping the other ranks', file=sys.stderr)
                    for o in alive:
                        procs[o].terminate()
            if timeout is not None and time.time() - t0 > timeout and alive:
                print(f'[launch] timeout after {timeout:.0f}s; stopping ranks {sorted(alive)}', file=sys.stderr)
                rc = rc or 124
                for o in alive:
                    procs[o].terminate()
                timeout = None
            if alive:
                time.sleep(poll_s)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
            p.wait()
    return rc
