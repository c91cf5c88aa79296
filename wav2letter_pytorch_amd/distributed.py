"""Data-parallel gradient averaging over RCCL (torch.distributed backend "nccl" on ROCm),
one process per GPU, overlapped with the backward pass.

The reference has no collective call site of its own; with PL's ``Trainer(gpus=N)`` it would
run torch DDP: replicas with unsynchronised BatchNorm and a sum-all-reduce of every parameter
gradient scaled by 1/world (README.md:40).  Here the step engine hands each weight gradient
to ``GradReducer.on_grad`` the moment its wgrad kernel has been enqueued; the reducer
makes a side HIP stream wait on an event recorded at that point and launches the
all-reduce (op AVG) there, so the collective of layer L runs under the dgrad/wgrad kernels
of layers < L.  xGMI is point-to-point (7 links per GPU): one large message per conv weight
(up to 93 MB fp32) keeps every ring step bandwidth-bound; the ~80 tiny per-channel gradients
are flattened into one message at the end.  ``finish()`` makes the compute stream wait for the
side stream before the gradients are handed to autograd / the optimizer.
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.distributed as dist

SMALL_BYTES = 1 << 20


def init_process_group_from_env(backend: Optional[str] = None, force: bool = False):
    """torchrun-style rendezvous (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT).
    ``force`` creates a 1-rank group too (used to exercise the RCCL path on a single GPU)."""
    if dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world <= 1 and not force:
        return 0, 1
    os.environ.setdefault('RANK', '0')
    os.environ.setdefault('WORLD_SIZE', '1')
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29500')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if backend is None:
        # W2L_DIST_BACKEND=gloo: rehearsal of a multi-rank run on a box with a single GPU (RCCL refuses two ranks on one
        # device); production is nccl (= RCCL on ROCm)
        backend = os.environ.get('W2L_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
    if backend == 'nccl':
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
    dist.init_process_group(backend=backend)
    return dist.get_rank(), dist.get_world_size()


class NativeComm:
    """An RCCL communicator held through the C ABI (include/w2l_hip.h, ``w2l_rccl_*``): the exchange step as a host
    without torch.distributed would drive it.  ``GradReducer(native=True)`` / ``W2L_DP_NATIVE=1`` routes the gradient
    collectives through it (no Work objects, no watchdog thread: a collective is one more asynchronous launch on the
    reducer's stream); torch.distributed is then only the channel that carries rank 0's unique id to the other ranks."""

    ID_BYTES = 128

    def __init__(self, rank: int, world: int, unique_id: bytes):
        import ctypes as C
        from ._lib import check, lib
        if len(unique_id) != self.ID_BYTES:
            raise ValueError(f'an RCCL unique id has {self.ID_BYTES} bytes, got {len(unique_id)}')
        self.rank, self.world = rank, world
        self.rehearsal = False
        self._lib, self._check = lib, check
        comm = C.c_void_p()
        check(lib.w2l_rccl_init(unique_id, rank, world, C.byref(comm)), 'w2l_rccl_init')   # collective over the ranks
        self._comm = comm

    @staticmethod
    def unique_id() -> bytes:
        import ctypes as C
        from ._lib import check, lib
        buf = C.create_string_buffer(NativeComm.ID_BYTES)
        check(lib.w2l_rccl_unique_id(buf), 'w2l_rccl_unique_id')
        return buf.raw

    @classmethod
    def from_process_group(cls, group=None):
        """rank 0 draws the id; the existing process group (any backend) carries it to the others.

        Collective and fail-together: if rank 0 cannot draw an id every rank raises (nobody is left waiting in the
        broadcast).  One-GPU rehearsal (``W2L_DIST_BACKEND=gloo``: all ranks share cuda:0, and RCCL refuses two ranks of one
        communicator on one device): every process gets a ONE-rank communicator of its own -- the launch path (C ABI call
        on the reducer's stream, events) is the production one, but nothing is averaged across the ranks (``rehearsal``)."""
        rank, world = (dist.get_rank(group), dist.get_world_size(group)) if dist.is_initialized() else (0, 1)
        if world > 1 and os.environ.get('W2L_DIST_BACKEND') == 'gloo':
            comm = cls(0, 1, cls.unique_id())
            comm.rehearsal = True
            return comm
        box = [None]
        if rank == 0:
            try:
                box = [cls.unique_id()]
            except Exception as e:          # noqa: BLE001 -- carried to the other ranks below, raised on all of them
                box = [e]
        if world > 1:
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        if not isinstance(box[0], (bytes, bytearray)):
            raise RuntimeError(f'rank 0 could not draw an RCCL unique id: {box[0]!r}')
        return cls(rank, world, bytes(box[0]))

    def _stream(self, stream):
        import ctypes as C
        s = stream if stream is not None else torch.cuda.current_stream()
        return C.c_void_p(s.cuda_stream)

    def all_reduce(self, t: torch.Tensor, average: bool = True, stream=None):
        """in place, asynchronous on ``stream`` (default: torch's current stream); fp32 or bf16, dense"""
        if not (t.is_cuda and t.is_contiguous()):
            raise ValueError('NativeComm.all_reduce needs a dense device tensor')
        code = {torch.float32: 0, torch.bfloat16: 1}.get(t.dtype)
        if code is None:
            raise TypeError(f'NativeComm.all_reduce: fp32 or bf16, got {t.dtype}')
        import ctypes as C
        self._check(self._lib.w2l_rccl_all_reduce(self._comm, C.c_void_p(t.data_ptr()), t.numel(), code, int(average),
                                                  self._stream(stream)), 'w2l_rccl_all_reduce')

    def broadcast(self, t: torch.Tensor, root: int = 0, stream=None):
        if not (t.is_cuda and t.is_contiguous()):
            raise ValueError('NativeComm.broadcast needs a dense device tensor')
        import ctypes as C
        self._check(self._lib.w2l_rccl_broadcast(self._comm, C.c_void_p(t.data_ptr()), t.numel() * t.element_size(), root,
                                                 self._stream(stream)), 'w2l_rccl_broadcast')

    def close(self):
        comm, self._comm = self._comm, None
        if comm is not None:
            self._check(self._lib.w2l_rccl_destroy(comm), 'w2l_rccl_destroy')

    def __del__(self):
        import sys
        if sys.is_finalizing():    # interpreter shutdown: the HIP runtime may be gone; the driver reclaims the communicator
            return
        try:
            self.close()
        except Exception:
            pass


class _StreamWork:
    """Work-like handle of a collective launched through NativeComm: wait() = the current stream waits for the event
    recorded behind the collective on the reducer's stream"""

    def __init__(self, event):
        self._event = event

    def wait(self):
        torch.cuda.current_stream().wait_event(self._event)


class _Pending:
    """one collective started by GradReducer.start()"""

    def __init__(self, reducer, entry):
        self._reducer, self._entry = reducer, entry

    def finish(self):
        work, buf, need_div = self._entry
        work.wait()
        if isinstance(buf, tuple):           # bf16 transport: widen the averaged copy back into the fp32 gradient
            buf, half = buf
            buf.copy_(half)
            if half.is_cuda:
                half.record_stream(torch.cuda.current_stream(half.device))
        if need_div and self._reducer.world > 1:
            buf.div_(self._reducer.world)


class GradReducer:
    """Averages gradients across the ranks of ``group`` as they become ready."""

    def __init__(self, group=None, small_bytes: int = SMALL_BYTES, force: bool = False, native: Optional[bool] = None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (force and dist.is_initialized())
        self.small_bytes = small_bytes
        self._stream = None
        self._works = []
        self._small: List[torch.Tensor] = []
        self._keep = []
        backend = dist.get_backend(group) if dist.is_initialized() else None
        self._avg = dist.ReduceOp.AVG if backend == 'nccl' else None
        # W2L_DP_SERIALIZE=1: the stream that produced a gradient (the weight-gradient side stream) also waits for that
        # gradient's collective, so collectives and weight-gradient kernels alternate instead of running side by side: at
        # most two kernels compete for the CUs at any time (on one GPU three concurrent heavy kernels cost 18.8 instead of
        # 13.6 ms/step).  Off by default: RCCL's kernels are light, and the serial chain (4.6 ms of weight gradients + the
        # collectives) must stay shorter than the backward pass to remain hidden.  To be decided on an 8-GPU node.
        self.serialize = os.environ.get('W2L_DP_SERIALIZE', '0') == '1'
        # W2L_DP_BF16=1: large gradients travel as a bf16 copy (half the bytes on the xGMI ring: 306 instead of 612 MB per
        # step for the full Wav2Letter table) and are widened back in finish(); the average is then accurate to bf16's 8
        # bits, which is NOT what fp32 DDP computes -- an option for link-bound nodes, off by default
        self.bf16 = os.environ.get('W2L_DP_BF16', '0') == '1'
        # W2L_DP_NATIVE=1 / native=True: the collectives go through the C ABI's RCCL helpers (NativeComm) instead of
        # torch's ProcessGroupNCCL; same streams, same events, same result (ncclAvg).  Device tensors only.
        if native is None:
            native = os.environ.get('W2L_DP_NATIVE', '0') == '1'
        self._comm: Optional[NativeComm] = None
        if native and self.active and torch.cuda.is_available():
            comm = NativeComm.from_process_group(group)
            if comm.rehearsal:
                # one-rank stand-ins (W2L_DIST_BACKEND=gloo on a shared GPU) average NOTHING across the ranks: fine for
                # bench.py's plumbing rehearsal, which asks for them explicitly through set_native(), never for a trainer
                comm.close()
                import warnings
                warnings.warn('W2L_DP_NATIVE=1 with W2L_DIST_BACKEND=gloo: RCCL cannot span two ranks on one device; the '
                              'gradient collectives stay on torch.distributed (gloo)')
            else:
                self._comm = comm

    def set_native(self, comm: Optional[NativeComm]):
        """route the gradient collectives through ``comm`` (the C ABI's RCCL helpers) from the next step on, or back through
        torch.distributed (None).  Call between steps, with nothing in flight (after ``finish()`` and a stream sync)."""
        if self._works or self._small:
            raise RuntimeError('GradReducer.set_native with collectives in flight')
        self._comm = comm

    def _side_stream(self, device):
        if self._stream is None:
            from .streams import concurrent_stream
            # with NativeComm the collectives themselves run on this stream; with torch.distributed it only carries the
            # event waits (ProcessGroupNCCL launches on an internal stream of its own, out of this package's reach)
            self._stream = concurrent_stream(device, 'collectives')
        return self._stream

    def _all_reduce(self, t: torch.Tensor):
        # payload handed to the collectives since the caller last reset the counter (bench.py: bytes per step; a ring moves
        # 2 (N-1)/N of it over every link)
        self.wire_bytes = getattr(self, 'wire_bytes', 0) + t.numel() * t.element_size()
        if self._comm is not None and t.is_cuda:
            side = torch.cuda.current_stream(t.device)           # (the caller has made the reducer's stream current)
            self._comm.all_reduce(t, average=True, stream=side)
            ev = torch.cuda.Event()
            ev.record(side)
            return _StreamWork(ev), False
        if self._avg is not None:
            return dist.all_reduce(t, op=self._avg, group=self.group, async_op=True), False
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True), True

    def on_grad(self, param, grad: torch.Tensor, storage: Optional[torch.Tensor] = None):
        """``grad`` may be a strided view; ``storage`` is the dense buffer it lives in."""
        if not self.active:
            return
        buf = storage if storage is not None else grad
        if not buf.is_contiguous():
            raise ValueError('GradReducer needs the dense storage of a strided gradient view')
        if buf.numel() * buf.element_size() < self.small_bytes:
            self._small.append(buf)
            return
        self._launch(buf)

    def on_flat(self, buf: torch.Tensor):
        """a dense buffer holding many small gradients (the step engine's per-channel pool): one collective, in place"""
        if self.active:
            self._launch(buf)

    def _launch(self, buf: torch.Tensor):
        if self.bf16 and buf.dtype == torch.float32 and buf.numel() * 4 >= self.small_bytes:
            return self._launch_bf16(buf)
        if buf.is_cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(buf.device))
            side = self._side_stream(buf.device)
            side.wait_event(ev)
            with torch.cuda.stream(side):
                work, need_div = self._all_reduce(buf)
            if self.serialize:
                work.wait()                  # the producing stream resumes only after the collective
            # no record_stream: buf stays referenced in self._works until finish() has made the consumer stream wait
        else:
            work, need_div = self._all_reduce(buf)
        self._works.append((work, buf, need_div))

    def _launch_bf16(self, buf: torch.Tensor):
        if buf.is_cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(buf.device))
            side = self._side_stream(buf.device)
            side.wait_event(ev)
            with torch.cuda.stream(side):
                half = buf.to(torch.bfloat16)
                work, need_div = self._all_reduce(half)
        else:
            half = buf.to(torch.bfloat16)
            work, need_div = self._all_reduce(half)
        self._works.append((work, (buf, half), need_div))
        return work, True                    # "not final until finish()": the fp32 buffer is rewritten there

    def start(self, buf: torch.Tensor) -> '_Pending':
        """launch the all-reduce of ONE dense gradient buffer now (ordered after the current stream's work) and return a
        handle whose ``finish()`` makes the then-current stream wait for the averaged result.  Outside the backward pass's
        on_grad / finish() cycle: the step engine's deferred weight gradients (engine.flush_deferred) use it."""
        self._launch(buf)
        return _Pending(self, self._works.pop())

    def finish(self):
        """Flush the small gradients, then make the current stream wait for every collective."""
        if not self.active:
            return
        flat = None
        if self._small:
            flat = torch.cat([t.reshape(-1) for t in self._small])
            self._launch(flat)
        for work, buf, need_div in self._works:
            work.wait()                      # stream-level wait for NCCL works; blocking for gloo
            if isinstance(buf, tuple):       # bf16 transport: widen the averaged copy back into the fp32 gradient
                buf, half = buf
                buf.copy_(half)
                if half.is_cuda:
                    half.record_stream(torch.cuda.current_stream(half.device))   # allocated on the reducer's stream, read here
            if need_div and self.world > 1:
                buf.div_(self.world)
        if flat is not None:
            off = 0
            for t in self._small:
                n = t.numel()
                t.copy_(flat[off:off + n].view_as(t))
                off += n
        self._works, self._small = [], []


def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None):
    """identical replicas at step 0 (DDP broadcasts rank 0's state at construction)."""
    if not dist.is_initialized() or dist.get_world_size(group) <= 1:
        return
    if torch.cuda.is_available():
        # nothing of this package may still be running when torch's communicator starts: optimizer updates stream on a side
        # stream, and a second (native) RCCL communicator may have collectives queued -- RCCL promises no progress for two
        # communicators with work in flight at once
        torch.cuda.synchronize()
    from .engine import invalidate_packed
    invalidate_packed(module)              # .data writes below do not bump Parameter._version: repack on the next forward
    for t in list(module.parameters()) + list(module.buffers()):
        if t.is_contiguous():
            dist.broadcast(t.data, src, group=group)
        else:                               # tap-major conv weights: broadcast the dense physical storage
            phys = t.data.permute(2, 0, 1)
            if not phys.is_contiguous():
                raise ValueError('unexpected parameter layout')
            dist.broadcast(phys, src, group=group)
