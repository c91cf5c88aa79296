"""ctypes binding of libw2l_hip.so (C ABI: include/w2l_hip.h).

The library is the product: there is NO CPU or PyTorch fallback.  Loading fails
loudly if the shared object is missing, and every compute call fails loudly if
its tensors are not on a HIP device.
"""
from __future__ import annotations

import ctypes as C
import os

# HIP multiplexes all streams of a process onto GPU_MAX_HW_QUEUES (default 4) hardware queues, and two streams that share a
# queue do not overlap.  A data-parallel step uses five (caller's stream, weight gradients, optimizer updates, the reducer's
# stream, RCCL's own): with 8 queues the 1-rank RCCL rehearsal runs 14.5 instead of 15.0 ms/step.  Must be in the environment
# before the HIP runtime initialises; an explicit user setting wins.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

import torch  # noqa: E402,F401  -- must be imported first so the .so binds to torch's bundled HIP runtime

_HERE = os.path.dirname(os.path.abspath(__file__))
# W2L_LIB=<path>: load another build of the library (A/B experiments: csrc/Makefile BUILD= EXTRA= OUT=)
LIB_PATH = os.environ.get('W2L_LIB') or os.path.join(_HERE, 'libw2l_hip.so')

c_p = C.c_void_p
c_i = C.c_int
c_i64 = C.c_int64
c_u64 = C.c_uint64
c_f = C.c_float


class BnActDesc(C.Structure):
    """w2l_bnact_t (include/w2l_hip.h)."""
    _fields_ = [
        ('N', C.c_int32), ('T', C.c_int32), ('C', C.c_int32),
        ('y', c_p), ('y_f32', C.c_int32),
        ('scale', c_p), ('shift', c_p), ('mean', c_p), ('invstd', c_p),
        ('y2', c_p), ('scale2', c_p), ('shift2', c_p), ('mean2', c_p), ('invstd2', c_p),
        ('act', C.c_int32), ('drop_p', c_f), ('seed', c_u64), ('offset', c_u64),
        ('mask', c_p), ('lens', c_p), ('offset_dev', c_p), ('q_clipped', c_p),
    ]


class WgradItem(C.Structure):
    """w2l_wgrad_item_t (include/w2l_hip.h): one layer of a grouped weight-gradient launch."""
    _fields_ = [('dy', c_p), ('dy_bstride', C.c_int64), ('xp', c_p), ('x_bstride', C.c_int64), ('x_rows_total', C.c_int64),
                ('dw', c_p), ('Cin', C.c_int32), ('Cout', C.c_int32), ('Kw', C.c_int32), ('pad_', C.c_int32)]


class BnFin(C.Structure):
    """w2l_bnfin_t (include/w2l_hip.h): the statistics finalize of one BatchNorm branch, folded into w2l_bn_act_fwd_fin."""
    _fields_ = [('partial', c_p), ('rows', C.c_int32), ('count', C.c_int64), ('gamma', c_p), ('beta', c_p), ('eps', c_f),
                ('momentum', c_f), ('running_mean', c_p), ('running_var', c_p), ('mean', c_p), ('invstd', c_p), ('scale', c_p),
                ('shift', c_p)]


class GradSrc(C.Structure):
    """w2l_gradsrc_t (include/w2l_hip.h)."""
    _fields_ = [('dxp', c_p), ('f32', C.c_int32), ('pad_l', C.c_int32), ('pad_r', C.c_int32),
                ('pad_mode', C.c_int32), ('rows', C.c_int32)]


_SIGNATURES = {
    'w2l_last_error': (C.c_char_p, []),
    'w2l_abi_version': (c_i, []),
    'w2l_tune_save': (c_i, [C.c_char_p]),
    'w2l_tune_load': (c_i, [C.c_char_p]),
    'w2l_pack_weights': (c_i, [c_p, c_i64, c_i64, c_i64, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p]),
    'w2l_sgd_pack': (c_i, [c_p, c_p, c_p, c_i, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_p]),
    'w2l_novograd_pack': (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p,
                                c_p]),
    'w2l_nct_to_ntc': (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p]),
    'w2l_pad_cast': (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p]),
    'w2l_conv_stat_tiles': (c_i, [c_i, c_i]),
    'w2l_conv1d_igemm': (c_i, [c_p, c_i64, c_i64, c_p, c_p, c_i, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    'w2l_conv1d_igemm_tune': (c_i, [c_p, c_i64, c_i64, c_p, c_p, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    'w2l_conv1d_igemm_ws': (c_i, [c_p, c_i64, c_i64, c_p, c_p, c_i, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_i64, c_p]),
    'w2l_conv1d_igemm_tune_ws': (c_i, [c_p, c_i64, c_i64, c_p, c_p, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p,
                                       c_i64, c_p]),
    'w2l_conv_splitk_workspace_bytes': (c_i64, [c_i, c_i, c_i]),
    'w2l_conv_streamk_ranges': (c_i, [c_i] * 8 + [c_i64]),
    'w2l_conv_streamk_pieces': (c_i, [c_i, c_i, c_i, c_p, c_i]),
    'w2l_conv_force_tile_config': (None, [c_i]),
    'w2l_conv_force_fp8_config': (None, [c_i]),
    'w2l_wgrad_force_plan': (None, [c_i, c_i]),
    'w2l_wgrad_deterministic': (None, [c_i]),
    'w2l_wgrad_needs_zero': (c_i, [c_i, c_i, c_i, c_i, c_i]),
    'w2l_conv1d_wgrad': (c_i, [c_p, c_i64, c_p, c_i64, c_i64, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    'w2l_conv1d_wgrad_ws': (c_i, [c_p, c_i64, c_p, c_i64, c_i64, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_i64, c_p]),
    'w2l_conv1d_wgrad_tune_ws': (c_i, [c_p, c_i64, c_p, c_i64, c_i64, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_i64, c_p]),
    'w2l_wgrad_needs_zero_ws': (c_i, [c_i, c_i, c_i, c_i, c_i, c_i64]),
    'w2l_wgrad_workspace_bytes': (c_i64, [c_i, c_i, c_i]),
    'w2l_wgrad_dealt_workspace_bytes': (c_i64, [c_i, c_i, c_i]),
    'w2l_conv1d_wgrad_tune_x': (c_i, [c_p, c_i64, c_p, c_i64, c_i64, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_i64, c_i, c_p]),
    'w2l_wgrad_needs_zero_x': (c_i, [c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i64]),
    'w2l_wgrad_dealt_segments': (c_i, [c_i, c_i, c_i, c_p, c_i]),
    'w2l_wgrad_plan': (c_i, [c_i, c_i, c_i, c_i, c_i]),
    'w2l_conv1d_wgrad_group': (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    'w2l_wgrad_group_tiles': (c_i, [c_i, c_i, c_i, c_i]),
    'w2l_conv1d_wgrad_tune': (c_i, [c_p, c_i64, c_p, c_i64, c_i64, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    'w2l_dwconv_fwd': (c_i, [c_p, c_p, c_i, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p]),
    'w2l_dwconv_dgrad': (c_i, [c_p, c_i, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p]),
    'w2l_dwconv_wgrad': (c_i, [c_p, c_i, c_i, c_p, c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p]),
    'w2l_bn_finalize': (c_i, [c_p, c_i, c_i, c_i64, c_p, c_p, c_f, c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    'w2l_bn_act_fwd': (c_i, [C.POINTER(BnActDesc), c_p, c_p, c_i, c_i, c_i, c_i, c_p]),
    'w2l_bn_act_fwd_q': (c_i, [C.POINTER(BnActDesc), c_p, c_p, c_p, c_f, c_i, c_i, c_i, c_i, c_p]),
    'w2l_bn_bwd_fast_ok': (c_i, [C.POINTER(BnActDesc), C.POINTER(GradSrc), C.POINTER(GradSrc)]),
    'w2l_bn_act_bwd_reduce_slots': (c_i, [C.POINTER(BnActDesc), C.POINTER(GradSrc), c_p, c_i, c_p]),
    'w2l_bn_act_bwd_apply_slots': (c_i, [C.POINTER(BnActDesc), C.POINTER(GradSrc), c_p, c_i, c_p, c_p, c_i, c_p, c_p]),
    'w2l_bn_act_fwd_fin': (c_i, [C.POINTER(BnActDesc), C.POINTER(BnFin), C.POINTER(BnFin), c_p, c_p, c_f, c_i, c_i, c_i, c_i, c_p]),
    'w2l_conv_stats_mode': (None, [c_i]),
    'w2l_quantize_e4m3': (c_i, [c_p, c_i, c_i64, c_f, c_p, c_p]),
    'w2l_quantize_e4m3_dyn': (c_i, [c_p, c_i64, c_p, c_p, c_p, c_p]),
    'w2l_bn_act_bwd_apply_amax': (c_i, [C.POINTER(BnActDesc), C.POINTER(GradSrc), C.POINTER(GradSrc), c_p, c_p, c_p, c_i,
                                        c_p, c_p, c_i, c_p, c_p]),
    'w2l_conv1d_igemm_fp8': (c_i, [c_p, c_i64, c_i64, c_p, c_p, c_i, c_f, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    'w2l_conv1d_igemm_fp8_tune': (c_i, [c_p, c_i64, c_i64, c_p, c_p, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    'w2l_bn_bwd_blocks': (c_i, [c_i, c_i, c_i]),
    'w2l_bn_act_bwd_reduce': (c_i, [C.POINTER(BnActDesc), C.POINTER(GradSrc), C.POINTER(GradSrc), c_p, c_p]),
    'w2l_bn_bwd_finalize': (c_i, [c_p, c_i, c_i, c_i, c_p, c_p]),
    'w2l_bn_act_bwd_apply': (c_i, [C.POINTER(BnActDesc), C.POINTER(GradSrc), C.POINTER(GradSrc), c_p, c_p, c_p, c_i,
                                   c_p, c_p, c_i, c_p]),
    'w2l_conv1d_dgrad_bnreduce_ws': (c_i, [c_p, c_i64, c_p, c_p, c_p, C.POINTER(BnActDesc), c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i,
                                           c_p, c_i64, c_p]),
    'w2l_conv1d_dgrad_bnreduce_tune_ws': (c_i, [c_p, c_i64, c_p, c_p, c_p, C.POINTER(BnActDesc), c_i, c_i, c_i, c_i, c_i, c_i, c_i,
                                                c_i, c_i, c_p, c_i64, c_p]),
    'w2l_bn_act_bwd_apply_fin': (c_i, [C.POINTER(BnActDesc), C.POINTER(GradSrc), C.POINTER(GradSrc), c_p, c_i, c_p, c_p, c_p, c_i,
                                       c_p, c_p, c_i, c_p, c_p]),
    'w2l_log_softmax_fwd': (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p]),
    'w2l_log_softmax_bwd': (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p]),
    'w2l_ctc_workspace_bytes': (c_i64, [c_i, c_i, c_i]),
    'w2l_ctc_loss': (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p]),
    'w2l_argmax': (c_i, [c_p, c_i64, c_i, c_p, c_p]),
    'w2l_logmel': (c_i, [c_p, c_p, c_p, c_f, c_f, c_i, c_i64, c_p, c_i, c_i, c_i, c_p, c_p, c_i, c_i, c_f, c_p, c_i, c_p]),
    'w2l_feature_normalize': (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_p, c_p, c_p, c_p]),
    'w2l_zero_rects': (c_i, [c_p, c_i, c_i, c_i, c_p, c_i, c_p]),
    'w2l_levenshtein_host': (c_i, [c_p, c_i, c_p, c_i]),
    'w2l_greedy_score_host': (c_i, [c_p, c_i, c_i, c_p, c_i, c_p, c_i, c_p, c_p, c_p, c_p, c_p]),
    'w2l_wgrad_fp8_needs_zero': (c_i, [c_i, c_i, c_i, c_i, c_i]),
    'w2l_conv1d_wgrad_fp8': (c_i, [c_p, c_i64, c_p, c_i64, c_i64, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_p, c_i, c_p]),
    'w2l_conv1d_wgrad_fp8_tune': (c_i, [c_p, c_i64, c_p, c_i64, c_i64, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    'w2l_stream_probe': (c_i, [c_p, c_p, c_p, c_i, c_i]),
    # RCCL helpers (data-parallel exchange for hosts without torch.distributed; distributed.NativeComm)
    'w2l_rccl_available': (c_i, []),
    'w2l_rccl_library': (C.c_char_p, []),
    'w2l_rccl_unique_id': (c_i, [c_p]),
    'w2l_rccl_init': (c_i, [c_p, c_i, c_i, c_p]),
    'w2l_rccl_world': (c_i, [c_p, c_p]),
    'w2l_rccl_all_reduce': (c_i, [c_p, c_p, c_i64, c_i, c_i, c_p]),
    'w2l_rccl_broadcast': (c_i, [c_p, c_p, c_i64, c_i, c_p]),
    'w2l_rccl_destroy': (c_i, [c_p]),
    # recorded launch lists (replay.py): the replay loop and the recordable primitives
    'w2l_replay_op': (c_i, [C.c_char_p]),
    'w2l_replay_arity': (c_i, [c_i]),
    'w2l_replay': (c_i, [c_p, c_i, c_p]),
    'w2l_event_create': (c_i, [c_p]),
    'w2l_event_destroy': (c_i, [c_p]),
    'w2l_event_record': (c_i, [c_p, c_p]),
    'w2l_stream_wait_event': (c_i, [c_p, c_p]),
    'w2l_event_query': (c_i, [c_p]),
    'w2l_event_synchronize': (c_i, [c_p]),
    'w2l_stream_wait_stream': (c_i, [c_p, c_p]),
    'w2l_fill_zero': (c_i, [c_p, c_i64, c_p]),
    'w2l_pad_vec_f32': (c_i, [c_p, c_i, c_p, c_i, c_f, c_p]),
    'w2l_counter_add': (c_i, [c_p, c_i64, c_p]),
    'w2l_add_i64_multi': (c_i, [c_p, c_i, c_i64, c_p]),
    'w2l_sgd_small_multi': (c_i, [c_p, c_i, c_i, c_f, c_f, c_f, c_i, c_p]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


class W2LError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f'{LIB_PATH} not found: build the HIP extension first '
            f'(python -c "import __graft_entry__ as g; g.build()" or make -C wav2letter_pytorch_amd/csrc). '
            f'There is no CPU fallback.')
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def check(rc: int, what: str = ''):
    if rc != 0:
        msg = lib.w2l_last_error()
        raise W2LError(f'{what} failed (code {rc}): {msg.decode() if msg else "?"}')


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr():
    """hipStream_t of torch's current stream on the current device (raw getters: ~20x cheaper than
    torch.cuda.current_stream(), and this is called for every launch)."""
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


def require_device(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise W2LError('wav2letter_pytorch_amd runs on MI355X only: got a CPU tensor '
                           '(there is no CPU fallback; move the model and batch to cuda)')


# ---------------------------------------------------------------------------------------------------------------- launch trace
# A timeline of the step WITHOUT a profiler attached (under rocprofv3 the host needs ~13 ms to enqueue a step instead of ~5
# and the step boundary shows bubbles the real run does not have): trace_launches(True) swaps every launching entry point
# for a wrapper that records a HIP event before and after the call on the stream it launches on; trace_dump() turns the
# pairs into rows with device timestamps (ns from the first event), in the column layout of rocprofv3's kernel trace, so
# tools/timeline.py reads either.  ~10 us of host time per launch; diagnostic only (bench.py --event-trace).
TRACE_NAMES = {
    'w2l_conv1d_igemm': 'conv_igemm_kernel', 'w2l_conv1d_igemm_ws': 'conv_igemm_kernel',
    'w2l_conv1d_dgrad_bnreduce_ws': 'conv_igemm_kernel/dgrad+bnreduce', 'w2l_conv1d_igemm_fp8': 'conv_igemm_fp8_kernel',
    'w2l_conv1d_wgrad': 'conv_wgrad_kernel', 'w2l_conv1d_wgrad_ws': 'conv_wgrad_kernel', 'w2l_conv1d_wgrad_group': 'conv_wgrad_kernel', 'w2l_conv1d_wgrad_fp8': 'conv_wgrad_fp8_kernel',
    'w2l_bn_finalize': 'bn_finalize_kernel', 'w2l_bn_act_fwd': 'bn_act_fwd_kernel', 'w2l_bn_act_fwd_q': 'bn_act_fwd_kernel',
    'w2l_bn_act_fwd_fin': 'bn_act_fwd_kernel', 'w2l_bn_act_bwd_reduce': 'bn_act_bwd_reduce_kernel',
    'w2l_bn_act_bwd_reduce_slots': 'bn_act_bwd_reduce_kernel', 'w2l_bn_act_bwd_apply_slots': 'bn_act_bwd_apply_kernel', 'w2l_bn_bwd_finalize': 'bn_bwd_finalize_kernel',
    'w2l_bn_act_bwd_apply': 'bn_act_bwd_apply_kernel', 'w2l_bn_act_bwd_apply_amax': 'bn_act_bwd_apply_kernel',
    'w2l_bn_act_bwd_apply_fin': 'bn_act_bwd_apply_kernel', 'w2l_sgd_pack': 'sgd_pack_kernel', 'w2l_pack_weights': 'pack_weights_kernel',
    'w2l_ctc_loss': 'ctc_kernels', 'w2l_log_softmax_fwd': 'log_softmax_fwd', 'w2l_log_softmax_bwd': 'log_softmax_bwd',
    'w2l_nct_to_ntc': 'nct_to_ntc_kernel', 'w2l_pad_cast': 'pad_cast_kernel', 'w2l_quantize_e4m3': 'quantize_e4m3',
    'w2l_quantize_e4m3_dyn': 'quantize_e4m3_dyn', 'w2l_dwconv_fwd': 'dw_fwd_kernel', 'w2l_dwconv_dgrad': 'dw_dgrad_kernel',
    'w2l_dwconv_wgrad': 'dw_wgrad_kernel', 'w2l_argmax': 'argmax_kernel', 'w2l_novograd_pack': 'novograd_pack_kernel',
}
_trace = {'rows': None, 'saved': {}, 'pool': []}


def trace_launches(enable: bool, capacity: int = 4096):
    """start (True) or stop (False) recording an event pair around every launching entry point; the events of the first
    ``capacity`` launches are created up front (hipEventCreate costs more than the launch it brackets)"""
    if enable and _trace['rows'] is None:
        _trace['rows'] = []
    if enable and not _trace['saved']:           # (re-)wrap whenever the entry points are not wrapped, whatever rows holds
        _trace['pool'] = [torch.cuda.Event(enable_timing=True) for _ in range(2 * capacity)]
        for name, label in TRACE_NAMES.items():
            fn = getattr(lib, name)
            _trace['saved'][name] = fn

            def wrapper(*args, _fn=fn, _label=label):
                pool = _trace['pool']
                s = pool.pop() if pool else torch.cuda.Event(enable_timing=True)
                e = pool.pop() if pool else torch.cuda.Event(enable_timing=True)
                s.record()
                rc = _fn(*args)
                e.record()
                _trace['rows'].append((_label, torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()), s, e))
                return rc

            setattr(lib, name, wrapper)
    elif not enable:
        for name, fn in _trace['saved'].items():
            setattr(lib, name, fn)
        _trace['saved'] = {}


def trace_dump(path: str) -> int:
    """write the recorded launches as a rocprofv3-style kernel trace (csv) and forget them; returns the row count.
    The device must be idle (torch.cuda.synchronize()) -- event times are read back."""
    rows, _trace['rows'] = _trace['rows'] or [], None
    _trace['pool'] = []
    if not rows:
        return 0
    origin = rows[0][2]
    with open(path, 'w') as f:
        f.write('Kernel_Name,Queue_Id,Start_Timestamp,End_Timestamp\n')
        for label, stream, s, e in rows:
            t0 = origin.elapsed_time(s)
            f.write('%s,%d,%d,%d\n' % (label, stream, int(t0 * 1e6), int((t0 + s.elapsed_time(e)) * 1e6)))
    return len(rows)


# ---------------------------------------------------------------------------------------------------------------- events
def raw_stream(stream=None) -> int:
    """hipStream_t (as an int) of a torch stream, or of torch's current stream"""
    if stream is None:
        return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())
    return stream.cuda_stream


class Event:
    """A HIP event behind the C ABI (w2l_event_*): what the step engine orders its streams with.  Unlike torch.cuda.Event
    every record / wait is an entry-point call, so a recorded launch list (replay.py) replays the stream order too.  Handles
    are pooled; an event created while a phase is being recorded belongs to that record (its handle is in the list)."""
    __slots__ = ('h', '__weakref__')
    _free: list = []

    def __init__(self):
        if Event._free:
            self.h = Event._free.pop()
        else:
            out = C.c_void_p()
            check(lib.w2l_event_create(C.byref(out)), 'w2l_event_create')
            self.h = out.value
        rec = _recorder[0]
        if rec is not None:
            rec.keep.append(self)

    def record(self, stream=None):
        check(lib.w2l_event_record(self.h, raw_stream(stream)), 'w2l_event_record')

    def wait(self, stream=None):
        """make ``stream`` (default: the current one) wait for this event"""
        check(lib.w2l_stream_wait_event(raw_stream(stream), self.h), 'w2l_stream_wait_event')

    def query(self) -> bool:
        rc = lib.w2l_event_query(self.h)
        if rc not in (0, 1):
            check(rc, 'w2l_event_query')
        return rc == 0

    def synchronize(self):
        check(lib.w2l_event_synchronize(self.h), 'w2l_event_synchronize')

    def __del__(self):
        try:
            Event._free.append(self.h)
        except Exception:          # interpreter shutdown
            pass


def stream_wait_stream(waiter, signaler):
    """``waiter`` waits for everything enqueued so far on ``signaler`` (torch streams or raw handles)"""
    a = waiter if isinstance(waiter, int) else waiter.cuda_stream
    b = signaler if isinstance(signaler, int) else signaler.cuda_stream
    check(lib.w2l_stream_wait_stream(a, b), 'w2l_stream_wait_stream')


# ---------------------------------------------------------------------------------------------------------------- recorder
class Slot(C.Union):
    """w2l_slot_t"""
    _fields_ = [('p', c_p), ('i', c_i64), ('d', C.c_double)]


REPLAY_MAX_ARGS = 24


class Call(C.Structure):
    """w2l_call_t"""
    _fields_ = [('op', C.c_int32), ('nargs', C.c_int32), ('a', Slot * REPLAY_MAX_ARGS)]


_recorder = [None]            # the Recorder of the phase being recorded (module-wide: a phase runs on ONE thread at a time)
_REPLAY_OPS = {}              # entry point name -> (op index, per-argument kind: 'p' / 'i' / 'd')


def _replay_ops():
    if not _REPLAY_OPS:
        for name, (_, args) in _SIGNATURES.items():
            op = lib.w2l_replay_op(name.encode())
            if op < 0:
                continue
            kinds = []
            for a in args:
                if a in (c_i, c_i64, c_u64, C.c_int32):
                    kinds.append('i')
                elif a in (c_f, C.c_double):
                    kinds.append('d')
                else:
                    kinds.append('p')
            assert lib.w2l_replay_arity(op) == len(kinds) <= REPLAY_MAX_ARGS, name
            _REPLAY_OPS[name] = (op, tuple(kinds))
    return _REPLAY_OPS


class Phase:
    """One recorded phase of a step: C segments (arrays of w2l_call_t, replayed by ONE w2l_replay call each) and -- only
    where the engine had to call back into Python between launches -- Python items run with their recorded current stream."""
    __slots__ = ('items', 'keep', 'n_calls')

    def __init__(self, items, keep, n_calls):
        self.items, self.keep, self.n_calls = items, keep, n_calls

    def replay(self):
        failed = C.c_int(-1)
        for it in self.items:
            if it[0] == 'c':
                rc = lib.w2l_replay(it[1], it[2], C.byref(failed))
                if rc != 0:
                    msg = lib.w2l_last_error()
                    raise W2LError(f'w2l_replay failed at record {failed.value} of {it[2]} (code {rc}): {msg.decode() if msg else "?"}')
            else:
                _, fn, args, stream = it
                if stream is None:
                    fn(*args)
                else:
                    with torch.cuda.stream(stream):
                        fn(*args)


class Recorder:
    """While active (``with Recorder() as rec``) every replayable entry point called through ``lib`` on this thread is executed
    AND appended to the list; ``python(fn, *args)`` runs and records a Python callback in sequence; ``poison(why)`` marks the
    phase as not replayable (a code path that still uses torch ops between launches).  ``finish()`` -> Phase or None."""

    def __init__(self):
        self.calls = []          # pending C calls of the current segment: (op, kinds, args)
        self.items = []
        self.keep = []
        self.poisoned = None
        self.n_calls = 0
        self._saved = {}
        self._thread = None

    def __enter__(self):
        import threading
        if _recorder[0] is not None:
            raise W2LError('a phase is already being recorded')
        self._thread = threading.get_ident()
        for name, (op, kinds) in _replay_ops().items():
            fn = getattr(lib, name)
            self._saved[name] = fn

            def wrapper(*args, _fn=fn, _op=op, _kinds=kinds, _name=name):
                rc = _fn(*args)
                if threading.get_ident() == self._thread and (rc == 0 or rc is None):
                    self.calls.append((_op, _kinds, args))
                return rc

            setattr(lib, name, wrapper)
        _recorder[0] = self
        return self

    def __exit__(self, *exc):
        for name, fn in self._saved.items():
            setattr(lib, name, fn)
        self._saved = {}
        _recorder[0] = None
        return False

    def poison(self, why: str):
        if self.poisoned is None:
            self.poisoned = why

    def python(self, fn, *args, stream=None):
        self._flush()
        fn(*args)
        self.items.append(('py', fn, args, stream))

    def _flush(self):
        if not self.calls:
            return
        arr = (Call * len(self.calls))()
        for c, (op, kinds, args) in zip(arr, self.calls):
            c.op, c.nargs = op, len(kinds)
            for slot, kind, v in zip(c.a, kinds, args):
                if kind == 'i':
                    slot.i = int(v)
                elif kind == 'd':
                    slot.d = float(v)
                elif v is None:
                    slot.p = None
                elif isinstance(v, int):
                    slot.p = v
                elif isinstance(v, C.c_void_p):
                    slot.p = v.value
                else:
                    # a by-reference struct (C.byref(obj)) or a ctypes array passed as a pointer: the record owns a copy
                    obj = getattr(v, '_obj', v)
                    copy = type(obj).from_buffer_copy(obj)
                    self.keep.append(copy)
                    slot.p = C.addressof(copy)
        self.items.append(('c', arr, len(self.calls)))
        self.n_calls += len(self.calls)
        self.calls = []

    def finish(self):
        self._flush()
        if self.poisoned is not None:
            return None
        return Phase(self.items, self.keep, self.n_calls)


def recording() -> 'Recorder | None':
    return _recorder[0]


def poison(why: str):
    """called from code paths a recorded launch list cannot reproduce (torch ops between launches, host synchronisation,
    measuring launches): the phase being recorded, if any, is dropped and the step stays eager"""
    rec = _recorder[0]
    if rec is not None:
        rec.poison(why)
