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
    'w2l_conv_force_tile_config': (None, [c_i]),
    'w2l_conv_force_fp8_config': (None, [c_i]),
    'w2l_wgrad_force_plan': (None, [c_i, c_i]),
    'w2l_wgrad_needs_zero': (c_i, [c_i, c_i, c_i, c_i, c_i]),
    'w2l_conv1d_wgrad': (c_i, [c_p, c_i64, c_p, c_i64, c_i64, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    'w2l_conv1d_wgrad_ws': (c_i, [c_p, c_i64, c_p, c_i64, c_i64, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_i64, c_p]),
    'w2l_conv1d_wgrad_tune_ws': (c_i, [c_p, c_i64, c_p, c_i64, c_i64, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_i64, c_p]),
    'w2l_wgrad_needs_zero_ws': (c_i, [c_i, c_i, c_i, c_i, c_i, c_i64]),
    'w2l_wgrad_workspace_bytes': (c_i64, [c_i, c_i, c_i]),
    'w2l_conv1d_wgrad_tune': (c_i, [c_p, c_i64, c_p, c_i64, c_i64, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    'w2l_dwconv_fwd': (c_i, [c_p, c_p, c_i, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p]),
    'w2l_dwconv_dgrad': (c_i, [c_p, c_i, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p]),
    'w2l_dwconv_wgrad': (c_i, [c_p, c_i, c_i, c_p, c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p]),
    'w2l_bn_finalize': (c_i, [c_p, c_i, c_i, c_i64, c_p, c_p, c_f, c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    'w2l_bn_act_fwd': (c_i, [C.POINTER(BnActDesc), c_p, c_p, c_i, c_i, c_i, c_i, c_p]),
    'w2l_bn_act_fwd_q': (c_i, [C.POINTER(BnActDesc), c_p, c_p, c_p, c_f, c_i, c_i, c_i, c_i, c_p]),
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
