"""MI355X-native Wav2Letter / Jasper CTC training path (drop-in for the model side of
assafmu/wav2letter_pytorch).  There is no CPU fallback: the first use of any public name imports ``_lib``, which loads
libw2l_hip.so and raises if it (or any symbol include/w2l_hip.h declares) is missing.

The names are resolved lazily (PEP 562) so that ``python -m wav2letter_pytorch_amd.train trainer.gpus=N`` -- whose parent
process only spawns the ranks (launch.py) -- never maps the HIP library: a process that has touched the GPU must not start
the ranks."""
import importlib

_PUBLIC = {
    'ConvCTCASR': '.base_asr_models', 'CTCLoss': '.ctc_loss', 'Decoder': '.decoder', 'GreedyDecoder': '.decoder',
    'Conv1dBlock': '.wav2letter', 'Wav2Letter': '.wav2letter', 'Jasper': '.jasper', 'JasperBlock': '.jasper',
    'MaskedConv1d': '.jasper',
}
__all__ = sorted(_PUBLIC)


def __getattr__(name):
    target = _PUBLIC.get(name)
    if target is None:
        if name.startswith('_') and name != '_lib':
            raise AttributeError(f'module {__name__!r} has no attribute {name!r}')
        try:                                   # submodules: ``wav2letter_pytorch_amd.engine`` after a bare package import
            return importlib.import_module('.' + name, __name__)
        except ModuleNotFoundError as e:
            if e.name != f'{__name__}.{name}':
                raise                          # the submodule exists; one of ITS imports is missing
            raise AttributeError(f'module {__name__!r} has no attribute {name!r}') from None
    importlib.import_module('._lib', __name__)           # fails loudly when the HIP library is missing
    value = getattr(importlib.import_module(target, __name__), name)
    globals()[name] = value
    return value


def __dir__():
    return sorted(set(globals()) | set(_PUBLIC))
