"""CPU-only checks: the C-ABI library loads and exports every symbol include/w2l_hip.h declares,
the host-side mirror of the reference interface (module tree, state-dict names, pad rule, config
loader, decoder string logic, Levenshtein), the loud failure on CPU tensors, and the data-parallel
reducer over gloo with world_size 2.  No kernel is launched here."""
import ast
import os
import re
import socket
import sys
import time

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden')


def test_abi_exports_every_declared_symbol():
    from wav2letter_pytorch_amd import _lib
    text = open(os.path.join(ROOT, 'include', 'w2l_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    declared = set(re.findall(r'\b(w2l_[a-z0-9_]+)\s*\(', text))
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(_lib.lib, name), f'{name} declared in the header but not exported'
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)
    assert _lib.lib.w2l_abi_version() == 2


def _cfg(mid_layers=20, **kw):
    from wav2letter_pytorch_amd.config import to_cfg
    from wav2letter_pytorch_amd.data import label_sets
    sys.path.insert(0, ROOT)
    from oracle.w2l_oracle import W2L_LAYERS
    labels = label_sets.labels_map['english_lowercase']
    d = dict(name='wav2letter', mid_layers=mid_layers, input_size=64, labels=labels,
             layers=[dict(output_size=c, kernel_size=k, stride=s, dilation=dl, dropout=p) for c, k, s, dl, p in W2L_LAYERS],
             audio_conf=dict(window='hamming', window_stride=0.01, window_size=0.02, sample_rate=16000),
             decoder=dict(_target_='decoder.GreedyDecoder', labels=labels),
             optimizer=dict(_target_='torch.optim.SGD', lr=1e-5, momentum=0.9, nesterov=True, weight_decay=1e-5),
             scheduler=dict(_target_='torch.optim.lr_scheduler.ExponentialLR', gamma=0.999))
    d.update(kw)
    return to_cfg(d)


def test_label_sets_match_reference_semantics():
    from wav2letter_pytorch_amd.data import label_sets
    lab = label_sets.labels_map['english_lowercase']
    assert len(lab) == 29 and lab[0] == '_' and lab[-1] == ' ' and lab[1] == "'" and lab[2] == 'a'
    assert label_sets.labels_map['english'][2] == 'A' and len(label_sets.labels_map['hebrew']) == 29


def test_wav2letter_module_surface_full_table():
    from wav2letter_pytorch_amd import Wav2Letter
    m = Wav2Letter(_cfg(20))
    sd = m.state_dict()
    n_params = sum(p.numel() for p in m.parameters())
    from oracle.w2l_oracle import W2L_LAYERS
    cin, expect = 64, 0
    for c, k, _, _, _ in W2L_LAYERS:
        expect += c * cin * k + c + 2 * c       # conv weight + bias, BN gamma + beta
        cin = c
    expect += 29 * cin + 29
    assert n_params == expect and round(n_params / 1e6, 2) == 153.07          # 153.07 M params
    assert m.scaling_factor == 2 and m.mid_layers == 20 and m.input_size == 64
    assert torch.equal(m.compute_output_lengths(torch.tensor([1000, 801])), torch.tensor([500, 400]))
    keys = list(sd.keys())
    assert keys[0] == 'conv1ds.conv1d_0.conv1.weight' and 'conv1ds.conv1d_0.batch_norm.num_batches_tracked' in sd
    assert 'conv1ds.conv1d_20.conv1.bias' in sd and 'conv1ds.conv1d_20.batch_norm.weight' not in sd
    assert sd['conv1ds.conv1d_16.conv1.weight'].shape == (896, 768, 29)
    pads = [(b.pad_l, b.pad_r) for b in m.conv1ds.children()]
    assert pads[0] == (4, 5) and pads[1] == (5, 5) and pads[16] == (28, 28) and pads[19] == (0, 0) and pads[20] == (0, 0)
    bn = m.conv1ds.conv1d_3.batch_norm
    assert bn.momentum == 0.9 and bn.eps == 0.001
    eng = m.engine()
    assert len(eng.units) == 20 and len(eng.parameters()) == len(list(m.parameters()))
    assert m.example_input_array[0].shape == (4, 64, 200)
    opt, sch = m.configure_optimizers()
    assert isinstance(opt[0], torch.optim.SGD) and opt[0].defaults['nesterov']
    # default shipped truncation: mid_layers = 1
    m1 = Wav2Letter(_cfg(1))
    assert sum(p.numel() for p in m1.parameters()) == 256 * 64 * 11 + 256 + 512 + 29 * 256 + 29


def test_weight_layout_and_checkpoint_roundtrip(tmp_path):
    from wav2letter_pytorch_amd import Wav2Letter
    z = np.load(os.path.join(GOLD, 'w2l_ml3.npz'), allow_pickle=True)
    meta = ast.literal_eval(str(z['meta']))
    m = Wav2Letter(_cfg(3, layers=meta['layers']))
    w = m.conv1ds.conv1d_1.conv1.weight
    assert w.shape == (96, 96, 11) and w.stride() == (96, 1, 96 * 96)      # logical [out,in,k], stored tap-major
    sd = {k[3:]: torch.from_numpy(z[k].copy()) for k in z.files if k.startswith('p0/')}
    m.load_state_dict(sd)                                                  # a reference state dict loads as is
    for k, v in m.state_dict().items():
        assert torch.equal(v.contiguous(), sd[k]), k
    torch.save(m.state_dict(), tmp_path / 'ck.pt')
    m2 = Wav2Letter(_cfg(3, layers=meta['layers']))
    m2.load_state_dict(torch.load(tmp_path / 'ck.pt'))
    assert torch.equal(m2.conv1ds.conv1d_1.conv1.weight, w)
    # the whole module pickles (torch.save(model)) and deep-copies: the cached engine, the operand packs and the load hook are
    # left behind, parameters and buffers travel
    import copy
    import pickle
    m2.engine()
    m2.conv1ds.conv1d_1.conv1.weight._w2l_pack = __import__('wav2letter_pytorch_amd.engine', fromlist=['x'])._Volatile(x=1)
    for clone in (pickle.loads(pickle.dumps(m2)), copy.deepcopy(m2)):
        assert '_engine_cache' not in clone.__dict__ or clone.__dict__['_engine_cache'] is None
        assert getattr(clone.conv1ds.conv1d_1.conv1.weight, '_w2l_pack', None) is None
        assert torch.equal(clone.conv1ds.conv1d_1.conv1.weight, w)
        clone.load_state_dict(m2.state_dict())              # the post-load hook came along and still works
        assert len(clone.engine().units) == 3
    from wav2letter_pytorch_amd.train import name_to_model
    assert 'jasper' in name_to_model and set(name_to_model) == {'jasper', 'wav2letter'} and name_to_model.get('nope') is None
    assert name_to_model['wav2letter'] is Wav2Letter and len(name_to_model) == 2
    # same seed -> same initial values as nn.Conv1d would draw into a contiguous tensor
    torch.manual_seed(3)
    a = Wav2Letter(_cfg(1))
    torch.manual_seed(3)
    ref = torch.nn.Conv1d(64, 256, 11, stride=2)
    # (the model draws the decoder/example input first, so only distribution bounds are compared)
    bound = 1 / (64 * 11) ** 0.5
    assert float(a.conv1ds.conv1d_0.conv1.weight.abs().max()) <= bound + 1e-6
    assert float(ref.weight.abs().max()) <= bound + 1e-6


def test_cpu_tensors_fail_loudly():
    from wav2letter_pytorch_amd import Wav2Letter, CTCLoss
    from wav2letter_pytorch_amd._lib import W2LError
    m = Wav2Letter(_cfg(1))
    with pytest.raises(W2LError):
        m(torch.randn(2, 64, 50), torch.tensor([50, 50]))
    with pytest.raises(W2LError):
        CTCLoss(0, 'mean', True)(torch.randn(10, 2, 29).log_softmax(-1), torch.ones(2, 3, dtype=torch.int32),
                                 torch.tensor([10, 10]), torch.tensor([3, 3]))
    with pytest.raises(RuntimeError):
        m.conv1ds.conv1d_0(torch.randn(1, 64, 20))
    with pytest.raises(NotImplementedError):
        CTCLoss(reduction='sum')


def test_decoder_host_logic_and_levenshtein():
    from wav2letter_pytorch_amd.decoder import GreedyDecoder
    z = np.load(os.path.join(GOLD, 'greedy_cases.npz'), allow_pickle=True)
    dec = GreedyDecoder(list(z['labels']), blank_index=0)
    strings, offsets = dec.convert_to_strings(torch.from_numpy(z['argmax']), z['sizes'], remove_repetitions=True,
                                              return_offsets=True)
    assert [s[0] for s in strings] == list(z['strings'])
    for o, ref in zip(offsets, z['offsets']):
        assert o[0].tolist() == list(ref)
    for (a, b), c, w in zip(z['pairs'], z['cer'], z['wer']):
        assert dec.cer_ratio(str(a), str(b)) == tuple(c)
        assert dec.wer_ratio(str(a), str(b)) == tuple(w)
    assert GreedyDecoder('english_lowercase').space_index == 28
    assert dec.cer('', 'abc') == 3 and dec.wer('a b', '') == 2


def test_score_batch_equals_the_python_string_path():
    """decoder.GreedyDecoder.score_batch (ONE host call: collapse + CER / WER totals, w2l_greedy_score_host) against the
    reference-shaped Python path (convert_to_strings + cer_ratio / wer_ratio per utterance, decoder.py:31-66,104-119) on random
    index matrices and transcripts: runs of blanks and repeats, ragged sizes, leading / double / trailing spaces, empty
    decodes, empty transcripts, tabs inside a transcript (str.split() whitespace), the Hebrew label set (non-ASCII labels)"""
    from wav2letter_pytorch_amd.data import label_sets
    from wav2letter_pytorch_amd.decoder import GreedyDecoder
    rng = np.random.default_rng(7)
    for name in ('english_lowercase', 'hebrew'):
        labels = label_sets.labels_map[name]
        dec = GreedyDecoder(labels)
        for trial in range(12):
            n, t = int(rng.integers(1, 9)), int(rng.integers(1, 120))
            # few distinct symbols -> many repeats; weight blank and space heavily
            pool = np.array([0, 0, 0, len(labels) - 1, len(labels) - 1] + list(rng.integers(1, len(labels), 4)))
            idx = pool[rng.integers(0, len(pool), (n, t))].astype(np.int32)
            sizes = torch.from_numpy(rng.integers(0, t + 1, n).astype(np.int32)) if trial % 3 else None
            texts = []
            for i in range(n):
                m = int(rng.integers(0, 40))
                chars = [labels[int(c)] for c in pool[rng.integers(0, len(pool), m)] if c != 0]
                txt = ''.join(chars)
                if trial % 4 == 1 and txt:
                    txt = txt[: len(txt) // 2] + '\t' + txt[len(txt) // 2:]
                texts.append(txt)
            if not any(t_.replace(' ', '') for t_ in texts):
                texts[0] = labels[2] + ' ' + labels[3]
            hyps, totals = dec.score_batch(torch.from_numpy(idx), sizes, texts)
            want = [s_[0] for s_ in dec.convert_to_strings(idx, sizes, remove_repetitions=True)]
            assert hyps == want
            cer = [dec.cer_ratio(r, h) for r, h in zip(texts, want)]
            wer = [dec.wer_ratio(r, h) for r, h in zip(texts, want)]
            assert totals == (sum(e for e, _ in cer), sum(d for _, d in cer), sum(e for e, _ in wer), sum(d for _, d in wer))
    # the known answer of unit_tests/decoder_test.py:40-42 through this path: all-blank frames decode to ''
    dec = GreedyDecoder(['_', 'A', 'B', ' '])
    hyps, totals = dec.score_batch(np.zeros((1, 2), dtype=np.int32), None, ['A B'])
    assert hyps == [''] and totals == (2, 2, 2, 2)
    with pytest.raises(Exception):
        dec.score_batch(np.full((1, 2), 9, dtype=np.int32), None, ['A'])


def test_dropin_module_names_resolve_in_a_fresh_interpreter():
    """the reference's OWN module names (train.py:12-19, configuration/config.yaml:14-16, optimizer `_target_`s) with
    ``dropin/`` on the path: `from wav2letter import Wav2Letter`, `from jasper import Jasper`, `import decoder`,
    `import base_asr_models`, `import novograd`, `from data import label_sets`, and `_target_: decoder.GreedyDecoder` /
    `novograd.Novograd` through config.instantiate -- in a subprocess, so nothing this test session imported can help"""
    import subprocess
    code = r"""
import sys
import torch
from wav2letter import Wav2Letter, Conv1dBlock
from jasper import Jasper, JasperBlock, MaskedConv1d
import decoder, base_asr_models, novograd
from data import label_sets
from data.data_loader import SpectrogramDataset, BatchAudioDataLoader
import wav2letter_pytorch_amd as pkg
from wav2letter_pytorch_amd.config import instantiate
assert Wav2Letter is pkg.Wav2Letter and Jasper is pkg.Jasper
assert issubclass(Wav2Letter, base_asr_models.ConvCTCASR) and issubclass(Jasper, base_asr_models.ConvCTCASR)
labels = label_sets.labels_map['english_lowercase']
dec = instantiate({'_target_': 'decoder.GreedyDecoder', 'labels': labels})
assert type(dec) is decoder.GreedyDecoder and dec.blank_index == 0 and dec.space_index == 28
p = [torch.nn.Parameter(torch.zeros(3))]
opt = instantiate({'_target_': 'novograd.Novograd', 'lr': 0.01}, params=p)
assert type(opt) is novograd.Novograd and opt.param_groups[0]['lr'] == 0.01
from wav2letter_pytorch_amd.defaults import wav2letter_model, jasper_model
m = Wav2Letter(wav2letter_model(2))
assert type(m.ctc_decoder) is decoder.GreedyDecoder and m.scaling_factor == 2
j = Jasper(jasper_model(2))
assert type(j.ctc_decoder) is decoder.GreedyDecoder and j.scaling_factor == 2
print('dropin ok')
"""
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, 'dropin'), ROOT]))
    out = subprocess.run([sys.executable, '-c', code], env=env, cwd=str(ROOT), capture_output=True, text=True, timeout=240)
    assert out.returncode == 0 and 'dropin ok' in out.stdout, out.stderr[-3000:]


def test_replay_table_matches_the_binding_and_rejects_bad_records():
    """recorded launch lists, host side only (no launch): every replayable entry point of the library's table (w2l_replay_op /
    w2l_replay_arity) has the arity the ctypes binding declares -- the recorder stores one slot per declared argument --, the
    layout of w2l_call_t / w2l_slot_t is the header's (8-byte slots, 24 of them behind two int32), an empty list replays to 0,
    and a record naming an unknown entry point or the wrong argument count is refused with its index (nothing is called)"""
    import ctypes as C
    from wav2letter_pytorch_amd import _lib
    ops = _lib._replay_ops()
    assert len(ops) >= 40 and 'w2l_conv1d_igemm_ws' in ops and 'w2l_event_record' in ops and 'w2l_sgd_small_multi' in ops
    for name, (op, kinds) in ops.items():
        assert _lib.lib.w2l_replay_arity(op) == len(_lib._SIGNATURES[name][1]) == len(kinds), name
        assert _lib.lib.w2l_replay_op(name.encode()) == op
    # host-only queries and the measuring entry points are NOT replayable: their results are recorded control flow
    for name in ('w2l_wgrad_needs_zero_x', 'w2l_conv1d_igemm_tune_ws', 'w2l_bn_bwd_fast_ok', 'w2l_tune_save', 'w2l_greedy_score_host'):
        assert _lib.lib.w2l_replay_op(name.encode()) == -1, name
    assert C.sizeof(_lib.Slot) == 8 and C.sizeof(_lib.Call) == 8 + 8 * _lib.REPLAY_MAX_ARGS
    failed = C.c_int(-1)
    assert _lib.lib.w2l_replay(None, 0, C.byref(failed)) == 0
    calls = (_lib.Call * 2)()
    calls[0].op, calls[0].nargs = ops['w2l_conv_stats_mode'][0], 1          # a harmless host-state call: stats mode 0
    calls[0].a[0].i = 0
    calls[1].op, calls[1].nargs = 9999, 0
    assert _lib.lib.w2l_replay(calls, 2, C.byref(failed)) != 0 and failed.value == 1
    calls[1].op, calls[1].nargs = ops['w2l_fill_zero'][0], 2                # wrong arity (3 declared)
    assert _lib.lib.w2l_replay(calls, 2, C.byref(failed)) != 0 and failed.value == 1
    assert b'replay' in _lib.lib.w2l_last_error()


def test_recorder_records_and_replays_host_state_calls():
    """the recorder itself, on the CPU: entry points that only move HOST state (w2l_conv_stats_mode, w2l_wgrad_deterministic)
    are in the replay table like any launch, so recording and replaying them needs no device.  Checked: calls are executed
    while recorded, land in C segments in call order, a Python callback splits the segments and is replayed in sequence, the
    library's functions are restored when the recorder exits (also on an exception), a poisoned recording yields no phase,
    and only one recorder can be active."""
    from wav2letter_pytorch_amd import _lib
    lib = _lib.lib
    orig = lib.w2l_conv_stats_mode
    seen = []
    with _lib.Recorder() as rec:
        assert _lib.recording() is rec and lib.w2l_conv_stats_mode is not orig
        lib.w2l_conv_stats_mode(3)
        lib.w2l_wgrad_deterministic(1)
        rec.python(seen.append, 'callback')
        lib.w2l_wgrad_deterministic(0)
        lib.w2l_conv_stats_mode(0)
        with pytest.raises(_lib.W2LError):
            with _lib.Recorder():
                pass
    assert _lib.recording() is None and lib.w2l_conv_stats_mode is orig
    phase = rec.finish()
    assert phase is not None and phase.n_calls == 4 and [it[0] for it in phase.items] == ['c', 'py', 'c']
    assert [phase.items[0][2], phase.items[2][2]] == [2, 2] and seen == ['callback']
    ops = _lib._replay_ops()
    first = phase.items[0][1]
    assert first[0].op == ops['w2l_conv_stats_mode'][0] and first[0].a[0].i == 3 and first[1].a[0].i == 1
    phase.replay()
    phase.replay()
    assert seen == ['callback'] * 3
    with _lib.Recorder() as rec2:
        lib.w2l_conv_stats_mode(0)
        _lib.poison('a torch op between launches')
    assert rec2.finish() is None and rec2.poisoned == 'a torch op between launches'
    with pytest.raises(ZeroDivisionError):
        with _lib.Recorder():
            1 / 0
    assert _lib.recording() is None and lib.w2l_conv_stats_mode is orig


def test_config_loader_hydra_tree(tmp_path):
    """defaults list, `# @package model` groups, ${a.b} interpolation, key=value and group overrides
    (the structure of configuration/config.yaml:1-28, rebuilt here from Python dicts)"""
    import yaml
    from wav2letter_pytorch_amd.config import instantiate, load_config
    (tmp_path / 'model').mkdir()
    (tmp_path / 'audio').mkdir()
    (tmp_path / 'optimizer').mkdir()
    (tmp_path / 'config.yaml').write_text(yaml.safe_dump({
        'defaults': [{'audio': 'standard_16k'}, {'optimizer': 'exp_lr_optimizer'}, {'model': 'wav2letter'}],
        'data': {'train_manifest': '???', 'batch_size': 4, 'mel_spec': '${model.input_size}', 'audio_conf': '${model.audio_conf}'},
        'model': {'input_size': 64, 'labels': 'english_lowercase',
                  'decoder': {'_target_': 'decoder.GreedyDecoder', 'labels': '${model.labels}'}},
        'trainer': {'default_root_dir': '.', 'max_epochs': 5, 'gpus': 0},
        'hydra': {'run': {'dir': '${trainer.default_root_dir}'}}}))
    (tmp_path / 'audio' / 'standard_16k.yaml').write_text('# @package model\naudio_conf:\n  window: hamming\n  sample_rate: 16000\n  window_size: 0.02\n')
    (tmp_path / 'optimizer' / 'exp_lr_optimizer.yaml').write_text(
        '# @package model\noptimizer:\n  _target_: torch.optim.SGD\n  lr: 1e-5\n  momentum: 0.9\n  nesterov: True\n'
        'scheduler:\n  _target_: torch.optim.lr_scheduler.ExponentialLR\n  gamma: 0.999\n')
    (tmp_path / 'model' / 'wav2letter.yaml').write_text(
        '# @package model\nname: wav2letter\nmid_layers: 1\nlayers:\n - output_size: 256\n   kernel_size: 11\n   stride: 2\n   dilation: 1\n   dropout: 0.2\n')
    (tmp_path / 'model' / 'jasper.yaml').write_text('# @package model\nname: jasper\nmid_layers: 1\njasper_blocks: []\n')
    cfg = load_config(str(tmp_path), ['model.mid_layers=3', 'trainer.gpus=1'])
    assert cfg.model.name == 'wav2letter' and cfg.model.mid_layers == 3 and cfg.trainer.gpus == 1
    assert cfg.data.mel_spec == 64 and cfg.data.audio_conf.sample_rate == 16000
    assert cfg.model.decoder.labels == 'english_lowercase' and cfg.model.layers[:1][0].kernel_size == 11
    assert cfg.model.get('print_decoded_prob', 0) == 0
    assert load_config(str(tmp_path), ['model=jasper']).model.name == 'jasper'
    dec = instantiate(cfg.model.decoder)
    assert type(dec).__name__ == 'GreedyDecoder' and len(dec.labels) == 29
    p = torch.nn.Parameter(torch.zeros(3))
    opt = instantiate(cfg.model.optimizer, params=[p])
    assert isinstance(opt, torch.optim.SGD)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _dp_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from wav2letter_pytorch_amd.distributed import GradReducer, broadcast_parameters, init_process_group_from_env
    r, w = init_process_group_from_env(backend='gloo')
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)
    lin = torch.nn.Linear(5, 3)
    conv_w = torch.nn.Parameter(torch.randn(4, 6, 8).permute(1, 2, 0))       # tap-major strided parameter
    mod = torch.nn.Module()
    mod.lin, mod.w = lin, conv_w
    broadcast_parameters(mod)
    red = GradReducer(small_bytes=64)
    g = torch.Generator().manual_seed(7 + rank)
    big = torch.randn(4, 6, 8, generator=g)                                   # dense storage of a conv gradient
    view = big.permute(1, 2, 0)
    smalls = [torch.randn(3, generator=g), torch.randn(5, generator=g)]
    red.on_grad(conv_w, view, big)
    for t in smalls:
        red.on_grad(None, t)
    pool = torch.randn(7, generator=g)                                         # the step engine's per-channel pool
    red.on_flat(pool)
    red.finish()
    # numpy payloads are pickled by value (torch tensors travel as shared-memory handles that die with this process)
    q.put((rank, mod.lin.weight.detach().numpy().copy(), conv_w.detach().numpy().copy(), big.numpy().copy(),
           [t.numpy().copy() for t in smalls] + [pool.numpy().copy()]))
    dist.barrier()
    dist.destroy_process_group()


def test_grad_reducer_gloo_world2():
    """N>1 path on CPU: identical replicas after broadcast, gradients averaged (large tensor on its own,
    small ones through the flattened bucket), strided gradient views reduced through their dense storage"""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res = [(r, torch.from_numpy(w), torch.from_numpy(c), torch.from_numpy(b), [torch.from_numpy(t) for t in s])
           for r, w, c, b, s in res]
    (_, w0, c0, big0, s0), (_, w1, c1, big1, s1) = res
    assert torch.equal(w0, w1) and torch.equal(c0, c1)                        # broadcast made the replicas identical
    assert torch.equal(big0, big1)
    exp = [torch.randn(4, 6, 8, generator=torch.Generator().manual_seed(7 + r)) for r in range(2)]
    assert torch.allclose(big0, (exp[0] + exp[1]) / 2, atol=1e-6)
    for a, b in zip(s0, s1):
        assert torch.equal(a, b)


def _dp4_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from wav2letter_pytorch_amd.distributed import GradReducer, init_process_group_from_env
    init_process_group_from_env(backend='gloo')
    red = GradReducer(small_bytes=64)
    out = []
    for step in range(3):
        g = torch.Generator().manual_seed(1000 * step + rank)
        # ---- "backward": the units top down; the top two units' weight gradients are held back (optim.FusedSGD.defer_wgrad),
        # the others are reduced as they appear; rank 1 is late at every other launch, rank 3 at the flush of the small ones
        held = [torch.randn(6, 5, 7, generator=g) for _ in range(2)]
        body = [torch.randn(4, 6, 8, generator=g) for _ in range(5)]
        smalls = [torch.randn(3 + i, generator=g) for i in range(4)]
        pool = torch.randn(33, generator=g)
        for i, t in enumerate(body):
            if rank == 1 and i % 2 == 0:
                time.sleep(0.05)
            red.on_grad(None, t.permute(1, 2, 0), t)
            red.on_grad(None, smalls[i % 4]) if i < 4 else None
        red.on_flat(pool)
        if rank == 3:
            time.sleep(0.08)
        red.finish()
        # ---- "next forward": the held-back gradients are reduced one by one in forward order (engine.flush_deferred), each
        # waited for before its update -- rank 2 arrives late, rank 0 is late between the two
        if rank == 2:
            time.sleep(0.06)
        pend = []
        for i, t in enumerate(reversed(held)):
            pend.append(red.start(t))
            if rank == 0 and i == 0:
                time.sleep(0.04)
        for p_ in pend:
            p_.finish()
        out.append([t.numpy().copy() for t in body + held + smalls + [pool]])
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_grad_reducer_gloo_world4_deferred_and_delayed():
    """four ranks, three steps of the data-parallel step's collective pattern with the deferral on -- large gradients as they
    appear, small ones through the flattened bucket at the end of the backward pass, the per-channel pool, then the held-back
    gradients one by one beside the 'next forward' -- while every rank is late somewhere else: nobody waits for a collective
    another rank has not started in the same order (the run finishes), and every buffer holds the mean over the ranks"""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp4_worker, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for step in range(3):
        gens = [torch.Generator().manual_seed(1000 * step + r) for r in range(4)]
        exp = None
        for g in gens:
            held = [torch.randn(6, 5, 7, generator=g) for _ in range(2)]
            body = [torch.randn(4, 6, 8, generator=g) for _ in range(5)]
            smalls = [torch.randn(3 + i, generator=g) for i in range(4)]
            pool = torch.randn(33, generator=g)
            cur = body + held + smalls + [pool]
            exp = cur if exp is None else [a + b for a, b in zip(exp, cur)]
        for r in range(4):
            for got, e in zip(res[r][step], exp):
                assert np.allclose(got, (e / 4).numpy(), atol=1e-6)


def test_novograd_matches_reference_fixture():
    """4 steps against tests/golden/novograd_cases.npz (generated from the reference's novograd.py)"""
    from wav2letter_pytorch_amd.novograd import Novograd
    z = np.load(os.path.join(GOLD, 'novograd_cases.npz'))
    for tag, kw in dict(plain=dict(lr=0.01, betas=(0.95, 0.5), weight_decay=1e-3, grad_averaging=True),
                        ams=dict(lr=0.02, betas=(0.9, 0.25), weight_decay=0.0, grad_averaging=False, amsgrad=True)).items():
        ps = [torch.nn.Parameter(torch.from_numpy(z[f'{tag}/p0_0'].copy())), torch.nn.Parameter(torch.from_numpy(z[f'{tag}/p0_1'].copy()))]
        opt = Novograd(ps, **kw)
        for it in range(4):
            ps[0].grad = torch.from_numpy(z[f'{tag}/grads0'][it].copy())
            ps[1].grad = torch.from_numpy(z[f'{tag}/grads1'][it].copy())
            opt.step()
        np.testing.assert_allclose(ps[0].detach().numpy(), z[f'{tag}/p4_0'], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(ps[1].detach().numpy(), z[f'{tag}/p4_1'], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(opt.state[ps[0]]['exp_avg_sq'].numpy(), z[f'{tag}/v_0'], rtol=1e-5)
    with pytest.raises(ValueError):
        Novograd(ps, betas=(1.0, 0))


def test_tune_cache_roundtrip(tmp_path):
    """w2l_tune_save / w2l_tune_load: host-only persistence of the measured block-shape / split-K choices."""
    from wav2letter_pytorch_amd import _lib as L
    src = tmp_path / 'in.txt'
    src.write_text('w2l-tune v2 gfx950\n'
                   'igemm 7 640 768 1000 21 1 1 1 15\n'
                   'igemm 7 640 768 1000 21 1 1 1 999\n'     # unknown configuration: skipped
                   'igemm 7 640 768 500 21 2 1 0 15\n'       # stride 2 only runs on shape 2: skipped
                   'wgrad 7 640 768 1000 21 3 1\n'
                   'wgrad 1 64 64 10 3 999 0\n'               # more splits than (n,t) steps: skipped
                   'wgradf8 7 640 768 1000 21 5 1\n'          # the e4m3 weight-gradient kernel's choices
                   'wgradf8 7 640 768 1000 21 5 4\n'          # no such block order: skipped
                   'igemmf8 7 640 768 1000 21 1 1 1 3\n'      # the e4m3 implicit-GEMM kernel's block shape
                   'igemmf8 7 640 768 1000 21 1 1 1 77\n'     # no such shape: skipped
                   'garbage\n')
    assert L.lib.w2l_tune_load(str(src).encode()) == 4
    out = tmp_path / 'out.txt'
    assert L.lib.w2l_tune_save(str(out).encode()) == 0
    lines = out.read_text().splitlines()
    assert lines[0] == 'w2l-tune v2 gfx950'
    assert 'igemm 7 640 768 1000 21 1 1 1 15' in lines and 'wgrad 7 640 768 1000 21 3 1' in lines
    assert 'wgradf8 7 640 768 1000 21 5 1' in lines and 'igemmf8 7 640 768 1000 21 1 1 1 3' in lines
    assert not any(' 99' in ln or ' 999' in ln for ln in lines)
    assert L.lib.w2l_tune_load(str(tmp_path / 'missing').encode()) == -1
    assert b'cannot open' in L.lib.w2l_last_error()
    bad = tmp_path / 'bad.txt'
    bad.write_text('something else\n')
    assert L.lib.w2l_tune_load(str(bad).encode()) == -1
    old = tmp_path / 'old.txt'              # a cache of the previous format (other configuration indices) is refused whole
    old.write_text('w2l-tune v1 gfx950\nigemm 7 640 768 1000 21 1 1 1 15\n')
    assert L.lib.w2l_tune_load(str(old).encode()) == -1


def test_data_loader_host_side(tmp_path):
    """load_audio (stdlib WAV path), manifest parsing, target mapping and the host collate (data_loader.py:20-31,90-158)"""
    import wave
    from wav2letter_pytorch_amd.data import data_loader as DL
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'features.npz'), allow_pickle=True)
    sig = (np.sin(np.arange(3200) * 0.01) * 0.5).astype(np.float32)
    p = str(tmp_path / 'a.wav')
    with wave.open(p, 'wb') as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
        w.writeframes((sig * 32767).astype('<i2').tobytes())
    a = DL.load_audio(p)
    assert a.dtype == np.float32 and a.shape == (3200,) and np.abs(a - sig).max() < 1e-4
    assert DL.load_audio(p, duration=0.05, offset=0.1).shape == (800,)
    assert DL._sample_rate(p) == 16000
    items = [(torch.from_numpy(z[f'spect{i}']), list(z['col_targets'][i]), 'f', 't') for i in range(int(z['n_cases']))]
    inputs, il, tg, tl, _, _ = DL._collator(items)
    np.testing.assert_array_equal(inputs.numpy(), z['col_inputs'])
    np.testing.assert_array_equal(il.numpy(), z['col_il'])
    np.testing.assert_array_equal(tg.numpy(), z['col_tg'])
    np.testing.assert_array_equal(tl.numpy(), z['col_tl'])
    with pytest.raises(RuntimeError):                   # no CPU feature path
        DL.SpectrogramExtractor(dict(window='hamming', window_stride=0.01, window_size=0.02, sample_rate=16000), 64)


def test_launcher_spawns_ranks_and_propagates_failure(tmp_path):
    """launch.spawn_ranks (what `bench.py --gpus N` and `train.py trainer.gpus=N` use when no launcher set the rendezvous
    variables): N ranks of one command line with torchrun's environment, a gloo all-reduce across them, rank 0's stdout
    forwarded; a failing rank stops the others and gives a non-zero exit code."""
    import subprocess
    import sys
    script = tmp_path / 'rank.py'
    script.write_text(
        'import os, sys, time\n'
        'import torch, torch.distributed as dist\n'
        'if sys.argv[1] == "fail":\n'
        '    if os.environ["RANK"] == "1":\n'
        '        sys.exit(3)\n'
        '    time.sleep(60)\n'
        'dist.init_process_group("gloo")\n'
        't = torch.tensor([float(dist.get_rank() + 1)])\n'
        'dist.all_reduce(t)\n'
        'print("SUM", float(t), os.environ["WORLD_SIZE"], os.environ["LOCAL_RANK"])\n'
        'dist.destroy_process_group()\n')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    drv = tmp_path / 'drv.py'
    drv.write_text(
        'import importlib.util, sys\n'
        f'spec = importlib.util.spec_from_file_location("l", {os.path.join(root, "wav2letter_pytorch_amd", "launch.py")!r})\n'
        'm = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)\n'
        f'sys.exit(m.spawn_ranks(int(sys.argv[1]), [sys.executable, {str(script)!r}, sys.argv[2]], timeout=100))\n')
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    ok = subprocess.run([sys.executable, str(drv), '3', 'ok'], capture_output=True, text=True, timeout=120, env=env)
    assert ok.returncode == 0, ok.stderr[-1000:]
    sums = [ln.split() for ln in ok.stdout.splitlines() if ln.startswith('SUM')]      # (gloo prints a banner of its own)
    assert sums == [['SUM', '6.0', '3', '0']]                      # only rank 0's stdout is forwarded
    t0 = time.time()
    bad = subprocess.run([sys.executable, str(drv), '2', 'fail'], capture_output=True, text=True, timeout=120, env=env)
    assert bad.returncode == 3 and time.time() - t0 < 45           # rank 0 (sleeping) was stopped, not waited for
    assert 'rank 1 exited with code 3' in bad.stderr


def test_init_weights_xavier_and_batchnorm_reset():
    """jasper.py:29-50: xavier_uniform(gain 1) on every Conv1d -- |w| <= sqrt(6 / (fan_in + fan_out)) with
    fan = channels * kernel width, the bound is approached -- and BatchNorm reset to mean 0 / var 1 / gamma 1 / beta 0,
    applied to the encoder and to the classifier (:434,:453)"""
    import math
    from wav2letter_pytorch_amd import Jasper
    from wav2letter_pytorch_amd.defaults import jasper_model
    from wav2letter_pytorch_amd.jasper import init_weights
    torch.manual_seed(1)
    m = Jasper(jasper_model(mid_layers=3))
    convs = [(k, p) for k, p in m.named_parameters() if k.endswith('conv.weight') or k == 'final_layer.0.weight']
    assert len(convs) == 3 * 2 + 2 + 1                       # 3 separable blocks (dw + pw), 2 residual convs, the head
    for k, w in convs:
        cout, cin_g, kw = w.shape
        bound = math.sqrt(6.0 / (cin_g * kw + cout * kw))
        assert float(w.abs().max()) <= bound * (1 + 1e-6), k
        assert float(w.abs().max()) > 0.9 * bound or w.numel() < 200, k
        assert abs(float(w.mean())) < 0.1 * bound
    with torch.no_grad():
        for k, t in list(m.state_dict().items()):
            if 'running' in k or k.endswith('.weight') and t.dim() == 1 or k.endswith('.bias') and 'mconv' in k:
                t.add_(3.0)
        bn = m.jasper_encoder[0].mconv[2]
        bn.num_batches_tracked.fill_(7)
    m.jasper_encoder.apply(init_weights)
    for k, t in m.state_dict().items():
        if 'running_mean' in k or ('mconv' in k and k.endswith('.bias')):
            assert float(t.abs().max()) == 0, k
        elif 'running_var' in k or ('mconv' in k and k.endswith('.weight') and t.dim() == 1):
            assert bool((t == 1).all()), k
        elif 'num_batches' in k:
            assert int(t) == 0, k
    with pytest.raises(ValueError):
        init_weights(m.final_layer[0], mode='kaiming')
    x, lens = m.create_example_input_array()                 # base_asr_models.py:27-31
    assert x.shape == (4, 64, 200) and float(x.min()) >= 0 and float(x.max()) < 1
    assert lens.shape == (4,) and int(lens.min()) >= 100 and int(lens.max()) < 200


def test_launch_parents_never_map_the_hip_library(tmp_path):
    """`python -m wav2letter_pytorch_amd.train ... trainer.gpus=N` and `python bench.py --gpus N` first run in a parent that
    only spawns the ranks (launch.py): that process must not have loaded libw2l_hip.so (nor torch), so that no HIP / HSA
    call can have happened before the children start.  Checked in a fresh interpreter up to the point where the parent
    would call spawn_ranks."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys\n"
        "import wav2letter_pytorch_amd.train as t\n"
        "cfg = t.build_config(['data.train_manifest=a.csv', 'data.val_manifest=b.csv', 'trainer.gpus=2', 'model=jasper'])\n"
        "assert t._requested_gpus(cfg) == 2 and cfg.model.name == 'jasper' and not t.under_launcher()\n"
        "bad = [m for m in sys.modules if m.endswith('._lib') or m == 'torch']\n"
        "assert not bad, bad\n"
        "import wav2letter_pytorch_amd as W\n"
        "assert W.Wav2Letter.__name__ == 'Wav2Letter' and any(m.endswith('._lib') for m in sys.modules)\n"
        "print('clean')\n")
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE')}
    out = subprocess.run([sys.executable, '-c', code], cwd=root, env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0 and 'clean' in out.stdout, out.stderr[-2000:]


def test_visible_gpu_count_needs_no_hip(monkeypatch):
    """the launch parent counts devices from the *_VISIBLE_DEVICES variables / sysfs, never through the runtime"""
    from wav2letter_pytorch_amd import launch
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', '0,1,2')
    assert launch.visible_gpu_count() == 3
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', '')
    assert launch.visible_gpu_count() == 0
    for var in ('HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES'):
        monkeypatch.delenv(var, raising=False)
    n = launch.visible_gpu_count()
    assert n is None or n >= 0


def test_synthetic_batch_ragged_layout():
    """SURVEY 8d's second measurement run: input lengths U{T/2..T} with the longest at T, spectrograms zero beyond their length
    (the collator's right padding, data_loader.py:149-158), every CTC alignment feasible; the full-length batch is unchanged
    by the option's existence"""
    from wav2letter_pytorch_amd.defaults import synthetic_batch
    x, il, tg, tl = synthetic_batch(16, 400, seed=5, ragged=True)
    assert int(il.max()) == 400 and int(il.min()) >= 200 and len(set(il.tolist())) > 4
    for n in range(16):
        assert not x[n, :, int(il[n]):].any() and x[n, :, :int(il[n])].abs().sum() > 0
        assert 2 * int(tl[n]) <= int(il[n]) // 2 and not tg[n, int(tl[n]):].any() and (tg[n, :int(tl[n])] > 0).all()
    xf, ilf, tgf, tlf = synthetic_batch(16, 400, seed=5)
    assert (ilf == 400).all() and torch.equal(xf[:, :, :200], synthetic_batch(16, 400, seed=5)[0][:, :, :200])


@pytest.mark.parametrize('tiles,S,G', [(175, 256, 512), (324, 256, 512), (490, 256, 512), (245, 256, 256), (96, 256, 512),
                                       (512, 256, 512), (1, 24, 7), (18, 18, 20), (100, 2000, 512), (36, 15, 256), (7, 3, 8)])
def test_dealt_stream_k_decomposition_covers_every_step_once(tiles, S, G):
    """the dealt stream-K plan of the weight gradient (include/w2l_hip.h: w2l_wgrad_dealt_segments runs the SAME arithmetic as
    the kernel and the launcher, csrc/conv_wgrad_kernel.h dealt_segment): every (tile, K step) belongs to exactly one block, the
    blocks of a tile are numbered 0 .. n-1 in step order (the order the last arriver sums their slabs in), the first G blocks
    are the ranges' first segments, the following ones the second segments, longest first, and every slot's two segments add
    up to one range"""
    import ctypes as C
    from wav2letter_pytorch_amd import _lib
    buf = (C.c_int * (6 * 2 * G))()
    n = _lib.lib.w2l_wgrad_dealt_segments(tiles, S, G, buf, 2 * G)
    assert G <= n <= 2 * G
    rows = np.array(buf[:6 * n]).reshape(n, 6)
    assert (rows[:G, 0] >= 0).all()
    # second segments are dealt per XCD (block G + j runs where blocks j % 8 ran, and takes one of THEIR ranges' second
    # segments, longest first; -1 = a padding position: that XCD has none left)
    q, rem = G >> 3, G & 7
    xcd_of = {}
    for x in range(8):
        base = x * (q + 1) if x < rem else rem * (q + 1) + (x - rem) * q
        for r in range(base, base + (q + 1 if x < rem else q)):
            xcd_of[r] = x
    for x in range(8):
        mine = rows[G + x::8]
        mine = mine[mine[:, 0] >= 0]
        assert all(xcd_of[int(r)] == x for r in mine[:, 0])
        ln = mine[:, 3] - mine[:, 2]
        assert (ln[:-1] >= ln[1:]).all() and (mine[:, 2] == 0).all()
    rows = rows[rows[:, 0] >= 0]
    cover = np.zeros((tiles, S), dtype=np.int32)
    per_tile = {}
    for r, t, s0, s1, place, cnt in rows:
        assert 0 <= t < tiles and 0 <= s0 < s1 <= S and 0 <= place < cnt
        cover[t, s0:s1] += 1
        per_tile.setdefault(int(t), []).append((int(s0), int(place), int(cnt), int(r)))
    assert (cover == 1).all()
    for t, segs in per_tile.items():
        segs.sort()
        assert [p for _, p, _, _ in segs] == list(range(len(segs))) and all(c == len(segs) for _, _, c, _ in segs)
        assert [r for _, _, _, r in segs] == list(range(segs[0][3], segs[0][3] + len(segs)))     # consecutive ranges: slab r + t
    # first segments of all G ranges (each range once), then second segments by descending length
    assert sorted(rows[:G, 0]) == list(range(G))
    length = {}
    for r, t, s0, s1, _, _ in rows:
        length[int(r)] = length.get(int(r), 0) + int(s1 - s0)
    W = tiles * S
    assert all(length[r] == W * (r + 1) // G - W * r // G for r in range(G))
    # no dealt form: more tiles than ranges, or fewer steps than ranges
    assert _lib.lib.w2l_wgrad_dealt_segments(G + 1, S, G, buf, 2 * G) == -1


def test_wgrad_group_planner():
    """wgrad_groups.plan (host arithmetic only): groups are consecutive among the groupable convolutions of the backward order,
    share dilation and (N, T'), respect the size cap, never contain a non-groupable entry; the Wav2Letter table with its top
    six units held back plans the partition that measured best in the step ({640 -> 768} with the classifier, {640 -> 640 x 2,
    512 -> 640}, {512 -> 512 x 2, 384 -> 512}); tile counts agree with the library's; overrides parse"""
    from wav2letter_pytorch_amd import _lib, wgrad_groups as G
    sys.path.insert(0, ROOT)
    from oracle.w2l_oracle import W2L_LAYERS
    cins = [64] + [l[0] for l in W2L_LAYERS]
    table = [(cins[i], l[0], l[1], l[2], l[3]) for i, l in enumerate(W2L_LAYERS)] + [(1024, 64, 1, 1, 1)]       # + the classifier
    rows = list(range(len(table) - 1, -1, -1))                      # backward order: classifier, then the units top down

    def seq_for(defer, N=32):
        out = []
        for r in rows:
            cin, cout, kw, s, d = table[r]
            held = r != len(table) - 1 and (len(table) - 2 - r) < defer
            out.append(None if (s != 1 or held) else (cin, cout, kw, d, (N, 500)))
        return out

    for defer in (0, 4, 6):
        for mx in (2, 3, 8):
            seq = seq_for(defer)
            groups = G.plan(seq, max_group=mx)
            flat = [i for g in groups for i in g]
            assert len(flat) == len(set(flat)) and all(2 <= len(g) <= mx for g in groups)
            ok = [i for i, e in enumerate(seq) if e is not None]
            for g in groups:
                assert all(seq[i] is not None for i in g)
                assert len({(seq[i][3], seq[i][4]) for i in g}) == 1
                pos = [ok.index(i) for i in g]
                assert pos == list(range(pos[0], pos[0] + len(g)))             # consecutive among the groupable entries
    best = [[rows[i] for i in g] for g in G.plan(seq_for(6), max_group=3)]
    assert best == [[20, 13], [12, 11, 10], [9, 8, 7]], best
    for cin, cout, kw in [(640, 640, 21), (512, 640, 21), (896, 896, 29), (1024, 64, 1), (64, 256, 11)]:
        for form in G.FORMS:
            assert G.tiles(cin, cout, kw, form) == _lib.lib.w2l_wgrad_group_tiles(cin, cout, kw, form)
    assert G.parse_override('auto', 10) is None and G.parse_override('0', 10) == [] and G.parse_override('off', 10) == []
    assert G.parse_override('8,9,10;11,12', 21) == [[8, 9, 10], [11, 12]]
    assert G.parse_override('3;4,99', 21) == []                        # singletons and out-of-range indices are dropped


@pytest.mark.parametrize('tiles,steps,G', [(140, 210, 256), (329, 406, 256), (60, 56, 512), (1, 300, 256), (700, 9, 256), (255, 257, 256)])
def test_igemm_stream_k_pieces_cover_every_step_once(tiles, steps, G):
    """w2l_conv_streamk_pieces runs the function the stream-K implicit GEMM runs per block (sk_piece, host + device): every
    (tile, step) belongs to exactly one piece; the pieces of a cut tile are numbered 0..n-1 in step order with consecutive slab
    ids, unique over the launch and below tiles + G (the workspace the launcher checks); whole tiles take no slab; every range
    owns floor(W (r+1) / G) - floor(W r / G) steps."""
    import numpy as np
    from wav2letter_pytorch_amd import _lib
    cap = tiles + 2 * G + 8
    buf = np.zeros(7 * cap, dtype=np.int32)
    n = _lib.lib.w2l_conv_streamk_pieces(tiles, steps, G, buf.ctypes.data, cap)
    assert n > 0
    pc = buf[:7 * n].reshape(n, 7)
    W = tiles * steps
    cover = np.zeros((tiles, steps), dtype=np.int32)
    per_range = np.zeros(G, dtype=np.int64)
    for r, t, b, e, ns, sp, slab in pc:
        assert 0 <= t < tiles and 0 <= b < e <= steps and 0 <= sp < ns
        cover[t, b:e] += 1
        per_range[r] += e - b
        assert (slab == -1) == (ns == 1)
    assert (cover == 1).all()
    assert (per_range == [W * (r + 1) // G - W * r // G for r in range(G)]).all()
    assert (np.diff(pc[:, 0]) >= 0).all()                       # range order
    slabs = pc[pc[:, 6] >= 0, 6]
    assert len(set(slabs.tolist())) == len(slabs) and (slabs < tiles + G).all()
    for t in range(tiles):
        mine = pc[pc[:, 1] == t]
        mine = mine[np.argsort(mine[:, 2])]
        assert (mine[:, 4] == len(mine)).all() and (mine[:, 5] == np.arange(len(mine))).all()
        assert mine[0, 2] == 0 and mine[-1, 3] == steps and (mine[1:, 2] == mine[:-1, 3]).all()
        if len(mine) > 1:
            assert (np.diff(mine[:, 6]) == 1).all()
    assert _lib.lib.w2l_conv_streamk_pieces(tiles, steps, G, buf.ctypes.data, 1) == (-1 if n > 1 else 1)
    assert _lib.lib.w2l_conv_streamk_pieces(1 << 20, 4096, 512, buf.ctypes.data, cap) == -1

