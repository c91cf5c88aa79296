"""Feature front-end and augmentation masks on the GPU vs the CPU oracle and the reference-generated
fixture tests/golden/features.npz (data/data_loader.py:33-88,149-158; data/augmentations.py)."""
import json
import os
import random
import wave

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
CONF = dict(window='hamming', window_stride=0.01, window_size=0.02, sample_rate=16000)
# fp32 features after mean/std normalisation are O(1); the device FFT (radix-2 in LDS) and torch's CPU FFT differ by
# ~1e-6 of the largest bin, which log1p + the division by the per-feature std turn into a few 1e-5 absolute
TOL = 2e-4


@pytest.fixture(scope='module')
def fx():
    return np.load(os.path.join(HERE, 'golden', 'features.npz'), allow_pickle=True)


@pytest.fixture(scope='module')
def ext():
    from wav2letter_pytorch_amd.data.data_loader import SpectrogramExtractor
    return SpectrogramExtractor(CONF, mel_spec=64)


def test_extract_matches_reference_fixture(fx, ext):
    assert ext.n_fft == int(fx['n_fft'])
    np.testing.assert_array_equal(ext.fb[0].cpu().numpy(), fx['fb'])
    np.testing.assert_allclose(ext.window.cpu().numpy(), fx['window'], rtol=0, atol=1e-7)
    for i in range(int(fx['n_cases'])):
        got = ext.extract(fx[f'audio{i}'], noise=fx[f'noise{i}']).cpu().numpy()
        want = fx[f'spect{i}']
        assert got.shape == want.shape
        assert np.abs(got - want).max() < TOL, (i, np.abs(got - want).max())


def test_mel_power_matches_oracle(fx, ext):
    from oracle import features_oracle as FO
    for i in range(int(fx['n_cases'])):
        got = ext._get_spect(fx[f'audio{i}'], noise=fx[f'noise{i}'])[0].cpu().numpy()
        want = FO.mel_power(fx[f'audio{i}'], fx[f'noise{i}'], CONF).numpy()
        assert got.shape == want.shape
        assert np.abs(got - want).max() <= 2e-5 * np.abs(want).max()


def test_batch_layout_is_the_collators(fx, ext):
    n = int(fx['n_cases'])
    audio = [fx[f'audio{i}'] for i in range(n)]
    noise = [fx[f'noise{i}'] for i in range(n)]
    inputs, lens = ext.extract_batch(audio, noise=noise)
    assert lens.dtype == torch.int32 and not lens.is_cuda
    np.testing.assert_array_equal(lens.numpy(), fx['col_il'])
    got = inputs.cpu().numpy()
    want = fx['col_inputs']
    assert got.shape == want.shape
    assert np.abs(got - want).max() < TOL
    for i in range(n):                                   # the padding is exactly zero
        assert not got[i, :, int(lens[i]):].any()


def test_collator_on_device_tensors(fx, ext):
    from wav2letter_pytorch_amd.data.data_loader import _collator
    n = int(fx['n_cases'])
    items = [(torch.from_numpy(fx[f'spect{i}']).cuda(), list(fx['col_targets'][i]), 'f%d.wav' % i, 't%d' % i) for i in range(n)]
    inputs, il, tg, tl, paths, texts = _collator(items)
    np.testing.assert_array_equal(inputs.cpu().numpy(), fx['col_inputs'])
    np.testing.assert_array_equal(il.numpy(), fx['col_il'])
    np.testing.assert_array_equal(tg.numpy(), fx['col_tg'])
    np.testing.assert_array_equal(tl.numpy(), fx['col_tl'])
    assert tg.dtype == torch.int32 and tl.dtype == torch.int32


def test_spec_augment_and_cutout_bit_exact(fx):
    from wav2letter_pytorch_amd.data.augmentations import Identity, SpecAugment, SpecCutout
    x = torch.from_numpy(fx['aug_x']).cuda()
    got = SpecAugment(freq_masks=2, time_masks=2, freq_width=15, time_width=50, rng=random.Random(11))(x)
    np.testing.assert_array_equal(got.cpu().numpy(), fx['specaug'])
    np.testing.assert_array_equal(x.cpu().numpy(), fx['aug_x'])          # input untouched (masked_fill semantics)
    got = SpecCutout(rect_masks=5, rect_time=60, rect_freq=25, rng=random.Random(12))(x)
    np.testing.assert_array_equal(got.cpu().numpy(), fx['speccut'])
    xs = torch.from_numpy(fx['aug_xs']).cuda()          # T < time_width: negative starts follow Python slice rules
    got = SpecAugment(freq_masks=1, time_masks=2, freq_width=15, time_width=50, rng=random.Random(13))(xs)
    np.testing.assert_array_equal(got.cpu().numpy(), fx['specaug_short'])
    assert Identity()(x) is x


def _write_wav(path, samples, sr=16000):
    with wave.open(path, 'wb') as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(sr)
        w.writeframes((np.clip(samples, -1, 1) * 32767).astype('<i2').tobytes())


def test_dataset_and_loader_end_to_end(tmp_path):
    from oracle import features_oracle as FO
    from wav2letter_pytorch_amd.data import label_sets
    from wav2letter_pytorch_amd.data.data_loader import BatchAudioDataLoader, SpectrogramDataset, load_audio
    labels = label_sets.labels_map['english_lowercase']
    g = np.random.default_rng(0)
    texts = ['hello world', 'a_b', 'mi three fifty five x', "it's"]
    rows = []
    for i, L in enumerate((8000, 12345, 4001, 16000)):
        p = str(tmp_path / f'u{i}.wav')
        _write_wav(p, 0.2 * g.standard_normal(L))
        rows.append({'audio_filepath': p, 'text': texts[i]})
    man = str(tmp_path / 'm.json')
    with open(man, 'w') as f:
        for r in rows:
            f.write(json.dumps(r) + '\n')
    ds = SpectrogramDataset(man, CONF, labels, mel_spec=64)
    ds.extractor.dithering = 0.0                        # the dither is random; everything else must match the oracle
    assert len(ds) == 4 and ds.data_channels() == 64
    spect, target, path, text = ds[1]
    want1 = FO.extract(load_audio(rows[1]['audio_filepath']), None, CONF)
    assert spect.shape == want1.shape and np.abs(spect.cpu().numpy() - want1).max() < TOL
    # '_' is the blank (index 0) and is dropped by filter(None, ...) like unknown characters (data_loader.py:127)
    assert target == [labels.index('a'), labels.index('b')]
    loader = BatchAudioDataLoader(ds, batch_size=3)
    batches = list(loader)
    assert [b[0].shape[0] for b in batches] == [3, 1]
    inputs, il, tg, tl, paths, txts = batches[0]
    assert inputs.is_cuda and inputs.dtype == torch.float32 and il.dtype == torch.int32
    assert tuple(paths) == tuple(r['audio_filepath'] for r in rows[:3]) and tuple(txts) == tuple(texts[:3])
    specs = [FO.extract(load_audio(r['audio_filepath']), None, CONF) for r in rows[:3]]
    tgt = [[labels.index(c) for c in t if c in labels and labels.index(c) != 0] for t in texts[:3]]
    wx, wil, wtg, wtl = FO.collate(specs, tgt)
    assert np.abs(inputs.cpu().numpy() - wx).max() < TOL
    np.testing.assert_array_equal(il.numpy(), wil)
    np.testing.assert_array_equal(tg.numpy(), wtg)
    np.testing.assert_array_equal(tl.numpy(), wtl)
    # csv manifest with offset / duration columns
    import pandas as pd
    pd.DataFrame([dict(r, offset=0.1, duration=0.3) for r in rows]).to_csv(str(tmp_path / 'm.csv'))
    ds2 = SpectrogramDataset(str(tmp_path / 'm.csv'), CONF, labels, mel_spec=64)
    ds2.extractor.dithering = 0.0
    a = load_audio(rows[0]['audio_filepath'], 0.3, 0.1)
    assert a.shape[0] == 4800
    assert np.abs(ds2[0][0].cpu().numpy() - FO.extract(a, None, CONF)).max() < TOL


def test_short_utterance_is_rejected(ext):
    with pytest.raises(ValueError):
        ext.extract(np.zeros(200, dtype=np.float32))


def test_full_size_batch_properties(ext):
    """BASELINE shape: 32 utterances x 10 s -> 64 mel x 1001 frames; size-independent properties."""
    g = torch.Generator().manual_seed(0)
    lens = [160000] + [int(v) for v in torch.randint(80000, 160001, (31,), generator=g)]
    audio = [(0.1 * torch.randn(L, generator=g)).numpy() for L in lens]
    inputs, il = ext.extract_batch(audio)
    assert inputs.shape == (32, 64, 1001)
    np.testing.assert_array_equal(il.numpy(), 1 + np.array(lens) // 160)
    x = inputs.cpu()
    assert torch.isfinite(x).all()
    for n in (0, 7, 31):
        t = int(il[n])
        v = x[n, :, :t]
        sd = v.std(1)                                   # = s / (s + 1e-5) with s the raw feature's std: at most 1
        assert v.mean(1).abs().max() < 1e-3 and sd.max() <= 1 + 1e-4 and sd.min() > 0.5
        assert not x[n, :, t:].any()
    # linearity of the un-normalised power path: scaling the audio by 2 scales mel power by 4
    a = audio[1][:20000]
    p1 = ext._get_spect(a, noise=np.zeros_like(a))
    p2 = ext._get_spect(2 * a, noise=np.zeros_like(a))
    assert ((p2 - 4 * p1).abs().max() <= 1e-5 * p2.abs().max())
    # per-utterance result does not depend on its batch neighbours
    solo = ext.extract(audio[5], noise=np.zeros_like(audio[5]))
    both, _ = ext.extract_batch([audio[5], audio[0]], noise=[np.zeros_like(audio[5]), np.zeros_like(audio[0])])
    assert torch.equal(solo, both[0, :, :solo.shape[1]])


def _toy_corpus(tmp_path, n=6):
    g = np.random.default_rng(3)
    words = ['speech', 'on', 'mi', 'three', 'fifty', 'five', 'x', 'runs', 'fast']
    rows = []
    for i in range(n):
        p = str(tmp_path / f'c{i}.wav')
        _write_wav(p, 0.2 * g.standard_normal(int(g.integers(9000, 16000))))
        rows.append({'audio_filepath': p, 'text': ' '.join(g.choice(words, 3))})
    import pandas as pd
    pd.DataFrame(rows[:4]).to_csv(str(tmp_path / 'train.csv'))
    pd.DataFrame(rows[4:]).to_csv(str(tmp_path / 'val.csv'))
    return str(tmp_path / 'train.csv'), str(tmp_path / 'val.csv')


def test_train_cli_end_to_end_builtin_config(tmp_path):
    """the reference's `python train.py data.train_manifest=... ` flow (train.py:28-41; BASELINE config 1 'plumbing'):
    manifests -> GPU features -> model chosen by cfg.model.name -> fit loop -> checkpoint with the reference's keys"""
    from wav2letter_pytorch_amd.train import main
    tr, va = _toy_corpus(tmp_path)
    out = tmp_path / 'run'
    trainer, model = main([f'data.train_manifest={tr}', f'data.val_manifest={va}', 'data.batch_size=2', 'model.mid_layers=2',
                           'trainer.max_epochs=2', f'trainer.default_root_dir={out}', 'model.optimizer.lr=0.01'])
    assert trainer.global_step == 4
    logs = trainer.logged[-1][1]
    assert {'train_loss', 'learning_rate', 'train_cer', 'train_wer'} <= set(logs) and np.isfinite(logs['train_loss'])
    assert {'val_loss', 'val_cer', 'val_wer'} <= set(model._logged)
    ck = [f for f in os.listdir(out) if f.endswith('.ckpt')]
    assert len(ck) == 2
    sd = torch.load(os.path.join(out, sorted(ck)[-1]))['state_dict']
    assert 'conv1ds.conv1d_0.conv1.weight' in sd and 'conv1ds.conv1d_1.batch_norm.running_var' in sd


def test_data_parallel_loaders_shard_the_manifest(tmp_path):
    """train.get_data_loaders(rank, world): the ranks of a data-parallel run read DISJOINT utterances that together cover the
    manifest (Lightning's DDP injects a DistributedSampler into the reference's loaders, train.py:21-26,34-37), and every
    rank takes the same number of steps"""
    from wav2letter_pytorch_amd.defaults import root_config
    from wav2letter_pytorch_amd.train import get_data_loaders
    tr, va = _toy_corpus(tmp_path, n=9)                    # 4 training utterances, 5 validation utterances
    cfg = root_config('wav2letter')
    cfg.data.train_manifest, cfg.data.val_manifest, cfg.data.batch_size = tr, va, 1
    labels = list(cfg.model.labels)
    seen = []
    for rank in range(2):
        tl, vl = get_data_loaders(labels, cfg.data, rank, 2)
        seen.append(([b[4][0] for b in tl], [b[4][0] for b in vl]))
    (t0, v0), (t1, v1) = seen
    assert len(t0) == len(t1) == 2 and not set(t0) & set(t1) and len(set(t0) | set(t1)) == 4
    assert len(v0) == len(v1) == 3 and len(set(v0) | set(v1)) == 5      # 5 over 2 ranks: one utterance repeated as padding
    one, _ = get_data_loaders(labels, cfg.data)
    assert [b[4][0] for b in one] == [t for pair in zip(t0, t1) for t in pair]     # rank r reads items r, r + world, ...


def test_stream_probe_and_measured_side_streams(monkeypatch):
    """w2l_stream_probe / streams.concurrent_stream: a stream probed against itself reads as queued (the stamp kernel starts
    after the fill kernel has ended), the stream handed out for a role runs beside the main stream and beside the roles
    chosen before it, and a role keeps its stream for the life of the process"""
    from wav2letter_pytorch_amd import streams as S
    dev = torch.device('cuda', 0)
    main = torch.cuda.current_stream(dev)
    assert S.overlap_fraction(main, main, dev) > 0.9
    # a registry of its own: the roles other tests of this process have filled stay theirs (four busy streams is what the
    # command processor runs side by side; a fifth role has nowhere to go)
    monkeypatch.setitem(S._chosen, 0, {})
    monkeypatch.setitem(S._main, 0, main)
    a = S.concurrent_stream(dev, 'test-role-a', main=main)
    b = S.concurrent_stream(dev, 'test-role-b')
    assert a is S.concurrent_stream(dev, 'test-role-a') and b is S.concurrent_stream(dev, 'test-role-b')
    assert len({a.cuda_stream, b.cuda_stream, main.cuda_stream}) == 3
    for x, y in ((main, a), (main, b), (a, b)):
        assert S.serialise(x, y, dev) < S.SERIAL_FRAC
    assert {'test-role-a', 'test-role-b'} <= set(S.chosen(dev))
    assert any(r[1] == 'test-role-b' and r[3] < S.SERIAL_FRAC for r in S.report)


def test_stream_probe_warns_when_nothing_runs_beside(monkeypatch):
    """if every candidate reads as queued behind a busy stream (more busy streams than the command processor runs side by
    side), the role still gets a stream -- the best seen -- and the user is told that the overlap is lost"""
    from wav2letter_pytorch_amd import streams as S
    dev = torch.device('cuda', 0)
    monkeypatch.setitem(S._chosen, 0, {})
    monkeypatch.setitem(S._main, 0, torch.cuda.current_stream(dev))
    monkeypatch.setattr(S, 'TRIES', 3)
    seen = iter([0.9, 0.6, 0.8])
    monkeypatch.setattr(S, 'serialise', lambda a, b, d, reps=2: next(seen))
    with pytest.warns(UserWarning, match='none of 3 candidate streams'):
        st = S.concurrent_stream(dev, 'test-role-crowded')
    assert isinstance(st, torch.cuda.Stream)
    assert S.report[-1][1:] == ('test-role-crowded', 3, 0.6)


def test_native_rccl_helpers_one_rank():
    """include/w2l_hip.h's RCCL helpers (w2l_rccl_unique_id / init / world / all_reduce / broadcast / destroy) on a 1-rank
    communicator, and a training step whose gradients go through them (GradReducer(native=True)) -- see the worker"""
    import socket
    import subprocess
    import sys
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    env.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'native_rccl_worker.py')
    res = subprocess.run([sys.executable, worker], env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert 'NATIVE_RCCL_OK' in res.stdout


def test_train_cli_two_ranks_and_resume(tmp_path):
    """`trainer.gpus=2` (the reference's Lightning flag): the command line starts two ranks itself (gloo rehearsal on one
    GPU), each trains on its shard, only rank 0 writes checkpoints; the checkpoint carries optimizer and scheduler state
    under Lightning's keys and a second run resumes from it."""
    import subprocess
    import sys
    tr, va = _toy_corpus(tmp_path)
    out = tmp_path / 'run'
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env['W2L_DIST_BACKEND'] = 'gloo'
    cmd = [sys.executable, '-m', 'wav2letter_pytorch_amd.train', f'data.train_manifest={tr}', f'data.val_manifest={va}',
           'data.batch_size=1', 'model.mid_layers=2', 'trainer.max_epochs=2', f'trainer.default_root_dir={out}',
           'model.optimizer.lr=0.01', 'trainer.gpus=2', 'trainer.log_every_n_steps=1']
    run = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=280)
    assert run.returncode == 0, run.stderr[-2000:]
    ck = sorted(f for f in os.listdir(out) if f.endswith('.ckpt'))
    assert ck == ['epoch=0-step=2.ckpt', 'epoch=1-step=4.ckpt']           # 4 utterances / 2 ranks / batch 1 = 2 steps an epoch
    assert run.stdout.count('epoch 0 done') == 1                          # rank 0 alone reports
    state = torch.load(os.path.join(out, ck[-1]))
    assert {'state_dict', 'epoch', 'global_step', 'optimizer_states', 'lr_schedulers'} <= set(state)
    mom = [v for v in state['optimizer_states'][0]['state'].values() if v.get('momentum_buffer') is not None]
    assert len(mom) == len(state['optimizer_states'][0]['param_groups'][0]['params'])
    assert abs(state['lr_schedulers'][0]['_last_lr'][0] - 0.01 * 0.999 ** 2) < 1e-9
    # resume (single process): continues at epoch 2 with the saved momentum and learning rate
    from wav2letter_pytorch_amd.train import main
    trainer, model = main([f'data.train_manifest={tr}', f'data.val_manifest={va}', 'data.batch_size=2', 'model.mid_layers=2',
                           'trainer.max_epochs=3', f'trainer.default_root_dir={out}', 'model.optimizer.lr=0.01',
                           f'trainer.resume_from_checkpoint={os.path.join(out, ck[-1])}'])
    assert trainer.current_epoch == 2 and trainer.global_step == 4 + 2
    assert abs(model.optimizers().param_groups[0]['lr'] - 0.01 * 0.999 ** 3) < 1e-9
    assert len(trainer.val_logged) == 1 and {'val_loss', 'val_cer', 'val_wer'} <= set(trainer.val_logged[0])


def test_train_cli_yaml_tree_and_jasper(tmp_path):
    """--config-dir: a Hydra-style tree with the reference's keys (written here), model=jasper group override"""
    import yaml
    from wav2letter_pytorch_amd.train import main
    tr, va = _toy_corpus(tmp_path)
    cd = tmp_path / 'configuration'
    for g in ('model', 'audio', 'optimizer'):
        (cd / g).mkdir(parents=True)
    (cd / 'config.yaml').write_text(yaml.safe_dump({
        'defaults': [{'audio': 'standard_16k'}, {'optimizer': 'exp_lr_optimizer'}, {'model': 'wav2letter'}],
        'data': {'train_manifest': '???', 'val_manifest': '???', 'batch_size': 2, 'mel_spec': '${model.input_size}',
                 'audio_conf': '${model.audio_conf}'},
        'model': {'input_size': 64, 'labels': 'english_lowercase',
                  'decoder': {'_target_': 'decoder.GreedyDecoder', 'labels': '${model.labels}'}},
        'trainer': {'default_root_dir': str(tmp_path / 'run2'), 'max_epochs': 1, 'max_steps': None, 'gpus': 0}}))
    (cd / 'audio' / 'standard_16k.yaml').write_text(
        '# @package model\naudio_conf:\n  window: hamming\n  window_stride: 0.01\n  window_size: 0.02\n  sample_rate: 16000\n')
    (cd / 'optimizer' / 'exp_lr_optimizer.yaml').write_text(
        '# @package model\noptimizer:\n  _target_: torch.optim.SGD\n  lr: 1e-3\n  momentum: 0.9\n  nesterov: True\n  weight_decay: 1e-5\n'
        'scheduler:\n  _target_: torch.optim.lr_scheduler.ExponentialLR\n  gamma: 0.999\n')
    (cd / 'model' / 'wav2letter.yaml').write_text('# @package model\nname: wav2letter\nmid_layers: 1\nlayers: []\n')
    (cd / 'model' / 'jasper.yaml').write_text('# @package model\n' + yaml.safe_dump({'name': 'jasper', 'mid_layers': 2, 'jasper_blocks': [
        {'layer_size': 64, 'kernel_size': 32, 'stride': 2, 'residual': False, 'separable': True},
        {'layer_size': 64, 'kernel_size': 32, 'stride': 1, 'residual': True, 'separable': True}]}))
    trainer, model = main(['--config-dir', str(cd), 'model=jasper', f'data.train_manifest={tr}', f'data.val_manifest={va}'])
    assert type(model).__name__ == 'Jasper' and trainer.global_step == 2
    assert np.isfinite(trainer.logged[-1][1]['train_loss'])


@pytest.mark.parametrize('window_size,window', [(0.016, 'hann'), (0.05, 'hamming')])
def test_other_fft_sizes_use_the_generic_kernel(fx, window_size, window):
    """n_fft = 256 (16 ms window: no zero padding of the window) and n_fft = 1024 (50 ms) go through the LDS radix-2 kernel"""
    from oracle import features_oracle as FO
    from wav2letter_pytorch_amd.data.data_loader import SpectrogramExtractor
    conf = dict(window=window, window_stride=0.01, window_size=window_size, sample_rate=16000)
    e = SpectrogramExtractor(conf, mel_spec=40)
    assert e.n_fft in (256, 1024)
    for i in (1, 3):
        got = e.extract(fx[f'audio{i}'], noise=fx[f'noise{i}']).cpu().numpy()
        want = FO.extract(fx[f'audio{i}'], fx[f'noise{i}'], conf, n_mels=40)
        assert got.shape == want.shape and np.abs(got - want).max() < TOL, np.abs(got - want).max()
