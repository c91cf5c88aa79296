"""Recorded launch lists (wav2letter_pytorch_amd/replay.py) against the eager step engine on a real MI355X: a training step
that is recorded once and then replayed through one w2l_replay call per phase must do EXACTLY what the eager step does -- same
losses, same parameters, bit for bit when the step's bit-reproducible kernels are selected -- across new batches (static input
buffer), ragged Jasper lengths (static length table), held-back weight gradients (phase X, two alternating record sets),
learning-rate changes (phase O re-recorded), gradient accumulation (fallback to the eager backward), dropout (fresh masks every
replay) and fp8 mode."""
import ast
import os

import numpy as np
import pytest
import torch

from gpu_helpers import build_jasper, build_w2l, scale_err

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def _bit_reproducible(monkeypatch):
    from wav2letter_pytorch_amd import engine as E
    monkeypatch.setattr(E, 'FOLD_BN_FWD', '0')
    monkeypatch.setattr(E, 'FAST_BN_BWD', False)
    monkeypatch.setattr(E, 'DETERMINISTIC_WGRAD', True)


def _run(make, batches, steps, replay_on, defer=0, lr_change_at=None, overlap=True, accumulate_at=None, lr=0.02):
    """``steps`` training steps over ``batches`` in rotation; returns (losses, final parameters, replay statistics)"""
    from wav2letter_pytorch_amd import replay
    from wav2letter_pytorch_amd.optim import FusedSGD
    replay.ENABLED = replay_on
    for k in ('recorded', 'replayed_F', 'replayed_B', 'replayed_O', 'replayed_X'):
        replay.STATS[k] = 0
    replay.STATS['poisoned'] = []
    try:
        torch.manual_seed(11)
        model = make().cuda().train()
        model.check_nan = False
        opt = FusedSGD.from_sgd(torch.optim.SGD(model.parameters(), lr=lr, momentum=0.9, nesterov=True, weight_decay=1e-4))
        opt.overlap = overlap
        if defer:
            opt.defer_wgrad(model, defer)
        losses = []
        for i in range(steps):
            x, il, tg, tl = batches[i % len(batches)]
            if lr_change_at is not None and i == lr_change_at:
                for g in opt.param_groups:
                    g['lr'] *= 0.5
            opt.zero_grad(set_to_none=True)
            out, ol = model(x, il)
            loss = model.criterion(out.transpose(0, 1), tg, ol, tl)
            loss.backward()
            if accumulate_at is not None and i == accumulate_at:          # a second backward pass before the step: gradients ADD
                out, ol = model(x, il)
                model.criterion(out.transpose(0, 1), tg, ol, tl).backward()
            opt.step()
            losses.append(float(loss))
        opt.join()
        torch.cuda.synchronize()
        params = {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}
        stats = dict(replay.STATS)
        rep = replay.report(model.engine())
        return losses, params, stats, rep
    finally:
        replay.ENABLED = True


def _w2l_case(dropout=False, layers=None):
    from oracle import w2l_oracle as O
    layers = layers or [(128, 11, 2, 1, 0.3), (192, 13, 1, 1, 0.3), (128, 29, 1, 2, 0.3)]
    sd = O.init_wav2letter_state(layers, seed=41)
    batches = []
    for b in range(3):
        x, il, tg, tl = O.synthetic_batch(4, 300, seed=50 + b, s_lo=8, s_hi=30)
        batches.append((x.cuda(), il, tg.cuda(), tl.cuda()))
    return (lambda: build_w2l(layers, sd, 'bf16', dropout=dropout)), batches


@pytest.mark.parametrize('defer', [0, 1, 2])
def test_replayed_steps_equal_eager_steps_w2l(defer, monkeypatch):
    """ten steps over three different batches: losses and final parameters of the replayed run equal the eager run's bit for
    bit (dropout off, bit-reproducible kernels), and the run really was replayed: two sets recorded, forward / backward /
    optimizer (and, with held-back weight gradients, phase X) replayed, every phase one C segment (no Python items)"""
    _bit_reproducible(monkeypatch)
    make, batches = _w2l_case()
    le, pe, _, _ = _run(make, batches, 10, False, defer=defer)
    lr_, pr, st, rep = _run(make, batches, 10, True, defer=defer)
    assert st['poisoned'] == [], st
    # (how many of the ten steps were replayed depends on how warm the process-wide kernel-plan tables are: at least three)
    assert st['recorded'] >= 2 and st['replayed_F'] >= 3 and st['replayed_B'] >= 3 and st['replayed_O'] >= 3, st
    if defer:
        assert st['replayed_X'] >= 2, st
    sets = max(rep['shapes'], key=lambda g: g['seen'])['sets']      # (the first step ran under another key: no weight events yet)
    assert all(s['F'] and s['B'] and s['O'] and s['python_items'] <= 1 for s in sets), rep
    assert le == lr_, (le, lr_)
    for k in pe:
        assert np.array_equal(pe[k], pr[k]), k


def test_replay_several_batch_shapes_interleaved(monkeypatch):
    """variable-length training: three batch shapes in rotation (plus one that shows up only twice).  Every recurring shape gets
    its own pair of record sets once it has come back, a shape met for the first time -- or still measuring its kernel plans --
    never disturbs the shapes already recorded, the one-off shape stays eager; same trajectory as the eager run, bit for bit."""
    from oracle import w2l_oracle as O
    _bit_reproducible(monkeypatch)
    layers = [(128, 11, 2, 1, 0.0), (192, 13, 1, 1, 0.0), (128, 29, 1, 2, 0.0)]
    sd = O.init_wav2letter_state(layers, seed=47)
    shapes = [(4, 300), (3, 260), (2, 340)]
    batches = []
    for i in range(21):
        n, t = shapes[i % 3] if i not in (7, 16) else (5, 220)
        x, il, tg, tl = O.synthetic_batch(n, t, seed=300 + i, s_lo=8, s_hi=25)
        batches.append((x.cuda(), il, tg.cuda(), tl.cuda()))
    make = lambda: build_w2l(layers, sd, 'bf16')          # noqa: E731
    le, pe, _, _ = _run(make, batches, 21, False, defer=1)
    lr_, pr, st, rep = _run(make, batches, 21, True, defer=1)
    recorded = [g for g in rep['shapes'] if all(s_['F'] and s_['B'] and s_['O'] for s_ in g['sets'])]
    assert len(recorded) == 3 and sorted(tuple(g['input'][::2]) for g in recorded) == sorted(shapes), rep
    assert not any(g['input'][0] == 5 for g in rep['shapes']), rep            # the one-off shape never got buffers
    assert st['replayed_F'] >= 3 and st['poisoned'] == [], st
    assert le == lr_, (le, lr_)
    for k in pe:
        assert np.array_equal(pe[k], pr[k]), k


def test_replay_learning_rate_change_and_accumulation(monkeypatch):
    """the learning rate is a by-value argument of the recorded optimizer phase: a scheduler step must re-record it (same
    trajectory as eager); a backward pass that finds p.grad set (gradient accumulation) takes the eager backward on the record's
    context and the eager optimizer step"""
    _bit_reproducible(monkeypatch)
    make, batches = _w2l_case()
    le, pe, _, _ = _run(make, batches, 12, False, defer=1, lr_change_at=8, accumulate_at=10)
    lr_, pr, st, _ = _run(make, batches, 12, True, defer=1, lr_change_at=8, accumulate_at=10)
    assert st['replayed_O'] >= 2 and st['replayed_F'] >= 3, st
    assert le == lr_, (le, lr_)
    for k in pe:
        assert np.array_equal(pe[k], pr[k]), k


def test_replay_jasper_ragged_lengths(monkeypatch):
    """Jasper blocks (masked convolutions, residual branches, two gradient sources per block input) with DIFFERENT ragged
    lengths in every batch: the length chain is re-evaluated on the host and uploaded into the record set's static table"""
    from oracle import w2l_oracle as O
    _bit_reproducible(monkeypatch)
    # (channel counts that are multiples of 64, like every shipped configuration: a padded channel count keeps torch ops in
    # the step -- the running statistics are copied back -- and such a step stays eager, see the last assertion)
    blocks = [dict(layer_size=64, kernel_size=11, stride=2, residual=False, separable=False),
              dict(layer_size=128, kernel_size=13, stride=1, residual=True, separable=False, repeat=2),
              dict(layer_size=64, kernel_size=29, stride=1, dilation=2, residual=True, separable=True, repeat=2)]
    from wav2letter_pytorch_amd import Jasper
    from wav2letter_pytorch_amd.config import to_cfg
    torch.manual_seed(9)
    sdj = {k: v.detach().clone() for k, v in Jasper(to_cfg(dict(
        name='jasper', mid_layers=3, jasper_blocks=blocks, input_size=64, labels=O.ENGLISH_LOWERCASE, precision='bf16',
        audio_conf=dict(window='hamming', window_stride=0.01, window_size=0.02, sample_rate=16000),
        decoder=dict(_target_='decoder.GreedyDecoder', labels=O.ENGLISH_LOWERCASE)))).state_dict().items()}
    meta = {'blocks': blocks}
    batches = []
    g = torch.Generator().manual_seed(3)
    for b in range(3):
        x, il, tg, tl = O.synthetic_batch(4, 240, seed=60 + b, s_lo=5, s_hi=15)
        il = torch.randint(120, 241, (4,), generator=g, dtype=torch.int32)
        il[b] = 240
        for n in range(4):
            x[n, :, int(il[n]):] = 0
        tl = torch.minimum(tl, (il // 8).to(torch.int32)).clamp(min=1)
        batches.append((x.cuda(), il, tg.cuda(), tl.cuda()))
    make = lambda: build_jasper(meta['blocks'], sdj, 'bf16')          # noqa: E731
    le, pe, _, _ = _run(make, batches, 9, False, defer=1)
    lr_, pr, st, rep = _run(make, batches, 9, True, defer=1)
    assert st['poisoned'] == [] and st['recorded'] >= 2 and st['replayed_F'] >= 3 and st['replayed_O'] >= 3, (st, rep)
    assert le == lr_, (le, lr_)
    for k in pe:
        assert np.array_equal(pe[k], pr[k]), k
    # the reference-generated fixture has 48-channel blocks (padded to 64): its step keeps torch ops between launches, the
    # recording is dropped, the shape is given up after three attempts and the run is the eager one -- same results
    z = np.load(os.path.join(GOLD, 'jasper_dense.npz'), allow_pickle=True)
    meta48 = ast.literal_eval(str(z['meta']))
    sd48 = {k[3:]: torch.from_numpy(z[k].copy()) for k in z.files if k.startswith('p0/')}
    make48 = lambda: build_jasper(meta48['blocks'], sd48, 'bf16')          # noqa: E731
    le, pe, _, _ = _run(make48, batches, 8, False)
    lr_, pr, st, rep = _run(make48, batches, 8, True)
    gave_up = [g['disabled'] for g in rep['shapes'] if g['disabled']]
    assert st['replayed_F'] == 0 and gave_up and 'padded' in gave_up[0], (st, rep)
    assert le == lr_
    for k in pe:
        assert np.array_equal(pe[k], pr[k]), k


def test_replay_draws_fresh_dropout_masks():
    """dropout on (p = 0.3): the recorded step's Philox offsets come from a device counter that a recorded w2l_counter_add
    moves, so two replays of the same set on the same batch draw different masks (different losses), as eager steps do"""
    from wav2letter_pytorch_amd import replay
    from wav2letter_pytorch_amd.optim import FusedSGD
    make, batches = _w2l_case(dropout=True)
    torch.manual_seed(5)
    model = make().cuda().train()
    opt = FusedSGD.from_sgd(torch.optim.SGD(model.parameters(), lr=0.0, momentum=0.9, nesterov=True))      # lr 0: weights stand still
    opt.overlap = True
    x, il, tg, tl = batches[0]
    before = replay.STATS['replayed_F']
    losses, masks = [], []
    for i in range(10):
        opt.zero_grad(set_to_none=True)
        out, ol = model(x, il)
        loss = model.criterion(out.transpose(0, 1), tg, ol, tl)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    opt.join()
    assert replay.STATS['replayed_F'] - before >= 4
    assert all(np.isfinite(losses))
    # BatchNorm running statistics move, the weights do not: equal masks would give (nearly) equal training-mode losses; the
    # spread of the replayed steps' losses must be that of the eager ones (steps 0-1), i.e. clearly non-zero
    tail = losses[4:]
    assert len(set(round(v, 6) for v in tail)) == len(tail), losses
    assert max(tail) - min(tail) > 1e-4, losses


def test_replay_fp8_eager_stretches_for_scale_upkeep(monkeypatch):
    """fp8 mode keeps its e4m3 weight scales current in EAGER steps (amax request, adoption eight weight versions later): a replayed
    run therefore drops to the eager step for a stretch every FP8_EAGER_EVERY replays, with the skipped versions added to every
    weight's age.  With the period shrunk to 3 replays and the rescale interval to 8 versions the cycle runs several times in 30
    steps: replays and eager stretches alternate, the loss keeps falling, requests are made and adopted."""
    from wav2letter_pytorch_amd import engine as E, replay
    from wav2letter_pytorch_amd.optim import FusedSGD
    from oracle import w2l_oracle as O
    monkeypatch.setattr(replay, 'FP8_EAGER_EVERY', 3)
    monkeypatch.setattr(replay, 'FP8_EAGER_STEPS', 4)
    monkeypatch.setattr(E, 'FP8_WEIGHT_RESCALE', 8)
    monkeypatch.setattr(E, 'FP8_RESCALE_LAG', 2)
    layers = [(128, 11, 2, 1, 0.0), (256, 13, 1, 1, 0.0), (128, 29, 1, 2, 0.0)]
    sd = O.init_wav2letter_state(layers, seed=45)
    x, il, tg, tl = O.synthetic_batch(8, 600, seed=46, s_lo=10, s_hi=40)
    x, tg, tl = x.cuda(), tg.cuda(), tl.cuda()
    model = build_w2l(layers, sd, 'fp8').cuda().train()
    opt = FusedSGD.from_sgd(torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9, nesterov=True))
    opt.overlap = True
    before = replay.STATS['replayed_F']
    losses, replayed = [], []
    for i in range(30):
        opt.zero_grad(set_to_none=True)
        out, ol = model(x, il)
        loss = model.criterion(out.transpose(0, 1), tg, ol, tl)
        loss.backward()
        opt.step()
        losses.append(float(loss))
        replayed.append(replay.STATS['replayed_F'] - before)
    opt.join()
    steps_replayed = [b - a for a, b in zip([0] + replayed[:-1], replayed)]
    assert sum(steps_replayed) >= 6 and steps_replayed.count(0) >= 10, steps_replayed      # both kinds of step, repeatedly
    runs = ''.join(str(v) for v in steps_replayed)
    assert '1110' in runs and '01' in runs[8:], runs                                        # ... alternating
    assert all(np.isfinite(losses)) and losses[-1] < 0.7 * losses[0], losses
    w = list(model.conv1ds.children())[1].conv1.weight
    assert w.__dict__['_w2l_fp8']['scale'] > 0


def test_replay_fp8_and_eval_in_between():
    """fp8 mode replays too (e4m3 operands, device-side dy scales), and an evaluation-mode forward between training steps
    (validation) runs eagerly without disturbing the records"""
    from wav2letter_pytorch_amd import replay
    from wav2letter_pytorch_amd.optim import FusedSGD
    from oracle import w2l_oracle as O
    layers = [(128, 11, 2, 1, 0.0), (256, 13, 1, 1, 0.0), (128, 29, 1, 2, 0.0)]
    sd = O.init_wav2letter_state(layers, seed=43)
    x, il, tg, tl = O.synthetic_batch(8, 600, seed=44, s_lo=10, s_hi=40)
    x, tg, tl = x.cuda(), tg.cuda(), tl.cuda()
    res = {}
    for on in (False, True):
        replay.ENABLED = on
        try:
            model = build_w2l(layers, sd, 'fp8').cuda().train()
            opt = FusedSGD.from_sgd(torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9, nesterov=True))
            opt.overlap = True
            opt.defer_wgrad(model, 1)
            before = replay.STATS['replayed_F']
            losses = []
            for i in range(9):
                if i == 6:
                    opt.join()
                    model.eval()
                    with torch.no_grad():
                        ev, _ = model(x, il)
                    assert torch.isfinite(ev).all()
                    model.train()
                opt.zero_grad(set_to_none=True)
                out, ol = model(x, il)
                loss = model.criterion(out.transpose(0, 1), tg, ol, tl)
                loss.backward()
                opt.step()
                losses.append(float(loss))
            opt.join()
            torch.cuda.synchronize()
            res[on] = (losses, replay.STATS['replayed_F'] - before)
        finally:
            replay.ENABLED = True
    assert res[True][1] >= 3 and res[False][1] == 0
    assert all(np.isfinite(res[True][0])) and res[True][0][-1] < res[True][0][0]
    # (fp32 atomics in the default mode: the two trajectories agree to rounding noise amplified by 9 low-precision steps)
    assert abs(res[True][0][-1] - res[False][0][-1]) < 0.05 * abs(res[False][0][-1]), res
