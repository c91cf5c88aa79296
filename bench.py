#!/usr/bin/env python3
"""Headline benchmark: audio-frames/sec (fwd + CTC + bwd [+ SGD step]) of Wav2Letter, full
21-layer table of configuration/model/wav2letter.yaml (mid_layers=20, 153 M params), bf16
operands / fp32 accumulate, synthetic 64-mel x 1000-frame spectrograms, batch 32 per GPU.

  python bench.py [--gpus N --steps K --warmup W]

N > 1: under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` each process is one rank; a plain
`python bench.py --gpus N` (no rendezvous variables in the environment) starts the N ranks itself
(wav2letter_pytorch_amd/launch.py: N subprocesses of this command line, before this process touches the GPU) and
forwards rank 0's line.  One rank per GPU, gradients averaged with RCCL (torch.distributed backend "nccl").

Prints ONE JSON line on rank 0 (see the contract in the task statement).  `roofline` is the
dominant kernel (conv_igemm_kernel: forward + data-gradient convolutions) timed with HIP events
on its launch stream in a separate instrumented pass of the same step; `cpu_baseline` is the
CPU oracle (torch-CPU restatement of the reference path) timed on this host's cores."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BF16_DENSE_PEAK_TFLOPS = 2500.0      # MI355X_MICROARCH.md: ~2.5 PFLOP/s dense bf16
FP8_DENSE_PEAK_TFLOPS = 5000.0       # ~5 PFLOP/s dense fp8 (block-scaled MFMA, K = 128)


def w2l_cfg(mid_layers, dropout=True, precision='bf16'):
    from wav2letter_pytorch_amd.defaults import wav2letter_model
    return wav2letter_model(mid_layers, dropout=dropout, precision=precision)


def jasper10x5_cfg(precision='bf16'):
    from wav2letter_pytorch_amd.defaults import jasper10x5_model
    return jasper10x5_model(precision=precision)


def cpu_baseline(budget_s=20.0, mid_layers=20, N=2, T=1000):
    """the oracle's training step (fp32, torch CPU ops = what the reference executes) on a bounded
    sample of the same workload: W2L mid_layers=20, N=2, T=1000; the shipped default
    ``mid_layers: 1`` at N=32 when the run is of that configuration."""
    from oracle import w2l_oracle as O
    cores = min(os.cpu_count() or 1, 32)       # torch CPU convolutions stop scaling (and regress) well before 256 threads
    torch.set_num_threads(cores)
    layers = [l[:4] + (0.0,) for l in O.W2L_LAYERS][:mid_layers]
    sd = O.init_wav2letter_state(layers, seed=0)
    x, il, tg, tl = O.synthetic_batch(N, T, seed=1234)
    t0 = time.perf_counter()
    O.wav2letter_step(x, il, tg, tl, sd, layers)            # warm-up, also sizes the sample
    warm = time.perf_counter() - t0
    nsteps = max(1, min(5, int(budget_s / max(warm, 1e-3))))
    times = []
    for _ in range(nsteps):
        t0 = time.perf_counter()
        O.wav2letter_step(x, il, tg, tl, sd, layers)
        times.append(time.perf_counter() - t0)
    best, mean = min(times), sum(times) / len(times)
    return {'value': round(N * T / best, 1), 'unit': 'frames/s', 'cores': cores, 'kind': 'port',
            'mean_value': round(N * T / mean, 1),
            'sample': f'W2L mid_layers={mid_layers} fp32 N={N} T={T}, dropout off (p = 0 in every layer; the GPU leg has the yaml dropout on): '
                      f'{nsteps} timed step(s) after 1 warm-up, best step {best:.3f}s (value), mean step {mean:.3f}s (mean_value), '
                      f'torch CPU ops on {cores} threads of {os.cpu_count()} host cores'}


def cpu_baseline_jasper(budget_s=25.0):
    """the same for the secondary workload (N=2 for Jasper): the oracle's Jasper 10x5 step, fp32, torch CPU ops"""
    from oracle import w2l_oracle as O
    from wav2letter_pytorch_amd import Jasper
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    cfg = jasper10x5_cfg()
    blocks = [dict(b) for b in cfg.jasper_blocks]
    torch.manual_seed(0)
    sd = {k: v.detach().clone() for k, v in Jasper(cfg).state_dict().items()}       # the module only initialises the weights
    N, T = 2, 1000
    x, il, tg, tl = O.synthetic_batch(N, T, seed=1234)
    t0 = time.perf_counter()
    O.jasper_step(x, il, tg, tl, sd, blocks)
    warm = time.perf_counter() - t0
    nsteps = max(1, min(5, int(budget_s / max(warm, 1e-3))))
    times = []
    for _ in range(nsteps):
        t0 = time.perf_counter()
        O.jasper_step(x, il, tg, tl, sd, blocks)
        times.append(time.perf_counter() - t0)
    best, mean = min(times), sum(times) / len(times)
    return {'value': round(N * T / best, 1), 'unit': 'frames/s', 'cores': cores, 'kind': 'port',
            'mean_value': round(N * T / mean, 1),
            'sample': f'Jasper 10x5 fp32 N={N} T={T} (no dropout in this configuration): {nsteps} timed step(s) after 1 warm-up, '
                      f'best step {best:.3f}s (value), mean step {mean:.3f}s (mean_value), torch CPU ops on {cores} threads of '
                      f'{os.cpu_count()} host cores'}


def csrc_fingerprint():
    """sha256 over the sources of the kernels whose counters profiles/*_pmc_bench.json holds -- the implicit-GEMM and the
    weight-gradient kernels (conv_igemm.hip, conv_wgrad.hip, conv_wgrad_kernel.h, conv_wgrad3_dev.hip, common.h) and the Makefile's flags: the stamp of that file, so
    that counters collected on other kernels are never quoted (tools/prof_summary.py writes the same stamp).  Entry points
    added elsewhere in the library do not change what these two kernels read and write."""
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(ROOT, 'wav2letter_pytorch_amd', 'csrc')
    for name in ('Makefile', 'common.h', 'conv_igemm.hip', 'conv_wgrad.hip', 'conv_wgrad_kernel.h', 'conv_wgrad3_dev.hip'):
        h.update(name.encode())
        h.update(open(os.path.join(src, name), 'rb').read())
    return h.hexdigest()[:16]


def share_tune_plans(E, dist, rank, world, fwd_bwd):
    """One set of kernel plans for all ranks.  Rank 0 runs ONE forward + backward alone (no collectives, no optimizer step:
    the engine measures and picks block shapes / split-K / block orders for every layer shape), writes the choices to a
    file, the other ranks wait at a barrier and load it.  Without this each of the N ranks would tune for itself (N x the
    measuring launches) and could pick different plans, i.e. different ms/step for no reason the line explains.  Returns
    (path, entries loaded or None, sha16 of this rank's plan table)."""
    import hashlib
    import tempfile
    path = os.environ.get('W2L_TUNE_CACHE') or os.path.join(
        tempfile.gettempdir(), 'w2l_tune_%s_%d.txt' % (os.environ.get('MASTER_PORT', '0'), os.getuid()))
    loaded = None
    if world > 1:
        dist.barrier()            # the process group's communicator (and its streams) exists on EVERY rank before any rank's
                                  # engine probes its side streams: all ranks create their streams in the same order
    if rank == 0:
        fwd_bwd()
        torch.cuda.synchronize()
        E.save_tune_cache(path)
    if world > 1:
        dist.barrier()
        if rank != 0:
            loaded = E.load_tune_cache(path)
    own = '%s.rank%d' % (path, rank)
    E.save_tune_cache(own)
    sha = hashlib.sha256(open(own, 'rb').read()).hexdigest()[:16]
    os.remove(own)
    return path, loaded, sha


def gather_objects(dist, world, obj):
    if world <= 1:
        return [obj]
    out = [None] * world
    dist.all_gather_object(out, obj)
    return out


def live_traffic(E, kernel, timeout_s=170):
    """HBM-side bytes per launch of ``kernel``, measured NOW: two child runs of this script (2 steps each) under
    `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes: the two counters do not fit one),
    started as ordinary subprocesses while this process idles, with this run's kernel plans handed over through a tune-cache
    file.  traffic = 2 * FETCH_SIZE KiB + WRITE_SIZE KiB (gfx950: FETCH_SIZE tallies wide streaming reads at half their
    bytes, MI355X_MICROARCH.md 'HBM').  Returns (bytes, how) or (None, why-not); never raises."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(exe):
        return None, 'rocprofv3 not found'
    tmp = tempfile.mkdtemp(prefix='w2l_pmc_', dir='/tmp')
    try:
        cache = os.path.join(tmp, 'tune.txt')
        E.save_tune_cache(cache)
        # (the counted launches are the same kernels with the same plans whichever way the step is driven: the child passes run
        # the eager step, without the deferral A/B and the trainer legs, to stay short under the profiler)
        env = dict(os.environ, W2L_TUNE_CACHE=cache, TMPDIR='/tmp', W2L_REPLAY='0')
        for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
            env.pop(k, None)
        vals = {}
        for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
            out = os.path.join(tmp, counter)
            cmd = [exe, '--kernel-trace', '--pmc', counter, '--output-format', 'csv', '-d', out, '--', sys.executable,
                   os.path.abspath(__file__), '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--no-live-traffic',
                   '--no-trainer-leg', '--defer-wgrad', '4']
            r = subprocess.run(cmd, cwd='/tmp', env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s)
            if r.returncode != 0:
                return None, f'rocprofv3 --pmc {counter} exited with {r.returncode}'
            v = [float(row['Counter_Value']) for f in glob.glob(out + '/**/*counter_collection.csv', recursive=True)
                 for row in csv.DictReader(open(f)) if row['Counter_Name'] == counter and kernel in row['Kernel_Name']]
            if not v:
                return None, f'no {counter} rows for {kernel}'
            vals[counter] = sum(v) / len(v)
        return 2 * vals['FETCH_SIZE'] * 1024 + vals['WRITE_SIZE'] * 1024, (
            'measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes of this command (2 steps each), mean over '
            'the kernel\'s launches, traffic = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 correction)')
    except Exception as e:            # noqa: BLE001 -- the bench line must come out whatever the profiler does
        return None, 'live PMC pass failed: ' + repr(e)[:160]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _launcher():
    """wav2letter_pytorch_amd/launch.py loaded by path: importing the package would load libw2l_hip.so (and the HIP
    runtime) into the parent, which only spawns the ranks"""
    import importlib.util
    spec = importlib.util.spec_from_file_location('w2l_launch', os.path.join(ROOT, 'wav2letter_pytorch_amd', 'launch.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def grad_byte_split(model, defer_k):
    """fp32 gradient bytes of the conv weights whose gradient is held back for the next forward pass (optim.FusedSGD.defer_wgrad)
    vs everything reduced inside the backward pass"""
    try:
        units = model.engine().units
        spec = [int(v) for v in str(defer_k).split(',')] if ',' in str(defer_k) else int(defer_k)
        n = len(units)
        held = set(range(n - spec, n)) if isinstance(spec, int) else {(u if u >= 0 else n + u) for u in spec}
        fwd = sum(c.weight.numel() * 4 for i, u in enumerate(units) if i in held for c in (u.main, u.res) if c is not None)
        total = sum(p.numel() * 4 for p in model.parameters())
        return {'total': total, 'reduced_beside_next_forward': fwd, 'reduced_in_backward': total - fwd,
                'fraction_beside_next_forward': round(fwd / total, 4)}
    except Exception as e:        # noqa: BLE001 -- a report field, never a reason to lose the record
        return {'error': repr(e)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--frames', type=int, default=1000)
    ap.add_argument('--mid-layers', type=int, default=20)
    ap.add_argument('--model', default='wav2letter', choices=['wav2letter', 'jasper10x5'],
                    help='wav2letter = the headline workload; jasper10x5 = BASELINE config 4 (secondary)')
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp8'],
                    help='fp8 = forward, data-gradient and weight-gradient convolutions on e4m3 operands (BASELINE config 5)')
    ap.add_argument('--ragged', action='store_true', help='input lengths U{T/2..T} instead of T for every utterance (SURVEY 8d: the second run)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-live-traffic', action='store_true',
                    help='do not start the two rocprofv3 --pmc child passes that measure roofline.traffic (the committed '
                         'profiles/r*_pmc_bench.json is quoted instead when it was measured on this build)')
    ap.add_argument('--defer-candidates', default=None, help="with --defer-wgrad auto: ';'-separated specs to measure instead of the built-in list")
    ap.add_argument('--no-optimizer', action='store_true')
    ap.add_argument('--graph', action='store_true', help='replay the step as a captured hipGraph (graph.GraphedTrainStep; Wav2Letter)')
    ap.add_argument('--no-sgd-overlap', action='store_true', help='keep the fused SGD updates on the main stream')
    ap.add_argument('--defer-wgrad', default=os.environ.get('W2L_DEFER_WGRAD', 'auto'),
                    help='weight gradients (+ fused update, + all-reduce) of the top K conv units run beside the NEXT forward pass '
                         '(optim.FusedSGD.defer_wgrad); an integer, or "auto": measured in this run after the warm-up (0 vs '
                         'the candidates, the fastest serves the timed region)')
    ap.add_argument('--force-dp', action='store_true', help='run the RCCL gradient path even with one rank (plumbing check)')
    ap.add_argument('--early-collective', action='store_true',
                    help='with --force-dp: run one collective before the first step, as the parameter broadcast of a multi-rank run '
                         'does (the communicator and its streams then exist before the engine creates its side streams)')
    ap.add_argument('--collective-ab', action='store_true',
                    help='data-parallel runs: after the complete record, time the same region once more with the gradient '
                         'collectives through the C ABI\'s RCCL helpers and keep the faster path.  OPT-IN: that path has never run '
                         'more than one rank, so the leg runs under a watchdog that prints the record in hand and ends the '
                         'process with exit code 3 ("record valid, native leg abandoned") should a collective never return')
    ap.add_argument('--no-collective-ab', action='store_true', help='(the default since round 5; kept for old command lines)')
    ap.add_argument('--serial-wgrad', action='store_true', help='keep weight gradients on the main stream (clean per-kernel durations for profiling)')
    ap.add_argument('--trace-steps', action='store_true', help='per-step host-enqueue vs GPU time (stderr), then exit')
    ap.add_argument('--event-trace', default=None, metavar='CSV',
                    help='HIP-event timeline of --steps steps without a profiler attached (_lib.trace_launches), written in the '
                         'column layout of a rocprofv3 kernel trace for tools/timeline.py; then exit')
    ap.add_argument('--lead-trace', action='store_true',
                    help='how far the host runs ahead of the GPU at four points of every step (stderr), then exit')
    ap.add_argument('--host-profile', action='store_true', help='cProfile of the host side of the step (stderr), then exit')
    ap.add_argument('--breakdown', action='store_true', help='print the per-kernel event timing table to stderr')
    ap.add_argument('--through-trainer', action='store_true',
                    help='the timed region (value, ms_per_step) runs the reference\'s loop body instead of the hand-rolled step: '
                         'training_step on _collator\'s 6-tuple (host spectrograms and integer tensors, transcript strings), i.e. '
                         'forward + CTC + per-step greedy decode + CER/WER + log_dict, then backward, optimizer step and the '
                         'on_train_batch_end hook (base_asr_models.py:78-85 driven as train.py:34-37 drives it)')
    ap.add_argument('--no-trainer-leg', action='store_true',
                    help='skip the extra timed regions that put trainer_ms_per_step (and its synchronous-metrics "before" figure) '
                         'into the record')
    args = ap.parse_args()

    L = _launcher()
    if args.gpus > 1 and not L.under_launcher():
        # parent of a self-launched run: no HIP / HSA call may happen here (the children are started from this process), so
        # the devices are counted from sysfs / the *_VISIBLE_DEVICES variables, never through torch.cuda
        seen = L.visible_gpu_count()
        if os.environ.get('W2L_DIST_BACKEND') != 'gloo' and seen is not None and seen < args.gpus:
            raise SystemExit(f'bench.py --gpus {args.gpus}: only {seen} GPU(s) visible')
        sys.exit(L.spawn_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]))
    if L.under_launcher() and int(os.environ['WORLD_SIZE']) != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} inside a {os.environ['WORLD_SIZE']}-rank launch: the two must agree")

    # stdout carries exactly ONE line, the JSON record: libraries that print there (RCCL's version banner on the first
    # communicator, for one) are sent to stderr for the life of the run, and the record goes to the saved descriptor
    sys.stdout.flush()
    record_fd = os.dup(1)
    os.dup2(2, 1)

    from wav2letter_pytorch_amd import Jasper, Wav2Letter, engine as E
    from wav2letter_pytorch_amd.distributed import GradReducer, NativeComm, broadcast_parameters, init_process_group_from_env
    import torch.distributed as dist
    from wav2letter_pytorch_amd.defaults import synthetic_batch      # the GPU leg never touches oracle/

    rank, world = init_process_group_from_env(force=args.force_dp)
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if os.environ.get('W2L_DIST_BACKEND') == 'gloo':      # one-GPU rehearsal of the multi-rank command line: every rank on cuda:0
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)

    torch.manual_seed(0)
    if args.model == 'jasper10x5':
        model = Jasper(jasper10x5_cfg(args.dtype)).to(dev).train()
        model.check_nan = False              # the reference's per-step NaN assert is a host sync
    else:
        model = Wav2Letter(w2l_cfg(args.mid_layers, precision=args.dtype)).to(dev).train()
    N, T = args.batch, args.frames
    x, il, tg, tl = synthetic_batch(N, T, seed=1234 + rank, ragged=args.ragged)
    x = x.to(dev)
    tg_d, tl_d = tg.to(dev), tl.to(dev)
    ol = model.compute_output_lengths(il).to(dev)
    lens_arg = il if args.model == 'jasper10x5' else None        # host lengths, as _collator hands them over

    def fwd_bwd():
        model.zero_grad(set_to_none=True)
        out, _ = model(x, lens_arg)
        loss = model.criterion(out.transpose(0, 1), tg_d, ol, tl_d)
        loss.backward()
        return loss

    dp = world > 1 or args.force_dp
    tune_path = tune_loaded = tune_sha = None
    if dp and E.AUTOTUNE:
        tune_path, tune_loaded, tune_sha = share_tune_plans(E, dist, rank, world, fwd_bwd)
    broadcast_parameters(model)          # (also undoes what rank 0's tuning pass did to its BatchNorm running statistics)
    if args.early_collective and dist.is_initialized():
        dist.all_reduce(torch.zeros(1, device=dev))
        torch.cuda.synchronize()
    if dp:
        model.grad_reducer = GradReducer(force=args.force_dp)
    if args.serial_wgrad:
        model._overlap_wgrad = False
    opt, _ = model.configure_optimizers()
    opt = opt[0]
    if hasattr(opt, 'overlap') and not args.no_sgd_overlap:
        opt.overlap = True            # conv-weight updates run on a side stream under the next step's forward (optim.FusedSGD)

    can_defer = hasattr(opt, 'defer_wgrad') and getattr(opt, 'overlap', False) and not args.graph and not args.no_optimizer
    defer_k = '0'            # 'k' = the top k units, 'a,b,c' = exactly those units (negative: counted from the top)

    def defer_spec(text):
        """'4' = the top 4 units; '-1,-3,-5' = exactly those units (counted from the top)"""
        return [int(v) for v in text.split(',')] if ',' in text else int(text)

    if can_defer and args.defer_wgrad != 'auto':
        defer_k = args.defer_wgrad
        opt.defer_wgrad(model, defer_spec(defer_k))

    def raw_step():
        opt.zero_grad(set_to_none=True)
        out, _ = model(x, lens_arg)
        loss = model.criterion(out.transpose(0, 1), tg_d, ol, tl_d)
        loss.backward()
        if not args.no_optimizer:
            opt.step()
        return loss

    # ---- the reference's loop body (base_asr_models.py:78-85 as pytorch_lightning's fit loop drives it, train.py:34-37):
    # _collator's 6-tuple -- spectrograms and integer tensors ON THE HOST (page-locked, as a DataLoader with pin_memory hands
    # them over), transcripts as strings (the synthetic targets spelled with the model's labels) -- through training_step
    # (forward, CTC, greedy decode + CER / WER of every batch, log_dict), backward, optimizer step, on_train_batch_end
    labels = list(model.labels)
    texts = tuple(''.join(labels[int(c)] for c in tg[n, : int(tl[n])]) for n in range(N))
    x_host = x.detach().cpu().pin_memory()
    batch6 = (x_host, il.to(torch.int32), tg.to(torch.int32), tl.to(torch.int32), tuple(f'synthetic_{n}.wav' for n in range(N)), texts)
    model._optimizers = opt                       # what Trainer.fit sets: training_step reads the learning rate from it
    trainer_step_no = [0]

    def trainer_step():
        opt.zero_grad(set_to_none=True)
        loss = model.training_step(batch6, trainer_step_no[0])
        loss.backward()
        if not args.no_optimizer:
            opt.step()
        model.on_train_batch_end(loss, batch6, trainer_step_no[0])
        trainer_step_no[0] += 1
        return loss

    step = trainer_step if args.through_trainer else raw_step

    if args.graph:
        from wav2letter_pytorch_amd.graph import GraphedTrainStep
        gstep = GraphedTrainStep(model, opt, x, il, tg_d, tl_d, warmup=max(args.warmup, 2))

        eager_step = step

        def step():                                   # noqa: F811
            return gstep()
    else:
        eager_step = step

    def fence():
        if hasattr(model, 'resolve_metrics'):
            model.resolve_metrics(wait_all=True)      # string metrics still to be scored belong to the steps just run
        if hasattr(opt, 'join'):
            opt.join()                    # updates still streaming on the optimizer's side stream belong to the step just run
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_ms(k):
        """k steps between two fences, ms per step, the slowest rank's"""
        fence()
        t0 = time.perf_counter()
        for _ in range(k):
            step()
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t)
        return dt / k * 1e3

    # steps a configuration needs before it is in its steady state: the eager warm steps of a step shape + the two steps that
    # record its two launch lists (wav2letter_pytorch_amd/replay.py) + one replayed step of each
    from wav2letter_pytorch_amd import replay as _replay
    settle = (_replay.WARM_STEPS + 4) if _replay.ENABLED else 2
    for _ in range(max(args.warmup, settle)):
        step()
    fence()

    # ---- how many of the top units' weight gradients to hold back for the next forward pass?  Measured here: 0 against the
    # candidates, a few steps each between fences, the slowest rank's time; all ranks take the same decision.
    defer_ab = None
    if can_defer and args.defer_wgrad == 'auto':
        n_units = len(model.engine().units)
        # (every-other-unit patterns, e.g. '-1,-3,-5,-7', were measured too: never ahead of the plain top-k sets)
        cands = ['0'] + [str(k) for k in (4, 6) if k <= n_units]
        if args.defer_candidates:
            cands = ['0'] + args.defer_candidates.split(';')
        defer_ab = {}
        for k in cands:
            opt.defer_wgrad(model, defer_spec(k))
            for _ in range(settle):          # a new configuration: warm steps, then the two steps that record it (replay.py)
                step()
            defer_ab[k] = round(timed_ms(5), 3)
        defer_k = min(defer_ab, key=lambda k: defer_ab[k])
        opt.defer_wgrad(model, defer_spec(defer_k))
        for _ in range(settle):
            step()
        fence()

    collective_paths = None
    if args.event_trace:
        from wav2letter_pytorch_amd import _lib
        fence()
        _lib.trace_launches(True)
        for _ in range(args.steps):
            step()
        fence()
        _lib.trace_launches(False)
        n = _lib.trace_dump(args.event_trace)
        print(f'{n} launches of {args.steps} steps -> {args.event_trace} (defer_wgrad {defer_k}: {defer_ab})', file=sys.stderr)
        return
    if args.trace_steps:
        # per-step host enqueue time vs GPU time (event to event): tells a host-bound box from a slow-clock box
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        host = []
        evs[0].record()
        for i in range(args.steps):
            h0 = time.perf_counter()
            step()
            evs[i + 1].record()
            host.append((time.perf_counter() - h0) * 1e3)
        torch.cuda.synchronize()
        gpu = [evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps)]
        print('load average: %s' % open('/proc/loadavg').read().strip(), file=sys.stderr)
        print('host enqueue ms/step: ' + ' '.join('%.1f' % v for v in host), file=sys.stderr)
        print('gpu ms/step:          ' + ' '.join('%.1f' % v for v in gpu), file=sys.stderr)
        return
    if args.lead_trace:
        # Is the GPU ever starved by the host?  At four program points of every step -- step start, forward enqueued, backward
        # enqueued, optimizer enqueued -- the host notes its clock and records an event on the main stream; afterwards
        # lead = (GPU time at which the event fired) - (host time at which it was recorded), both measured from a common
        # synchronised origin.  A lead near zero means the GPU reached that point as soon as the host had enqueued it: the
        # stream was running dry there.
        names = ('step start', 'forward enqueued', 'backward enqueued', 'optimizer enqueued')
        torch.cuda.synchronize()
        origin = torch.cuda.Event(enable_timing=True)
        origin.record()
        torch.cuda.synchronize()
        h0 = time.perf_counter()
        marks = []

        def mark():
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            marks.append((time.perf_counter() - h0, e))

        for _ in range(args.steps):
            mark()
            opt.zero_grad(set_to_none=True)
            out, _ = model(x, lens_arg)
            loss = model.criterion(out.transpose(0, 1), tg_d, ol, tl_d)
            mark()
            loss.backward()
            mark()
            opt.step()
            mark()
        torch.cuda.synchronize()
        print('lead of the host over the GPU in ms (rows = steps):', file=sys.stderr)
        print('  ' + ' | '.join('%-18s' % n for n in names) + ' | host ms in step | gpu ms in step', file=sys.stderr)
        for i in range(args.steps):
            row = marks[4 * i: 4 * i + 4]
            leads = [origin.elapsed_time(e) - h * 1e3 for h, e in row]
            nxt = marks[4 * i + 4] if i + 1 < args.steps else None
            host_ms = (nxt[0] - row[0][0]) * 1e3 if nxt else float('nan')
            gpu_ms = row[0][1].elapsed_time(nxt[1]) if nxt else float('nan')
            print('  ' + ' | '.join('%18.3f' % v for v in leads) + ' | %15.3f | %14.3f' % (host_ms, gpu_ms), file=sys.stderr)
        return
    if args.host_profile:
        import cProfile
        import pstats
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print('enqueue %.3f ms/step, with drain %.3f ms/step' % ((t1 - t0) / args.steps * 1e3, (t2 - t0) / args.steps * 1e3),
              file=sys.stderr)
        pr = cProfile.Profile()
        with torch.autograd.set_multithreading_enabled(False):    # backward on this thread so the profile sees it
            pr.enable()
            for _ in range(args.steps):
                step()
            pr.disable()
        torch.cuda.synchronize()
        pstats.Stats(pr, stream=sys.stderr).sort_stats('tottime').print_stats(50)
        return
    def timed_region(fn=None):
        """EXACTLY --steps steps between two fences (barrier + device sync on both sides); the job is as fast as its slowest rank"""
        fn = fn or step
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss_ = fn()
        fence()
        elapsed_ = time.perf_counter() - t0
        rank_ms_ = [elapsed_ / args.steps * 1e3]
        if world > 1:
            t = torch.tensor([elapsed_], device=dev, dtype=torch.float64)
            every = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(every, t)
            rank_ms_ = [float(v) / args.steps * 1e3 for v in every]
            elapsed_ = max(float(v) for v in every)
        return elapsed_ / args.steps * 1e3, rank_ms_, loss_

    if getattr(model, 'grad_reducer', None) is not None:
        model.grad_reducer.wire_bytes = 0
    ms, rank_ms, loss = timed_region()
    value = world * N * T / (ms * 1e-3)
    comm_bytes = None
    if getattr(model, 'grad_reducer', None) is not None and model.grad_reducer.active:
        # payload handed to the gradient collectives per step in the timed region (fp32 unless W2L_DP_BF16=1: then the large
        # gradients travel as bf16 copies)
        comm_bytes = {'per_step': int(model.grad_reducer.wire_bytes // args.steps),
                      'transport': 'bf16 copies of the large gradients (W2L_DP_BF16=1)' if model.grad_reducer.bf16 else 'fp32'}

    # ---- the same number of steps through the reference's loop body (see trainer_step above): what a train.py user gets.
    # Two legs: as shipped (metrics enqueued in training_step, scored in on_train_batch_end) and -- the "before" figure -- with
    # the reference's order restored (async_metrics = False: decode, CER / WER and float(loss) inside training_step, i.e.
    # before backward() is enqueued).  Host enqueue time per step of each leg rides along.
    trainer = None
    if not args.no_trainer_leg and not args.graph and not args.no_optimizer:
        trainer = {}
        legs = (('raw_loop', raw_step, True),) if args.through_trainer else ()
        legs += (('async_metrics', trainer_step, True), ('sync_metrics_before', trainer_step, False))
        for name, fn, async_on in legs:
            model.async_metrics = async_on
            for _ in range(settle):
                fn()
            fence()
            h0 = time.perf_counter()
            for _ in range(args.steps):
                fn()
            host_ms = (time.perf_counter() - h0) / args.steps * 1e3
            fence()
            t_ms, _, _ = timed_region(fn)
            trainer[name] = {'ms_per_step': round(t_ms, 3), 'host_enqueue_ms_per_step': round(host_ms, 3)}
        model.async_metrics = True
        logged = {k: round(float(v), 6) for k, v in getattr(model, '_logged', {}).items()}
        trainer['logged_last_step'] = logged
        fence()

    # ---- exposed communication: the same step with the gradient reducer detached (no collectives), same run, per rank ----
    exposed_comm_ms = exposed_by_rank = solo_ms = bf16_leg = None
    reducer = getattr(model, 'grad_reducer', None)
    if reducer is not None and not args.graph:
        model.grad_reducer = None
        _replay_was, _replay.ENABLED = _replay.ENABLED, False      # the data-parallel step runs eagerly (its collectives are not
        k2 = max(2, min(args.steps, 10))                            # entry points): so does the step it is compared with
        step()
        fence()
        t0 = time.perf_counter()
        for _ in range(k2):
            step()
        fence()
        solo = time.perf_counter() - t0
        solo_ms = solo / k2 * 1e3
        own = rank_ms[rank] - solo_ms                         # this rank's step with the collectives minus its step without
        exposed_by_rank = [round(v, 3) for v in gather_objects(dist, world, own)]
        exposed_comm_ms = max(exposed_by_rank)
        model.grad_reducer = reducer
        _replay.ENABLED = _replay_was
        broadcast_parameters(model)                          # the replicas drifted apart while stepping alone
        # ---- when more than a tenth of the step is exposed communication: the same region once more with the large gradients
        # travelling as bf16 copies (half the bytes on every xGMI link; the average is then accurate to bf16's 8 bits, which is
        # NOT what fp32 DDP computes -- reported beside the fp32 figure, never instead of it).  All ranks take the same branch
        # (the decision is the maximum over the ranks, gathered above).
        if exposed_comm_ms > 0.1 * ms and not reducer.bf16 and os.environ.get('W2L_BENCH_BF16_LEG', '1') != '0':
            reducer.bf16 = True
            step()
            step()
            fence()
            reducer.wire_bytes = 0
            ms_b, rank_ms_b, _ = timed_region()
            bf16_leg = {'ms_per_step': round(ms_b, 3), 'value': round(world * N * T / (ms_b * 1e-3), 1),
                        'comm_bytes_on_wire_per_step': int(reducer.wire_bytes // args.steps),
                        'exposed_comm_ms': round(max(gather_objects(dist, world, rank_ms_b[rank] - solo_ms)), 3),
                        'note': 'W2L_DP_BF16=1: not the headline (bf16-accurate averages); shown because the fp32 exchange '
                                'left more than 10 % of the step exposed'}
            reducer.bf16 = False
            broadcast_parameters(model)
    tune_shas = gather_objects(dist, world, tune_sha) if tune_sha is not None else None

    # ---- instrumented pass: HIP events around every conv kernel launch (same stream) ----
    roof = None
    # every rank runs these steps (a data-parallel step contains collectives: a rank stepping alone would wait forever);
    # only rank 0 records and reports the events
    if rank == 0:
        E.KERNEL_TIMER = []
    model._overlap_wgrad = False      # serialise the side stream so per-launch durations are not shared-GPU times
    for _ in range(3):
        eager_step()                  # (the captured graph carries no timing events)
    if hasattr(opt, 'join'):
        opt.join()                    # the last step's deferred weight gradients are launched (and timed) here
    torch.cuda.synchronize()
    model._overlap_wgrad = not args.serial_wgrad
    if rank == 0:
        agg = {}
        fused = [0.0, 0.0, 0]
        by_shape = {}
        for name, flops, s, e in E.KERNEL_TIMER:
            b = by_shape.setdefault((name, flops), [0.0, 0])
            b[0] += s.elapsed_time(e) * 1e-3
            b[1] += 1
            if name.endswith('/dgrad+bnreduce'):          # same kernel family; the launches whose epilogue also forms the
                name = name.split('/')[0]                 # BatchNorm-backward sums are additionally reported on their own
                fused[0] += flops
                fused[1] += s.elapsed_time(e) * 1e-3
                fused[2] += 1
            a = agg.setdefault(name, [0.0, 0.0, 0])
            a[0] += flops
            a[1] += s.elapsed_time(e) * 1e-3
            a[2] += 1
        E.KERNEL_TIMER = None
        name = 'conv_igemm_kernel'
        fl, tt, cnt = agg[name]
        ach = fl / tt / 1e12
        roof = {'kernel': name, 'bound': 'mfma', 'achieved': round(ach, 1), 'peak': BF16_DENSE_PEAK_TFLOPS,
                'unit': 'TFLOP/s', 'frac': round(ach / BF16_DENSE_PEAK_TFLOPS, 4), 'traffic': None,
                'launches_per_step': cnt // 3, 'avg_launch_ms': round(tt / cnt * 1e3, 4),
                'alg_gflop_per_launch': round(fl / cnt / 1e9, 2)}
        if fused[2]:
            roof['dgrad_with_fused_bn_reduce'] = {
                'note': 'data-gradient launches of the same kernel whose epilogue also forms the BatchNorm-backward sums '
                        '(replaces bn_act_bwd_reduce_kernel); conv FLOPs only are counted',
                'achieved': round(fused[0] / fused[1] / 1e12, 1), 'launches_per_step': fused[2] // 3,
                'avg_launch_ms': round(fused[1] / fused[2] * 1e3, 4)}
            rest = (fl - fused[0], tt - fused[1], cnt - fused[2])
            if rest[2] > 0:
                roof['other_launches'] = {'achieved': round(rest[0] / rest[1] / 1e12, 1), 'launches_per_step': rest[2] // 3,
                                          'avg_launch_ms': round(rest[1] / rest[2] * 1e3, 4)}
        headline = args.model == 'wav2letter' and args.mid_layers == 20 and args.batch == 32 and args.dtype == 'bf16'
        profiled = 'rocprof' in os.environ.get('LD_PRELOAD', '') or any(k.startswith('ROCPROF') for k in os.environ)
        if headline and world == 1 and not args.no_live_traffic and not args.graph and not profiled:
            # HBM-side bytes per launch of the dominant kernel, measured by this very run (two profiled child passes)
            roof['traffic'], roof['traffic_source'] = live_traffic(E, name)
            if roof['traffic'] is not None:
                roof['traffic'] = round(roof['traffic'])
        if headline and roof['traffic'] is None:
            # fall back to the committed PMC passes of this same command (tools/make_profiles.sh): FETCH_SIZE and WRITE_SIZE in
            # separate runs, gfx950 FETCH correction applied (tools/prof_summary.py).  Each file carries the fingerprint of the
            # kernel sources it was measured on: only a file measured on THIS build is quoted (newest round first).
            import glob
            sha = csrc_fingerprint()
            why = roof.get('traffic_source')
            roof['traffic_source'] = ('none: no profiles/r*_pmc_bench.json was measured on these kernel sources (' + sha + ')'
                                      + ('; ' + why if why else ''))
            for pmc_file in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_bench.json')), reverse=True):
                pmc = json.load(open(pmc_file))
                k = pmc.get('kernels', {}).get(name, {})
                if pmc.get('csrc_sha') == sha and 'traffic_bytes_per_launch' in k:
                    roof['traffic'] = round(k['traffic_bytes_per_launch'])
                    roof['traffic_source'] = ('profiles/' + os.path.basename(pmc_file) + ' (rocprofv3 --pmc, mean over the '
                                              'step\'s launches; kernel sources ' + sha + ')' + ('; ' + why if why else ''))
                    break
        if 'conv_igemm_fp8_kernel' in agg:
            # fp8 mode: the forward convolutions of the units ran on e4m3 operands; they are priced against the fp8 peak, the
            # bf16 launches left in `roof` (first layer, classifier, data gradients) against the bf16 peak
            fl8, tt8, cnt8 = agg['conv_igemm_fp8_kernel']
            roof['fp8_kernel'] = {'kernel': 'conv_igemm_kernel<..., F8>', 'achieved': round(fl8 / tt8 / 1e12, 1),
                                  'peak': FP8_DENSE_PEAK_TFLOPS, 'frac': round(fl8 / tt8 / 1e12 / FP8_DENSE_PEAK_TFLOPS, 4),
                                  'avg_launch_ms': round(tt8 / cnt8 * 1e3, 4), 'launches_per_step': cnt8 // 3,
                                  'alg_gflop_per_launch': round(fl8 / cnt8 / 1e9, 2)}
        if args.dtype == 'fp8' and 'fp8_kernel' in roof:
            # the dominant kernel of an fp8 run is the e4m3 instantiation: it becomes the line's `roofline`, priced against
            # the fp8 peak; the few bf16 launches of the same family (first layer, classifier) move to `bf16_launches`
            f8 = roof.pop('fp8_kernel')
            bf = {k: roof[k] for k in ('achieved', 'frac', 'launches_per_step', 'avg_launch_ms', 'alg_gflop_per_launch')}
            roof.update(kernel='conv_igemm_kernel<..., F8> (e4m3 operands, v_mfma_scale_f32_16x16x128_f8f6f4)', peak=f8['peak'],
                        achieved=f8['achieved'], frac=f8['frac'], launches_per_step=f8['launches_per_step'],
                        avg_launch_ms=f8['avg_launch_ms'], alg_gflop_per_launch=f8['alg_gflop_per_launch'])
            roof['bf16_launches'] = bf
        if 'conv_wgrad_kernel' in agg or 'conv_wgrad3_kernel' in agg:
            # bf16 weight gradients: the two kernel families side by side (conv_wgrad_kernel<..> = two taps per wave;
            # w2l_wgrad3*_kernel = the three-tap code object with its accumulators in AGPRs), and their total.  A launch may
            # cover a GROUP of layers (w2l_conv1d_wgrad_group): FLOPs of all its layers, one launch.
            fam = {}
            for key, label in (('conv_wgrad_kernel', 'two_tap_kernels'), ('conv_wgrad3_kernel', 'three_tap_agpr_kernels')):
                if key in agg:
                    f_, t_, c_ = agg[key]
                    fam[label] = {'achieved': round(f_ / t_ / 1e12, 1), 'frac': round(f_ / t_ / 1e12 / BF16_DENSE_PEAK_TFLOPS, 4),
                                  'avg_launch_ms': round(t_ / c_ * 1e3, 4), 'launches_per_step': round(c_ / 3, 1),
                                  'gflop_per_step': round(f_ / 3 / 1e9, 1)}
            fl2 = sum(agg[k][0] for k in ('conv_wgrad_kernel', 'conv_wgrad3_kernel') if k in agg)
            tt2 = sum(agg[k][1] for k in ('conv_wgrad_kernel', 'conv_wgrad3_kernel') if k in agg)
            cnt2 = sum(agg[k][2] for k in ('conv_wgrad_kernel', 'conv_wgrad3_kernel') if k in agg)
            roof['wgrad_kernel'] = {'achieved': round(fl2 / tt2 / 1e12, 1), 'frac': round(fl2 / tt2 / 1e12 / BF16_DENSE_PEAK_TFLOPS, 4),
                                    'avg_launch_ms': round(tt2 / cnt2 * 1e3, 4), 'launches_per_step': round(cnt2 / 3, 1),
                                    'by_kernel_family': fam}
            fl8, tt8, _ = agg.get('conv_igemm_fp8_kernel', (0.0, 0.0, 0))
            flw8, ttw8, cntw8 = agg.get('conv_wgrad_fp8_kernel', (0.0, 0.0, 0))
            if cntw8:
                roof['wgrad_fp8_kernel'] = {'kernel': 'conv_wgrad_fp8_kernel (e4m3 operands, ds_read_b64_tr_b8 fragments)',
                                            'achieved': round(flw8 / ttw8 / 1e12, 1), 'peak': FP8_DENSE_PEAK_TFLOPS,
                                            'frac': round(flw8 / ttw8 / 1e12 / FP8_DENSE_PEAK_TFLOPS, 4),
                                            'avg_launch_ms': round(ttw8 / cntw8 * 1e3, 4), 'launches_per_step': cntw8 // 3}
            roof['conv_ms_per_step'] = round((tt + tt2 + tt8 + ttw8) / 3 * 1e3, 3)
            # all conv launches of the step, each priced against the dense peak of ITS operand type (bf16 2.5 PF, e4m3 5 PF):
            # the time the launches would take at their peaks over the time they took -- a fraction, never above 1
            ideal_s = (fl + fl2) / (BF16_DENSE_PEAK_TFLOPS * 1e12) + (fl8 + flw8) / (FP8_DENSE_PEAK_TFLOPS * 1e12)
            roof['conv_stack_frac_of_peak'] = round(ideal_s / (tt + tt2 + tt8 + ttw8), 4)
            if fl8 + flw8 > 0:
                roof['conv_stack_bf16_equivalent_tflops'] = round((fl + fl2 + fl8 + flw8) / (tt + tt2 + tt8 + ttw8) / 1e12, 1)
        if args.breakdown:
            for k, (f, t_, c) in agg.items():
                print(f'{k}: {c // 3} launches/step, {t_ / 3 * 1e3:.3f} ms/step, {f / t_ / 1e12:.1f} TFLOP/s', file=sys.stderr)
            # per problem size (launches of one kernel family with the same algorithmic FLOPs = the same layer shape)
            for (k, f), (t_, c) in sorted(by_shape.items(), key=lambda kv: -kv[1][0]):
                print(f'  {k:34s} {f / 1e9:9.2f} GFLOP x {c // 3:3d}/step: {t_ / c * 1e6:8.1f} us each, {f * c / t_ / 1e12:7.1f} TFLOP/s, '
                      f'{t_ / 3 * 1e3:6.3f} ms/step', file=sys.stderr)

    if rank == 0:
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            if args.model != 'wav2letter':
                cpu = cpu_baseline_jasper()
            elif args.mid_layers >= 20:
                cpu = cpu_baseline(T=args.frames)
            else:          # truncated stacks (the shipped default mid_layers: 1): the run's own batch, SURVEY 8d
                cpu = cpu_baseline(mid_layers=args.mid_layers, N=args.batch, T=args.frames)
        line = {
            'metric': (f'audio-frames/sec/GPU (fwd+bwd+CTC), Wav2Letter 64-mel x 1000-frame {args.dtype}' if args.model == 'wav2letter'
                       else f'audio-frames/sec/GPU (fwd+bwd+CTC), Jasper 10x5 {args.dtype} (secondary workload)'),
            'value': round(value, 1), 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(ms, 3),
            # the reference's loop body: training_step (forward + CTC + greedy decode + CER / WER + log_dict on a host batch) ->
            # backward -> optimizer step -> on_train_batch_end, same steps, same fences (None: leg skipped)
            'trainer_ms_per_step': (round(ms, 3) if args.through_trainer else
                                    (trainer or {}).get('async_metrics', {}).get('ms_per_step')),
            'trainer_loop': trainer,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': ('Jasper 10x5 (13 dense blocks, repeat 5, 322 M params), ' if args.model == 'jasper10x5' else '')
                                   + f'Wav2Letter mid_layers={args.mid_layers} (configuration/model/wav2letter.yaml table), ' * (args.model == 'wav2letter')
                                   +
                                   f'N={N}/GPU x T={T} x 64 mel' + (' (ragged: lengths U{T/2..T}, frames counted at T)' if args.ragged else '')
                                   + ', dropout on, fwd+CTC+bwd'
                                   + ('' if args.no_optimizer else '+fused SGD(nesterov) step')
                                   + (', through training_step (per-step greedy decode + CER/WER, host batch)' if args.through_trainer else '')
                                   + (', step replayed as a hipGraph' if args.graph else ''),
                       'global_batch': world * N, 'frames': T, 'parallelism': f'dp{world}',
                       'value_is': 'whole-job frames/s (per-GPU = value / n_gpus)', 'loss': round(float(loss.detach()), 4)},
            'roofline': roof, 'cpu_baseline': cpu,
            'rccl_world': dist.get_world_size() if dist.is_initialized() else 1,
            'backend': dist.get_backend() if dist.is_initialized() else None,
            'side_streams': [{'role': r, 'candidates_tried': n, 'overlap_fraction_vs_busy_streams': f}
                             for _, r, n, f in __import__('wav2letter_pytorch_amd.streams', fromlist=['report']).report],
            'collectives_via': (None if getattr(model, 'grad_reducer', None) is None or not model.grad_reducer.active else
                                'w2l_rccl_* (C ABI)' if model.grad_reducer._comm is not None else 'torch.distributed'),
            'collective_paths_ms': collective_paths,
            'defer_wgrad': {'units': defer_k, 'measured_ms_per_step': defer_ab,
                            'note': 'weight gradients + fused updates of the top units run beside the next forward pass; every '
                                    'deferred launch of a timed step is inside the timed region (the fences flush them)'},
            'rank_ms_per_step': [round(v, 3) for v in rank_ms],
            'exposed_comm_ms': None if exposed_comm_ms is None else round(exposed_comm_ms, 3),
            'comm_bytes_on_wire': comm_bytes,
            'bf16_transport_leg': bf16_leg,
            # what a bandwidth-bound ring all-reduce of this step's gradients would take on xGMI: 2 (N-1)/N x bytes over ONE
            # link's ~153 GB/s (point-to-point links: a ring is per-link bound) -- to read exposed_comm_ms against
            'expected_ring_ms': (None if world < 2 else round(
                2.0 * (world - 1) / world * sum(p.numel() for p in model.parameters()) * 4 / 153e9 * 1e3, 3)),
            # where the gradient bytes are reduced: the deferred units' weight gradients (and their all-reduces) run beside the
            # NEXT forward pass, everything else inside the backward pass -- so a first multi-GPU curve explains itself
            'grad_bytes': grad_byte_split(model, defer_k),
            'exposed_comm_ms_by_rank': exposed_by_rank,
            'tune_plans': (None if tune_shas is None else
                           {'shared_from_rank0': tune_path, 'sha16_by_rank': tune_shas, 'identical': len(set(tune_shas)) == 1}),
            'per_gpu_value': round(value / world, 1),
            # recorded launch lists (replay.py): which step shapes are replayed through one w2l_replay call per phase
            'replay': _replay.report(model.engine()),
        }
    else:
        line = None

    def emit():
        if rank == 0:
            sys.stdout.flush()
            os.write(record_fd, (json.dumps(line) + '\n').encode())

    # ---- which collective path?  Measured here, in this run, on this node: nobody is there to flip a switch on the first
    # 8-GPU run.  The record above is complete and was measured with the gradient collectives through torch.distributed
    # (ProcessGroupNCCL: its own internal stream, the one stream of the step streams.py cannot probe).  Now the same timed
    # region once more through the C ABI's RCCL helpers (NativeComm: a collective is a launch on the reducer's probed stream);
    # if that is faster it becomes the record.  RCCL through this path has never run more than one rank, so the leg is bounded:
    # the leg is opt-in (--collective-ab), the communicator is created on a helper thread with a deadline, and a watchdog prints
    # the record already in hand and ends the process with exit code 3 should a native collective never return.  The
    # two communicators never have work in flight together: every switch sits between two fences (device sync + barrier).
    reducer0 = getattr(model, 'grad_reducer', None)
    forced_native = os.environ.get('W2L_DP_NATIVE')
    if (reducer0 is not None and reducer0.active and args.collective_ab and not args.no_collective_ab and not args.graph
            and forced_native is None):
        import threading
        leg_s = float(os.environ.get('W2L_AB_TIMEOUT', '150'))
        native, via_torch = 'w2l_rccl_* (C ABI)', 'torch.distributed'
        collective_paths = {via_torch: round(ms, 3), native: None, 'kept': via_torch}
        if line is not None:
            line['collective_paths_ms'] = collective_paths

        def give_up():
            collective_paths['native_unavailable'] = f'the native-collective leg did not finish within {leg_s:.0f} s: abandoned'
            emit()
            os._exit(3)       # never 0: a rank stuck in a collective is not a clean exit, whatever it printed first

        watchdog = threading.Timer(leg_s, give_up)
        watchdog.daemon = True
        watchdog.start()
        box = {}

        def create():
            try:
                box['comm'] = NativeComm.from_process_group()
            except Exception as e:        # noqa: BLE001 -- reported in the line; the run keeps torch.distributed
                box['why'] = repr(e)

        th = threading.Thread(target=create, daemon=True)
        th.start()
        th.join(min(60.0, leg_s / 2))
        comm = box.get('comm') if not th.is_alive() else None
        why = box.get('why') or ('communicator creation timed out' if th.is_alive() else None)
        ok = torch.tensor([1.0 if comm is not None else 0.0], device=dev)
        if world > 1:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)            # all ranks or none
        if float(ok) == 1.0:
            fence()
            reducer0.set_native(comm)
            step()
            step()                        # the communicator's first collectives (channel set-up) stay out of the timing
            fence()
            ms_n, rank_ms_n, loss_n = timed_region()
            collective_paths[native] = round(ms_n, 3)
            if comm.rehearsal:
                collective_paths['note'] = ('one-GPU rehearsal (W2L_DIST_BACKEND=gloo): the native path ran on one-rank '
                                            'communicators, averaging nothing')
            if ms_n < ms and line is not None:
                collective_paths['kept'] = native
                line.update(value=round(world * N * T / (ms_n * 1e-3), 1), ms_per_step=round(ms_n, 3),
                            rank_ms_per_step=[round(v, 3) for v in rank_ms_n], collectives_via=native,
                            per_gpu_value=round(N * T / (ms_n * 1e-3), 1))
                line['config']['loss'] = round(float(loss_n.detach()), 4)
            if solo_ms is not None and ms_n < ms:
                by = [round(v, 3) for v in gather_objects(dist, world, rank_ms_n[rank] - solo_ms)]
                if line is not None:
                    line.update(exposed_comm_ms=max(by), exposed_comm_ms_by_rank=by)
            elif solo_ms is not None:
                gather_objects(dist, world, 0.0)          # (keeps the ranks' collective sequences identical)
        else:
            collective_paths['native_unavailable'] = why or 'another rank could not create its communicator'
            if comm is not None:
                comm.close()
        watchdog.cancel()
    emit()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
