#!/usr/bin/env python3
"""Timeline of a training step from a rocprofv3 kernel trace (one row per dispatch with start / end timestamps).

  rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline
  python3 tools/timeline.py OUT/*/*_kernel_trace.csv [--steps 4] [--from 4]

Kernels are grouped into families (MFMA-bound convolutions vs HBM-bound element-wise passes vs the rest); for the last
`--steps` occurrences of the step's first kernel the tool prints, per step: wall time, time with no kernel running, time in
which ONLY non-MFMA kernels run (the exposed tail), time in which two MFMA kernels overlap, and the stretch of every family
(sum of its launches' durations inside the step vs the same family's minimum-duration launches)."""
import csv
import sys
from collections import defaultdict

MFMA = ('conv_igemm', 'conv_wgrad')          # (w2l_wgrad3* -- the three-tap code object's kernels -- count as conv_wgrad)


def family(name):
    if 'w2l_wgrad3' in name:
        return 'conv_wgrad'
    if 'bn_bwd_reduce_fast' in name or 'bn_bwd_reduce_loop' in name:      # (round 5's two-launch backward chain, one-shot / looped)
        return 'bn_act_bwd_reduce'
    if 'bn_bwd_apply_fast' in name or 'bn_bwd_apply_loop' in name:
        return 'bn_act_bwd_apply'
    for key in ('conv_igemm_fp8', 'conv_wgrad_fp8', 'conv_igemm', 'conv_wgrad', 'bn_act_fwd', 'bn_act_bwd_reduce', 'bn_act_bwd_apply',
                'bn_bwd_finalize', 'bn_finalize', 'sgd_pack', 'ctc_', 'log_softmax', 'nct_to_ntc', 'pad_cast', 'quantize', 'dw_',
                'probe_', 'Cijk', 'ncclDevKernel', 'rccl'):
        if key in name:
            return key.rstrip('_')
    if 'elementwise' in name or 'vectorized' in name or 'fill' in name.lower():
        return 'torch_elementwise'
    return 'other'


def union_len(iv):
    iv = sorted(iv)
    tot, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


def overlap2(iv):
    """time covered by at least two intervals"""
    ev = []
    for s, e in iv:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    depth, last, tot = 0, None, 0
    for t, d in ev:
        if depth >= 2:
            tot += t - last
        depth += d
        last = t
    return tot


def main():
    path = sys.argv[1]
    nsteps = int(sys.argv[sys.argv.index('--steps') + 1]) if '--steps' in sys.argv else 4
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', r.get('Stream_Id', '0'))))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if 'nct_to_ntc' in r[2]]          # first kernel of a forward pass
    if len(starts) < nsteps + 1:
        print('not enough steps in the trace (%d forward passes found)' % len(starts))
        return
    if '--from' in sys.argv:                       # index of the first forward pass to analyse (bench.py: warm-up steps come first,
        first = int(sys.argv[sys.argv.index('--from') + 1])      # the 3 steps of its instrumented pass -- serialised -- last)
        starts = starts[first:first + nsteps + 1]
    else:
        starts = starts[-(nsteps + 1):]
    agg = defaultdict(lambda: [0.0, 0])
    tot = defaultdict(float)
    for a, b in zip(starts[:-1], starts[1:]):
        seg = rows[a:b]
        t0, t1 = seg[0][0], rows[b][0]
        allv = [(s, min(e, t1)) for s, e, *_ in seg]
        mf = [(s, min(e, t1)) for s, e, n, _ in seg if family(n).startswith(MFMA)]
        busy, mbusy = union_len(allv), union_len(mf)
        tot['wall'] += (t1 - t0) / 1e3
        tot['idle'] += (t1 - t0 - busy) / 1e3
        tot['non-MFMA only'] += (busy - mbusy) / 1e3
        tot['two MFMA kernels'] += overlap2(mf) / 1e3
        tot['MFMA busy'] += mbusy / 1e3
        for s, e, n, _ in seg:
            fam = family(n)
            agg[fam][0] += (e - s) / 1e3
            agg[fam][1] += 1
    if '--gaps' in sys.argv:
        # the intervals of ONE step (the first analysed) in which no MFMA-bound kernel runs: when (us from the step's start),
        # how long, what runs meanwhile, and the MFMA kernels on either side
        a, b = starts[0], starts[1]
        seg = rows[a:b]
        t0, t1 = seg[0][0], rows[b][0]
        mf = sorted((s, min(e, t1), n) for s, e, n, _ in seg if family(n).startswith(MFMA))
        merged = []
        for s, e, n in mf:
            if merged and s <= merged[-1][1]:
                merged[-1][1] = max(merged[-1][1], e)
            else:
                merged.append([s, e])
        gaps = [(t0, merged[0][0])] + [(merged[i][1], merged[i + 1][0]) for i in range(len(merged) - 1)] + [(merged[-1][1], t1)]
        def short(n):
            n = n.split('(')[0]
            return n[n.rfind('::') + 2:] if '::' in n else n
        print('gaps of step 0 (no MFMA kernel running), us from the step start; total %.1f us in %d gaps:' %
              (sum(e - s for s, e in gaps) / 1e3, len(gaps)))
        # where the CTC kernels sit splits the step into forward and backward
        ctc0 = min((s for s, e, n, _ in seg if 'ctc' in n), default=t1)
        fwd = sum(min(e, ctc0) - s for s, e in gaps if s < ctc0) / 1e3
        print('  before the first CTC kernel (forward): %.1f us; after: %.1f us; forward lasts %.1f us' %
              (fwd, sum(e - s for s, e in gaps) / 1e3 - fwd, (ctc0 - t0) / 1e3))
        for s, e in gaps:
            if e - s < 3000:
                continue
            inside = [short(n) + ':%.0f' % ((min(ke, e) - max(ks, s)) / 1e3) for ks, ke, n, _ in seg if ks < e and ke > s]
            print('  @%8.1f  %7.1f us  %s' % ((s - t0) / 1e3, (e - s) / 1e3, ' '.join(inside)[:180]))
    n = len(starts) - 1
    print('per step over %d steps (us):' % n)
    for k in ('wall', 'MFMA busy', 'two MFMA kernels', 'non-MFMA only', 'idle'):
        print('  %-20s %9.1f' % (k, tot[k] / n))
    print('families (sum of launch durations per step, launches per step, mean us):')
    for fam, (us, cnt) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        print('  %-22s %9.1f us  %5.1f launches  %8.1f us each' % (fam, us / n, cnt / n, us / cnt))


if __name__ == '__main__':
    main()
