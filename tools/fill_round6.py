#!/usr/bin/env python3
"""Fills the R6_* placeholders of DESIGN.md / README.md from the round's profile files (profiles/r06_*), so that every number
in the prose is the one in the committed record.  Run after copying gpurun_out/final/* to profiles/r06_*.
(One-shot: the templates with the R6_* markers are the DESIGN.md / README.md of commit "r06 profiles (final code); DESIGN / README
drafts"; the files in the tree are its output.)"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, 'profiles')


def rec(name):
    return json.loads(open(os.path.join(P, name)).read().strip().splitlines()[-1])


def ab(cfg):
    """(replayed 'dev / host', eager 'dev / host') of one tools/replay_ab.py block of r06_replay_ab.txt"""
    text = open(os.path.join(P, 'r06_replay_ab.txt')).read()
    blk = text.split('== tools/replay_ab.py ' + cfg + '\n')[1].split('== tools/replay_ab.py')[0]
    out = {}
    for kind in ('replay', 'eager'):
        m = re.search(kind + r'\s+ms/step .*?\(mean ([\d.]+)\)\s+host enqueue ms/step .*?\(mean ([\d.]+)\)', blk)
        out[kind] = '%.2f / %.2f' % (float(m.group(1)), float(m.group(2)))
    return '%s | %s' % (out['replay'], out['eager'])


def lead(name):
    rows = [l.split('|') for l in open(os.path.join(P, name)) if l.count('|') == 5]
    vals = sorted(float(r[4]) for r in rows[1:] if r[4].strip() not in ('nan', 'host ms in step'))
    return '%.1f' % vals[len(vals) // 2]                # median host ms per step


def main():
    head = rec('r06_bench.json')
    tr = {k: rec('r06_bench_through_trainer%s.json' % s) for k, s in (('W2L', ''), ('J', '_jasper10x5'), ('J8', '_jasper10x5_fp8'))}
    m1 = rec('r06_bench_w2l_mid1_cpu_baseline.json')
    jc = rec('r06_bench_jasper10x5_cpu_baseline.json')
    sub = {}
    for k, d in tr.items():
        loop = d['trainer_loop']
        raw, asy, syn = loop['raw_loop']['ms_per_step'], loop['async_metrics']['ms_per_step'], loop['sync_metrics_before']['ms_per_step']
        sub['R6_%s_RAW' % k] = '%.2f' % raw
        sub['R6_%s_TR' % k] = '%.2f' % asy
        sub['R6_%s_RATIO' % k] = '%.3fx' % (asy / raw)
        sub['R6_%s_SYNC' % k] = '%.2f' % syn
    sub['R6_M1_RAW'] = '%.2f' % m1['ms_per_step']
    sub['R6_M1_TR'] = '%.2f' % m1['trainer_loop']['async_metrics']['ms_per_step']
    sub['R6_M1_SYNC'] = '%.2f' % m1['trainer_loop']['sync_metrics_before']['ms_per_step']
    sub['R6_CPU_M1'] = '%.0f' % m1['cpu_baseline']['value']
    sub['R6_CPU_J'] = '%.0f frames/s' % jc['cpu_baseline']['value']
    sub['R6_HEADV'] = '%.2f' % (head['value'] / 1e6)
    sub['R6_HEAD'] = '%.2f' % head['ms_per_step']
    sub['R6_FRAC'] = '%.1f %%' % (100 * head['roofline']['frac'])
    sub['R6_TWO_TAP'] = '%.2f' % head['roofline']['wgrad_kernel']['by_kernel_family']['two_tap_kernels']['frac']
    for key, cfg in (('W2L', ''), ('N16', '--batch 16'), ('N8', '--batch 8'), ('F8', '--dtype fp8'), ('M1', '--mid-layers 1'),
                     ('J', '--model jasper10x5 --batch 16'), ('J8', '--model jasper10x5 --batch 16 --dtype fp8')):
        sub['R6_AB_' + key] = ab(cfg)
    sub['R6_LEAD_WE'] = lead('r06_lead_trace_eager.txt')
    sub['R6_LEAD_W'] = lead('r06_lead_trace.txt')
    sub['R6_LEAD_JE'] = lead('r06_lead_trace_jasper10x5_eager.txt')
    sub['R6_LEAD_J'] = lead('r06_lead_trace_jasper10x5.txt')
    bn = open(os.path.join(P, 'r06_bn_kernels.txt')).read()
    m = re.search(r'these 13 layers: ([\d.]+) ms for ([\d.]+) GB = ([\d.]+) TB/s', bn)
    sub['R6_SGD_ALONE'] = '%s ms for the %s GB of 13 table layers = %s TB/s' % m.groups() if m else 'n/a'
    for path in ('DESIGN.md', 'README.md'):
        text = open(os.path.join(ROOT, path)).read()
        for k in sorted(sub, key=len, reverse=True):
            text = text.replace(k, sub[k])
        left = sorted(set(re.findall(r'R6_[A-Z0-9_]+', text)))
        if left:
            print(path, 'unfilled:', left, file=sys.stderr)
        open(os.path.join(ROOT, path), 'w').write(text)
    print(json.dumps(sub, indent=1))


if __name__ == '__main__':
    main()
