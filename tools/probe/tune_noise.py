"""How much of the run-to-run spread of bench.py is the measured plan selection?  N fresh runs, each saving its plans
(W2L_TUNE_CACHE), then every saved plan file is run again: the spread between files vs. the spread of one file.
    python tools/probe/tune_noise.py [--runs 4] [--out DIR]      (bench.py runs as child processes only)"""
import argparse
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def bench(cache, extra=()):
    env = dict(os.environ, W2L_TUNE_CACHE=cache)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--defer-wgrad', '4', '--no-cpu-baseline', '--no-live-traffic', *extra],
                         env=env, capture_output=True, text=True, timeout=900)
    for line in reversed(out.stdout.splitlines()):
        if line.startswith('{'):
            return json.loads(line)
    raise RuntimeError(out.stderr[-2000:])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--runs', type=int, default=4)
    ap.add_argument('--out', default='/tmp/tune_noise')
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    res = {}
    for i in range(args.runs):
        c = os.path.join(args.out, f'c{i}.txt')
        if os.path.exists(c):
            os.remove(c)
        r = bench(c)
        res[i] = [r['ms_per_step']]
        print('fresh', i, r['ms_per_step'], flush=True)
    for rep in range(2):
        for i in range(args.runs):
            r = bench(os.path.join(args.out, f'c{i}.txt'))
            res[i].append(r['ms_per_step'])
            print('again', i, r['ms_per_step'], flush=True)
    print(json.dumps(res))


if __name__ == '__main__':
    main()
