"""Same-box A/B of weight-gradient plans INSIDE the training step (not the isolated-launch time the tuner minimises):
bench.py runs once to measure and save the plans, then again with copies of that cache whose `wgrad` lines for the wide
stride-1 layers are overwritten by a forced (split, order), and the ms/step of every variant is printed.

    python tools/probe/plan_ab.py [--steps 20] [--rounds 2] [--variants 'name:split,order;...']

Runs bench.py as child processes only (this process never touches the GPU)."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def bench(cache, steps, extra=()):
    env = dict(os.environ, W2L_TUNE_CACHE=cache)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', str(steps), '--warmup', '5', '--defer-wgrad', '4',
                          '--no-cpu-baseline', '--no-live-traffic', *extra], env=env, capture_output=True, text=True, timeout=900)
    for line in reversed(out.stdout.splitlines()):
        if line.startswith('{'):
            return json.loads(line)
    raise RuntimeError(out.stderr[-2000:])


def variant(src, dst, split, order, min_c):
    with open(src) as f, open(dst, 'w') as g:
        for line in f:
            v = line.split()
            if v and v[0] == 'wgrad' and int(v[5]) >= 3 and min(int(v[2]), int(v[3])) >= min_c and int(v[2]) % 128 == 0 and int(v[3]) % 128 == 0:
                v[6], v[7] = str(split), str(order)
                line = ' '.join(v) + '\n'
            g.write(line)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--rounds', type=int, default=2)
    ap.add_argument('--min-c', type=int, default=384)
    ap.add_argument('--variants', default='t3:1,17;t6:1,21;t2:1,1;t4:1,5')
    ap.add_argument('--dir', default='/tmp/plan_ab')
    args = ap.parse_args()
    os.makedirs(args.dir, exist_ok=True)
    base = os.path.join(args.dir, 'base.txt')
    if os.path.exists(base):
        os.remove(base)
    r = bench(base, 5)
    print('measured plans:', r['ms_per_step'], flush=True)
    names = ['base']
    for spec in args.variants.split(';'):
        name, so = spec.split(':')
        s, o = (int(x) for x in so.split(','))
        variant(base, os.path.join(args.dir, name + '.txt'), s, o, args.min_c)
        names.append(name)
    res = {n: [] for n in names}
    for _ in range(args.rounds):
        for n in names:
            r = bench(os.path.join(args.dir, n + '.txt'), args.steps)
            res[n].append(r['ms_per_step'])
            print(n, r['ms_per_step'], 'wgrad', r['roofline'].get('wgrad_kernel', {}).get('achieved'), flush=True)
    print(json.dumps(res))


if __name__ == '__main__':
    main()
