"""Condenses rocprofv3 output into the small files committed under profiles/.

  prof_summary.py stats <kernel_stats.csv>            -> per kernel FAMILY (template instances merged) table, stdout
  prof_summary.py pmc <out.json> <dir> [<dir> ...]     -> per-family mean counters per launch + corrected HBM traffic

FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE tallies wide streaming reads at half their bytes
(/opt/skills/guides/MI355X_MICROARCH.md, "HBM"), so traffic = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 bytes.
"""
import collections
import csv
import glob
import json
import re
import sys


def family(name):
    m = re.search(r'(\w+)(<.*?>)?\(', name)
    return m.group(1) if m else name[:48]


def stats(path):
    agg = collections.OrderedDict()
    total = 0.0
    for r in csv.DictReader(open(path)):
        f = family(r['Name'])
        a = agg.setdefault(f, [0, 0.0])
        a[0] += int(r['Calls'])
        a[1] += float(r['TotalDurationNs'])
        total += float(r['TotalDurationNs'])
    print('%-44s %8s %12s %12s %7s' % ('kernel family', 'calls', 'total ms', 'avg us', '%'))
    for f, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print('%-44s %8d %12.3f %12.2f %7.2f' % (f, c, t / 1e6, t / c / 1e3, 100 * t / total))


def pmc(out, dirs):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
            for r in csv.DictReader(open(f)):
                agg[family(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
    res = {}
    for k, d in agg.items():
        e = {c: sum(v) / len(v) for c, v in d.items()}
        e['launches'] = max(len(v) for v in d.values())
        if 'FETCH_SIZE' in e and 'WRITE_SIZE' in e:
            e['traffic_bytes_per_launch'] = 2 * e['FETCH_SIZE'] * 1024 + e['WRITE_SIZE'] * 1024
        res[k] = e
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import csrc_fingerprint                    # the kernel sources these counters were measured on
    json.dump({'note': 'per-launch means; traffic = 2*FETCH_SIZE KiB + WRITE_SIZE KiB (gfx950 correction)',
               'csrc_sha': csrc_fingerprint(), 'kernels': res}, open(out, 'w'), indent=1, sort_keys=True)


if __name__ == '__main__':
    if sys.argv[1] == 'stats':
        stats(sys.argv[2])
    else:
        pmc(sys.argv[2], sys.argv[3:])
