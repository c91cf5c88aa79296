"""Loops of every kernel in an llvm-objdump -d listing that contain MFMAs: instruction count, MFMAs, scratch accesses,
lane moves (SGPR spills), v_accvgpr moves, LDS-DMA pieces.   python tools/probe/loop_stats.py listing.s"""
import re
import sys

funcs, cur, start = {}, None, {}
for l in open(sys.argv[1]):
    m = re.match(r'^([0-9a-f]+) <(\w+)>:', l)
    if m:
        cur = m.group(2)
        funcs[cur] = []
        start[cur] = int(m.group(1), 16)
        continue
    m = re.match(r'^\s+(\S+)\s*(.*?)//\s*([0-9A-Fa-f]+):(.*)$', l)
    if m and cur:
        funcs[cur].append((int(m.group(3), 16), m.group(1), m.group(2) + m.group(4)))
for name, ins in funcs.items():
    out = []
    for a, op, rest in ins:
        if op.startswith('s_cbranch') or op == 's_branch':
            m = re.search(r'<\w+\+0x([0-9a-f]+)>', rest)
            if m:
                t = int(m.group(1), 16) + start[name]
                if t < a:
                    b = [x for x in ins if t <= x[0] <= a]
                    n = lambda pre: sum(1 for x in b if x[1].startswith(pre))
                    if n('v_mfma'):
                        out.append({'ins': len(b), 'mfma': n('v_mfma'), 'scratch': n('scratch_'), 'lane': n('v_readlane') + n('v_writelane'),
                                    'accvgpr': n('v_accvgpr'), 'dma': n('global_load_lds'), 'ds_read': n('ds_read'), 'at': hex(t)})
    print(name)
    for o in out:
        print('   ', o)
