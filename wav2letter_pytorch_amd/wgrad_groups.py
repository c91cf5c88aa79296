"""Which weight gradients of a backward pass share a launch (w2l_conv1d_wgrad_group, include/w2l_hip.h).

A weight-gradient launch is a pool of [128 co x 128 ci] x tap-group tiles on 512 (4-wave blocks) or 256 (8-wave blocks) block
slots; a layer whose own tiles fill a fraction of a round -- 640 -> 640, k21: 175 three-tap tiles -- either idles the rest of the
chip or splits its reduction and pays for the partial tiles (fp32 atomics + a zero-filled dw, or slabs: both cost what the split
wins).  Consecutive layers of the backward pass that agree in (N, Tout, stride 1, dilation) can share ONE
launch instead: 640 -> 640 twice + 512 -> 640 are 490 three-tap tiles, 96 % of a round with no split at all (measured 1 276
against 1 032 TFLOP/s one by one).  This module only PLANS (host arithmetic, no device work): a dynamic programme over the
backward order with a cost model in units of "one tap of one tile", whole rounds; the step engine then MEASURES every planned
group once, during warm-up, in each block form against its members launched one by one, and keeps what is faster
(engine.StackEngine._wgrad_group_now).

Replaces nothing in the reference by itself: it schedules the weight half of aten::convolution_backward for the nn.Conv1d call
sites wav2letter.py:35-36,42 / jasper.py:96-105,127."""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

# block forms of a group launch: plan order bits (include/w2l_hip.h) -> (taps per block, block slots on the chip, relative speed
# of a full round: measured on the 896 -> 896 / 768 -> 768 layers, profiles/r05_wgrad_groups.txt)
FORMS = {
    17: (3, 512, 0.95),      # three taps per wave (AGPR accumulators), two 4-wave blocks per CU
    21: (6, 256, 1.00),      # two such tap groups: one 8-wave block per CU
    5: (4, 256, 0.97),       # two taps per wave, two tap groups: one 8-wave block per CU
    1: (2, 512, 0.90),       # two taps per wave, two 4-wave blocks per CU
}
MAX_GROUP = 8                # W2L_WGRAD_GROUP_MAX
THREE_TAP_MAX_DIL = 4        # the three-tap kernels' static LDS window


def tiles(cin: int, cout: int, kw: int, form: int) -> int:
    """w2l_wgrad_group_tiles: the tiles a layer adds to a launch of this form"""
    kwblk = FORMS[form][0] if (kw > 1 or form & 16) else 1
    return -(-cout // 128) * -(-cin // 128) * -(-kw // kwblk)


def forms_for(dil: int) -> List[int]:
    return [f for f in FORMS if not (f & 16 and dil > THREE_TAP_MAX_DIL)]


# one launch costs its own 12 us or so whatever it computes (a full round of 1 536 tap-tiles takes ~590 us: 0.38 us per unit), a
# split launch a zero fill of dw on top, and its partial tiles (fp32 atomics) ~15 % of its time: what makes the narrow layers'
# own plans (512 -> 512, k17: 908 TFLOP/s at split 3) lose to a shared launch.  (The model is deliberately pessimistic about
# sparsely filled rounds: groups of the NARROWEST layers -- 384 / 256 wide -- win 30 % as launches and lose in the step, where
# their long-lived blocks hold CUs the main stream's kernels wait for: measured, profiles/r05_step_ab.txt)
LAUNCH_UNITS = 31.0
FILL_UNITS = 26.0
SPLIT_PENALTY = 1.15


def launch_cost(layers: Sequence[Tuple[int, int, int]], form: int, splits: int = 1) -> float:
    """cost of one launch in tap-tile units: whole rounds of the form's slots + the launch itself (+ the zero fill and the
    partial tiles of a split)"""
    kwblk, slots, speed = FORMS[form]
    n = sum(tiles(ci, co, kw, form) for ci, co, kw in layers) * splits
    rounds = -(-n // slots)
    work = rounds * slots * kwblk / splits / speed
    return work * (SPLIT_PENALTY if splits > 1 else 1.0) + LAUNCH_UNITS + (FILL_UNITS if splits > 1 else 0.0)


def best_cost(layers: Sequence[Tuple[int, int, int]], dil: int, allow_split: bool) -> Tuple[float, int]:
    best = (float('inf'), 0)
    for f in forms_for(dil):
        for s in ((1, 2, 3, 4, 6, 8, 10) if allow_split else (1,)):
            c = launch_cost(layers, f, s)
            if c < best[0]:
                best = (c, f)
    return best


def plan(seq: Sequence[Optional[Tuple[int, int, int, int, int]]], max_group: int = MAX_GROUP) -> List[List[int]]:
    """seq: the convolutions of a backward pass in launch order, each (cin_padded, cout_padded, kw, dil, compat) or None for one
    that cannot join a group (strided, deferred, fp8, ...); ``compat`` is any hashable the members of a group must share (N, Tout).
    Returns the groups (lists of indices into seq, consecutive among the groupable entries, at least two members each): the
    partition of minimum modelled cost, where a layer on its own may split its reduction and a group may not."""
    idx = [i for i, e in enumerate(seq) if e is not None]
    n = len(idx)
    if n < 2:
        return []
    single = [best_cost([seq[i][:3]], seq[i][3], True)[0] for i in idx]
    INF = float('inf')
    cost = [0.0] + [INF] * n
    back = [0] * (n + 1)
    for j in range(1, n + 1):
        cost[j] = cost[j - 1] + single[j - 1]
        back[j] = j - 1
        for i in range(j - 2, max(j - 1 - max_group, -1), -1):          # group idx[i:j]
            members = [seq[idx[k]] for k in range(i, j)]
            if any(m[3] != members[0][3] or m[4] != members[0][4] for m in members):
                break
            c = cost[i] + best_cost([m[:3] for m in members], members[0][3], False)[0]
            if c < cost[j] - 1e-9:
                cost[j], back[j] = c, i
    groups = []
    j = n
    while j > 0:
        i = back[j]
        if j - i >= 2:
            groups.append([idx[k] for k in range(i, j)])
        j = i
    groups.reverse()
    return groups


def parse_override(text: str, n: int) -> Optional[List[List[int]]]:
    """W2L_WGRAD_GROUPS: '0' / 'off' = none, 'auto' = the plan, or explicit groups of backward-order indices 'i,j,k;l,m'"""
    t = text.strip().lower()
    if t in ('auto', '1', ''):
        return None
    if t in ('0', 'off', 'none'):
        return []
    out = []
    for part in t.split(';'):
        g = sorted(int(v) for v in part.split(',') if v.strip() != '')
        if len(g) >= 2 and all(0 <= v < n for v in g):
            out.append(g)
    return out


def setting() -> str:
    return os.environ.get('W2L_WGRAD_GROUPS', 'auto')
