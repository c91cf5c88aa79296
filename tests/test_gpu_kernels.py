"""Kernel-level parity on a real MI355X: every C-ABI entry point against the CPU oracle
(oracle/w2l_oracle.py, torch-CPU fp32 / numpy) on the same seeded inputs.
Tolerances: bf16-operand kernels are compared on bf16-rounded inputs (fp32 accumulate) at
2e-3 relative to the output scale; fp32 elementwise kernels at 1e-5."""
import ctypes as C
import ctypes as C_
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


@pytest.fixture(scope='module')
def L():
    from wav2letter_pytorch_amd import _lib
    assert torch.cuda.is_available()
    return _lib


def bf(x):
    return x.to(torch.bfloat16).float()


def relerr(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def to_ntc_padded(x_nct, pad_l, pad_r, mode, cp=None):
    """reference construction of the padded channels-last buffer on the CPU"""
    n, c, t = x_nct.shape
    cp = cp or c
    if pad_l or pad_r:
        x_nct = F.pad(x_nct, (pad_l, pad_r), mode='reflect' if mode == 1 else 'constant')
    out = torch.zeros(n, t + pad_l + pad_r, cp)
    out[:, :, :c] = x_nct.transpose(1, 2)
    return out


@pytest.mark.parametrize('N,C,T,CP,pl,pr,mode', [(3, 64, 100, 64, 4, 5, 1), (2, 40, 37, 64, 28, 28, 1), (2, 64, 50, 64, 3, 3, 0),
                                                 (1, 29, 33, 64, 0, 0, 0)])
def test_nct_to_ntc(L, N, C, T, CP, pl, pr, mode):
    torch.manual_seed(0)
    x = torch.randn(N, C, T)
    xd = x.cuda()
    hi = torch.empty(N, pl + T + pr, CP, dtype=torch.bfloat16, device='cuda')
    lo = torch.empty_like(hi)
    L.check(L.lib.w2l_nct_to_ntc(L.ptr(xd), N, C, T, CP, pl, pr, mode, None, L.ptr(hi), L.ptr(lo), L.stream_ptr()))
    ref = to_ntc_padded(x, pl, pr, mode, CP)
    assert torch.equal(hi.float().cpu(), bf(ref))
    rec = hi.float().cpu() + lo.float().cpu()
    assert (rec - ref).abs().max() <= 2 ** -16 * ref.abs().max()
    # length masking
    lens = torch.tensor([T - 7] + [T] * (N - 1), dtype=torch.int32).cuda()
    L.check(L.lib.w2l_nct_to_ntc(L.ptr(xd), N, C, T, CP, pl, pr, 0, L.ptr(lens), L.ptr(hi), None, L.stream_ptr()))
    xm = x.clone()
    xm[0, :, T - 7:] = 0
    assert torch.equal(hi.float().cpu(), bf(to_ntc_padded(xm, pl, pr, 0, CP)))


def pack(L, w, precise=False):
    cout, cin, kw = w.shape
    coutp, cinp = (cout + 63) // 64 * 64, (cin + 63) // 64 * 64
    wd = w.cuda()
    fh = torch.empty(kw, coutp, cinp, dtype=torch.bfloat16, device='cuda')
    dh = torch.empty(kw, cinp, coutp, dtype=torch.bfloat16, device='cuda')
    fl = torch.empty_like(fh) if precise else None
    dl = torch.empty_like(dh) if precise else None
    L.check(L.lib.w2l_pack_weights(L.ptr(wd), wd.stride(0), wd.stride(1), wd.stride(2), cout, cin, kw, coutp, cinp,
                                   L.ptr(fh), L.ptr(fl), L.ptr(dh), L.ptr(dl), L.stream_ptr()))
    return fh, fl, dh, dl, coutp, cinp


def test_pack_weights(L):
    torch.manual_seed(1)
    for (cout, cin, kw) in [(96, 40, 5), (128, 64, 11), (29, 128, 1)]:
        w = torch.randn(cout, cin, kw)
        for wsrc in (w, torch.randn(kw, cout, cin).permute(1, 2, 0)):
            fh, fl, dh, dl, coutp, cinp = pack(L, wsrc, precise=True)
            ref = torch.zeros(kw, coutp, cinp)
            ref[:, :cout, :cin] = wsrc.permute(2, 0, 1)
            assert torch.equal(fh.float().cpu(), bf(ref))
            refd = torch.zeros(kw, cinp, coutp)
            refd[:, :cin, :cout] = wsrc.flip(2).permute(2, 1, 0)
            assert torch.equal(dh.float().cpu(), bf(refd))
            assert ((fh.float() + fl.float()).cpu() - ref).abs().max() <= 2 ** -16 * ref.abs().max()
            assert ((dh.float() + dl.float()).cpu() - refd).abs().max() <= 2 ** -16 * refd.abs().max()


CONV_CASES = [
    # N, Cin, Cout, Kw, s, d, T(valid in), pad_l, pad_r
    (2, 64, 128, 11, 2, 1, 100, 4, 5),      # layer-0 shape: stride 2, asymmetric reflect pad
    (3, 128, 128, 11, 1, 1, 137, 5, 5),
    (2, 64, 192, 29, 1, 2, 150, 28, 28),    # dilation 2, Cout not a multiple of the 128 tile
    (2, 192, 64, 1, 1, 1, 70, 0, 0),        # 1x1 (classifier-like, Cout < tile)
    (1, 256, 256, 13, 1, 1, 500, 6, 6),     # T spans several 128-row tiles
    (2, 64, 64, 29, 1, 2, 75, 28, 28),      # short utterances: K steps run into the zero gap / next utterance
]


def conv_inputs(N, Cin, Cout, Kw, T, pl, pr, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, Cin, T, generator=g)
    w = torch.randn(Cout, Cin, Kw, generator=g) / (Cin * Kw) ** 0.5
    b = torch.randn(Cout, generator=g)
    return x, w, b


@pytest.mark.parametrize('case', CONV_CASES)
@pytest.mark.parametrize('precise', [False, True])
def test_conv_igemm_forward(L, case, precise):
    N, Cin, Cout, Kw, s, d, T, pl, pr = case
    x, w, b = conv_inputs(N, Cin, Cout, Kw, T, pl, pr, 3)
    xp = to_ntc_padded(x, pl, pr, 1)
    rows = xp.shape[1]
    Tout = (rows - (Kw - 1) * d - 1) // s + 1
    fh, fl, _, _, coutp, cinp = pack(L, w, precise)
    xh = xp.to(torch.bfloat16).cuda()
    xl = (xp - xh.float().cpu()).to(torch.bfloat16).cuda()
    y = torch.full((N, Tout, coutp), float('nan'), dtype=torch.float32 if precise else torch.bfloat16, device='cuda')
    tiles = L.lib.w2l_conv_stat_tiles(N, Tout)
    stats = torch.zeros(tiles, 2, coutp, device='cuda')
    bd = torch.zeros(coutp)
    bd[:Cout] = b
    bd = bd.cuda()
    st = L.stream_ptr()
    args = (N, cinp, coutp, Tout, Kw, s, d, st)
    if not precise:
        L.check(L.lib.w2l_conv1d_igemm(L.ptr(xh), rows * cinp, N * rows, L.ptr(fh), L.ptr(y), 0, 0, L.ptr(bd),
                                       L.ptr(stats), *args))
        ref = F.conv1d(F.pad(bf(x), (pl, pr), mode='reflect'), bf(w), b, stride=s, dilation=d)
        tol = 1e-2          # bf16 output rounding (2^-8 relative per element)
    else:
        L.check(L.lib.w2l_conv1d_igemm(L.ptr(xh), rows * cinp, N * rows, L.ptr(fh), L.ptr(y), 1, 0, L.ptr(bd), None, *args))
        L.check(L.lib.w2l_conv1d_igemm(L.ptr(xh), rows * cinp, N * rows, L.ptr(fl), L.ptr(y), 1, 1, None, None, *args))
        L.check(L.lib.w2l_conv1d_igemm(L.ptr(xl), rows * cinp, N * rows, L.ptr(fh), L.ptr(y), 1, 1, None, L.ptr(stats), *args))
        ref = F.conv1d(F.pad(x, (pl, pr), mode='reflect'), w, b, stride=s, dilation=d)
        tol = 1e-4
    torch.cuda.synchronize()
    got = y.float().cpu()[:, :, :Cout].transpose(1, 2)
    assert torch.isfinite(got).all()
    assert relerr(got, ref) < tol, relerr(got, ref)
    if coutp > Cout:
        assert (y.float().cpu()[:, :, Cout:] == 0).all()
    # BatchNorm partial statistics: sums over (n, t) of y and y^2
    s1 = stats[:, 0, :Cout].sum(0).cpu()
    s2 = stats[:, 1, :Cout].sum(0).cpu()
    r1, r2 = ref.sum((0, 2)), (ref * ref).sum((0, 2))
    assert (s1 - r1).abs().max() <= 5e-3 * r2.sqrt().max() * (N * Tout) ** 0.5 * (1 if not precise else 0.05)
    assert relerr(s2, r2) < (2e-2 if not precise else 1e-3)


def test_conv_igemm_slot_statistics_every_block_shape(L):
    """w2l_conv_stats_mode(8): the statistics of EVERY block shape -- also the 144 / 288 / 384 / 448-column ones that the
    one-row-per-128-column layout excludes --, both K-loop structures, split-K and stream-K forms: the column sums added onto
    the 8 slot rows equal the reference's, the output equals the launch without statistics"""
    N, Cin, Cout, T, Kw, d = 3, 256, 320, 620, 7, 2
    pl = pr = (Kw - 1) * d // 2
    x, w, b = conv_inputs(N, Cin, Cout, Kw, T, pl, pr, 14)
    xp = to_ntc_padded(x, pl, pr, 1)
    rows = xp.shape[1]
    Tout = rows - (Kw - 1) * d
    fh, _, _, _, coutp, cinp = pack(L, w, False)
    xh = xp.to(torch.bfloat16).cuda()
    bd = b.cuda()
    ref = F.conv1d(F.pad(bf(x), (pl, pr), mode='reflect'), bf(w), b, dilation=d)
    r1, r2 = ref.sum((0, 2)), (ref * ref).sum((0, 2))
    ws = torch.zeros(int(L.lib.w2l_conv_splitk_workspace_bytes(N, coutp, Tout)), dtype=torch.uint8, device='cuda')

    def run(idx, stats):
        y = torch.full((N, Tout, coutp), float('nan'), dtype=torch.bfloat16, device='cuda')
        L.lib.w2l_conv_force_tile_config(idx)
        L.lib.w2l_conv_stats_mode(8)
        try:
            rc = L.lib.w2l_conv1d_igemm_ws(L.ptr(xh), rows * cinp, N * rows, L.ptr(fh), L.ptr(y), 0, 0, L.ptr(bd),
                                           L.ptr(stats) if stats is not None else None, N, cinp, coutp, Tout, Kw, 1, d,
                                           L.ptr(ws), ws.numel(), L.stream_ptr())
        finally:
            L.lib.w2l_conv_stats_mode(0)
            L.lib.w2l_conv_force_tile_config(-1)
        torch.cuda.synchronize()
        return rc, y

    ran = wide = 0
    for idx in list(range(52)) + list(range(52, 52 * 8, 5)):
        slots = torch.zeros(8, 2, coutp, device='cuda')
        rc, y = run(idx, slots)
        if rc != 0:
            continue                                       # (does not fit LDS / the spilling shape)
        ran += 1
        wide += idx % 26 in (6, 7, 8, 9, 17, 18, 19, 21, 22, 23, 24, 25)
        _, y0 = run(idx, None)
        assert torch.equal(y, y0), idx
        assert relerr(y.float().cpu().transpose(1, 2)[:, :Cout], ref) < 1e-2, idx
        got = slots.sum(0).cpu()
        assert relerr(got[1, :Cout], r2) < 2e-2, idx
        assert (got[0, :Cout] - r1).abs().max() <= 5e-3 * r2.sqrt().max() * (N * Tout) ** 0.5, idx
        assert not got[:, Cout:].any(), idx
    assert ran >= 90 and wide >= 40, (ran, wide)


@pytest.mark.parametrize('with_stats', [True, False])
def test_conv_igemm_stream_k(L, with_stats):
    """the stream-K form of every block shape x both K-loop structures (split option 7: one block per resident slot, equal
    shares of all (tile, step) pairs; tiles cut by a range boundary are combined through slabs in range order) on a problem
    whose tiles are ragged in both dimensions: against the fp32 reference and against the one-block-per-tile launch of the
    same shape, twice on the same workspace (bit-identical: the order is fixed; tickets back at zero), and on other data."""
    N, Cin, Cout, T, Kw, d = 4, 512, 576, 620, 7, 2
    pl = pr = (Kw - 1) * d // 2
    x, w, b = conv_inputs(N, Cin, Cout, Kw, T, pl, pr, 12)
    xp = to_ntc_padded(x, pl, pr, 1)
    rows = xp.shape[1]
    Tout = rows - (Kw - 1) * d
    fh, _, _, _, coutp, cinp = pack(L, w, False)
    xh = xp.to(torch.bfloat16).cuda()
    bd = b.cuda()
    ref = F.conv1d(F.pad(bf(x), (pl, pr), mode='reflect'), bf(w), b, dilation=d)
    r1, r2 = ref.sum((0, 2)), (ref * ref).sum((0, 2))
    tiles = L.lib.w2l_conv_stat_tiles(N, Tout)
    ws = torch.zeros(int(L.lib.w2l_conv_splitk_workspace_bytes(N, coutp, Tout)), dtype=torch.uint8, device='cuda')

    def run(idx, xin, bias, stats_on):
        y = torch.full((N, Tout, coutp), float('nan'), dtype=torch.bfloat16, device='cuda')
        stats = torch.zeros(tiles, 2, coutp, device='cuda')
        L.lib.w2l_conv_force_tile_config(idx)
        try:
            rc = L.lib.w2l_conv1d_igemm_ws(L.ptr(xin), rows * cinp, N * rows, L.ptr(fh), L.ptr(y), 0, 0, bias,
                                           L.ptr(stats) if stats_on else None, N, cinp, coutp, Tout, Kw, 1, d,
                                           L.ptr(ws), ws.numel(), L.stream_ptr())
        finally:
            L.lib.w2l_conv_force_tile_config(-1)
        torch.cuda.synchronize()
        return rc, y, stats

    ran = 0
    for idx in range(52 * 7, 52 * 8):
        if not L.lib.w2l_conv_streamk_ranges(idx, N, cinp, coutp, Tout, Kw, 1, d, ws.numel()):
            continue
        rc, y, stats = run(idx, xh, L.ptr(bd), with_stats)
        if rc != 0:
            continue                           # (statistics need 128-column tiles)
        ran += 1
        got = y.float().cpu().transpose(1, 2)
        assert torch.isfinite(got).all(), idx
        assert relerr(got, ref) < 1e-2, (idx, relerr(got, ref))
        _, y1, _ = run(idx - 52 * 7, xh, L.ptr(bd), with_stats)           # one block per tile: the same sums, other fp32 order
        assert (y.float() - y1.float()).abs().max() <= 2 ** -6 * ref.abs().max(), idx
        if with_stats:
            assert relerr(stats[:, 1].sum(0).cpu(), r2) < 2e-2, idx
            assert (stats[:, 0].sum(0).cpu() - r1).abs().max() <= 5e-3 * r2.sqrt().max() * (N * Tout) ** 0.5, idx
        _, yb, sb = run(idx, xh, L.ptr(bd), with_stats)
        assert torch.equal(y, yb) and torch.equal(stats, sb), idx
        assert not ws[:65536].any(), idx
        _, ya, _ = run(idx, xh, None, False)
        _, y2, _ = run(idx, xh * 2, None, False)
        assert torch.equal(y2, ya * 2), idx
        if ran % 8 == 1:
            # fp32 output accumulated onto what is there (the split-bf16 passes of the fp32 mode): prefill + the same sums
            pre = torch.randn(N, Tout, coutp, device='cuda')
            yf = pre.clone()
            L.lib.w2l_conv_force_tile_config(idx)
            try:
                L.check(L.lib.w2l_conv1d_igemm_ws(L.ptr(xh), rows * cinp, N * rows, L.ptr(fh), L.ptr(yf), 1, 1, None, None, N, cinp, coutp,
                                                  Tout, Kw, 1, d, L.ptr(ws), ws.numel(), L.stream_ptr()))
            finally:
                L.lib.w2l_conv_force_tile_config(-1)
            torch.cuda.synchronize()
            want = pre[:, :, :Cout] + (ref - b[None, :, None]).transpose(1, 2).cuda()
            assert (yf[:, :, :Cout] - want).abs().max() <= 2e-3 * ref.abs().max(), idx
            assert not ws[:65536].any(), idx
    assert ran >= (10 if with_stats else 30), ran


@pytest.mark.parametrize('with_stats', [True, False])
def test_conv_igemm_every_configuration(L, with_stats):
    """every block shape x both K-loop structures x every split-K factor the autotuner may pick, on a problem with ragged
    edges in both tile dimensions (Cout = 320: 2.5 / 1.25 / 0.6 tiles; T = 300), stride 1 and the stride-2 shape.  Split
    launches run twice on the same workspace: bit-identical results (partials are summed in split order) and the tickets
    are back at zero afterwards."""
    ran = split_ran = 0
    for (s, d, Kw, pl, pr) in [(1, 2, 5, 4, 4), (2, 1, 11, 4, 5)]:
        N, Cin, Cout, T = 2, 256, 320, 300
        x, w, b = conv_inputs(N, Cin, Cout, Kw, T, pl, pr, 11)
        xp = to_ntc_padded(x, pl, pr, 1)
        rows = xp.shape[1]
        Tout = (rows - (Kw - 1) * d - 1) // s + 1
        fh, _, _, _, coutp, cinp = pack(L, w, False)
        xh = xp.to(torch.bfloat16).cuda()
        bd = b.cuda()
        ref = F.conv1d(F.pad(bf(x), (pl, pr), mode='reflect'), bf(w), b, stride=s, dilation=d)
        r1, r2 = ref.sum((0, 2)), (ref * ref).sum((0, 2))
        tiles = L.lib.w2l_conv_stat_tiles(N, Tout)
        ws = torch.zeros(int(L.lib.w2l_conv_splitk_workspace_bytes(N, coutp, Tout)), dtype=torch.uint8, device='cuda')
        for idx in range(52 * 7):                        # (26 block shapes x 2 K-loop structures) x 7 split options
            outs = []
            for rep in range(2 if idx >= 52 else 1):
                y = torch.full((N, Tout, coutp), float('nan'), dtype=torch.bfloat16, device='cuda')
                stats = torch.zeros(tiles, 2, coutp, device='cuda')
                L.lib.w2l_conv_force_tile_config(idx)
                try:
                    rc = L.lib.w2l_conv1d_igemm_ws(L.ptr(xh), rows * cinp, N * rows, L.ptr(fh), L.ptr(y), 0, 0, L.ptr(bd),
                                                   L.ptr(stats) if with_stats else None, N, cinp, coutp, Tout, Kw, s, d,
                                                   L.ptr(ws), ws.numel(), L.stream_ptr())
                finally:
                    L.lib.w2l_conv_force_tile_config(-1)
                if rc != 0:
                    break                     # this configuration cannot run the problem (statistics need 128-column tiles, ...)
                torch.cuda.synchronize()
                outs.append((y, stats))
            if not outs:
                continue
            ran += 1
            split_ran += idx >= 52
            y, stats = outs[0]
            got = y.float().cpu().transpose(1, 2)
            assert torch.isfinite(got).all(), idx
            assert relerr(got, ref) < 1e-2, (idx, relerr(got, ref))
            if with_stats:
                assert relerr(stats[:, 1].sum(0).cpu(), r2) < 2e-2, idx
                assert (stats[:, 0].sum(0).cpu() - r1).abs().max() <= 5e-3 * r2.sqrt().max() * (N * Tout) ** 0.5, idx
            if len(outs) == 2:
                assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), idx
                assert not ws[:65536].any(), idx
                if idx % 13 == 0:
                    # the same split plan and workspace on OTHER data (x * 2, no bias: exact): a slab line left in a cache by
                    # the launch before would show -- the slabs are stored and loaded past the non-coherent caches (sc1)
                    ys = []
                    for xin in (xh, xh * 2):
                        y2 = torch.full((N, Tout, coutp), float('nan'), dtype=torch.bfloat16, device='cuda')
                        L.lib.w2l_conv_force_tile_config(idx)
                        try:
                            L.check(L.lib.w2l_conv1d_igemm_ws(L.ptr(xin), rows * cinp, N * rows, L.ptr(fh), L.ptr(y2), 0, 0, None, None,
                                                              N, cinp, coutp, Tout, Kw, s, d, L.ptr(ws), ws.numel(), L.stream_ptr()))
                        finally:
                            L.lib.w2l_conv_force_tile_config(-1)
                        torch.cuda.synchronize()
                        ys.append(y2)
                    assert torch.equal(ys[1], ys[0] * 2), idx
            if with_stats and idx % 7 == 0:
                # w2l_conv_stats_mode(8): the per-tile statistics ADDED onto 8 zero rows (fp32 atomics) -- the same column sums
                slots = torch.zeros(8, 2, coutp, device='cuda')
                y3 = torch.empty(N, Tout, coutp, dtype=torch.bfloat16, device='cuda')
                L.lib.w2l_conv_force_tile_config(idx)
                L.lib.w2l_conv_stats_mode(8)
                try:
                    L.check(L.lib.w2l_conv1d_igemm_ws(L.ptr(xh), rows * cinp, N * rows, L.ptr(fh), L.ptr(y3), 0, 0, L.ptr(bd), L.ptr(slots),
                                                      N, cinp, coutp, Tout, Kw, s, d, L.ptr(ws), ws.numel(), L.stream_ptr()))
                finally:
                    L.lib.w2l_conv_stats_mode(0)
                    L.lib.w2l_conv_force_tile_config(-1)
                torch.cuda.synchronize()
                assert torch.equal(y3, y), idx
                assert relerr(slots.sum(0), stats.sum(0)) < 1e-5, idx
    assert ran >= (28 if with_stats else 40) * 6, ran
    assert split_ran >= 100, split_ran


@pytest.mark.parametrize('Kw,s,d', [(5, 1, 2), (6, 1, 1), (7, 1, 2), (8, 1, 1), (11, 2, 1), (7, 1, 3), (7, 1, 4), (7, 1, 5)])
def test_conv_wgrad_every_plan(L, Kw, s, d):
    """split counts x both block orders x {one tap group, two tap groups, 32x32x16 MFMA fragments (order bit 3; stride 1),
    three taps per block with the accumulators in AGPRs (order bit 4: the code object of conv_wgrad3_dev.hip; its last tap
    group holds 2 taps at Kw = 5 and 8, 3 at Kw = 6, 1 at Kw = 7)} per block (the autotuner's search space).  Tap counts
    4k+1 .. 4k+4: with two tap groups (the 8-wave kernel, order bit 2) the last block then has an idle tap group and a
    one-tap wave (Kw = 5), an idle tap group (6), a one-tap second group (7), or is full (8); stride 2 takes the other
    instantiation; dilation 3 and 4 fill the three-tap kernels' static LDS windows (72 / 88 rows) to the last row, dilation 5
    is beyond them: those plans take the two-tap kernels"""
    N, Cin, Cout, T, pl, pr = 3, 192, 320, 333, 4, 4
    x, w, b = conv_inputs(N, Cin, Cout, Kw, T, pl, pr, 12)
    xp = to_ntc_padded(x, pl, pr, 1)
    rows = xp.shape[1]
    Tout = (rows - (Kw - 1) * d - 1) // s + 1
    dy = torch.randn(N, Cout, Tout, generator=torch.Generator().manual_seed(13))
    hb = (Kw - 1) * d
    ha = max(hb, (Tout + 63) // 64 * 64 - Tout)
    dyp = to_ntc_padded(dy, hb, ha, 0, Cout)
    drows = dyp.shape[1]
    dyh, xh = dyp.to(torch.bfloat16).cuda(), xp.to(torch.bfloat16).cuda()
    dyh_neg = dyh * 2          # (a power of two: exact; a sign flip is not -- the matrix cores do not round symmetrically)
    wr = bf(w).requires_grad_(True)
    F.conv1d(F.pad(bf(x), (pl, pr), mode='reflect'), wr, None, stride=s, dilation=d).backward(bf(dy))
    ws = torch.zeros(int(L.lib.w2l_wgrad_workspace_bytes(Cin, Cout, Kw)), dtype=torch.uint8, device='cuda')
    for splits in (1, 2, 3, 5, 18):
        # 16, 17: three taps per block, AGPR accumulators (stride 1; else the two-tap kernel); 20, 21: two such tap groups (8 waves)
        for order in (0, 1, 4, 5, 8, 9, 16, 17, 20, 21):
            # (a) without a workspace: fp32 atomics into a zero-filled dw
            dw = torch.zeros(Kw, Cout, Cin, device='cuda')
            L.lib.w2l_wgrad_force_plan(splits, order)
            try:
                assert L.lib.w2l_wgrad_needs_zero(N, Cin, Cout, Tout, Kw) == int(splits > 1)
                L.check(L.lib.w2l_conv1d_wgrad(C.c_void_p(dyh.data_ptr() + hb * Cout * 2), drows * Cout, L.ptr(xh), rows * Cin,
                                               N * rows, L.ptr(dw), N, Cin, Cout, Tout, Kw, s, d, 0, L.stream_ptr()))
                torch.cuda.synchronize()
                got = dw.cpu().permute(1, 2, 0)
                assert relerr(got, wr.grad) < 2e-3, (splits, order, relerr(got, wr.grad))
                # (b) with the workspace: slabs + ticket, plain stores over a NaN-filled dw (nothing may survive), twice:
                # bit-identical results, tickets back at zero
                assert L.lib.w2l_wgrad_needs_zero_ws(N, Cin, Cout, Tout, Kw, ws.numel()) == 0
                outs = []
                for rep in range(2):
                    dw2 = torch.full((Kw, Cout, Cin), float('nan'), device='cuda')
                    L.check(L.lib.w2l_conv1d_wgrad_ws(C.c_void_p(dyh.data_ptr() + hb * Cout * 2), drows * Cout, L.ptr(xh),
                                                      rows * Cin, N * rows, L.ptr(dw2), N, Cin, Cout, Tout, Kw, s, d, 0, L.ptr(ws),
                                                      ws.numel(), L.stream_ptr()))
                    torch.cuda.synchronize()
                    outs.append(dw2)
                assert torch.equal(outs[0], outs[1]) and not ws[:65536].any(), (splits, order)
                assert relerr(outs[0].cpu().permute(1, 2, 0), wr.grad) < 2e-3, (splits, order)
                # the same plan and workspace on OTHER data: a slab line left in a cache by the launch before would show (the
                # slabs are stored and loaded past the non-coherent caches: sc1); dy * 2 is exact, so is the result
                dw3 = torch.full((Kw, Cout, Cin), float('nan'), device='cuda')
                L.check(L.lib.w2l_conv1d_wgrad_ws(C.c_void_p(dyh_neg.data_ptr() + hb * Cout * 2), drows * Cout, L.ptr(xh),
                                                  rows * Cin, N * rows, L.ptr(dw3), N, Cin, Cout, Tout, Kw, s, d, 0, L.ptr(ws),
                                                  ws.numel(), L.stream_ptr()))
                torch.cuda.synchronize()
                assert torch.equal(dw3, outs[0] * 2), (splits, order)
                # accumulate = 1 adds to what is there
                L.check(L.lib.w2l_conv1d_wgrad_ws(C.c_void_p(dyh.data_ptr() + hb * Cout * 2), drows * Cout, L.ptr(xh), rows * Cin,
                                                  N * rows, L.ptr(outs[1]), N, Cin, Cout, Tout, Kw, s, d, 1, L.ptr(ws), ws.numel(),
                                                  L.stream_ptr()))
                torch.cuda.synchronize()
                assert relerr(outs[1].cpu().permute(1, 2, 0), 2 * wr.grad) < 2e-3, (splits, order)
            finally:
                L.lib.w2l_wgrad_force_plan(0, -1)
    # dealt stream-K (order bit 5; the split field is the range count G): the (tile, step) space cut into G ranges whose one or
    # two segments are blocks of their own, partial tiles through slabs + ticket -- stores over a NaN-filled dw, twice:
    # bit-identical, tickets back at zero.  G = 40 / 97 / 256 against 6 .. 36 tiles x 18 steps: ranges shorter and longer than a
    # tile's third, one-step segments, whole tiles; a form without a dealt geometry here (fewer steps than ranges, more
    # tiles than ranges) must say so through w2l_wgrad_needs_zero_x and run its atomic fallback correctly
    dws = torch.zeros(max(1 << 20, int(L.lib.w2l_wgrad_dealt_workspace_bytes(Cin, Cout, Kw))), dtype=torch.uint8, device='cuda')
    dealt_ran = 0
    for G in (40, 97, 256):
        for order in (32, 33, 36, 37, 40, 41, 48, 49, 52, 53):
            L.lib.w2l_wgrad_force_plan(G, order)
            try:
                zero = L.lib.w2l_wgrad_needs_zero_x(N, Cin, Cout, Tout, Kw, s, d, ws.numel())
                assert L.lib.w2l_wgrad_needs_zero(N, Cin, Cout, Tout, Kw) == 1          # no workspace: the atomic fallback
                outs = []
                for rep in range(2):
                    dw2 = torch.zeros(Kw, Cout, Cin, device='cuda') if zero else torch.full((Kw, Cout, Cin), float('nan'), device='cuda')
                    L.check(L.lib.w2l_conv1d_wgrad_ws(C.c_void_p(dyh.data_ptr() + hb * Cout * 2), drows * Cout, L.ptr(xh),
                                                      rows * Cin, N * rows, L.ptr(dw2), N, Cin, Cout, Tout, Kw, s, d, 0, L.ptr(ws),
                                                      ws.numel(), L.stream_ptr()))
                    torch.cuda.synchronize()
                    outs.append(dw2)
                assert relerr(outs[0].cpu().permute(1, 2, 0), wr.grad) < 2e-3, (G, order, zero)
                assert not ws[:65536].any(), (G, order)
                if not zero:
                    dealt_ran += 1
                    assert torch.equal(outs[0], outs[1]), (G, order)
                    dw3 = torch.full((Kw, Cout, Cin), float('nan'), device='cuda')       # other data, same slabs: nothing stale
                    L.check(L.lib.w2l_conv1d_wgrad_ws(C.c_void_p(dyh_neg.data_ptr() + hb * Cout * 2), drows * Cout, L.ptr(xh),
                                                      rows * Cin, N * rows, L.ptr(dw3), N, Cin, Cout, Tout, Kw, s, d, 0, L.ptr(ws),
                                                      ws.numel(), L.stream_ptr()))
                    torch.cuda.synchronize()
                    assert torch.equal(dw3, outs[0] * 2), (G, order)
                    L.check(L.lib.w2l_conv1d_wgrad_ws(C.c_void_p(dyh.data_ptr() + hb * Cout * 2), drows * Cout, L.ptr(xh), rows * Cin,
                                                      N * rows, L.ptr(outs[1]), N, Cin, Cout, Tout, Kw, s, d, 1, L.ptr(ws), ws.numel(),
                                                      L.stream_ptr()))
                    torch.cuda.synchronize()
                    assert relerr(outs[1].cpu().permute(1, 2, 0), 2 * wr.grad) < 2e-3, (G, order)
                # the workspace sized for the dealt plans of this layer (what the step engine passes by default) is enough
                # for every dealt form of G <= 256 ranges of it
                if not zero and G <= 256:
                    assert L.lib.w2l_wgrad_needs_zero_x(N, Cin, Cout, Tout, Kw, s, d, dws.numel()) == 0, (G, order)
            finally:
                L.lib.w2l_wgrad_force_plan(0, -1)
    assert dealt_ran >= 12, dealt_ran
    # bit 6: a classic split adds atomically although a workspace is given (the default mode's classic plans)
    L.lib.w2l_wgrad_force_plan(3, 64 | 1)
    try:
        assert L.lib.w2l_wgrad_needs_zero_x(N, Cin, Cout, Tout, Kw, s, d, ws.numel()) == 1
        dw = torch.zeros(Kw, Cout, Cin, device='cuda')
        L.check(L.lib.w2l_conv1d_wgrad_ws(C.c_void_p(dyh.data_ptr() + hb * Cout * 2), drows * Cout, L.ptr(xh), rows * Cin, N * rows,
                                          L.ptr(dw), N, Cin, Cout, Tout, Kw, s, d, 0, L.ptr(ws), ws.numel(), L.stream_ptr()))
        torch.cuda.synchronize()
        assert relerr(dw.cpu().permute(1, 2, 0), wr.grad) < 2e-3
    finally:
        L.lib.w2l_wgrad_force_plan(0, -1)
    # stream-K decomposition (order bit 1): persistent blocks cut the (tile, step) space into equal ranges that straddle
    # tile boundaries (18 tiles x 18 steps over 20 blocks here); whole tiles are stored, pieces are added atomically
    # 6, 7: the 8-wave (two tap groups) stream-K form; 18: stream-K asked of the three-tap form -> the two-tap stream-K kernel
    for order in (2, 3) + ((6, 7, 18) if s == 1 and Kw > 2 else ()):
        L.lib.w2l_wgrad_force_plan(0, order)
        try:
            assert L.lib.w2l_wgrad_needs_zero(N, Cin, Cout, Tout, Kw) == 1
            assert L.lib.w2l_wgrad_needs_zero_ws(N, Cin, Cout, Tout, Kw, ws.numel()) == 0      # a workspace means the slab plan
            dw = torch.zeros(Kw, Cout, Cin, device='cuda')
            for rep in (1, 2):               # the second launch accumulates on top of the first
                L.check(L.lib.w2l_conv1d_wgrad(C.c_void_p(dyh.data_ptr() + hb * Cout * 2), drows * Cout, L.ptr(xh), rows * Cin,
                                               N * rows, L.ptr(dw), N, Cin, Cout, Tout, Kw, s, d, int(rep == 2), L.stream_ptr()))
                torch.cuda.synchronize()
                assert relerr(dw.cpu().permute(1, 2, 0), rep * wr.grad) < 2e-3, (order, rep)
        finally:
            L.lib.w2l_wgrad_force_plan(0, -1)


def test_wgrad_slabs_stress_bit_exact(L):
    """the slab + ticket reduction of the split weight gradients (classic split and dealt stream-K) under load: 240 launches
    back to back on a side stream, alternating between two inputs so that every launch re-writes every slab with OTHER
    data, while the main stream streams 256 MB copies through HBM and the L2s -- every result must equal, bit for bit, what
    the same plan produced alone (conv_wgrad_kernel.h: the slabs move through sc1 = agent-scope accesses only; a stale line
    would show as a wrong tile).  Also: w2l_wgrad_deterministic(1) strips plan bit 6 (atomics although a workspace is given)."""
    N, Cin, Cout, T, Kw, d = 4, 256, 384, 500, 7, 1
    g = torch.Generator().manual_seed(5)
    hb = (Kw - 1) * d
    Tout = T
    ha = max(hb, (Tout + 63) // 64 * 64 - Tout)
    xh = torch.randn(N, T + hb, Cin, generator=g).to(torch.bfloat16).cuda()
    dys = [torch.zeros(N, hb + Tout + ha, Cout), torch.zeros(N, hb + Tout + ha, Cout)]
    for t in dys:
        t[:, hb:hb + Tout] = torch.randn(N, Tout, Cout, generator=g)
    dys = [t.to(torch.bfloat16).cuda() for t in dys]
    rows, drows = xh.shape[1], dys[0].shape[1]
    ws = torch.zeros(max(int(L.lib.w2l_wgrad_workspace_bytes(Cin, Cout, Kw)), int(L.lib.w2l_wgrad_dealt_workspace_bytes(Cin, Cout, Kw))),
                     dtype=torch.uint8, device='cuda')
    side = torch.cuda.Stream()
    big_a = torch.randn(64 << 20, device='cuda')
    big_b = torch.empty_like(big_a)

    def launch(dy, dw, stream):
        L.check(L.lib.w2l_conv1d_wgrad_ws(C.c_void_p(dy.data_ptr() + hb * Cout * 2), drows * Cout, L.ptr(xh), rows * Cin, N * rows,
                                          L.ptr(dw), N, Cin, Cout, Tout, Kw, 1, d, 0, L.ptr(ws), ws.numel(), C.c_void_p(stream.cuda_stream)))

    for splits, order in ((5, 1), (7, 17), (256, 33), (256, 49)):
        L.lib.w2l_wgrad_force_plan(splits, order)
        try:
            assert L.lib.w2l_wgrad_needs_zero_x(N, Cin, Cout, Tout, Kw, 1, d, ws.numel()) == 0, (splits, order)
            want = []
            for dy in dys:
                dw = torch.full((Kw, Cout, Cin), float('nan'), device='cuda')
                launch(dy, dw, torch.cuda.current_stream())
                torch.cuda.synchronize()
                want.append(dw)
            assert not torch.equal(want[0], want[1])
            outs = [torch.full((Kw, Cout, Cin), float('nan'), device='cuda') for _ in range(60)]
            torch.cuda.synchronize()
            for i, dw in enumerate(outs):
                launch(dys[i & 1], dw, side)               # no synchronisation between launches: one workspace, one stream
                if i % 4 == 0:
                    big_b.copy_(big_a)                     # main stream: 512 MB of HBM traffic beside them
            torch.cuda.synchronize()
            for i, dw in enumerate(outs):
                assert torch.equal(dw, want[i & 1]), (splits, order, i)
            assert not ws[:65536].any(), (splits, order)
        finally:
            L.lib.w2l_wgrad_force_plan(0, -1)
    L.lib.w2l_wgrad_force_plan(3, 64 | 1)
    try:
        assert L.lib.w2l_wgrad_needs_zero_x(N, Cin, Cout, Tout, Kw, 1, d, ws.numel()) == 1
        L.lib.w2l_wgrad_deterministic(1)
        assert L.lib.w2l_wgrad_needs_zero_x(N, Cin, Cout, Tout, Kw, 1, d, ws.numel()) == 0
    finally:
        L.lib.w2l_wgrad_deterministic(0)
        L.lib.w2l_wgrad_force_plan(0, -1)


@pytest.mark.parametrize('d', [1, 2, 4])
def test_conv_wgrad_group_launch(L, d):
    """w2l_conv1d_wgrad_group: the weight gradients of several layers (different Cin / Cout / Kw, same N, Tout, dilation) in ONE
    launch whose tiles form one pool -- every block form (two taps, two tap groups, 32x32x16 fragments, three taps in AGPRs,
    six taps) and both block orders, over NaN-filled outputs, against autograd of nn.Conv1d per layer; one- and four-layer
    groups; a tile count per layer as w2l_wgrad_group_tiles says"""
    from wav2letter_pytorch_amd._lib import WgradItem
    N, T = 3, 150
    shapes = [(192, 320, 7), (128, 128, 5), (320, 192, 9), (64, 256, 1)]
    items, refs, outs, keep = [], [], [], []
    for i, (Cin, Cout, Kw) in enumerate(shapes):
        pl = pr = (Kw - 1) * d // 2
        pr += (Kw - 1) * d - pl - pr
        x, w, b = conv_inputs(N, Cin, Cout, Kw, T, pl, pr, 30 + i)
        xp = to_ntc_padded(x, pl, pr, 1)
        rows = xp.shape[1]
        Tout = rows - (Kw - 1) * d
        assert Tout == T
        dy = torch.randn(N, Cout, Tout, generator=torch.Generator().manual_seed(40 + i))
        hb = (Kw - 1) * d
        ha = max(hb, (Tout + 63) // 64 * 64 - Tout)
        dyp = to_ntc_padded(dy, hb, ha, 0, Cout)
        drows = dyp.shape[1]
        dyh, xh = dyp.to(torch.bfloat16).cuda(), xp.to(torch.bfloat16).cuda()
        wr = bf(w).requires_grad_(True)
        F.conv1d(F.pad(bf(x), (pl, pr), mode='reflect') if pl or pr else bf(x), wr, None, dilation=d).backward(bf(dy))
        dw = torch.empty(Kw, Cout, Cin, device='cuda')
        it = WgradItem(dyh.data_ptr() + hb * Cout * 2, drows * Cout, xh.data_ptr(), rows * Cin, N * rows, dw.data_ptr(), Cin, Cout, Kw, 0)
        items.append(it)
        refs.append(wr.grad)
        outs.append(dw)
        keep += [dyh, xh]
    for form in (0, 1, 4, 5, 8, 9, 16, 17, 20, 21):
        kwblk = (3 if form & 16 else 2) * (2 if form & 4 else 1)
        for sel in ([0], [0, 1, 2, 3], [3, 2], [1, 0, 2]):
            for i in sel:
                Cin, Cout, Kw = shapes[i]
                want = -(-Cout // 128) * -(-Cin // 128) * -(-Kw // (kwblk if Kw > 1 or form & 16 else 1))
                assert L.lib.w2l_wgrad_group_tiles(Cin, Cout, Kw, form) == (want if Kw > 1 else -(-Cout // 128) * -(-Cin // 128))
                outs[i].fill_(float('nan'))
            arr = (WgradItem * len(sel))(*[items[i] for i in sel])
            L.check(L.lib.w2l_conv1d_wgrad_group(arr, len(sel), N, T, d, form, L.stream_ptr()), 'w2l_conv1d_wgrad_group')
            torch.cuda.synchronize()
            for i in sel:
                got = outs[i].cpu().permute(1, 2, 0)
                assert torch.isfinite(got).all(), (form, sel, i)
                assert relerr(got, refs[i]) < 2e-3, (form, sel, i, relerr(got, refs[i]))
    # more than W2L_WGRAD_GROUP_MAX layers, or a form that is not a block form: refused with a message
    arr = (WgradItem * 9)(*([items[0]] * 9))
    assert L.lib.w2l_conv1d_wgrad_group(arr, 9, N, T, d, 1, L.stream_ptr()) != 0
    assert L.lib.w2l_conv1d_wgrad_group(arr, 1, N, T, d, 2, L.stream_ptr()) != 0


@pytest.mark.parametrize('order', [0, 4, 16, 20, 21])
def test_conv_wgrad_workspace_holds_the_largest_split(L, order):
    """w2l_wgrad_workspace_bytes must cover the slabs of EVERY block form at the largest split count the tuner tries (32):
    the six-tap form (order 20 / 21) rounds 7 taps up to 12, more than the two-, three- and four-tap forms.  The workspace is
    carved out of a larger buffer whose tail is a sentinel: nothing may be written past the reported size."""
    N, Cin, Cout, Kw, T, s, d = 8, 128, 128, 7, 1000, 1, 1           # 8 x 16 steps: 32 ranges of 4 (what the tuner tries from 128 steps)
    x, w, b = conv_inputs(N, Cin, Cout, Kw, T, 3, 3, 21)
    xp = to_ntc_padded(x, 3, 3, 1)
    rows = xp.shape[1]
    Tout = rows - (Kw - 1) * d
    dy = torch.randn(N, Cout, Tout, generator=torch.Generator().manual_seed(22))
    hb = (Kw - 1) * d
    ha = max(hb, (Tout + 63) // 64 * 64 - Tout)
    dyp = to_ntc_padded(dy, hb, ha, 0, Cout)
    drows = dyp.shape[1]
    dyh, xh = dyp.to(torch.bfloat16).cuda(), xp.to(torch.bfloat16).cuda()
    wr = bf(w).requires_grad_(True)
    F.conv1d(F.pad(bf(x), (3, 3), mode='reflect'), wr, None).backward(bf(dy))
    need = int(L.lib.w2l_wgrad_workspace_bytes(Cin, Cout, Kw))
    guard = 8 << 20
    buf = torch.zeros(need + guard, dtype=torch.uint8, device='cuda')
    buf[need:] = 0xA5
    assert N * ((Tout + 63) // 64) >= 4 * 32
    L.lib.w2l_wgrad_force_plan(32, order)
    try:
        assert L.lib.w2l_wgrad_needs_zero_ws(N, Cin, Cout, Tout, Kw, need) == 0       # the slab path, not atomics
        dw = torch.full((Kw, Cout, Cin), float('nan'), device='cuda')
        L.check(L.lib.w2l_conv1d_wgrad_ws(C.c_void_p(dyh.data_ptr() + hb * Cout * 2), drows * Cout, L.ptr(xh), rows * Cin, N * rows,
                                          L.ptr(dw), N, Cin, Cout, Tout, Kw, s, d, 0, L.ptr(buf), need, L.stream_ptr()))
        torch.cuda.synchronize()
    finally:
        L.lib.w2l_wgrad_force_plan(0, -1)
    assert relerr(dw.cpu().permute(1, 2, 0), wr.grad) < 2e-3
    assert bool((buf[need:] == 0xA5).all()), 'slabs written past the reported workspace size'


@pytest.mark.parametrize('case', [c for c in CONV_CASES if c[4] == 1])
@pytest.mark.parametrize('precise', [False, True])
def test_conv_dgrad(L, case, precise):
    """dgrad = the same kernel over zero-haloed dy with flipped/transposed weights"""
    N, Cin, Cout, Kw, s, d, T, pl, pr = case
    x, w, b = conv_inputs(N, Cin, Cout, Kw, T, pl, pr, 4)
    Tp = T + pl + pr
    Tout = Tp - (Kw - 1) * d
    g = torch.Generator().manual_seed(5)
    dy = torch.randn(N, Cout, Tout, generator=g)
    _, _, dh, dl, coutp, cinp = pack(L, w, precise)
    hb = (Kw - 1) * d
    ha = max(hb, (Tout + 63) // 64 * 64 - Tout)
    dyp = to_ntc_padded(dy, hb, ha, 0, coutp)
    rows = dyp.shape[1]
    dyh = dyp.to(torch.bfloat16).cuda()
    dyl = (dyp - dyh.float().cpu()).to(torch.bfloat16).cuda()
    dx = torch.full((N, Tp, cinp), float('nan'), dtype=torch.float32 if precise else torch.bfloat16, device='cuda')
    st = L.stream_ptr()
    args = (N, coutp, cinp, Tp, Kw, 1, d, st)
    if not precise:
        L.check(L.lib.w2l_conv1d_igemm(L.ptr(dyh), rows * coutp, N * rows, L.ptr(dh), L.ptr(dx), 0, 0, None, None, *args))
        xr = F.pad(bf(x), (pl, pr), mode='reflect').requires_grad_(True)
        F.conv1d(xr, bf(w), None, dilation=d).backward(bf(dy))
        tol = 1e-2
    else:
        L.check(L.lib.w2l_conv1d_igemm(L.ptr(dyh), rows * coutp, N * rows, L.ptr(dh), L.ptr(dx), 1, 0, None, None, *args))
        L.check(L.lib.w2l_conv1d_igemm(L.ptr(dyh), rows * coutp, N * rows, L.ptr(dl), L.ptr(dx), 1, 1, None, None, *args))
        L.check(L.lib.w2l_conv1d_igemm(L.ptr(dyl), rows * coutp, N * rows, L.ptr(dh), L.ptr(dx), 1, 1, None, None, *args))
        xr = F.pad(x, (pl, pr), mode='reflect').requires_grad_(True)
        F.conv1d(xr, w, None, dilation=d).backward(dy)
        tol = 1e-4
    torch.cuda.synchronize()
    got = dx.float().cpu()[:, :, :Cin].transpose(1, 2)
    assert relerr(got, xr.grad) < tol, relerr(got, xr.grad)


@pytest.mark.parametrize('case', CONV_CASES)
@pytest.mark.parametrize('precise', [False, True])
def test_conv_wgrad(L, case, precise):
    N, Cin, Cout, Kw, s, d, T, pl, pr = case
    x, w, b = conv_inputs(N, Cin, Cout, Kw, T, pl, pr, 6)
    xp = to_ntc_padded(x, pl, pr, 1)
    rows = xp.shape[1]
    Tout = (rows - (Kw - 1) * d - 1) // s + 1
    g = torch.Generator().manual_seed(7)
    dy = torch.randn(N, Cout, Tout, generator=g)
    coutp, cinp = (Cout + 63) // 64 * 64, (Cin + 63) // 64 * 64
    hb = (Kw - 1) * d
    ha = max(hb, (Tout + 63) // 64 * 64 - Tout)
    dyp = to_ntc_padded(dy, hb, ha, 0, coutp)
    drows = dyp.shape[1]
    dyh = dyp.to(torch.bfloat16).cuda()
    dyl = (dyp - dyh.float().cpu()).to(torch.bfloat16).cuda()
    xh = xp.to(torch.bfloat16).cuda()
    xl = (xp - xh.float().cpu()).to(torch.bfloat16).cuda()
    dw = torch.zeros(Kw, coutp, cinp, device='cuda')
    st = L.stream_ptr()

    def run(dyt, xt, acc):
        L.check(L.lib.w2l_conv1d_wgrad(C.c_void_p(dyt.data_ptr() + hb * coutp * 2), drows * coutp, L.ptr(xt), rows * cinp,
                                       N * rows, L.ptr(dw), N, cinp, coutp, Tout, Kw, s, d, acc, st))

    if not precise:
        run(dyh, xh, 0)
        wr = bf(w).requires_grad_(True)
        F.conv1d(F.pad(bf(x), (pl, pr), mode='reflect'), wr, None, stride=s, dilation=d).backward(bf(dy))
        tol = 2e-3
    else:
        run(dyh, xh, 1)
        run(dyh, xl, 1)
        run(dyl, xh, 1)
        wr = w.clone().requires_grad_(True)
        F.conv1d(F.pad(x, (pl, pr), mode='reflect'), wr, None, stride=s, dilation=d).backward(dy)
        tol = 1e-4
    torch.cuda.synchronize()
    got = dw.cpu()[:, :Cout, :Cin].permute(1, 2, 0)
    assert relerr(got, wr.grad) < tol, relerr(got, wr.grad)


def test_log_softmax_and_argmax(L):
    torch.manual_seed(8)
    N, T, Cn, CP = 3, 50, 29, 64
    logits = torch.randn(N, T, CP) * 3
    ld = logits.cuda()
    for mode in (0, 1):
        out = torch.empty(N, T, Cn, device='cuda')
        L.check(L.lib.w2l_log_softmax_fwd(L.ptr(ld), N, T, Cn, CP, mode, L.ptr(out), L.stream_ptr()))
        lr = logits[:, :, :Cn].clone().requires_grad_(True)
        ref = F.log_softmax(lr, -1) if mode == 0 else F.softmax(lr, -1)
        assert (out.cpu() - ref).abs().max() < 1e-5
        g = torch.randn(N, T, Cn)
        ref.backward(g)
        gl = torch.empty(N, T, Cn, device='cuda')
        g_d = g.cuda()
        L.check(L.lib.w2l_log_softmax_bwd(L.ptr(g_d), L.ptr(out), N, T, Cn, mode, L.ptr(gl), L.stream_ptr()))
        assert (gl.cpu() - lr.grad).abs().max() < 1e-5
    z = np.load(os.path.join(GOLD, 'greedy_cases.npz'), allow_pickle=True)
    probs = torch.from_numpy(z['probs']).cuda()
    idx = torch.empty(probs.shape[0], probs.shape[1], dtype=torch.int32, device='cuda')
    L.check(L.lib.w2l_argmax(L.ptr(probs), probs.shape[0] * probs.shape[1], probs.shape[2], L.ptr(idx), L.stream_ptr()))
    np.testing.assert_array_equal(idx.cpu().numpy(), z['argmax'])       # bit-exact, ties -> lowest index


def test_ctc_golden_cases(L):
    """loss / nll / grad against the reference's criterion (tests/golden/ctc_cases.npz): ragged input
    lengths, empty target, repeated labels, infeasible alignments under zero_infinity."""
    from wav2letter_pytorch_amd.ctc_loss import CTCLoss
    z = np.load(os.path.join(GOLD, 'ctc_cases.npz'))
    lp = torch.from_numpy(z['log_probs']).cuda().requires_grad_(True)
    crit = CTCLoss(blank=0, reduction='mean', zero_infinity=True)
    loss = crit(lp.transpose(0, 1), torch.from_numpy(z['targets']), torch.from_numpy(z['in_lens']),
                torch.from_numpy(z['target_lens']))
    loss.backward()
    assert abs(float(loss) - float(z['loss'])) < 1e-4
    g = lp.grad.cpu().numpy()
    np.testing.assert_allclose(g, z['grad'], rtol=1e-3, atol=2e-6)
    assert np.all(g[1, 33:] == 0) and np.all(g[3] == 0) and np.all(g[5] == 0)


@pytest.mark.parametrize('N,T,S', [(4, 250, 100), (2, 500, 160), (3, 64, 5), (2, 600, 250), (3, 700, 300), (2, 40, 1)])
def test_ctc_vs_oracle_random(L, N, T, S):
    from wav2letter_pytorch_amd.ctc_loss import CTCLoss
    g = torch.Generator().manual_seed(N * 1000 + T)
    lp = torch.log_softmax(torch.randn(N, T, 29, generator=g) * 2, -1)
    tl = torch.randint(max(1, S // 2), S + 1, (N,), generator=g, dtype=torch.int32)
    tg = torch.randint(1, 29, (N, S), generator=g, dtype=torch.int32)
    il = torch.randint(T - T // 4, T + 1, (N,), generator=g, dtype=torch.int32)
    lr = lp.clone().requires_grad_(True)
    ref = F.ctc_loss(lr.transpose(0, 1), tg, il, tl, blank=0, reduction='mean', zero_infinity=True)
    ref.backward()
    ld = lp.cuda().requires_grad_(True)
    loss = CTCLoss(0, 'mean', True)(ld.transpose(0, 1), tg, il, tl)
    loss.backward()
    assert abs(float(loss) - float(ref)) < 1e-4 * max(1.0, abs(float(ref)))
    np.testing.assert_allclose(ld.grad.cpu().numpy(), lr.grad.numpy(), rtol=2e-3, atol=1e-6)


def test_ctc_extreme_emissions(L):
    """the alpha / beta recursion at the edges of fp32: (a) very peaked distributions -- log-probs down to -60; (b) frames where
    EVERY label an alignment may emit has a log-prob of -100 (its probability underflows fp32: only a log-domain recursion keeps
    such an utterance finite -- a linear-domain recursion with one power-of-two scale per frame was built in round 6 and dropped
    for exactly this: states 2^-126 below a column's maximum flush to zero, and with peaked emissions they are the ones whose
    beta is large, DESIGN Appendix B); (c) in the same batch an infeasible alignment (target longer than the input) ends at
    +inf -> 0 under zero_infinity, and ordinary utterances are untouched.  Loss and gradients against torch CPU."""
    from wav2letter_pytorch_amd.ctc_loss import CTCLoss
    g = torch.Generator().manual_seed(77)
    N, T, S = 5, 120, 20
    logits = torch.randn(N, T, 29, generator=g) * 2
    logits[0] *= 12                                      # (a) peaked: log-probs of the losers around -50
    tg = torch.randint(1, 29, (N, S), generator=g, dtype=torch.int32)
    tl = torch.tensor([S, S, 12, S, 7], dtype=torch.int32)
    il = torch.tensor([T, T, 10, T, 90], dtype=torch.int32)        # utterance 2: 12 labels in 10 frames -- infeasible
    lp = torch.log_softmax(logits, -1)
    # (b) utterance 1, frames 40-42: every target label and the blank at -100 (renormalised below by a junk label at ~0)
    junk = next(c for c in range(1, 29) if c not in set(int(v) for v in tg[1]))
    for t in (40, 41, 42):
        lp[1, t, :] = -100.0
        lp[1, t, junk] = 0.0
    lr = lp.clone().requires_grad_(True)
    ref = F.ctc_loss(lr.transpose(0, 1), tg, il, tl, blank=0, reduction='mean', zero_infinity=True)
    ref.backward()
    per = F.ctc_loss(lp.transpose(0, 1), tg, il, tl, blank=0, reduction='none', zero_infinity=False)
    assert torch.isinf(per[2]) and float(per[1]) > 250 and torch.isfinite(per[[0, 1, 3, 4]]).all()
    ld = lp.cuda().requires_grad_(True)
    loss = CTCLoss(0, 'mean', True)(ld.transpose(0, 1), tg, il, tl)
    loss.backward()
    assert abs(float(loss) - float(ref)) < 1e-4 * max(1.0, abs(float(ref))), (float(loss), float(ref))
    got, want = ld.grad.cpu().numpy(), lr.grad.numpy()
    assert np.isfinite(got).all()
    np.testing.assert_allclose(got, want, rtol=2e-3, atol=2e-6)


def _bnact_desc(L, N, T, C, y, scale, shift, mean, invstd, act, p=0.0, mask=None, lens=None, y2=None, bn2=None):
    d = L.BnActDesc()
    d.N, d.T, d.C = N, T, C
    d.y = y.data_ptr()
    d.y_f32 = int(y.dtype == torch.float32)
    d.scale = scale.data_ptr() if scale is not None else None
    d.shift = shift.data_ptr() if shift is not None else None
    d.mean = mean.data_ptr() if mean is not None else None
    d.invstd = invstd.data_ptr() if invstd is not None else None
    if y2 is not None:
        d.y2 = y2.data_ptr()
        d.scale2, d.shift2, d.mean2, d.invstd2 = (t.data_ptr() for t in bn2)
    d.act = act
    d.drop_p = p
    d.seed, d.offset = 1234, 1
    d.mask = mask.data_ptr() if mask is not None else None
    d.lens = lens.data_ptr() if lens is not None else None
    return d


@pytest.mark.parametrize('N,T,C,pl,pr,mode,act,f32', [
    (2, 150, 128, 12, 12, 1, 1, True), (2, 60, 128, 6, 6, 1, 1, True), (3, 100, 64, 4, 5, 1, 1, False),
    (2, 75, 192, 28, 28, 1, 1, True), (2, 90, 64, 3, 3, 0, 2, True), (1, 33, 640, 0, 0, 1, 0, True)])
def test_bn_act_fwd_bwd(L, N, T, C, pl, pr, mode, act, f32):
    """BN(train) + activation forward into a padded buffer, and its backward from a gradient given in
    the padded coordinates (reflect fold), against torch autograd on the CPU."""
    g = torch.Generator().manual_seed(N * 100 + T)
    y = torch.randn(N, C, T, generator=g) * 3 + 1
    if not f32:
        y = bf(y)
    gamma = torch.rand(C, generator=g) + 0.5
    beta = torch.randn(C, generator=g)
    yr = y.clone().requires_grad_(True)
    z = F.batch_norm(yr, None, None, gamma, beta, True, 0.9, 1e-3)
    a = z.clamp(0, 20) if act == 1 else (F.relu(z) if act == 2 else z)
    ap = F.pad(a, (pl, pr), mode='reflect' if mode == 1 else 'constant') if (pl or pr) else a
    gp = torch.randn(ap.shape, generator=g)
    ap.backward(gp)
    # device side
    yd = y.transpose(1, 2).contiguous().cuda()
    if not f32:
        yd = yd.to(torch.bfloat16)
    s1 = yd.float().sum((0, 1))
    s2 = (yd.float() ** 2).sum((0, 1))
    part = torch.stack([s1, s2]).unsqueeze(0).contiguous()
    scale, shift, mean, invstd = (torch.empty(C, device='cuda') for _ in range(4))
    rm, rv = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda')
    st = L.stream_ptr()
    gam_d, bet_d = gamma.cuda(), beta.cuda()
    L.check(L.lib.w2l_bn_finalize(L.ptr(part), 1, C, N * T, L.ptr(gam_d), L.ptr(bet_d), 1e-3, 0.9, L.ptr(rm),
                                  L.ptr(rv), L.ptr(mean), L.ptr(invstd), L.ptr(scale), L.ptr(shift), st))
    mref = y.mean((0, 2))
    vref = y.var((0, 2), unbiased=True)
    assert (rm.cpu() - 0.9 * mref).abs().max() < 1e-4 and relerr(rv.cpu(), 0.1 + 0.9 * vref) < 1e-4
    d = _bnact_desc(L, N, T, C, yd, scale, shift, mean, invstd, act)
    R = pl + T + pr
    out_hi = torch.empty(N, R, C, dtype=torch.bfloat16, device='cuda')
    out_lo = torch.empty_like(out_hi)
    L.check(L.lib.w2l_bn_act_fwd(C_.byref(d), L.ptr(out_hi), L.ptr(out_lo), R, pl, pr, mode, st))
    got = (out_hi.float() + out_lo.float()).cpu().transpose(1, 2)
    assert relerr(got, ap.detach()) < 2e-5
    # backward
    gd = gp.transpose(1, 2).contiguous().cuda()
    gs = L.GradSrc()
    gs.dxp, gs.f32, gs.pad_l, gs.pad_r, gs.pad_mode, gs.rows = gd.data_ptr(), 1, pl, pr, mode, pl + T + pr
    nb = L.lib.w2l_bn_bwd_blocks(N, T, C)
    partial = torch.empty(nb, 2, C, device='cuda')
    L.check(L.lib.w2l_bn_act_bwd_reduce(C_.byref(d), C_.byref(gs), None, L.ptr(partial), st))
    sums = torch.empty(4, C, device='cuda')
    L.check(L.lib.w2l_bn_bwd_finalize(L.ptr(partial), nb, C, 2, L.ptr(sums), st))
    h = 13                                            # shared-halo layout: h + N*(T+h) rows
    dy_hi = torch.full((h + N * (T + h), C), float('nan'), dtype=torch.bfloat16, device='cuda')
    dy_lo = torch.full_like(dy_hi, float('nan'))
    L.check(L.lib.w2l_bn_act_bwd_apply(C_.byref(d), C_.byref(gs), None, L.ptr(sums), L.ptr(dy_hi), L.ptr(dy_lo), h, None,
                                       None, 0, st))
    # the same in one launch (finalize folded into the dy kernel, slab form), NaN-prefilled: every row incl. the halos
    # must be written; sums equal up to fp32 summation order; amax = max |dy| left in device memory
    sums2 = torch.full((4, C), float('nan'), device='cuda')
    dy2_hi = torch.full_like(dy_hi, float('nan'))
    dy2_lo = torch.full_like(dy_hi, float('nan'))
    amax = torch.zeros(2, 64, device='cuda')                 # [dy, dy2][W2L_AMAX_SLOTS]
    L.check(L.lib.w2l_bn_act_bwd_apply_fin(C_.byref(d), C_.byref(gs), None, L.ptr(partial), nb, L.ptr(sums2), L.ptr(dy2_hi),
                                           L.ptr(dy2_lo), h, None, None, 0, L.ptr(amax), st))
    torch.cuda.synchronize()
    assert relerr(sums2[:2].cpu(), sums[:2].cpu()) < 1e-5
    a_, b_ = dy_hi.float() + dy_lo.float(), dy2_hi.float() + dy2_lo.float()
    assert torch.isfinite(b_).all() and relerr(b_.cpu(), a_.cpu()) < 1e-5
    assert abs(float(amax[0].max()) - float(b_.abs().max())) <= 1e-4 * float(amax[0].max()) and not amax[1].any()
    # d beta / d gamma via autograd on the same graph
    gam = gamma.clone().requires_grad_(True)
    bet = beta.clone().requires_grad_(True)
    z2 = F.batch_norm(y, None, None, gam, bet, True, 0.9, 1e-3)
    a2 = z2.clamp(0, 20) if act == 1 else (F.relu(z2) if act == 2 else z2)
    (F.pad(a2, (pl, pr), mode='reflect' if mode == 1 else 'constant') if (pl or pr) else a2).backward(gp)
    assert relerr(sums[0].cpu(), bet.grad) < 1e-4, relerr(sums[0].cpu(), bet.grad)
    assert relerr(sums[1].cpu(), gam.grad) < 1e-4
    dy = (dy_hi.float() + dy_lo.float()).cpu()
    assert (dy[:h] == 0).all()
    dyu = dy[h:].view(N, T + h, C)
    assert (dyu[:, T:] == 0).all()
    assert relerr(dyu[:, :T].transpose(1, 2), yr.grad) < 1e-4, relerr(dyu[:, :T].transpose(1, 2), yr.grad)


@pytest.mark.parametrize('N,T,C,pl,pr,mode,act,p,lens,h', [
    (3, 150, 128, 12, 12, 1, 1, 0.0, False, 24), (2, 333, 192, 4, 5, 1, 1, 0.3, False, 51), (2, 90, 64, 3, 3, 0, 2, 0.0, True, 38),
    (4, 500, 896, 28, 28, 1, 1, 0.4, False, 56), (1, 40, 64, 7, 7, 1, 0, 0.2, True, 24),
    # 12 000 rows and 8 M elements on: the LOOPED kernels (a wave walks four row groups, the next one's loads in flight) -- utterance
    # boundaries inside a wave's rows (T = 500 and 997 are no multiples of 8 x 4), reflect folds, length masks, ReLU / clamp / none
    (26, 500, 640, 20, 20, 1, 1, 0.2, False, 40), (13, 997, 640, 6, 6, 0, 2, 0.0, True, 12), (25, 500, 704, 28, 28, 1, 0, 0.3, True, 56)])
def test_bn_bwd_two_launch_chain_equals_three(L, N, T, C, pl, pr, mode, act, p, lens, h):
    """the backward chain's fast path (w2l_bn_act_bwd_reduce_slots + w2l_bn_act_bwd_apply_slots: sums added onto 8 slot rows,
    the finalize folded into the dy pass) against reduce + finalize + apply: the same sums (up to the order of the fp32
    additions), the same dy incl. the zero halo rows of its shared-halo buffer, the same amax; reflect folds, dropout masks
    and length masks included"""
    g = torch.Generator().manual_seed(N * 100 + T + C)
    st = L.stream_ptr()
    R = pl + T + pr
    y = (torch.randn(N, T, C, generator=g) * 3 + 1).to(torch.bfloat16).cuda()
    scale, shift, mean, invstd = ((torch.rand(C, generator=g) + 0.5).cuda() for _ in range(4))
    mask = (torch.randint(0, 256, (N * T * (C // 8),), generator=g, dtype=torch.int32).to(torch.uint8)).cuda() if p > 0 else None
    lens_d = torch.tensor([T - 3 * i for i in range(N)], dtype=torch.int32).cuda() if lens else None
    gp = torch.randn(N, R, C, generator=g).to(torch.bfloat16).cuda()
    d = _bnact_desc(L, N, T, C, y, scale, shift, mean, invstd, act, p=p, mask=mask, lens=lens_d)
    gs = L.GradSrc()
    gs.dxp, gs.f32, gs.pad_l, gs.pad_r, gs.pad_mode, gs.rows = gp.data_ptr(), 0, pl, pr, mode, R
    assert L.lib.w2l_bn_bwd_fast_ok(C_.byref(d), C_.byref(gs), None) == 1
    # three launches
    nb = L.lib.w2l_bn_bwd_blocks(N, T, C)
    part = torch.empty(nb, 2, C, device='cuda')
    sums_a = torch.zeros(4, C, device='cuda')
    dy_a = torch.full((h + N * (T + h), C), float('nan'), dtype=torch.bfloat16, device='cuda')
    amax_a = torch.zeros(2, 64, device='cuda')
    L.check(L.lib.w2l_bn_act_bwd_reduce(C_.byref(d), C_.byref(gs), None, L.ptr(part), st))
    L.check(L.lib.w2l_bn_bwd_finalize(L.ptr(part), nb, C, 2, L.ptr(sums_a), st))
    L.check(L.lib.w2l_bn_act_bwd_apply_amax(C_.byref(d), C_.byref(gs), None, L.ptr(sums_a), L.ptr(dy_a), None, h, None, None, 0,
                                            L.ptr(amax_a), st))
    # two launches
    slots = torch.zeros(8, 2, C, device='cuda')
    sums_b = torch.zeros(4, C, device='cuda')
    dy_b = torch.full((h + N * (T + h), C), float('nan'), dtype=torch.bfloat16, device='cuda')
    amax_b = torch.zeros(2, 64, device='cuda')
    L.check(L.lib.w2l_bn_act_bwd_reduce_slots(C_.byref(d), C_.byref(gs), L.ptr(slots), 8, st), 'reduce_slots')
    L.check(L.lib.w2l_bn_act_bwd_apply_slots(C_.byref(d), C_.byref(gs), L.ptr(slots), 8, L.ptr(sums_b), L.ptr(dy_b), h, L.ptr(amax_b),
                                             st), 'apply_slots')
    torch.cuda.synchronize()
    assert relerr(slots.sum(0), sums_a[:2]) < 2e-5 and relerr(sums_b[:2], sums_a[:2]) < 2e-5
    a, b = dy_a.float(), dy_b.float()
    assert torch.isfinite(b).all()
    assert (a - b).abs().max() <= 1.2e-2 * a.abs().max() and relerr(b, a) < 2e-3        # (a bf16 ulp where the sums' last bits differ)
    zr = torch.ones(h + N * (T + h), dtype=torch.bool)
    zr[h:].view(N, T + h)[:, :T] = False
    assert not dy_b[zr.cuda()].any()
    assert abs(float(amax_a[0].max()) - float(amax_b[0].max())) <= 1.2e-2 * float(amax_a[0].max())
    # a unit outside the fast path says so
    d2 = _bnact_desc(L, N, T, C, y.float(), scale, shift, mean, invstd, act)
    assert L.lib.w2l_bn_bwd_fast_ok(C_.byref(d2), C_.byref(gs), None) == 0
    assert L.lib.w2l_bn_bwd_fast_ok(C_.byref(d), C_.byref(gs), C_.byref(gs)) == 0


@pytest.mark.parametrize('N,T,C,pl,pr,mode,act,p,res,lens,q', [
    (3, 150, 128, 12, 12, 1, 1, 0.0, False, False, False), (2, 333, 192, 4, 5, 1, 1, 0.3, False, False, False),
    (2, 90, 64, 3, 3, 0, 2, 0.0, True, True, False), (4, 500, 896, 28, 28, 1, 1, 0.4, False, False, True),
    (2, 75, 256, 0, 0, 0, 2, 0.2, True, False, True), (1, 40, 64, 7, 7, 1, 0, 0.0, False, True, False)])
def test_bn_act_fwd_fin_equals_finalize_plus_apply(L, N, T, C, pl, pr, mode, act, p, res, lens, q):
    """w2l_bn_act_fwd_fin (the statistics finalize folded into the apply pass: every block re-reduces the few partial rows of
    its 64 channels) against w2l_bn_finalize + w2l_bn_act_fwd_q on the same partial rows: bit-identical padded activation,
    dropout mask and e4m3 copy, the same mean / invstd / scale / shift and running statistics -- with dropout, a residual
    branch with its own BatchNorm, length masks, reflect and zero padding, 1 .. 8 partial rows"""
    g = torch.Generator().manual_seed(N * 100 + T + C)
    st = L.stream_ptr()
    R = pl + T + pr

    def branch(rows):
        y = (torch.randn(N, T, C, generator=g) * 3 + 1).to(torch.bfloat16).cuda()
        # partial rows as the convolutions leave them under w2l_conv_stats_mode(rows): the tile sums folded onto `rows` rows
        yf = y.float().reshape(-1, C)
        part = torch.zeros(rows, 2, C, device='cuda')
        for r in range(rows):
            part[r, 0] = yf[r::rows].sum(0)
            part[r, 1] = (yf[r::rows] ** 2).sum(0)
        gam, bet = (torch.rand(C, generator=g) + 0.5).cuda(), torch.randn(C, generator=g).cuda()
        return y, part, gam, bet

    y1, part1, gam1, bet1 = branch(8 if C > 64 else 3)
    b2 = branch(1 if C > 64 else 5) if res else None
    lens_d = torch.tensor([T - 3 * i for i in range(N)], dtype=torch.int32).cuda() if lens else None

    def run(folded):
        vec = lambda: [torch.empty(C, device='cuda') for _ in range(4)]       # noqa: E731
        s1, s2 = vec(), vec()
        rm1, rv1 = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda')
        rm2, rv2 = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda')
        mask = torch.zeros(N * T * (C // 8), dtype=torch.uint8, device='cuda') if p > 0 else None
        out = torch.full((N, R, C), float('nan'), dtype=torch.bfloat16, device='cuda')
        outq = torch.full((N, R, C), 77, dtype=torch.uint8, device='cuda') if q else None
        clipped = torch.zeros(1, dtype=torch.int64, device='cuda')
        if not folded:
            L.check(L.lib.w2l_bn_finalize(L.ptr(part1), part1.shape[0], C, N * T, L.ptr(gam1), L.ptr(bet1), 1e-3, 0.9, L.ptr(rm1),
                                          L.ptr(rv1), L.ptr(s1[2]), L.ptr(s1[3]), L.ptr(s1[0]), L.ptr(s1[1]), st))
            if res:
                L.check(L.lib.w2l_bn_finalize(L.ptr(b2[1]), b2[1].shape[0], C, N * T, L.ptr(b2[2]), L.ptr(b2[3]), 1e-3, 0.1,
                                              L.ptr(rm2), L.ptr(rv2), L.ptr(s2[2]), L.ptr(s2[3]), L.ptr(s2[0]), L.ptr(s2[1]), st))
        d = _bnact_desc(L, N, T, C, y1, s1[0], s1[1], s1[2], s1[3], act, p=p, mask=mask, lens=lens_d,
                        y2=b2[0] if res else None, bn2=s2 if res else None)
        if q:
            d.q_clipped = clipped.data_ptr()
        if folded:
            def rec(part, gam, bet, mom, rm, rv, s):
                f = L.BnFin()
                f.partial, f.rows, f.count = part.data_ptr(), part.shape[0], N * T
                f.gamma, f.beta, f.eps, f.momentum = gam.data_ptr(), bet.data_ptr(), 1e-3, mom
                f.running_mean, f.running_var = rm.data_ptr(), rv.data_ptr()
                f.scale, f.shift, f.mean, f.invstd = (t.data_ptr() for t in s)
                return f
            f1 = rec(part1, gam1, bet1, 0.9, rm1, rv1, s1)
            f2 = rec(b2[1], b2[2], b2[3], 0.1, rm2, rv2, s2) if res else None
            d.scale = d.shift = d.scale2 = d.shift2 = None            # (ignored for a branch with a record)
            L.check(L.lib.w2l_bn_act_fwd_fin(C_.byref(d), C_.byref(f1), C_.byref(f2) if res else None, L.ptr(out), L.ptr(outq),
                                             4.0, R, pl, pr, mode, st), 'w2l_bn_act_fwd_fin')
        else:
            L.check(L.lib.w2l_bn_act_fwd_q(C_.byref(d), L.ptr(out), None, L.ptr(outq), 4.0, R, pl, pr, mode, st))
        torch.cuda.synchronize()
        return out, outq, mask, s1, s2, (rm1, rv1, rm2, rv2), clipped

    a, b = run(False), run(True)
    assert torch.equal(a[0].view(torch.int16), b[0].view(torch.int16))
    assert not torch.isnan(b[0].float()).any()
    if q:
        assert torch.equal(a[1], b[1]) and int(a[6]) == int(b[6])
    if p > 0:
        assert torch.equal(a[2], b[2]) and 0.5 * (1 - p) < float((a[2] != 0).float().mean()) <= 1.0
    for u, v in zip(a[3] + (a[4] if res else []) + list(a[5][: 4 if res else 2]), b[3] + (b[4] if res else []) + list(b[5][: 4 if res else 2])):
        assert torch.equal(u, v)


@pytest.mark.parametrize('N,T,S', [(2, 2500, 700), (2, 8000, 2600), (1, 8000, 4000)])
def test_ctc_long_transcripts(L, N, T, S):
    """transcripts beyond 511 labels (1024-thread blocks, 2-8 states per thread) and frames beyond LDS capacity:
    the T' = 8 000 utterances of BASELINE config 5.  Loss 1e-4 relative, gradient 1e-3 of its scale, vs torch CPU CTC."""
    g = torch.Generator().manual_seed(N * 1000 + S)
    lp = torch.randn(N, T, 29, generator=g).log_softmax(-1)
    tl = torch.tensor([S] + [max(1, S - 37 * (i + 1)) for i in range(N - 1)], dtype=torch.int32)
    tg = torch.randint(1, 29, (N, S), generator=g, dtype=torch.int32)
    il = torch.tensor([T] + [T - 13 * (i + 1) for i in range(N - 1)], dtype=torch.int32)
    def cpu(dtype):
        r = lp.detach().clone().to(dtype).requires_grad_(True)
        v = torch.nn.functional.ctc_loss(r.transpose(0, 1), tg, il, tl, blank=0, reduction='mean', zero_infinity=True)
        v.backward()
        return float(v), r.grad

    ref, gref = cpu(torch.float32)           # what the reference executes
    ref64, g64 = cpu(torch.float64)
    from wav2letter_pytorch_amd.ctc_loss import CTCLoss
    dlp = lp.cuda().requires_grad_(True)
    loss = CTCLoss(blank=0, reduction='mean', zero_infinity=True)(dlp.transpose(0, 1), tg, il, tl)
    loss.backward()
    assert abs(float(loss) - ref) <= 1e-4 * abs(ref)
    # alpha/beta reach ~ -T * 3.5 nats: one fp32 ulp there is ~1e-3 of a posterior, so two correct fp32 evaluations
    # differ by a few 1e-3 of the gradient's scale on these lengths.  Judge both against float64: the device must be no
    # further from it than 1e-3 of scale or 1.5x the reference's own fp32 error, whichever is larger.
    scale = float(g64.abs().max())
    err_dev = float((dlp.grad.cpu().double() - g64).abs().max())
    err_ref = float((gref.double() - g64).abs().max())
    assert err_dev <= max(1e-3 * scale, 1.5 * err_ref), (err_dev / scale, err_ref / scale)
