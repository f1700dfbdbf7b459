#!/usr/bin/env python
"""Round 4: the F = 8 LT layer under image layouts / kernel forms (development aid).  One process = one build of the library
(LT_LIB=tools/libamar_hip_old.so for the round-3 build).  `python tools/exp_lt8.py <scale> <F> <config> [<config> ...]` with
config = layout:window:sub:pairs:pace:spread[:variant] (variant = AMAR_LT_VARIANT of the launch: development ablations), e.g. deal:0:0:1:0:0 (the round-3 image), deal:0:0:0:0:3 (no pairs, repeats spread), defer:2048:512:0:0:0.
Prints per config: build statistics, plain product and fused layer (bias + ReLU + concat slice + next X.W) ms per launch,
max |diff| against the XCD-sliced form and bitwise reproducibility."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch


def timeit(fn, reps=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    scale, F = int(sys.argv[1]), int(sys.argv[2])
    configs = sys.argv[3:] or ['deal:0:0:1:0:0']
    from deep_cbrs_amar_renaissance_amd import capi
    from deep_cbrs_amar_renaissance_amd.data import synthetic
    from deep_cbrs_amar_renaissance_amd.utilities import lds_tiled
    from deep_cbrs_amar_renaissance_amd.utilities.math import gcn_filter_device, _unit_entries
    if os.environ.get('LT_LIB'):
        capi.LIB_PATH = os.path.join(ROOT, os.environ['LT_LIB'])
    capi.load()
    dev = torch.device('cuda')
    data = synthetic.ml1m_device(scale, device=dev)
    n = data['n_users'] + data['n_items']
    a = gcn_filter_device(data['train_pos'][:, 0], data['train_pos'][:, 1], n)
    xs = a.xcd_sliced()
    x = torch.randn((n, F), device=dev)
    xs_tab = torch.empty_like(x)
    capi.row_affine(x, xs.col_scale, xs_tab)
    bias = torch.randn(F, device=dev) * 0.1
    y_xs, y_lt = torch.empty((n, F), device=dev), torch.empty((n, F), device=dev)
    capi.spmm_xs(xs, xs_tab, y_xs, prescaled=True)
    print('lib %s scale %d F %d: N %d nnz %d' % (os.environ.get('LT_LIB', 'current'), scale, F, n, a.nnz), flush=True)
    rows, cols, diag, off = _unit_entries(a, True)
    col_scale = a.dinv.to(torch.float32).contiguous()
    cat = torch.empty((n, 3 * F), device=dev)
    wn = torch.randn((F, F), device=dev) * 0.3
    hn = torch.empty((n, F), device=dev)
    for cfg in configs:
        layout, window, sub, pairs, pace, spread, variant = (cfg.split(':') + ['0', '0'])[:7]
        os.environ.pop('AMAR_LT_VARIANT', None)
        kw = {}
        if int(spread):
            kw['spread'] = int(spread)
        if layout.endswith('+c'):
            layout, kw['colsort'] = layout[:-2], True
        if layout != 'deal':
            kw.update(layout=layout, sub_window=int(sub) or None)
        t0 = time.perf_counter()
        lt = lds_tiled.LdsTiled.build(rows, cols, n, n, F, diag, col_scale, col_scale, off, window_entries=int(window) or None,
                                      row_breaks=getattr(a, 'row_breaks', ()), pairs=bool(int(pairs)), **kw)
        if int(pace):
            lt.pace_every = int(pace)
        if os.environ.get('EXP_NTILES'):                         # only the first tiles of the image (half the CUs idle: how much of a tile's time is the shared L2?)
            lt.n_tiles = min(lt.n_tiles, int(os.environ['EXP_NTILES']))
        torch.cuda.synchronize()
        t_build = time.perf_counter() - t0
        if int(variant):
            os.environ['AMAR_LT_VARIANT'] = variant
        y_lt.zero_()
        capi.spmm_lt(lt, xs_tab, y_lt, prescaled=True)
        torch.cuda.synchronize()
        err = float((y_lt - y_xs).abs().max())
        y2 = torch.empty_like(y_lt)
        capi.spmm_lt(lt, xs_tab, y2, prescaled=True)
        same = bool(torch.equal(y2, y_lt))
        t_a = timeit(lambda: capi.spmm_lt(lt, xs_tab, y_lt, prescaled=True))
        t_d = float('nan') if int(variant) else timeit(lambda: capi.spmm_lt(lt, xs_tab, cat[:, F:2 * F], bias=bias, relu=True, Wnext=wn, Hnext=hn, prescaled=True, scale_next=True))
        os.environ.pop('AMAR_LT_VARIANT', None)
        print('  %-22s window %5d pace %d: plain %.4f ms, fused layer %.4f ms | max|diff| %.2e reproducible %s | tiles %d pairs %.3f%% '
              'flagged %.3f%% build %.1f s' % (cfg, lt.window_entries, lt.pace_every, t_a, t_d, err, same, lt.n_tiles,
                                               100.0 * lt.n_pairs / max(1, lt.n_entries), 100.0 * lt.n_flagged / max(1, lt.n_entries), t_build), flush=True)
        del lt


if __name__ == '__main__':
    main()
