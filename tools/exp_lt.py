#!/usr/bin/env python
"""LDS-tiled SpMM (amar_spmm_lt_f32) against the XCD-sliced form on ml1m(s): parity and time per launch, by window size
and kernel variant (development aid).  Variants are selected per process (AMAR_LT_VARIANT is read once by the library):
run as `python tools/exp_lt.py <scale> <F> <variant> [window ...]`."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch


def timeit(fn, reps=20, warm=3):
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
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    F = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    variants = (sys.argv[3] if len(sys.argv) > 3 else '0').split(',')
    windows = [int(w) for w in sys.argv[4:]] or [0]
    from deep_cbrs_amar_renaissance_amd import capi
    from deep_cbrs_amar_renaissance_amd.data import synthetic
    from deep_cbrs_amar_renaissance_amd.utilities import lds_tiled
    from deep_cbrs_amar_renaissance_amd.utilities.math import gcn_filter_device, _unit_entries
    if os.environ.get('LT_LIB'):                                 # same-box A/B of another build of the library
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
    t_xs = timeit(lambda: capi.spmm_xs(xs, xs_tab, y_xs, prescaled=True))
    y_csr = torch.empty_like(y_xs)
    t_csr = timeit(lambda: capi.spmm_csr(a.rowptr, a.colidx, a.vals, x, y_csr))
    print('scale %d F %d: N %d, nnz %d; XS %.4f ms; CSR row-stream kernel %.4f ms' % (scale, F, n, a.nnz, t_xs, t_csr), flush=True)
    rows, cols, diag, off = _unit_entries(a, True)
    col_scale = a.dinv.to(torch.float32).contiguous()
    for w in windows:
        t0 = time.perf_counter()
        lt = lds_tiled.LdsTiled.build(rows, cols, n, n, F, diag, col_scale, col_scale, off, window_entries=w or None,
                                      split=int(os.environ.get('LT_SPLIT', lds_tiled.SPLIT)), row_breaks=getattr(a, 'row_breaks', ()))
        torch.cuda.synchronize()
        t_build = time.perf_counter() - t0
        for variant in variants:
            os.environ['AMAR_LT_VARIANT'] = variant
            y_lt.zero_()
            capi.spmm_lt(lt, xs_tab, y_lt, prescaled=True)
            torch.cuda.synchronize()
            capi.spmm_xs(xs, xs_tab, y_xs, prescaled=True)
            err = float((y_lt - y_xs).abs().max())
            rel = err / float(y_xs.abs().max())
            y2 = torch.empty_like(y_lt)
            capi.spmm_lt(lt, xs_tab, y2, prescaled=True)
            same = bool(torch.equal(y2, y_lt))
            t_lt = timeit(lambda: capi.spmm_lt(lt, xs_tab, y_lt, prescaled=True))
            rows_t = (lt.tile_row0[1:] - lt.tile_row0[:-1])
            print('  variant %s window %5d: LT %.4f ms (%.2fx XS)  max|diff| %.2e (rel %.1e) reproducible %s | tiles %d (rows %d..%d) '
                  'vrows <= %d windows/tile %d pairs %.3f%% flagged %.3f%% pad %.2f%% build %.1f s' %
                  (variant, lt.window_entries, t_lt, t_xs / t_lt, err, rel, same, lt.n_tiles, int(rows_t.min()), int(rows_t.max()),
                   int(lt.vcount.max()), lt.maxwin1 - 1, 100.0 * lt.n_pairs / max(1, lt.n_entries), 100.0 * lt.n_flagged / max(1, lt.n_entries),
                   100.0 * (lt.words.numel() - lt.n_entries) / max(1, lt.n_entries), t_build), flush=True)
        os.environ.pop('AMAR_LT_VARIANT', None)
        # the call as the fused GCN chain makes it: bias + ReLU, Y into a column slice of the [N, 3F] concat buffer, next layer's X.W
        cat = torch.empty((n, 3 * F), device=dev)
        wn = torch.randn((F, F), device=dev) * 0.3
        hn = torch.empty((n, F), device=dev)
        # cold caches: 600 MB written between launches (what the pair stage does to the layer inside a bench step)
        scratch = torch.empty(150_000_000, device=dev)
        colds = []
        for _ in range(8):
            scratch.fill_(1.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); capi.spmm_lt(lt, xs_tab, y_lt, prescaled=True); e1.record(); e1.synchronize()
            colds.append(e0.elapsed_time(e1))
        print('  from cold caches: LT %.4f ms (median of 8)' % sorted(colds)[4], flush=True)
        del scratch
        t_a = timeit(lambda: capi.spmm_lt(lt, xs_tab, y_lt, prescaled=True))
        t_b = timeit(lambda: capi.spmm_lt(lt, xs_tab, y_lt, bias=bias, relu=True, prescaled=True))
        t_c = timeit(lambda: capi.spmm_lt(lt, xs_tab, cat[:, F:2 * F], bias=bias, relu=True, prescaled=True))
        t_d = timeit(lambda: capi.spmm_lt(lt, xs_tab, cat[:, F:2 * F], bias=bias, relu=True, Wnext=wn, Hnext=hn, prescaled=True, scale_next=True))
        print('  epilogue forms: plain %.4f | + bias, ReLU %.4f | ... into a concat slice %.4f | ... + next X.W %.4f ms' % (t_a, t_b, t_c, t_d), flush=True)
        del lt


if __name__ == '__main__':
    main()
