#!/usr/bin/env python
"""Time the XS SpMM with only ONE column slice populated (development aid): are some XCDs' slices slower per entry?"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tools.exp_xs_floor import timeit


def main():
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    F = 8
    from deep_cbrs_amar_renaissance_amd import capi
    from deep_cbrs_amar_renaissance_amd.data import synthetic
    from deep_cbrs_amar_renaissance_amd.utilities.math import gcn_filter_device
    capi.load()
    dev = torch.device('cuda')
    data = synthetic.ml1m_device(scale, device=dev)
    n = data['n_users'] + data['n_items']
    a = gcn_filter_device(data['train_pos'][:, 0], data['train_pos'][:, 1], n)
    xs = a.xcd_sliced()
    x = torch.randn((n, F), device=dev)
    y = torch.empty((n, F), device=dev)
    rp = xs.rowptr.clone()
    print('all slices: %.3f ms; bounds %s' % (timeit(lambda: capi.spmm_xs(xs, x, y)), xs.bounds), flush=True)
    for k in range(xs.n_slices):
        r = rp.clone()
        lo, hi = k * n, (k + 1) * n
        r[:lo] = rp[lo]                       # everything before slice k: empty ranges ending where k starts
        r[hi + 1:] = rp[hi]                   # everything after: empty
        xs.rowptr = r
        t = timeit(lambda: capi.spmm_xs(xs, x, y))
        print('slice %d only: %.3f ms  (%d entries, %d columns)' % (k, t, int(rp[hi] - rp[lo]), xs.bounds[k + 1] - xs.bounds[k]), flush=True)
    xs.rowptr = rp


if __name__ == '__main__':
    main()
