#!/usr/bin/env python
"""Runs the SpMM a few times on ml1m(s) for profiling under rocprofv3 (development aid)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch


def main():
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    F = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    kind = sys.argv[4] if len(sys.argv) > 4 else 'csr'
    from deep_cbrs_amar_renaissance_amd import capi
    from deep_cbrs_amar_renaissance_amd.data import synthetic
    from deep_cbrs_amar_renaissance_amd.utilities.math import gcn_filter_device
    capi.load()
    dev = torch.device('cuda')
    data = synthetic.ml1m_device(scale, device=dev)
    n = data['n_users'] + data['n_items']
    a = gcn_filter_device(data['train_pos'][:, 0], data['train_pos'][:, 1], n)
    x = torch.randn((n, F), device=dev)
    y = torch.empty((n, F), device=dev)
    torch.cuda.synchronize()
    if kind == 'zero':
        a.colidx.zero_()
    sj = a.sliced(F) if kind == 'sj' else None
    xs = a.xcd_sliced() if kind == 'xs' else None
    lt = a.lds_tiled(F) if kind == 'lt' else None                    # AMAR_LT_WINDOW / AMAR_LT_VARIANT apply
    torch.cuda.synchronize()
    for _ in range(reps):
        if kind == 'sj':
            capi.spmm_sj(sj, x, y)
        elif kind == 'xs':
            capi.spmm_xs(xs, x, y, prescaled=xs.row_scale is not None)
        elif kind == 'lt':
            capi.spmm_lt(lt, x, y, prescaled=True)
        else:
            capi.spmm_csr(a.rowptr, a.colidx, a.vals, x, y)
    torch.cuda.synchronize()
    print('done', a.nnz)


if __name__ == '__main__':
    main()
