#!/usr/bin/env python
"""Hypothesis test: column-sliced SpMM (each slice of X fits the 4 MB per-XCD L2), run as S sequential
accumulating launches of the production kernel. Upper bound for a phase-major fused kernel."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tools.profile_step import timeit


def main():
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    F = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    from deep_cbrs_amar_renaissance_amd import capi
    from deep_cbrs_amar_renaissance_amd.data import synthetic
    from deep_cbrs_amar_renaissance_amd.utilities.math import gcn_filter_device
    capi.load()
    dev = torch.device('cuda')
    data = synthetic.ml1m_device(scale, device=dev)
    n = data['n_users'] + data['n_items']
    a = gcn_filter_device(data['train_pos'][:, 0], data['train_pos'][:, 1], n)
    nnz = a.nnz
    x = torch.randn((n, F), device=dev)
    y_ref = torch.empty((n, F), device=dev)
    capi.spmm_csr(a.rowptr, a.colidx, a.vals, x, y_ref)
    med, _ = timeit(lambda: capi.spmm_csr(a.rowptr, a.colidx, a.vals, x, y_ref))
    print('unsliced: {:.3f} ms'.format(med), flush=True)
    rows = torch.repeat_interleave(torch.arange(n, device=dev), (a.rowptr[1:] - a.rowptr[:-1]).long())
    cols = a.colidx.long()
    for S in (2, 4, 8, 16, 32):
        # slice boundaries with equal nnz
        order_cols = torch.sort(cols).values
        bounds = [0] + [int(order_cols[(nnz * k) // S]) for k in range(1, S)] + [n]
        subs = []
        for k in range(S):
            m = (cols >= bounds[k]) & (cols < bounds[k + 1])
            r, c, v = rows[m], a.colidx[m], a.vals[m]
            rp = torch.zeros(n + 1, dtype=torch.int64, device=dev)
            rp[1:] = torch.cumsum(torch.bincount(r, minlength=n), 0)
            subs.append((rp.to(torch.int32), c.contiguous(), v.contiguous()))
        y = torch.empty((n, F), device=dev)

        def run():
            for k, (rp, c, v) in enumerate(subs):
                if k == 0:
                    capi.spmm_csr(rp, c, v, x, y)
                else:
                    capi.spmm_csr(rp, c, v, x, None, acc_in=y, acc_out=y)
        run()
        err = float((y - y_ref).abs().max())
        med, _ = timeit(run)
        per = timeit(lambda: capi.spmm_csr(subs[S // 2][0], subs[S // 2][1], subs[S // 2][2], x, None, acc_in=y, acc_out=y))[0]
        print('S={:2d} slices (X slice {:.1f} MB): total {:.3f} ms, one middle slice {:.3f} ms, max err {:.1e}'.format(
            S, n * F * 4 / S / 1e6, med, per, err), flush=True)


if __name__ == '__main__':
    main()
