#!/usr/bin/env python
"""Hypothesis test: XCD-affine column slices. Virtual CSR with 8N rows: 64-row block b holds slice (b % 8) of rows
chunk (b // 8); with round-robin block->XCD dispatch every XCD then gathers from ONE slice of X only."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tools.profile_step import timeit


def build_virtual(a, S, balanced=True):
    from deep_cbrs_amar_renaissance_amd.utilities.math import DeviceCSR
    dev = a.rowptr.device
    n = a.shape[0]
    nnz = a.nnz
    deg = (a.rowptr[1:] - a.rowptr[:-1]).long()
    rows = torch.repeat_interleave(torch.arange(n, device=dev), deg)
    cols = a.colidx.long()
    if balanced:
        sc = torch.sort(cols).values
        bounds = torch.tensor([0] + [int(sc[(nnz * k) // S]) for k in range(1, S)] + [n], device=dev)
    else:
        bounds = torch.tensor([(n * k) // S for k in range(S + 1)], device=dev)
    sl = torch.searchsorted(bounds, cols, right=True) - 1
    nchunks = (n + 63) // 64
    vr = ((rows // 64) * S + sl) * 64 + (rows % 64)
    order = torch.argsort(vr * n + cols)
    nv = nchunks * S * 64
    rp = torch.zeros(nv + 1, dtype=torch.int64, device=dev)
    rp[1:] = torch.cumsum(torch.bincount(vr, minlength=nv), 0)
    return DeviceCSR(rp.to(torch.int32), a.colidx[order].contiguous(), a.vals[order].contiguous(), (nv, n)), bounds


def main():
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    F = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    mode = sys.argv[3] if len(sys.argv) > 3 else 'time'
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
    capi.spmm_csr(a.rowptr, a.colidx, a.vals, x, y)
    if mode == 'time':
        med, _ = timeit(lambda: capi.spmm_csr(a.rowptr, a.colidx, a.vals, x, y))
        print('plain: {:.3f} ms'.format(med), flush=True)
    for S, bal in ((8, True), (8, False), (16, True)):
        v, bounds = build_virtual(a, S, bal)
        p = torch.empty((v.shape[0], F), device=dev)
        capi.spmm_csr(v.rowptr, v.colidx, v.vals, x, p)
        nch = v.shape[0] // (S * 64)
        ysum = p.view(nch, S, 64, F).sum(1).reshape(-1, F)[:n]
        err = float((ysum - y).abs().max())
        if mode == 'time':
            med, best = timeit(lambda: capi.spmm_csr(v.rowptr, v.colidx, v.vals, x, p))
            print('virtual S={} balanced={}: {:.3f} ms (best {:.3f}), partial rows {}, max err {:.1e}, slice widths {}'.format(
                S, bal, med, best, v.shape[0], err, (bounds[1:] - bounds[:-1]).tolist()), flush=True)
        else:
            for _ in range(3):
                capi.spmm_csr(v.rowptr, v.colidx, v.vals, x, p)
            torch.cuda.synchronize()
            print('ran S', S, bal)


if __name__ == '__main__':
    main()
