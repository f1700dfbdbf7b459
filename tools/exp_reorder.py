#!/usr/bin/env python
"""Hypothesis test: degree-descending renumbering of nodes (within users / within items) raises the L2 hit rate of the
row gathers because every cached 128-B line then holds four hot rows. Runs the production CSR kernel on both orders."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tools.profile_step import timeit


def permuted_csr(a, new_of_old):
    from deep_cbrs_amar_renaissance_amd.utilities.math import DeviceCSR
    n = a.shape[0]
    dev = a.rowptr.device
    deg = (a.rowptr[1:] - a.rowptr[:-1]).long()
    rows = torch.repeat_interleave(torch.arange(n, device=dev), deg)
    r2, c2 = new_of_old[rows], new_of_old[a.colidx.long()]
    key = r2 * n + c2
    order = torch.argsort(key)
    rp = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    rp[1:] = torch.cumsum(torch.bincount(r2, minlength=n), 0)
    return DeviceCSR(rp.to(torch.int32), c2[order].to(torch.int32), a.vals[order].contiguous(), a.shape)


def main():
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    from deep_cbrs_amar_renaissance_amd import capi
    from deep_cbrs_amar_renaissance_amd.data import synthetic
    from deep_cbrs_amar_renaissance_amd.utilities.math import gcn_filter_device
    capi.load()
    dev = torch.device('cuda')
    data = synthetic.ml1m_device(scale, device=dev)
    nu, ni = data['n_users'], data['n_items']
    n = nu + ni
    a = gcn_filter_device(data['train_pos'][:, 0], data['train_pos'][:, 1], n)
    deg = (a.rowptr[1:] - a.rowptr[:-1]).long()
    variants = {}
    variants['original'] = a
    # degree-descending within users and within items
    ou = torch.argsort(deg[:nu], descending=True)
    oi = torch.argsort(deg[nu:], descending=True) + nu
    old_of_new = torch.cat([ou, oi])
    new_of_old = torch.empty_like(old_of_new); new_of_old[old_of_new] = torch.arange(n, device=dev)
    variants['degree-desc within type'] = permuted_csr(a, new_of_old)
    # global degree-descending (mixes types)
    og = torch.argsort(deg, descending=True)
    ng = torch.empty_like(og); ng[og] = torch.arange(n, device=dev)
    variants['degree-desc global'] = permuted_csr(a, ng)
    # random permutation (control)
    orr = torch.randperm(n, device=dev)
    variants['random permutation'] = permuted_csr(a, orr)
    for F in (8, 16, 32):
        x = torch.randn((n, F), device=dev)
        y = torch.empty((n, F), device=dev)
        for name, m in variants.items():
            med, best = timeit(lambda: capi.spmm_csr(m.rowptr, m.colidx, m.vals, x, y), reps=15)
            alg = m.nnz * 8 + (n + 1) * 4 + 2 * n * F * 4
            print('F={:2d} {:<26s}: {:7.3f} ms  {:7.1f} GB/s alg'.format(F, name, med, alg / med / 1e6), flush=True)


if __name__ == '__main__':
    main()
