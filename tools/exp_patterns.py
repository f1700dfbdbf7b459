#!/usr/bin/env python
"""Gather-pattern experiments on the PRODUCTION SpMM kernel (same rowptr/vals, synthetic column patterns)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tools.profile_step import timeit


def main():
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    Fs = [int(f) for f in (sys.argv[2] if len(sys.argv) > 2 else '8').split(',')]
    from deep_cbrs_amar_renaissance_amd import capi
    from deep_cbrs_amar_renaissance_amd.data import synthetic
    from deep_cbrs_amar_renaissance_amd.utilities.math import gcn_filter_device
    capi.load()
    dev = torch.device('cuda')
    data = synthetic.ml1m_device(scale, device=dev)
    n = data['n_users'] + data['n_items']
    a = gcn_filter_device(data['train_pos'][:, 0], data['train_pos'][:, 1], n)
    nnz = a.nnz
    g = torch.Generator(device=dev); g.manual_seed(1)
    pos = torch.arange(nnz, device=dev, dtype=torch.int64)
    rnd = lambda hi: torch.randint(0, hi, (nnz,), device=dev, generator=g).to(torch.int32)
    pats = [('real graph', a.colidx), ('sequential (p mod N)', (pos % n).to(torch.int32)),
            ('uniform random over N', rnd(n)), ('random in 2^17 rows', rnd(1 << 17)), ('random in 2^16 rows', rnd(1 << 16)),
            ('random in 2^13 rows', rnd(1 << 13)), ('random in 2^9 rows', rnd(1 << 9)),
            ('all zero column', torch.zeros(nnz, device=dev, dtype=torch.int32))]
    for F in Fs:
        x = torch.randn((n, F), device=dev)
        y = torch.empty((n, F), device=dev)
        for label, c in pats:
            med, best = timeit(lambda: capi.spmm_csr(a.rowptr, c, a.vals, x, y), reps=15)
            print('F={:2d} {:<26s}: {:7.3f} ms  {:6.1f} Gnnz/s'.format(F, label, med, nnz / med / 1e6), flush=True)


if __name__ == '__main__':
    main()
