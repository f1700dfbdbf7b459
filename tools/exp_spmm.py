#!/usr/bin/env python
"""Gather-limit experiments for the CSR SpMM (development aid).

Same rowptr/vals as the real ml1m(s) A_hat, but column indices replaced by synthetic patterns, to
separate the cost of the CSR stream from the cost of the row gathers at each cache level; plus
cache-policy / unroll variants of the gather load on the real graph.
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from tools.profile_step import timeit


def main():
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    from deep_cbrs_amar_renaissance_amd import capi
    from deep_cbrs_amar_renaissance_amd.data import synthetic
    from deep_cbrs_amar_renaissance_amd.utilities.math import gcn_filter_device
    capi.load()
    lib = ctypes.CDLL(os.path.join(ROOT, 'tools', 'libexp_spmm.so'))
    lib.exp_spmm.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 6 + [ctypes.c_void_p]
    dev = torch.device('cuda')
    data = synthetic.ml1m_device(scale, device=dev)
    n = data['n_users'] + data['n_items']
    a = gcn_filter_device(data['train_pos'][:, 0], data['train_pos'][:, 1], n)
    nnz = a.nnz
    print('scale', scale, 'N', n, 'nnz', nnz, flush=True)
    g = torch.Generator(device=dev); g.manual_seed(1)
    pos = torch.arange(nnz, device=dev, dtype=torch.int64)

    def run(colidx, F, policy=0, unroll=2, nt=0, label=''):
        torch.manual_seed(F)
        x = torch.randn((n, F), device=dev)
        y = torch.empty((n, F), device=dev)
        st = torch.cuda.current_stream().cuda_stream
        fn = lambda: lib.exp_spmm(a.rowptr.data_ptr(), colidx.data_ptr(), a.vals.data_ptr(), x.data_ptr(), y.data_ptr(),
                                  n, n, F, policy, unroll, nt, st)
        assert fn() == 0
        med, best = timeit(fn, reps=15)
        alg = nnz * 8 + (n + 1) * 4 + 2 * n * F * 4
        print('  {:<46s} F={:2d} pol={} unr={} nt={}: {:7.3f} ms  {:6.1f} Gnnz/s  {:6.1f} GB/s alg'.format(
            label, F, policy, unroll, nt, med, nnz / med / 1e6, alg / med / 1e6), flush=True)
        return y

    real = a.colidx
    pats = {
        'real graph': real,
        'sequential (p mod N): no gather misses': (pos % n).to(torch.int32),
        'uniform random over N': torch.randint(0, n, (nnz,), device=dev, generator=g, dtype=torch.int64).to(torch.int32),
        'random in 2^19 rows (16 MB @F8)': torch.randint(0, min(n, 1 << 19), (nnz,), device=dev, generator=g).to(torch.int32),
        'random in 2^16 rows (2 MB @F8: L2)': torch.randint(0, 1 << 16, (nnz,), device=dev, generator=g).to(torch.int32),
        'random in 2^13 rows (256 KB @F8)': torch.randint(0, 1 << 13, (nnz,), device=dev, generator=g).to(torch.int32),
        'random in 2^9 rows (16 KB @F8: L1)': torch.randint(0, 1 << 9, (nnz,), device=dev, generator=g).to(torch.int32),
        'all zero column (broadcast)': torch.zeros(nnz, device=dev, dtype=torch.int32),
    }
    print('-- gather patterns (plain loads)')
    print('-- cache policy / unroll / nt-stream on the real graph, F=8')
    ref = run(real, 8, 0, 2, 0, 'ref')
    for pol in (0, 1, 2, 3, 4):
        for unr in (2, 4):
            for nt in (0, 1):
                y = run(real, 8, pol, unr, nt, 'variant')
                assert torch.allclose(y, ref, rtol=1e-5, atol=1e-5)
    print('-- cache policy, L2-resident pattern')
    for pol in (0, 1, 2, 3, 4):
        run(pats['random in 2^16 rows (2 MB @F8: L2)'], 8, pol, 4, 1, 'L2-resident')
    for pol in (0, 1, 2, 3, 4):
        run(real, 32, pol, 2, 0, 'real F=32')
    print('-- all-zero column (stream floor), unroll/nt variants')
    for unr in (2, 4):
        for nt in (0, 1):
            run(pats['all zero column (broadcast)'], 8, 0, unr, nt, 'floor')


if __name__ == '__main__':
    main()
