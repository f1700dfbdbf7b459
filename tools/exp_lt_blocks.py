#!/usr/bin/env python
"""LT against XS on the row block a rank owns in a world-W node-range partition of ml1m(s) (development aid): which image the
partitioned runner should use per world size.  One GPU: the blocks of rank 0 and of the last rank are timed in turn."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tools.exp_lt import timeit


def main():
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    F = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    from deep_cbrs_amar_renaissance_amd import capi, parallel
    from deep_cbrs_amar_renaissance_amd.data import synthetic
    from deep_cbrs_amar_renaissance_amd.utilities.math import gcn_filter_device
    capi.load()
    dev = torch.device('cuda')
    data = synthetic.ml1m_device(scale, device=dev)
    n = data['n_users'] + data['n_items']
    a = gcn_filter_device(data['train_pos'][:, 0], data['train_pos'][:, 1], n)
    for world in (1, 2, 4, 8):
        part = parallel.TypedPartition([0, data['n_users'], n], world)
        for rank in sorted({0, world - 1}):
            blk = part.local_block(a, rank)
            x = torch.randn((world * part.R, F), device=dev)
            y = torch.empty((blk.shape[0], F), device=dev)
            os.environ['AMAR_SPMM_LT'] = '0'
            xs = blk.xcd_sliced()
            t_xs = timeit(lambda: capi.spmm_xs(xs, x, y, prescaled=True))
            os.environ['AMAR_SPMM_LT'] = '1'
            lt = blk.lds_tiled(F)
            y2 = torch.empty_like(y)
            capi.spmm_xs(lt, x, y2, prescaled=True)
            capi.spmm_xs(xs, x, y, prescaled=True)
            err = float((y - y2).abs().max())
            t_lt = timeit(lambda: capi.spmm_xs(lt, x, y2, prescaled=True))
            print('world %d rank %d: rows %d nnz %d | XS %.4f ms  LT %.4f ms (tiles %d, density %.3f entries/(tile.col))  max|diff| %.1e' %
                  (world, rank, blk.shape[0], blk.nnz, t_xs, t_lt, lt.n_tiles, blk.nnz / (lt.n_tiles * blk.shape[1]), err), flush=True)
            del xs, lt, blk
    os.environ.pop('AMAR_SPMM_LT', None)


if __name__ == '__main__':
    main()
