#!/usr/bin/env python
"""GraphSAGE / GAT layer timing on ml1m(s) (development aid): the value-free row kernels against their algorithmic bytes
(nnz*4 + (N+1)*4 + 2*N*F*4, + nnz*4 for GAT's scalar gather)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tools.exp_xs_floor import timeit


def main():
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    F = 8
    from deep_cbrs_amar_renaissance_amd import capi, engine
    from deep_cbrs_amar_renaissance_amd.data import synthetic
    from deep_cbrs_amar_renaissance_amd.models import basic
    from deep_cbrs_amar_renaissance_amd.utilities.math import DeviceCSR
    capi.load()
    dev = torch.device('cuda')
    data = synthetic.ml1m_device(scale, device=dev)
    n = data['n_users'] + data['n_items']
    r, c = data['train_pos'][:, 0], data['train_pos'][:, 1]
    rows, cols = torch.cat([r, c]), torch.cat([c, r])
    order = torch.argsort(rows * n + cols)
    rows, cols = rows[order], cols[order]
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    rowptr[1:] = torch.cumsum(torch.bincount(rows, minlength=n), 0)
    a = DeviceCSR(rowptr.to(torch.int32), cols.to(torch.int32), None, (n, n))
    nnz = a.nnz
    x = torch.randn((n, F), device=dev)
    y = torch.empty((n, F), device=dev)
    w = torch.randn((2 * F, F), device=dev) * 0.3
    b = torch.zeros(F, device=dev)
    t = timeit(lambda: capi.sage_layer(a.rowptr, a.colidx, x, w, b, y, self_loop=True))
    alg = nnz * 4 + (n + 1) * 4 + 2 * n * F * 4
    print('sage_row_kernel<8>: %.3f ms, %.0f MB algorithmic -> %.0f GB/s (%.1f %% of 8 TB/s)' % (t, alg / 1e6, alg / t / 1e6, alg / t / 1e6 / 80), flush=True)
    xs = a.xcd_sliced_mean(True)
    xa = torch.empty((n, 2 * F), device=dev); z = torch.empty((n, F), device=dev); nrm = torch.empty((n, F), device=dev); inv = torch.empty(n, device=dev)

    def sage_xs():
        capi.copy_columns(x, xa[:, :F])
        capi.spmm_xs(xs, x, xa[:, F:], prescaled=True)
        capi.dense(xa, w, b, z, act=None)
        capi.l2norm_fwd(z, nrm, inv, y, act='relu')
    y_row = y.clone()
    sage_xs()
    print('  XS form max |diff| vs row kernel: %.2e' % float((y - y_row).abs().max()))
    t = timeit(sage_xs)
    print('sage on XS (mean aggregate + dense + l2norm): %.3f ms -> %.0f GB/s (%.1f %% of 8 TB/s)' % (t, (alg) / t / 1e6, alg / t / 1e6 / 80), flush=True)
    h = torch.randn((n, F), device=dev)
    ss, sn = torch.randn(n, device=dev), torch.randn(n, device=dev)
    t = timeit(lambda: capi.gat_layer(a.rowptr, a.colidx, h, ss, sn, b, y, self_loop=True))
    alg += nnz * 4
    print('gat_row_kernel<8>:  %.3f ms, %.0f MB algorithmic -> %.0f GB/s (%.1f %% of 8 TB/s)' % (t, alg / 1e6, alg / t / 1e6, alg / t / 1e6 / 80), flush=True)
    y_row = y.clone()
    xs_e = a.xcd_sliced()
    capi.gat_xs(xs_e, h, ss, sn, b, y, self_loop=True)
    print('  GAT XS form max |diff| vs row kernel: %.2e' % float((y - y_row).abs().max()))
    t = timeit(lambda: capi.gat_xs(xs_e, h, ss, sn, b, y, self_loop=True))
    print('gat on XS (pack + partial + combine): %.3f ms -> %.0f GB/s (%.1f %% of 8 TB/s)' % (t, alg / t / 1e6, alg / t / 1e6 / 80), flush=True)
    t = timeit(lambda: capi.spmm_csr(a.rowptr, a.colidx, None, x, y))
    print('value-free spmm_csr (stream kernel): %.3f ms' % t, flush=True)
    # GraphSAGE's mean aggregate on the LDS-tiled image, and GAT at the wider grids (C = 16, 32): row kernel vs XS form
    os.environ['AMAR_SPMM_LT'] = '1'
    img = a.tiled_mean_image(F, True)
    agg = torch.empty((n, F), device=dev)
    t = timeit(lambda: capi.spmm_xs(img, x, agg, prescaled=True))
    t2 = timeit(lambda: capi.spmm_xs(xs, x, agg, prescaled=True))
    print('sage mean aggregate alone: LT %.3f ms, XS %.3f ms' % (t, t2), flush=True)
    for C in (16, 32):
        hC = torch.randn((n, C), device=dev)
        bC = torch.zeros(C, device=dev)
        yr, yx = torch.empty((n, C), device=dev), torch.empty((n, C), device=dev)
        t_row = timeit(lambda: capi.gat_layer(a.rowptr, a.colidx, hC, ss, sn, bC, yr, self_loop=True))
        capi.gat_xs(xs_e, hC, ss, sn, bC, yx, self_loop=True)
        t_xs = timeit(lambda: capi.gat_xs(xs_e, hC, ss, sn, bC, yx, self_loop=True))
        print('GAT C=%d: row kernel %.3f ms, XS form %.3f ms, max |diff| %.2e' % (C, t_row, t_xs, float((yr - yx).abs().max())), flush=True)


if __name__ == '__main__':
    main()
