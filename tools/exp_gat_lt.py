#!/usr/bin/env python
"""GAT layer on the LDS-tiled image (amar_gat_lt_f32) against the row kernel and the XCD-sliced online-softmax form on
ml1m(s): parity and time per layer (development aid).  `python tools/exp_gat_lt.py <scale> [C ...]`."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tools.exp_xs_floor import timeit


def edge_csr(data, dev):
    from deep_cbrs_amar_renaissance_amd.utilities.math import DeviceCSR
    n = data['n_users'] + data['n_items']
    r, c = data['train_pos'][:, 0], data['train_pos'][:, 1]
    rows, cols = torch.cat([r, c]), torch.cat([c, r])
    order = torch.argsort(rows * n + cols)
    rows, cols = rows[order], cols[order]
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    rowptr[1:] = torch.cumsum(torch.bincount(rows, minlength=n), 0)
    a = DeviceCSR(rowptr.to(torch.int32), cols.to(torch.int32), None, (n, n))
    a.row_breaks = (data['n_users'],)
    return a


def main():
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    widths = [int(v) for v in sys.argv[2:]] or [8, 16, 32]
    from deep_cbrs_amar_renaissance_amd import capi
    from deep_cbrs_amar_renaissance_amd.data import synthetic
    capi.load()
    dev = torch.device('cuda')
    a = edge_csr(synthetic.ml1m_device(scale, device=dev), dev)
    n = a.shape[0]
    os.environ['AMAR_SPMM_LT'] = '1'
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    for C in widths:
        h = torch.randn((n, C), device=dev, generator=gen)
        b = torch.randn(C, device=dev, generator=gen) * 0.1
        for spread in (1.0, 40.0):                                  # attention scalars of unit size / far apart (rows fall back)
            ss = torch.randn(n, device=dev, generator=gen) * spread
            sn = torch.randn(n, device=dev, generator=gen) * spread
            y_row, y_lt = torch.empty((n, C), device=dev), torch.zeros((n, C), device=dev)
            capi.gat_layer(a.rowptr, a.colidx, h, ss, sn, b, y_row, self_loop=True)
            lt = a.tiled_gat_image(C)
            capi.gat_lt(lt, a, h, ss, sn, b, y_lt, self_loop=True)
            torch.cuda.synchronize()
            err = float((y_row - y_lt).abs().max())
            t_lt = timeit(lambda: capi.gat_lt(lt, a, h, ss, sn, b, y_lt, self_loop=True))
            print('C=%d spread %.0f: LT %.4f ms, max |diff| vs row kernel %.2e (tiles %d, vrows <= %d, window %d, flagged %.2f %%)' %
                  (C, spread, t_lt, err, lt.n_tiles, int(lt.vcount.max()), lt.window_entries, 100.0 * lt.n_flagged / max(1, lt.n_entries)), flush=True)
        t_row = timeit(lambda: capi.gat_layer(a.rowptr, a.colidx, h, ss, sn, b, y_row, self_loop=True))
        xs = a.xcd_sliced()
        y_xs = torch.empty((n, C), device=dev)
        capi.gat_xs(xs, h, ss, sn, b, y_xs, self_loop=True)
        t_xs = timeit(lambda: capi.gat_xs(xs, h, ss, sn, b, y_xs, self_loop=True))
        print('C=%d: row kernel %.4f ms, XS %.4f ms (max |diff| vs row %.2e)' % (C, t_row, t_xs, float((y_row - y_xs).abs().max())), flush=True)


if __name__ == '__main__':
    main()
