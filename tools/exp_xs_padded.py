#!/usr/bin/env python
"""Does row-aligned padding pay for the XS SpMM?  (development aid)
Pads every (slice, row) run of the VALUED image to a multiple of 8 entries (zero-valued repeats of the run's last entry),
so that a lane's 8 consecutive entries never straddle a row: no in-lane flushes, but more entries."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tools.exp_xs_floor import timeit


def main():
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    os.environ['AMAR_XS_VALUES'] = '1'
    from deep_cbrs_amar_renaissance_amd import capi
    from deep_cbrs_amar_renaissance_amd.data import synthetic
    from deep_cbrs_amar_renaissance_amd.utilities.math import XcdSliced, gcn_filter_device
    capi.load()
    dev = torch.device('cuda')
    data = synthetic.ml1m_device(scale, device=dev)
    n = data['n_users'] + data['n_items']
    a = gcn_filter_device(data['train_pos'][:, 0], data['train_pos'][:, 1], n)
    xs = XcdSliced.from_csr(a)
    assert xs.vals is not None
    x = torch.randn((n, 8), device=dev)
    y0, y1 = torch.empty((n, 8), device=dev), torch.empty((n, 8), device=dev)
    t0 = timeit(lambda: capi.spmm_xs(xs, x, y0))
    rp = xs.rowptr.long()
    cnt = rp[1:] - rp[:-1]
    pcnt = (cnt + 7) // 8 * 8
    new_rp = torch.zeros_like(rp)
    new_rp[1:] = torch.cumsum(pcnt, 0)
    m_new = int(new_rp[-1])
    seg = torch.repeat_interleave(torch.arange(cnt.numel(), device=dev), cnt)           # run of every old entry
    pos_in_run = torch.arange(int(rp[-1]), device=dev) - rp[seg]
    dst = new_rp[seg] + pos_in_run
    col_new = torch.zeros(m_new, dtype=torch.int32, device=dev)
    val_new = torch.zeros(m_new, dtype=torch.float32, device=dev)
    # pads: repeat the run's last entry (same key, same column: an L1 hit) with value 0
    pseg = torch.repeat_interleave(torch.arange(cnt.numel(), device=dev), pcnt)
    last_old = (rp[pseg] + cnt[pseg] - 1).clamp(min=0)
    col_new[:] = xs.colidx[last_old]
    col_new[dst] = xs.colidx
    val_new[dst] = xs.vals
    print('entries %d -> %d (+%.1f %%)' % (int(rp[-1]), m_new, 100.0 * (m_new / int(rp[-1]) - 1)))
    xs_p = XcdSliced(xs.diag, new_rp.to(torch.int32), col_new, val_new, xs.bounds, xs.shape)
    t1 = timeit(lambda: capi.spmm_xs(xs_p, x, y1))
    print('valued XS: %.3f ms;  row-aligned padded valued XS: %.3f ms;  max |diff| %.2e' % (t0, t1, float((y0 - y1).abs().max())))


if __name__ == '__main__':
    main()
