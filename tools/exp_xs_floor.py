#!/usr/bin/env python
"""Where does the XS partial kernel's time go?  (development aid)

Times amar_spmm_xs_f32 on ml1m(s) with the gather columns rewritten so that the cache path changes
while the instruction stream stays the same:
  real      the real columns
  l1        column & 1023 (32 KB of X: every gather an L1 hit)  -> instruction / issue floor
  l2line    column & ~3 | (entry index & 3)?? no: column rounded to a 128-B line (4 rows) -> same lines, same misses
  sorted4   columns of a row-run made consecutive (run start + j): 4 entries share a line -> 1/4 of the line fills
"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


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
    xs = a.xcd_sliced()
    x = torch.randn((n, F), device=dev)
    y = torch.empty((n, F), device=dev)
    real = xs.colidx.clone()
    key = real & ~0x3ffffff
    col = real & 0x3ffffff
    m = col.numel()
    idx = torch.arange(m, device=dev, dtype=torch.int32)
    variants = {
        'real': real,
        'l1': key | (col & 1023),
        'l2_64k': key | (col & 65535),                 # 2 MB of X: every gather an L2 hit, L1 mostly misses
        'seq': key | (idx % n).to(torch.int32),        # consecutive entries -> consecutive rows: 4 per line, coalesced
    }
    for name, c in variants.items():
        xs.colidx = c.contiguous()
        t = timeit(lambda: capi.spmm_xs(xs, x, y))
        print('%-8s %.3f ms' % (name, t), flush=True)
    xs.colidx = real
    # combine alone: an image without off-diagonal entries
    rp = xs.rowptr.clone()
    xs.rowptr = torch.zeros_like(rp)
    t = timeit(lambda: capi.spmm_xs(xs, x, y))
    print('%-8s %.3f ms (all tiles empty: launch + combine only)' % ('empty', t), flush=True)
    xs.rowptr = rp


if __name__ == '__main__':
    main()
