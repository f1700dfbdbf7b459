#!/usr/bin/env python
"""How much of each XCD slice's entries do its H hottest columns cover?  (Would an LDS-resident hot set of X rows pay?)
    python tools/exp_hot_columns.py [scale]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch


def main():
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    from deep_cbrs_amar_renaissance_amd.data import synthetic
    from deep_cbrs_amar_renaissance_amd.utilities.math import gcn_filter_device
    dev = torch.device('cuda')
    data = synthetic.ml1m_device(scale, device=dev)
    n = data['n_users'] + data['n_items']
    a = gcn_filter_device(data['train_pos'][:, 0], data['train_pos'][:, 1], n)
    xs = a.xcd_sliced()
    bounds = [int(b) for b in xs.bounds]
    cols = a.colidx.long()
    counts = torch.bincount(cols, minlength=n)
    print('n =', n, 'nnz =', a.nnz, 'slices:', bounds, ' (users: columns 0..%d)' % (data['n_users'] - 1))
    tot_hot = {h: 0 for h in (1024, 2048, 4096, 8192)}
    for k in range(len(bounds) - 1):
        c = counts[bounds[k]:bounds[k + 1]]
        srt = torch.sort(c, descending=True).values
        tot = int(c.sum())
        line = 'slice %d: %7d columns, %9d entries;' % (k, c.numel(), tot)
        for h in tot_hot:
            cov = int(srt[:h].sum())
            tot_hot[h] += cov
            line += '  top %d: %4.1f %%' % (h, 100.0 * cov / max(tot, 1))
        print(line)
    print('all slices:', {h: round(100.0 * v / a.nnz, 1) for h, v in tot_hot.items()}, '% of the entries')


if __name__ == '__main__':
    main()
