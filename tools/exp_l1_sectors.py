#!/usr/bin/env python
"""Round 4 (CPU only, ~1 min): how many DISTINCT 128- / 64- / 32-byte pieces of the gathered table a row tile of the ml1m(s=64) graph touches per
entry (F = 8: 32-byte rows), with the ids as they are and with the columns relabelled by degree — the L1 miss rate an LDS-tiled launch cannot go
below, to hold against the L2 requests the counters report (DESIGN.md 4a "Round 4").  256 tiles of equal entry count, 128 per node type."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep_cbrs_amar_renaissance_amd.data import synthetic
d = synthetic.ml1m_device(64, device=torch.device('cpu'))
tp = d['train_pos'].numpy(); nu, ni = d['n_users'], d['n_items']
u, i = tp[:, 0], tp[:, 1] - nu
du = np.bincount(u, minlength=nu); di = np.bincount(i, minlength=ni)
pu = np.empty(nu, np.int64); pu[np.argsort(-du, kind='stable')] = np.arange(nu)
pi = np.empty(ni, np.int64); pi[np.argsort(-di, kind='stable')] = np.arange(ni)
# cost-balanced tiles approximated: user tiles 2300*... use entries-balanced contiguous tiles with random ids: 256 tiles split by entries
def run(label, ru, ri, cu, ci, cpl):
    tot_e = tot_l = 0
    for rows, cols, nr, nc, nt in ((ru, ci, nu, ni, 128), (ri, cu, ni, nu, 128)):
        deg = np.bincount(rows, minlength=nr); cum = np.cumsum(deg)
        tb = np.searchsorted(cum, np.arange(1, nt) * cum[-1] / nt)
        t = np.searchsorted(tb, rows, side='right')
        nl_ = (nc + cpl - 1) // cpl + 1
        n = len(np.unique(t * nl_ + cols // cpl))
        tot_e += len(rows); tot_l += n
    print('%s, %d-B lines: %.3f distinct lines per entry' % (label, 32 * cpl, tot_l / tot_e), flush=True)
for cpl in (4, 2, 1):
    run('random ids', u, i, u, i, cpl)
    # dealt rows: row label such that contiguous tiles contain a degree mix: label = random (keep original row ids), columns sorted
    run('sorted columns, rows as they are', u, i, pu[u], pi[i], cpl)
