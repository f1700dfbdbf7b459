#!/usr/bin/env python
"""Round 4: what one replayed predict() pass at the reference's real size (ml1m(s=1)) is made of — run under
`rocprofv3 --kernel-trace --stats`; EXP_MODE=hoisted|faithful, 500 / 20 replays after the capture.  usage: see tools/profile_s1.sh"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench

from deep_cbrs_amar_renaissance_amd import capi, engine
from deep_cbrs_amar_renaissance_amd.data import synthetic
from deep_cbrs_amar_renaissance_amd.models import basic
from deep_cbrs_amar_renaissance_amd.utilities.math import gcn_filter_device
capi.load()
dev = torch.device('cuda', 0)
data = synthetic.ml1m_device(1, device=dev)
n = data['n_users'] + data['n_items']
a_hat = gcn_filter_device(data['train_pos'][:, 0], data['train_pos'][:, 1], n)
engine.set_seed(42)
model = basic.BasicGCN(a_hat, **bench.GRID1)
model.n_users, model.n_items = data['n_users'], data['n_items']
u_np = data['test'][:, 0].cpu().numpy().astype(np.int64)
i_np = data['test'][:, 1].cpu().numpy().astype(np.int64)
p = len(u_np)


class Seq:
    order_version = 0

    def __len__(self):
        return (p + 2047) // 2048

    def __getitem__(self, b):
        return (u_np[b * 2048:(b + 1) * 2048], i_np[b * 2048:(b + 1) * 2048]), np.zeros(min(2048, p - b * 2048))


seq = Seq()
hoist = os.environ.get('EXP_MODE', 'hoisted') == 'hoisted'
model._predict_graphed(seq, hoist)
torch.cuda.synchronize()
import time
reps = 500 if hoist else 20
t0 = time.perf_counter()
for _ in range(reps):
    model._predict_graphed(seq, hoist)
torch.cuda.synchronize()
print('%s: %.4f ms per pass over %d pairs (%d replays)' % ('hoisted' if hoist else 'faithful', 1e3 * (time.perf_counter() - t0) / reps, p, reps))
