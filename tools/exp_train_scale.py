#!/usr/bin/env python
"""Round 4: the training batch (training.py: full-graph propagation, BCE + L2, reverse pass, Adam — one replayed hipGraph) on LARGER
graphs than the reference's ml1m(s=1): where does a batch's time go once the propagation is no longer launch-bound?
usage: python tools/exp_train_scale.py <scale> [<batches> [<batch size> [<d>]]]      (BasicGCN d x 2, dense [3d, 3d], clf [48, 48])
Prints ms per batch (graph replays, device-resident ids) and, with EXP_EAGER=1, HIP-event times of the eager phases."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch


def main():
    scale = int(sys.argv[1])
    batches = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    bs = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
    d = int(sys.argv[4]) if len(sys.argv) > 4 else 8
    from deep_cbrs_amar_renaissance_amd import capi, engine, training
    from deep_cbrs_amar_renaissance_amd.data import synthetic
    from deep_cbrs_amar_renaissance_amd.models import basic
    from deep_cbrs_amar_renaissance_amd.utilities.math import gcn_filter_device
    capi.load()
    engine.set_seed(42)
    dev = torch.device('cuda')
    data = synthetic.ml1m_device(scale, device=dev)
    n = data['n_users'] + data['n_items']
    a = gcn_filter_device(data['train_pos'][:, 0], data['train_pos'][:, 1], n)
    model = basic.BasicGCN(a, embedding_dim=d, n_hiddens=[d, d], dense_units=[3 * d, 3 * d], clf_units=[48, 48], l2_regularizer=float(os.environ.get('EXP_L2', '1e-4')))
    model.n_users, model.n_items = data['n_users'], data['n_items']
    tr = training.Trainer(model, learning_rate=1e-3)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1)
    pairs = data['train_pos']
    print('ml1m(s=%d): N %d nnz %d, batch %d, d %d' % (scale, n, a.nnz, bs, d), flush=True)

    def batch(k):
        idx = torch.randint(0, pairs.shape[0], (bs,), device=dev, generator=gen)
        u = pairs[idx, 0].to(torch.int32)
        i = pairs[idx, 1].to(torch.int32)
        y = (torch.rand(bs, device=dev, generator=gen) < 0.57).to(torch.float32)
        return u, i, y
    prepared = [batch(k) for k in range(8)]
    for k in range(4):                                               # eager batch, capture, two replays
        tr.train_batch_graphed(*prepared[k % 8])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(batches):
        tr.train_batch_graphed(*prepared[k % 8])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / batches
    print('  %.3f ms per batch (%d replays): %.0f pairs/s; loss sum %.4f' % (1e3 * dt, batches, bs / dt, tr.pop_loss_sum()), flush=True)


if __name__ == '__main__':
    main()
