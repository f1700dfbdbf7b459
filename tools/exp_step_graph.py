#!/usr/bin/env python
"""The bench's single-GPU step replayed from a hipGraph vs launched kernel by kernel (development aid)."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch


def main():
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    from deep_cbrs_amar_renaissance_amd import capi, engine, parallel
    from deep_cbrs_amar_renaissance_amd.data import synthetic
    from deep_cbrs_amar_renaissance_amd.models import basic
    from deep_cbrs_amar_renaissance_amd.utilities.math import gcn_filter_device
    capi.load()
    dev = torch.device('cuda')
    data = synthetic.ml1m_device(scale, device=dev)
    n = data['n_users'] + data['n_items']
    a = gcn_filter_device(data['train_pos'][:, 0], data['train_pos'][:, 1], n)
    engine.set_seed(42)
    model = basic.BasicGCN(a, embedding_dim=8, n_hiddens=[8, 8], n_layers=2, dense_units=[24, 24], clf_units=[48, 48])
    model.n_users, model.n_items = data['n_users'], data['n_items']
    perm = torch.randperm(data['test'].shape[0], device=dev)
    u = data['test'][perm, 0].to(torch.int32).contiguous()
    i = data['test'][perm, 1].to(torch.int32).contiguous()
    runner = parallel.SingleRunner(model, u, i)
    for _ in range(3):
        runner.step()
    torch.cuda.synchronize()

    def timed(fn, reps=20):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3
    eager = timed(runner.step)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        runner.step()
    graph.replay()
    replay = timed(graph.replay)
    print('scale %d: step launched kernel by kernel %.3f ms, replayed from a hipGraph %.3f ms' % (scale, eager, replay))


if __name__ == '__main__':
    main()
