#!/usr/bin/env python
"""PCIe-inclusive rate of the scoring call (development aid): the boundary hands over host int64 id arrays
(datasets.py:199-203) and takes host fp32 scores back; bench.py's `value` keeps ids resident, as its contract says."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch


def main():
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    from deep_cbrs_amar_renaissance_amd import capi, engine
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
    nu = data['n_users']
    u_host, i_host = data['test'][:, 0].cpu(), data['test'][:, 1].cpu()          # int64, pageable: what a Sequence delivers
    u_pin, i_pin = u_host.pin_memory(), i_host.pin_memory()
    p = u_host.numel()
    out_pin = torch.empty((p, 1), dtype=torch.float32).pin_memory()

    def call(u_src, i_src, out):
        u = u_src.to(dev, non_blocking=True).to(torch.int32)
        i = i_src.to(dev, non_blocking=True).to(torch.int32)
        emb = model.gnn(None)
        scores = model.rs.score_towers(model.rs.towers(emb[:nu], emb[nu:nu + model.n_items]), u, i, 0, nu)
        if out is None:
            return scores.cpu()
        out.copy_(scores, non_blocking=True)
        torch.cuda.synchronize()
        return out

    for name, args in (('pageable ids, pageable scores', (u_host, i_host, None)), ('pinned ids, pinned scores', (u_pin, i_pin, out_pin))):
        call(*args)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            call(*args)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print('%-32s: %.2f ms per call for %d pairs = %.2e pairs/s (ids %.0f MB in, scores %.0f MB out)' %
              (name, dt * 1e3, p, p / dt, 2 * p * 8 / 1e6, p * 4 / 1e6), flush=True)


if __name__ == '__main__':
    main()
