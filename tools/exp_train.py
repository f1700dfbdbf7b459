#!/usr/bin/env python
"""Training wall-time on ml1m(s=1) for the config the reference publishes a time for (doc.pdf p.22 Table 5:
BasicRS-GCN 16ch x 2L, dense [48,48], clf [64,64], batch 1024, Adam 1e-3: 211 s for 25 epochs on an RTX 3060)."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch


def main():
    epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    only_table5 = len(sys.argv) > 2 and sys.argv[2] == 'table5'          # (tools/profile_train.sh)
    from deep_cbrs_amar_renaissance_amd import capi, engine
    from deep_cbrs_amar_renaissance_amd.data.datasets import UserItemGraph
    from deep_cbrs_amar_renaissance_amd.experiment import Adam
    from deep_cbrs_amar_renaissance_amd.models import basic
    from tests import helpers
    capi.load()
    g = helpers.ml1m_indexed(1)
    engine.set_seed(42)
    for name, cfg in (('BasicGCN 16x2 (Table 5)', dict(embedding_dim=16, n_hiddens=[16, 16], dense_units=[48, 48], clf_units=[64, 64], l2_regularizer=1e-4)),
                      ('BasicGCN 8x2 (grid1)', dict(embedding_dim=8, n_hiddens=[8, 8], dense_units=[24, 24], clf_units=[48, 48], l2_regularizer=1e-4)),
                      ('BasicLightGCN 8x2', dict(embedding_dim=8, n_layers=2, dense_units=[24, 24], clf_units=[48, 48], l2_regularizer=1e-4)),
                      ('BasicGraphSage 8x2', dict(embedding_dim=8, n_hiddens=[8, 8], dense_units=[24, 24], clf_units=[48, 48], l2_regularizer=1e-4)),
                      ('BasicGAT 8x2', dict(embedding_dim=8, n_hiddens=[8, 8], dense_units=[24, 24], clf_units=[48, 48], l2_regularizer=1e-4))):
        if only_table5 and 'Table 5' not in name:
            continue
        cls = basic.BasicLightGCN if 'Light' in name else basic.BasicGraphSage if 'Sage' in name else basic.BasicGAT if 'GAT' in name else basic.BasicGCN
        model = cls(g['adj_ui'], **cfg)
        model.compile(loss='binary_crossentropy', optimizer=Adam(learning_rate=1e-3), metrics=['accuracy'])
        train = UserItemGraph(g['train'], g['users'], g['items'], g['adj_ui'], batch_size=1024, shuffle=True)
        test = UserItemGraph(g['test'], g['users'], g['items'], g['adj_ui'], batch_size=2048, shuffle=False)
        model.fit(train, epochs=1, verbose=False)                       # warm-up epoch (allocator, packing caches)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        hist = model.fit(train, epochs=epochs, verbose=False)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / epochs
        loss, acc = model.evaluate(test)
        print('{}: {:.2f} s/epoch ({} batches of 1024) -> {:.0f} s per 25 epochs; {:.0f} train pairs/s; loss {:.4f} -> {:.4f}; test loss {:.4f} acc {:.3f}'.format(
            name, dt, len(train), 25 * dt, len(g['train']) / dt, hist['loss'][0], hist['loss'][-1], loss, acc), flush=True)


if __name__ == '__main__':
    main()
