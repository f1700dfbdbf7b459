#!/usr/bin/env python
"""Round 4: fit() of a HYBRID model at ML-1M size (econfigs/hybrid-gnn.yaml grid1 dims: HybridBertGCN d=8 x 2, dense [[24,24],[256,64],[64,64]],
clf [64,64], 768-d BERT rows per batch, batch 1 024): s per epoch and, under rocprofv3 (tools/profile_train.sh's recipe), the launches of a batch.
usage: python tools/exp_train_hybrid.py [epochs]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch


def main():
    epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    from deep_cbrs_amar_renaissance_amd import capi, engine
    from deep_cbrs_amar_renaissance_amd.experiment import Adam
    from deep_cbrs_amar_renaissance_amd.models import hybrid
    from tests import helpers
    capi.load()
    g = helpers.ml1m_indexed(1)
    engine.set_seed(42)
    model = hybrid.HybridBertGCN(g['adj_ui'], embedding_dim=8, n_hiddens=[8, 8], dense_units=[[24, 24], [256, 64], [64, 64]], clf_units=[64, 64],
                                 feature_based=True, l2_regularizer=1e-4)
    model.compile(loss='binary_crossentropy', optimizer=Adam(learning_rate=1e-3), metrics=['accuracy'])
    n = g['adj_ui'].shape[0]
    rng = np.random.default_rng(0)
    table = rng.standard_normal((n, 768)).astype(np.float32)
    tr = np.asarray(g['train'])
    u_all, i_all, y_all = tr[:, 0].astype(np.int64), tr[:, 1].astype(np.int64), tr[:, 2].astype(np.float32)
    bs = 1024
    nb = len(u_all) // bs

    class Seq:
        def __len__(self):
            return nb

        def __getitem__(self, b):
            s = slice(b * bs, (b + 1) * bs)
            if resident:
                return (u_all[s], i_all[s]), y_all[s]
            return (u_all[s], i_all[s], table[u_all[s]], table[i_all[s]]), y_all[s]
    resident = os.environ.get('EXP_TABLE') == '1'                      # the BERT table registered once on the device (model.set_bert_table): batches carry ids only
    if resident:
        model.set_bert_table(table)
    seq = Seq()
    if os.environ.get('EXP_SEQ') == 'reference':                       # the reference's own Sequence class (ids + host-gathered BERT rows per batch)
        from deep_cbrs_amar_renaissance_amd.data.datasets import UserItemGraphEmbeddings
        seq = UserItemGraphEmbeddings(tr, g['users'], g['items'], g['adj_ui'], table, batch_size=bs, shuffle=True)
        nb = len(seq)
    model.fit(seq, epochs=1, verbose=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    hist = model.fit(seq, epochs=epochs, verbose=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / epochs
    print('HybridBertGCN grid1 (' + ('reference Sequence, AMAR_RESIDENT_BERT=' + os.environ.get('AMAR_RESIDENT_BERT', '1') if os.environ.get('EXP_SEQ') == 'reference' else 'resident BERT table' if resident else 'BERT rows from the host per batch') + '): %.2f s/epoch (%d batches of %d): %.3f ms per batch, %.0f pairs/s; loss %.4f' % (dt, nb, bs, 1e3 * dt / nb, nb * bs / dt, hist['loss'][-1]), flush=True)


def predict_times():
    """predict() of the test pairs through the reference's hybrid Sequence, BERT rows per batch against the registered table."""
    from deep_cbrs_amar_renaissance_amd import capi, engine
    from deep_cbrs_amar_renaissance_amd.data.datasets import UserItemGraphEmbeddings
    from deep_cbrs_amar_renaissance_amd.models import hybrid
    from tests import helpers
    capi.load()
    g = helpers.ml1m_indexed(1)
    engine.set_seed(42)
    model = hybrid.HybridBertGCN(g['adj_ui'], embedding_dim=8, n_hiddens=[8, 8], dense_units=[[24, 24], [256, 64], [64, 64]], clf_units=[64, 64],
                                 feature_based=True, l2_regularizer=1e-4)
    table = np.random.default_rng(0).standard_normal((g['adj_ui'].shape[0], 768)).astype(np.float32)
    seq = UserItemGraphEmbeddings(np.asarray(g['test']), g['users'], g['items'], g['adj_ui'], table, batch_size=2048, shuffle=False)
    for mode in ('0', '1'):
        os.environ['AMAR_RESIDENT_BERT'] = mode
        model.predict(seq)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = model.predict(seq)
        torch.cuda.synchronize()
        print('predict(), %d pairs in %d batches, AMAR_RESIDENT_BERT=%s: %.1f ms' % (len(out), len(seq), mode, 1e3 * (time.perf_counter() - t0)), flush=True)


if __name__ == '__main__':
    if os.environ.get('EXP_PREDICT') == '1':
        predict_times()
        sys.exit(0)
    main()
