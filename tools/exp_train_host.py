"""Host vs device share of a training epoch (development aid): fit() with the graph replay stubbed out."""
import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from deep_cbrs_amar_renaissance_amd import capi, engine
from deep_cbrs_amar_renaissance_amd.data.datasets import UserItemGraph
from deep_cbrs_amar_renaissance_amd.experiment import Adam
from deep_cbrs_amar_renaissance_amd.models import basic
from tests import helpers
capi.load()
g = helpers.ml1m_indexed(1)
engine.set_seed(42)
model = basic.BasicGCN(g['adj_ui'], embedding_dim=16, n_hiddens=[16, 16], dense_units=[48, 48], clf_units=[64, 64], l2_regularizer=1e-4)
model.compile(loss='binary_crossentropy', optimizer=Adam(learning_rate=1e-3), metrics=['accuracy'])
train = UserItemGraph(g['train'], g['users'], g['items'], g['adj_ui'], batch_size=1024, shuffle=True)
model.fit(train, epochs=1, verbose=False)
torch.cuda.synchronize(); t0 = time.perf_counter(); model.fit(train, epochs=2, verbose=False); torch.cuda.synchronize()
full = (time.perf_counter() - t0) / 2
orig = torch.cuda.CUDAGraph.replay
torch.cuda.CUDAGraph.replay = lambda self: None
torch.cuda.synchronize(); t0 = time.perf_counter(); model.fit(train, epochs=2, verbose=False); torch.cuda.synchronize()
host = (time.perf_counter() - t0) / 2
torch.cuda.CUDAGraph.replay = orig
t0 = time.perf_counter()
for b in range(len(train)): _ = train[b]
seq = time.perf_counter() - t0
print('epoch %.3f s; without replay (host only) %.3f s; Sequence indexing alone %.3f s; batches %d' % (full, host, seq, len(train)))
