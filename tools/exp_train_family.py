#!/usr/bin/env python
"""Two fit() epochs of ONE Basic* model family at ml1m(s=1) (d = 8 x 2, dense [24, 24], clf [48, 48], batch 1 024) — meant to run under
rocprofv3 --kernel-trace --stats so that tools/train_launches.py <dir> 1482 lists the launches of a replayed batch.
usage: python tools/exp_train_family.py <BasicGCN|BasicLightGCN|BasicGraphSage|BasicGAT>"""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from deep_cbrs_amar_renaissance_amd import capi, engine
from deep_cbrs_amar_renaissance_amd.data.datasets import UserItemGraph
from deep_cbrs_amar_renaissance_amd.experiment import Adam
from deep_cbrs_amar_renaissance_amd.models import basic
from tests import helpers
capi.load()
g = helpers.ml1m_indexed(1)
engine.set_seed(42)
name = sys.argv[1]
cls = getattr(basic, name)
cfg = dict(embedding_dim=8, dense_units=[24, 24], clf_units=[48, 48], l2_regularizer=1e-4)
if name == 'BasicLightGCN': cfg['n_layers'] = 2
else: cfg['n_hiddens'] = [8, 8]
model = cls(g['adj_ui'], **cfg)
model.compile(loss='binary_crossentropy', optimizer=Adam(learning_rate=1e-3), metrics=['accuracy'])
train = UserItemGraph(g['train'], g['users'], g['items'], g['adj_ui'], batch_size=1024, shuffle=True)
model.fit(train, epochs=2, verbose=False)
torch.cuda.synchronize()
