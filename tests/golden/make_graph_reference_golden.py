#!/usr/bin/env python
"""Golden vectors for rows A1 / A1' (ratings -> contiguous ids -> adjacency) produced by the REFERENCE'S OWN functions, run in
the build container.

The modules that hold them import TensorFlow at their top (`src/utilities/math.py:2`, `src/data/loaders.py:6` via data.datasets)
and so cannot be imported here (ModuleNotFoundError, an ordinary import error) — but the four functions themselves are plain
numpy / pandas / scipy:

    symmetrize_matrix          /root/reference/src/utilities/math.py:6-21
    get_user_properties        /root/reference/src/data/preprocess.py:9-41
    build_adjacency_matrix     /root/reference/src/data/preprocess.py:44-170
    load_train_test_ratings    /root/reference/src/data/loaders.py:11-82

This script reads those files as text, takes exactly these function definitions out of the syntax tree and executes THEM,
unchanged, in a namespace that holds numpy, pandas and scipy.sparse — nothing of the reference is copied into the repo, nothing
is stubbed.  It then writes small rating / property files, calls `load_train_test_ratings` on them for every adjacency type of
the hot path, and stores inputs and outputs in tests/golden/graph_reference.npz.  The reference tree does not travel; the fixture does.

    python tests/golden/make_graph_reference_golden.py      (needs /root/reference; CPU only)
"""
import ast
import os
import sys
import tempfile

import numpy as np
import pandas as pd
from scipy import sparse

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/src'
WANTED = {'utilities/math.py': ['symmetrize_matrix'],
          'data/preprocess.py': ['get_user_properties', 'build_adjacency_matrix'],
          'data/loaders.py': ['load_train_test_ratings']}


def reference_functions():
    ns = {'np': np, 'pd': pd, 'sparse': sparse}
    for rel, names in WANTED.items():
        path = os.path.join(REF, rel)
        tree = ast.parse(open(path).read(), filename=path)
        for node in tree.body:
            if isinstance(node, ast.FunctionDef) and node.name in names:
                exec(compile(ast.Module(body=[node], type_ignores=[]), path, 'exec'), ns)
    assert all(n in ns for names in WANTED.values() for n in names)
    return ns


def tiny_files(d, rng):
    """Ratings with sparse raw ids (every test user / item also in train), a property file with a relation column, duplicate
    (item, property) pairs across relations, only train items (datasets/README.md:21-22, preprocess.py:173-198)."""
    n_users, n_items, n_props = 37, 29, 50
    raw_u = np.sort(rng.choice(400, n_users, replace=False))
    raw_i = np.sort(rng.choice(900, n_items, replace=False))
    raw_p = np.sort(rng.choice(3000, n_props, replace=False))
    keys = rng.choice(n_users * n_items, 420, replace=False)
    u, i = keys // n_items, keys % n_items
    r = (rng.random(len(u)) < 0.6).astype(np.int64)
    test = rng.random(len(u)) < 0.25
    test[np.unique(u, return_index=True)[1]] = False
    test[np.unique(i, return_index=True)[1]] = False                 # first occurrence of every user and item stays in train
    rows = np.stack([raw_u[u], raw_i[i], r], axis=1)
    train, tst = rows[~test], rows[test]
    train = train[rng.permutation(len(train))]
    tst = tst[rng.permutation(len(tst))]
    it = rng.choice(np.unique(train[:, 1]), 140)
    pr = raw_p[rng.integers(0, n_props, 140)]
    rel = rng.integers(0, 11, 140)
    props = np.stack([it, pr, rel], axis=1)
    props = np.concatenate([props, np.stack([props[:9, 0], props[:9, 1], (props[:9, 2] + 1) % 11], axis=1)])   # same pair, other relation
    paths = {k: os.path.join(d, k + '.tsv') for k in ('train', 'test', 'props')}
    for k, a in (('train', train), ('test', tst), ('props', props)):
        np.savetxt(paths[k], a, fmt='%d', delimiter='\t')
    return paths, train, tst, props


def coo(m):
    m = m.tocoo() if not isinstance(m, sparse.coo_matrix) else m
    return m.row.astype(np.int64), m.col.astype(np.int64), np.asarray(m.data), np.array(m.shape, dtype=np.int64)


def main():
    fn = reference_functions()
    rng = np.random.default_rng(20240102)
    out = {}
    with tempfile.TemporaryDirectory() as d:
        paths, train, tst, props = tiny_files(d, rng)
        out.update(raw_train=train, raw_test=tst, raw_props=props)
        (tr, te), (users, items) = fn['load_train_test_ratings'](paths['train'], paths['test'])
        out.update(train_indexed=tr, test_indexed=te, users=users, items=items)
        for kind in ('unary', 'binary', 'unary-uip', 'unary-kg'):
            for sym in (True, False):
                res = fn['load_train_test_ratings'](paths['train'], paths['test'], paths['props'] if 'unary-' in kind else None,
                                                    return_adjacency=True, type_adjacency=kind, symmetric_adjacency=sym)
                (tr2, te2), (u2, i2), adj = res
                assert np.array_equal(tr2, tr) and np.array_equal(te2, te) and np.array_equal(u2, users) and np.array_equal(i2, items)
                tag = '{}_{}'.format(kind.replace('-', '_'), 'sym' if sym else 'raw')
                mats = adj if isinstance(adj, tuple) else (adj,)
                for j, m in enumerate(mats):
                    r, c, v, shp = coo(m)
                    out.update({'{}_{}_row'.format(tag, j): r, '{}_{}_col'.format(tag, j): c, '{}_{}_val'.format(tag, j): v,
                                '{}_{}_shape'.format(tag, j): shp, '{}_{}_dtype'.format(tag, j): np.array(str(m.dtype))})
                if kind == 'unary-kg' and sym:                       # TwoWay's user-property graph out of the two matrices
                    up = fn['get_user_properties'](mats[0], mats[1], len(users), len(items))
                    r, c, v, shp = coo(up)
                    out.update(user_props_row=r, user_props_col=c, user_props_val=v, user_props_shape=shp,
                               user_props_dtype=np.array(str(up.dtype)))
    np.savez_compressed(os.path.join(HERE, 'graph_reference.npz'), **out)
    print('wrote graph_reference.npz:', len(out), 'arrays;', {k: out[k].shape for k in ('raw_train', 'raw_test', 'raw_props', 'users', 'items')})


if __name__ == '__main__':
    main()
