#!/usr/bin/env python
"""Golden vectors for rows A1 / A1' (ratings -> contiguous ids -> adjacency) produced by the REFERENCE'S OWN functions, run in
the build container.

The modules that hold them import TensorFlow at their top (`src/utilities/math.py:2`, `src/data/loaders.py:6` via data.datasets)
and so cannot be imported here (ModuleNotFoundError, an ordinary import error) — but the four functions themselves are plain
numpy / pandas / scipy:

    symmetrize_matrix               /root/reference/src/utilities/math.py:6-21
    get_user_properties             /root/reference/src/data/preprocess.py:9-41
    build_adjacency_matrix          /root/reference/src/data/preprocess.py:44-170
    process_item_properties_graph   /root/reference/src/data/preprocess.py:173-198
    load_train_test_ratings         /root/reference/src/data/loaders.py:11-82
    json_load_graph_embeddings, json_load_bert_embeddings, load_graph_user_item_embeddings,
    load_bert_user_item_embeddings  /root/reference/src/data/loaders.py:85-144

This script reads those files as text, takes exactly these function definitions out of the syntax tree and executes THEM,
unchanged, in a namespace that holds json, numpy, pandas and scipy.sparse — nothing of the reference is copied into the repo, nothing
is stubbed.  It then writes small rating / property files, calls `load_train_test_ratings` on them for every adjacency type of
the hot path, and stores inputs and outputs in tests/golden/graph_reference.npz; the embedding loaders and the offline property
filter run on small JSON / TSV files of the reference's formats -> tests/golden/loaders_reference.npz.  The reference
tree does not travel; the fixtures do.

    python tests/golden/make_graph_reference_golden.py      (needs /root/reference; CPU only)
"""
import ast
import json
import os
import sys
import tempfile

import numpy as np
import pandas as pd
from scipy import sparse

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/src'
WANTED = {'utilities/math.py': ['symmetrize_matrix'],
          'data/preprocess.py': ['get_user_properties', 'build_adjacency_matrix', 'process_item_properties_graph'],
          'data/loaders.py': ['load_train_test_ratings', 'json_load_graph_embeddings', 'json_load_bert_embeddings',
                              'load_graph_user_item_embeddings', 'load_bert_user_item_embeddings']}


def reference_functions():
    ns = {'json': json, 'np': np, 'pd': pd, 'sparse': sparse}
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
        # --- embedding files and the offline property filter, in the reference's on-disk formats
        lo = {}
        dim = 6
        n_ent = int(max(users.max(), items.max())) + 1
        kge = rng.uniform(-0.1, 0.1, (n_ent, dim)).astype(np.float32)
        kge_path = os.path.join(d, 'kge.json')
        json.dump({'ent_embeddings': kge.tolist()}, open(kge_path, 'w'))
        ub = (0.5 * rng.standard_normal((len(users), dim))).astype(np.float32)
        ib = (0.5 * rng.standard_normal((len(items), dim))).astype(np.float32)
        pu, pi = rng.permutation(len(users)), rng.permutation(len(items))                # file order is not id order
        ub_path, ib_path = os.path.join(d, 'user-lastlayer.json'), os.path.join(d, 'item-lastlayer.json')
        json.dump([{'ID_OpenKE': int(users[k]), 'profile_embedding': ub[k].tolist()} for k in pu], open(ub_path, 'w'))
        json.dump([{'ID_OpenKE': int(items[k]), 'embedding': ib[k].tolist()} for k in pi], open(ib_path, 'w'))
        lo.update(kge_table=kge, bert_users=ub, bert_items=ib, bert_user_file_order=pu, bert_item_file_order=pi, users=users, items=items)
        lo['kge_rows'] = fn['load_graph_user_item_embeddings'](kge_path, users, items)
        lo['bert_rows'] = fn['load_bert_user_item_embeddings'](ub_path, ib_path, users, items)
        # graph file: a header line, the train ratings, then KG triples — some about items that never occur in train
        extra = np.stack([rng.choice(np.setdiff1d(np.arange(900), items), 12), rng.integers(0, 3000, 12), rng.integers(0, 11, 12)], axis=1)
        kg_rows = np.concatenate([props, extra])[rng.permutation(len(props) + 12)]
        graph_path, kg_out = os.path.join(d, 'graph.tsv'), os.path.join(d, 'kg_out.tsv')
        with open(graph_path, 'w') as fp:
            fp.write('head\ttail\trel\n')
            np.savetxt(fp, np.concatenate([train, kg_rows]), fmt='%d', delimiter='\t')
        fn['process_item_properties_graph'](paths['train'], graph_path, kg_out)
        lo.update(filter_ratings=train, filter_kg_rows=kg_rows, filter_output=np.loadtxt(kg_out, dtype=np.int64, delimiter='\t').reshape(-1, 3))
    np.savez_compressed(os.path.join(HERE, 'graph_reference.npz'), **out)
    np.savez_compressed(os.path.join(HERE, 'loaders_reference.npz'), **lo)
    print('wrote graph_reference.npz:', len(out), 'arrays;', {k: out[k].shape for k in ('raw_train', 'raw_test', 'raw_props', 'users', 'items')})
    print('wrote loaders_reference.npz:', {k: v.shape for k, v in lo.items()})


if __name__ == '__main__':
    main()
