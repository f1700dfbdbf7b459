"""Regenerates tests/golden/*.npz: seeded small inputs + the oracle's outputs for them.

The reference cannot run here (TensorFlow/Keras/Spektral are not installed: ordinary
ModuleNotFoundError) and has no fixtures of its own, so these vectors come from the CPU oracle
(oracle/), which is itself pinned by the hand-computed cases and doc.pdf parameter counts in
tests/test_oracle.py.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import models as om, weights as ow          # noqa: E402
from tests import helpers                                # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def flatten(prefix, tree, out):
    if isinstance(tree, np.ndarray):
        out[prefix] = tree
    elif isinstance(tree, dict):
        for k, v in tree.items():
            if not isinstance(v, str):
                flatten(prefix + '.' + k, v, out)
    elif isinstance(tree, (list, tuple)):
        for k, v in enumerate(tree):
            flatten(prefix + '.' + str(k), v, out)


def main():
    for kind in ('gcn', 'lightgcn', 'sage', 'gat'):
        for graph in ('ui', 'uip'):
            g = helpers.tiny_graph(n_users=30, n_items=24, n_ratings=260, seed=21,
                                   n_props=12 if graph == 'uip' else 0, n_links=40 if graph == 'uip' else 0)
            rng = np.random.default_rng(100 + len(kind))
            n = g['adj'].shape[0]
            gnn = ow.gnn(rng, kind, n, 8, (8, 8), 2, bias_range=0.05)
            head = ow.basic_head(rng, ow.gnn_out_dim(gnn), [24, 24], [48, 48], bias_range=0.05)
            out = {'adj_row': g['adj'].row, 'adj_col': g['adj'].col, 'adj_data': g['adj'].data, 'n': np.int64(n),
                   'u_ids': g['u_ids'], 'i_ids': g['i_ids'], 'users': g['users'], 'items': g['items']}
            flatten('gnn', gnn, out)
            flatten('head', head, out)
            out['emb_f32'] = om.propagate(g['adj'], gnn, np.float32)
            out['emb_f64'] = om.propagate(g['adj'], gnn, np.float64)
            out['scores_f32'] = om.basic_gnn_scores(g['adj'], gnn, head, g['u_ids'], g['i_ids'], np.float32)
            out['scores_f64'] = om.basic_gnn_scores(g['adj'], gnn, head, g['u_ids'], g['i_ids'], np.float64)
            for k in (5, 10):
                tu, ti, ts = om.top_k(g['u_ids'], g['i_ids'], out['scores_f64'], g['users'], g['items'], k)
                out['top{}_users'.format(k)], out['top{}_items'.format(k)] = tu, ti
            np.savez_compressed(os.path.join(HERE, 'basic_{}_{}.npz'.format(kind, graph)), **out)
            print(kind, graph, n, out['scores_f64'][:3].ravel())
    extra()


def extra():
    """Families added after the first set: DGCF (dgcf_conv.py) and the hybrid-gnn-tweaks heads (attention fusion, residual)."""
    g = helpers.tiny_graph(n_users=30, n_items=24, n_ratings=260, seed=21, n_props=12, n_links=40)
    n = g['adj'].shape[0]
    base = {'adj_row': g['adj'].row, 'adj_col': g['adj'].col, 'adj_data': g['adj'].data, 'n': np.int64(n),
            'u_ids': g['u_ids'], 'i_ids': g['i_ids'], 'users': g['users'], 'items': g['items']}
    rng = np.random.default_rng(777)
    gnn = ow.gnn(rng, 'dgcf', n, 8, n_layers=2, bias_range=0.05)
    head = ow.basic_head(rng, 8, [24, 24], [48, 48], bias_range=0.05)
    out = dict(base)
    flatten('gnn', gnn, out)
    flatten('head', head, out)
    out['emb_f64'] = om.propagate(g['adj'], gnn, np.float64)
    out['scores_f64'] = om.basic_gnn_scores(g['adj'], gnn, head, g['u_ids'], g['i_ids'], np.float64)
    np.savez_compressed(os.path.join(HERE, 'extra_dgcf_uip.npz'), **out)
    print('dgcf', out['scores_f64'][:3].ravel())
    bert = rng.standard_normal((n, 40)).astype(np.float32) * 0.5
    for name, fb, fusion, residual in (('attention', True, 'attention', False), ('residual', True, 'concatenate', True),
                                       ('entity-attention', False, 'attention', False)):
        gnn = ow.gnn(rng, 'gcn', n, 8, (8, 8), 2, bias_range=0.05)
        head = ow.hybrid_head_tweaked(rng, ow.gnn_out_dim(gnn), 40, ([24, 16], [32, 24], [16, 16]), [24, 16], bias_range=0.05,
                                      fusion_method=fusion, residual=residual, feature_based=fb)
        out = dict(base, bert=bert, feature_based=np.int64(fb))
        flatten('gnn', gnn, out)
        flatten('head', head, out)
        out['scores_f64'] = om.hybrid_gnn_scores(g['adj'], gnn, head, g['u_ids'], g['i_ids'], bert, np.float64, feature_based=fb)
        np.savez_compressed(os.path.join(HERE, 'extra_hybrid_{}.npz'.format(name)), **out)
        print('hybrid', name, out['scores_f64'][:3].ravel())


if __name__ == '__main__':
    main()
