"""Seeded Keras-style initialisers for fixtures.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

glorot_uniform = U(+-sqrt(6 / (fan_in + fan_out))) with Keras' `_compute_fans`: for a 2-D
kernel fan_in, fan_out = shape; for rank > 2 the leading dims are a receptive field.  The
embedding table is a plain [N, d] weight (gnn.py:41-46), so fan_in = N.  Biases are zeros in
the reference; fixtures draw them U(+-0.05) so that bias paths are exercised.
"""
import numpy as np


def glorot_uniform(rng, shape):
    shape = tuple(int(s) for s in shape)
    if len(shape) == 1:
        fan_in = fan_out = shape[0]
    elif len(shape) == 2:
        fan_in, fan_out = shape
    else:
        rf = int(np.prod(shape[:-2]))
        fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
    limit = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-limit, limit, size=shape).astype(np.float32)


def _bias(rng, n, bias_range):
    if bias_range:
        return rng.uniform(-bias_range, bias_range, size=n).astype(np.float32)
    return np.zeros(n, dtype=np.float32)


def dense_net(rng, in_dim, units, bias_range=0.0):
    layers = []
    for u in units:
        layers.append((glorot_uniform(rng, (in_dim, u)), _bias(rng, u, bias_range)))
        in_dim = u
    return layers


def gnn(rng, kind, n_nodes, embedding_dim=8, n_hiddens=(8, 8), n_layers=2, final_node='concatenation',
        bias_range=0.0):
    w = {'kind': kind, 'embeddings': glorot_uniform(rng, (n_nodes, embedding_dim)), 'layers': [],
         'final_node': final_node}
    f_in = embedding_dim
    if kind == 'lightgcn':
        w['layers'] = [{} for _ in range(n_layers)]
        w['final_node'] = 'mean'
        return w
    if kind == 'dgcf':                                   # LocalityAdaptive: ones [N, 1] (dgcf_conv.py:93-99); fixtures perturb them
        w['layers'] = [{'w': (1.0 + (rng.uniform(-bias_range, bias_range, size=(n_nodes, 1)) if bias_range else 0.0)
                              * 10).astype(np.float32) * np.ones((n_nodes, 1), dtype=np.float32)} for _ in range(n_layers)]
        w['final_node'] = 'mean'
        return w
    for c in n_hiddens:
        if kind == 'gcn':
            lw = {'kernel': glorot_uniform(rng, (f_in, c)), 'bias': _bias(rng, c, bias_range)}
        elif kind == 'sage':
            lw = {'kernel': glorot_uniform(rng, (2 * f_in, c)), 'bias': _bias(rng, c, bias_range)}
        elif kind == 'gat':
            # Spektral shapes: kernel [F, heads=1, C]; attn kernels [C, heads=1, 1]
            lw = {'kernel': glorot_uniform(rng, (f_in, 1, c)).reshape(f_in, c),
                  'attn_self': glorot_uniform(rng, (c, 1, 1)).reshape(c),
                  'attn_neigh': glorot_uniform(rng, (c, 1, 1)).reshape(c),
                  'bias': _bias(rng, c, bias_range)}
        else:
            raise ValueError(kind)
        w['layers'].append(lw)
        f_in = c
    return w


def gnn_out_dim(w):
    d = w['embeddings'].shape[1]
    if w['kind'] == 'lightgcn' or w['final_node'] in ('mean', 'sum'):
        return d
    widths = [d] + [lw['kernel'].shape[1] for lw in w['layers']]
    return sum(widths) if w['final_node'] == 'concatenation' else widths[-1]


def basic_head(rng, in_dim, dense_units, clf_units, bias_range=0.0):
    return {'unet': dense_net(rng, in_dim, dense_units, bias_range),
            'inet': dense_net(rng, in_dim, dense_units, bias_range),
            'clf': dense_net(rng, 2 * dense_units[-1], list(clf_units) + [1], bias_range)}


def hybrid_head(rng, g_dim, b_dim, dense_units, clf_units, bias_range=0.0):
    d1, d2, d3 = dense_units
    return {'dense1a': dense_net(rng, g_dim, d1, bias_range), 'dense1b': dense_net(rng, g_dim, d1, bias_range),
            'dense2a': dense_net(rng, b_dim, d2, bias_range), 'dense2b': dense_net(rng, b_dim, d2, bias_range),
            'dense3a': dense_net(rng, 2 * d1[-1], d3, bias_range), 'dense3b': dense_net(rng, 2 * d2[-1], d3, bias_range),
            'clf': dense_net(rng, 2 * d3[-1], list(clf_units) + [1], bias_range)}


def attention_fuser(rng, da, db):
    """FusionLayer('attention') weights (fusion.py:19-47): att_weight [D, D], D = max(da, db); proj_weight when da != db."""
    d = max(da, db)
    fw = {'att_weight': glorot_uniform(rng, (d, d))}
    if da != db:
        fw['proj_weight'] = glorot_uniform(rng, (min(da, db), d))
    return fw


def hybrid_head_tweaked(rng, g_dim, b_dim, dense_units, clf_units, bias_range=0.0, fusion_method='attention', residual=False,
                        feature_based=True):
    """Heads of econfigs/hybrid-gnn-tweaks*.yaml: attention fusion and / or the residual classifier (hybrid.py:42-67)."""
    d1, d2, d3 = dense_units
    head = {'dense1a': dense_net(rng, g_dim, d1, bias_range), 'dense1b': dense_net(rng, g_dim, d1, bias_range),
            'dense2a': dense_net(rng, b_dim, d2, bias_range), 'dense2b': dense_net(rng, b_dim, d2, bias_range)}
    att = fusion_method == 'attention'
    if feature_based:
        ins = ((d1[-1], d1[-1]), (d2[-1], d2[-1]))
        first_att, last_att = False, att
    else:
        ins = ((d1[-1], d2[-1]), (d1[-1], d2[-1]))
        first_att, last_att = att, False
    for name, (da, db) in zip(('fuse1a', 'fuse1b'), ins):
        if first_att:
            head[name] = attention_fuser(rng, da, db)
    w3 = [max(a, b) if first_att else a + b for a, b in ins]
    head['dense3a'], head['dense3b'] = dense_net(rng, w3[0], d3, bias_range), dense_net(rng, w3[1], d3, bias_range)
    if last_att:
        head['fuse2'] = attention_fuser(rng, d3[-1], d3[-1])
    fused = d3[-1] if last_att else 2 * d3[-1]
    if residual:
        head['residual'] = dense_net(rng, fused, list(clf_units), bias_range)
        head['clf'] = dense_net(rng, clf_units[-1], [1], bias_range)
    else:
        head['clf'] = dense_net(rng, fused, list(clf_units) + [1], bias_range)
    return head


def _stack(rng, kind, n_rows, embedding_dim, hiddens, n_layers, final_node, bias_range, table=True):
    """One SequentialGNN-like stack; LightGCN / DGCF keep the caller's final_node (only the single-graph classes force
    'mean').  `n_rows` sizes the trainable table (None: FullInput stack, no table) — DGCF gates are per graph node."""
    n_table, n_nodes = n_rows
    w = gnn(rng, kind, n_nodes, embedding_dim=embedding_dim, n_hiddens=hiddens, n_layers=n_layers, final_node=final_node,
            bias_range=bias_range)
    w['final_node'] = final_node
    if table:
        w['embeddings'] = glorot_uniform(rng, (n_table, embedding_dim))
    else:
        del w['embeddings']
    return w


def two_step(rng, kind, n_users, n_items, n_props, embedding_dim=8, n_hiddens=(8, 8), n_layers=2, item_node='mean',
             final_node='concatenation', bias_range=0.0):
    """TwoStepGNN weights (tsgnn.py:53-81): step one [|I|+|P|, d] over the item-property graph; step two trains the
    [|U|, d2] user rows.  n_hiddens are the widths of step one; step two's continue the list the way tsgnn.py:65-75
    does (d (L+1) each for item_node 'concatenation', else d)."""
    hops = len(n_hiddens) if kind in ('gcn', 'sage', 'gat') else n_layers
    d2 = embedding_dim * (hops + 1) if (item_node == 'concatenation' and kind in ('gcn', 'sage', 'gat')) else embedding_dim
    if kind in ('lightgcn', 'dgcf'):
        final_node = 'mean'                                  # tsgnn.py:222, 252
    one = _stack(rng, kind, (n_items + n_props, n_items + n_props), embedding_dim, list(n_hiddens), hops, item_node, bias_range)
    two = _stack(rng, kind, (n_users, n_users + n_items), d2, [d2] * hops, hops, final_node, bias_range)
    return {'step_one': one, 'step_two': two}


def two_way(rng, kind, n_users, n_items, n_props, embedding_dim=8, n_hiddens=(8, 8), n_layers=2, user_item_node='mean',
            final_node='concatenation', bias_range=0.0):
    """TwoWayGNN weights (twgnn.py:53-85): way one [|U|+|P|, d] over the user-property graph, way two [|I|+|P|, d] over
    the item-property graph (same layer widths, separate weights), and the table-less user-item stack."""
    hops = len(n_hiddens) if kind in ('gcn', 'sage', 'gat') else n_layers
    d2 = embedding_dim * (hops + 1) if (user_item_node == 'concatenation' and kind in ('gcn', 'sage', 'gat')) else embedding_dim
    if kind in ('lightgcn', 'dgcf'):
        final_node = 'mean'                                  # twgnn.py:227, 257
    one = _stack(rng, kind, (n_users + n_props, n_users + n_props), embedding_dim, list(n_hiddens), hops, user_item_node, bias_range)
    two = _stack(rng, kind, (n_items + n_props, n_items + n_props), embedding_dim, list(n_hiddens), hops, user_item_node, bias_range)
    # the user-item stack's first layer consumes whatever the ways hand over
    tmp = dict(one)
    in_dim = gnn_out_dim(tmp) if kind in ('gcn', 'sage', 'gat') else embedding_dim
    three = _stack(rng, kind, (0, n_users + n_items), in_dim, [d2] * hops, hops, final_node, bias_range, table=False)
    return {'way_one': one, 'way_two': two, 'step_two': three}
