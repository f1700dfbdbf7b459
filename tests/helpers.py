"""Shared test plumbing: synthetic graphs, and product-model -> oracle weight export."""
import numpy as np


def seed_offset():
    """AMAR_TEST_SEED_OFFSET shifts the seeds of the randomised sweeps (stress runs: `AMAR_TEST_SEED_OFFSET=3 pytest -m gpu -k randomised`)."""
    import os
    return int(os.environ.get('AMAR_TEST_SEED_OFFSET', '0'))


def tiny_graph(n_users=40, n_items=30, n_ratings=400, seed=0, n_props=0, n_links=0):
    """Random bipartite (optionally tripartite) rating graph in contiguous ids + a pair list to score."""
    from deep_cbrs_amar_renaissance_amd.data.preprocess import build_adjacency_matrix
    rng = np.random.default_rng(seed)
    keys = rng.choice(n_users * n_items, size=min(n_ratings, n_users * n_items), replace=False)
    u, i = keys // n_items, keys % n_items + n_users
    ratings = np.stack([u, i, (rng.random(len(u)) < 0.6).astype(np.int64)], axis=1)
    users, items = np.arange(n_users) * 3 + 1, np.arange(n_items) * 5 + 2
    triples = props = None
    kind = 'unary'
    if n_props:
        it = rng.integers(0, n_items, size=n_links)
        pr = rng.integers(0, n_props, size=n_links) + n_items
        triples = np.stack([it, pr, np.ones(n_links, dtype=np.int64)], axis=1)
        triples = np.concatenate([triples, triples[: max(1, n_links // 10)]])      # duplicate (item, prop) links
        props = np.arange(n_props)
        kind = 'unary-uip'
    adj = build_adjacency_matrix(ratings, users, items, triples, props, type_adjacency=kind)
    pairs = rng.choice(n_users * n_items, size=min(300, n_users * n_items), replace=False)
    return {'adj': adj, 'ratings': ratings, 'users': users, 'items': items, 'triples': triples, 'props': props,
            'u_ids': pairs // n_items, 'i_ids': pairs % n_items + n_users, 'n_users': n_users, 'n_items': n_items}


def ml1m_indexed(scale=1):
    from deep_cbrs_amar_renaissance_amd.data import synthetic, loaders, preprocess
    ds = synthetic.ml1m(scale)
    (train, test), (users, items) = loaders.index_ratings(ds.train, ds.test)
    triples, props = loaders.index_props(ds.props, items)
    return {'train': train, 'test': test, 'users': users, 'items': items, 'triples': triples, 'props': props,
            'adj_ui': preprocess.build_adjacency_matrix(train, users, items),
            'adj_uip': preprocess.build_adjacency_matrix(train, users, items, triples, props, 'unary-uip'),
            'raw': ds}


def randomize_biases(model, seed=0, scale=0.05):
    """Biases are zero-initialised in the reference; draw them U(+-0.05) so bias paths are tested."""
    import torch
    rng = np.random.default_rng(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith('bias'):
                p.copy_(torch.from_numpy(rng.uniform(-scale, scale, size=tuple(p.shape)).astype(np.float32)))


def spread_scores(model, factor=30.0):
    """Seed-initialised heads put every score within ~1e-2 of 0.5, so rankings hinge on fp32 rounding.
    Scaling the output layer spreads the scores over (0, 1) like a trained model's, leaving few near-ties."""
    import torch
    with torch.no_grad():
        model.rs.clf.layers[-1].kernel.mul_(factor)


def _np(p):
    return p.detach().cpu().numpy().copy()


def gnn_to_oracle(gnn):
    """models.gnn.GNN instance -> oracle weight dict (oracle/models.py)."""
    return seq_to_oracle(gnn.gnn_layers)


def seq_to_oracle(seq):
    """One SequentialGNN / HalfInput / FullInputSequentialGNN stack -> oracle weight dict."""
    from deep_cbrs_amar_renaissance_amd.layers.gcn_conv import GCNConv
    from deep_cbrs_amar_renaissance_amd.layers.graphsage_conv import GraphSageConv
    from deep_cbrs_amar_renaissance_amd.layers.gat_conv import GATConv
    from deep_cbrs_amar_renaissance_amd.layers.lightgcn_conv import LightGCNConv
    from deep_cbrs_amar_renaissance_amd.layers.dgcf_conv import DGCFConv
    kinds = {GCNConv: 'gcn', GraphSageConv: 'sage', GATConv: 'gat', LightGCNConv: 'lightgcn', DGCFConv: 'dgcf'}
    kind = kinds[type(seq.seq_layers[0])]
    layers = []
    for l in seq.seq_layers:
        if kind == 'lightgcn':
            layers.append({})
        elif kind == 'dgcf':
            layers.append({'w': _np(l.w)})
        elif kind == 'gat':
            c = l.channels
            layers.append({'kernel': _np(l.kernel).reshape(-1, c), 'attn_self': _np(l.attn_kernel_self).reshape(c),
                           'attn_neigh': _np(l.attn_kernel_neighs).reshape(c), 'bias': _np(l.bias)})
        else:
            layers.append({'kernel': _np(l.kernel), 'bias': _np(l.bias)})
    out = {'kind': kind, 'layers': layers, 'final_node': seq.final_node}
    if seq.final_node == 'w-sum':
        out['reduction_w'] = _np(seq.reduce.w).reshape(-1)
    if getattr(seq, 'embeddings', None) is not None:
        out['embeddings'] = _np(seq.embeddings)
    return out


def two_step_to_oracle(gnn):
    return {'step_one': seq_to_oracle(gnn.step_one_gnn_layers), 'step_two': seq_to_oracle(gnn.step_two_gnn_layers)}


def two_way_to_oracle(gnn):
    return {'way_one': seq_to_oracle(gnn.way_one_gnn_layers), 'way_two': seq_to_oracle(gnn.way_two_gnn_layers),
            'step_two': seq_to_oracle(gnn.step_two_gnn_layers)}


def kg_graph(n_users=40, n_items=30, n_props=25, n_ratings=400, n_links=90, seed=0, symmetric=True):
    """tiny_graph's ratings + item-property links as the ('unary-kg') pair of graphs TwoStep / TwoWay models take, plus
    the two-hop user-property graph."""
    from deep_cbrs_amar_renaissance_amd.data.preprocess import build_adjacency_matrix, get_user_properties
    g = tiny_graph(n_users, n_items, n_ratings, seed, n_props=n_props, n_links=n_links)
    ui, ip = build_adjacency_matrix(g['ratings'], g['users'], g['items'], g['triples'], g['props'], type_adjacency='unary-kg',
                                    symmetric_adjacency=symmetric)
    g.update({'adj_ui': ui, 'adj_ip': ip, 'adj_up': get_user_properties(ui, ip, n_users, n_items), 'n_props': n_props})
    return g


def _net(seq):
    return [(_np(l.kernel), _np(l.bias)) for l in seq.layers]


def basic_head_to_oracle(rs):
    return {'unet': _net(rs.unet), 'inet': _net(rs.inet), 'clf': _net(rs.clf)}


def hybrid_head_to_oracle(rs):
    head = {k: _net(getattr(rs, k)) for k in ('dense1a', 'dense1b', 'dense2a', 'dense2b', 'dense3a', 'dense3b', 'clf')}
    for name in ('fuse1a', 'fuse1b', 'fuse2'):                      # attention fusions carry weights (fusion.py:19-47)
        fuse = getattr(rs, name)
        if fuse.method == 'attention':
            head[name] = {'att_weight': _np(fuse.att_weight)}
            if fuse.proj_weight is not None:
                head[name]['proj_weight'] = _np(fuse.proj_weight)
    if getattr(rs, 'residual', None) is not None:
        head['residual'] = _net(rs.residual)
    return head


def rel_err(got, want):
    return float(np.abs(np.asarray(got, dtype=np.float64) - want).max() / max(1e-30, np.abs(want).max()))


def load_oracle_weights(model, gnn, head):
    """Copy an oracle weight dict (oracle/models.py layout) into a product Basic* model in place."""
    import torch

    def put(param, value):
        with torch.no_grad():
            param.copy_(torch.from_numpy(np.ascontiguousarray(value, dtype=np.float32)).reshape(param.shape))

    seq = model.gnn.gnn_layers
    put(seq.embeddings, gnn['embeddings'])
    for layer, lw in zip(seq.seq_layers, gnn['layers']):
        if 'kernel' in lw:
            put(layer.kernel, lw['kernel'])
            put(layer.bias, lw['bias'])
        if 'attn_self' in lw:
            put(layer.attn_kernel_self, lw['attn_self'])
            put(layer.attn_kernel_neighs, lw['attn_neigh'])
        if 'w' in lw:
            put(layer.w, lw['w'])
    for name in head:
        if name.startswith('fuse'):
            for key, value in head[name].items():
                put(getattr(getattr(model.rs, name), key), value)
            continue
        for layer, (w, b) in zip(getattr(model.rs, name).layers, head[name]):
            put(layer.kernel, w)
            put(layer.bias, b)
