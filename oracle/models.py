"""Oracle rows A4 (SequentialGNN), A7 (gather), A8 (Basic / Hybrid heads), A9 (per-user top-k).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Weights travel as plain dicts of numpy arrays:

    gnn     {'kind': 'gcn'|'lightgcn'|'sage'|'gat'|'dgcf', 'embeddings': [N,d],
             'layers': [ {'kernel','bias'} | {} | {'kernel','bias'} | {'kernel','attn_self','attn_neigh','bias'} | {'w'} ],
             'final_node': 'concatenation'|'mean'|'sum'|'last'|'w-sum' (+ 'reduction_w' [n_layers] for 'w-sum')}
    basic   {'unet': [(W,b)..], 'inet': [(W,b)..], 'clf': [(W,b).., (W_out,b_out)]}
    hybrid  {'dense1a','dense1b','dense2a','dense2b','dense3a','dense3b': [(W,b)..], 'clf': [...]}
"""
import numpy as np

from oracle import graph as ograph
from oracle import layers as olayers


def propagate(adj, gnn, dtype=np.float32, self_loops=True):
    """SequentialGNN.call (gnn.py:74-84) under GNN.call (gnn.py:263-264): inputs are ignored.

    `adj` is the raw symmetric COO adjacency from build_adjacency_matrix; GCN and LightGCN first
    run gcn_filter on it (gnn.py:283,381), GraphSAGE and GAT consume it as is (gnn.py:316-319,349-352).
    """
    return propagate_from(adj, gnn['embeddings'], gnn, dtype, self_loops)


def propagate_from(adj, x0, gnn, dtype=np.float32, self_loops=True, force_mean=True):
    """The layer loop + reduction of SequentialGNN / HalfInput / FullInputSequentialGNN.call (gnn.py:74-84, 141-150,
    197-207) over the node table `x0`.  `force_mean`: the single-graph LightGCN / DGCF classes override final_node with
    'mean' (gnn.py:378, 405); the TwoStep / TwoWay classes override only their LAST stack's (tsgnn.py:222, twgnn.py:227)."""
    kind = gnn['kind']
    x = x0.astype(dtype)
    hs = [x]
    if kind in ('gcn', 'lightgcn'):
        a_hat = ograph.gcn_filter(adj)
        for lw in gnn['layers']:
            if kind == 'gcn':
                x = olayers.gcn_conv(x, a_hat, lw['kernel'].astype(dtype), lw['bias'].astype(dtype))
            else:
                x = olayers.lightgcn_conv(x, a_hat)
            hs.append(x)
    elif kind == 'dgcf':
        a_dgcf = ograph.dgcf_adjacency(adj)                          # gnn.py:408
        for lw in gnn['layers']:
            x = olayers.dgcf_conv(x, a_dgcf, lw['w'].astype(dtype))
            hs.append(x)
    elif kind in ('sage', 'gat'):
        row, col, _ = ograph.reordered_coo(adj)
        for lw in gnn['layers']:
            if kind == 'sage':
                x = olayers.sage_conv(x, row, col, lw['kernel'].astype(dtype), lw['bias'].astype(dtype),
                                      self_loops=self_loops)
            else:
                x, _ = olayers.gat_conv(x, row, col, lw['kernel'].astype(dtype), lw['attn_self'].astype(dtype),
                                        lw['attn_neigh'].astype(dtype), lw['bias'].astype(dtype),
                                        self_loops=self_loops)
            hs.append(x)
    else:
        raise ValueError("Unknown GNN kind {}".format(kind))
    final_node = gnn.get('final_node', 'concatenation')
    if force_mean and kind in ('lightgcn', 'dgcf'):
        final_node = 'mean'
    return olayers.reduce_layers(hs, final_node, gnn.get('reduction_w'))


def two_step(adjs, ts, n_users, n_items, dtype=np.float32):
    """TwoStepGNN.call (tsgnn.py:99-101): step one over the item-property graph, its first |I| rows appended to the
    trainable user rows (HalfInputSequentialGNN.call, gnn.py:141-142), step two over the user-item graph.
    adjs = (user-item, item-property) RAW adjacencies; ts = {'step_one': gnn dict, 'step_two': gnn dict}."""
    adj_ui, adj_kg = adjs
    x = propagate_from(adj_kg, ts['step_one']['embeddings'], ts['step_one'], dtype, force_mean=False)
    x0 = np.concatenate([ts['step_two']['embeddings'].astype(dtype), x[:n_items]], axis=0)
    return propagate_from(adj_ui, x0, ts['step_two'], dtype, force_mean=False)


def two_way(adjs, tw, n_users, n_items, dtype=np.float32):
    """TwoWayGNN.call (twgnn.py:98-105): users out of the user-property stack, items out of the item-property stack,
    stacked and propagated over the user-item graph (FullInputSequentialGNN).
    adjs = (user-item, item-property, user-property); tw = {'way_one', 'way_two', 'step_two'} gnn dicts (step_two has
    no 'embeddings')."""
    adj_ui, adj_ip, adj_up = adjs
    users = propagate_from(adj_up, tw['way_one']['embeddings'], tw['way_one'], dtype, force_mean=False)
    items = propagate_from(adj_ip, tw['way_two']['embeddings'], tw['way_two'], dtype, force_mean=False)
    x0 = np.concatenate([users[:n_users], items[:n_items]], axis=0)
    return propagate_from(adj_ui, x0, tw['step_two'], dtype, force_mean=False)


def _cast_net(net, dtype):
    return [(w.astype(dtype), b.astype(dtype)) for w, b in net]


def basic_rs(u, i, head, activation='relu'):
    """BasicRS.call (basic.py:31-37): unet(u), inet(i), concat, clf -> [B,1]."""
    dtype = u.dtype
    u = olayers.dense_network(u, _cast_net(head['unet'], dtype), activation)
    i = olayers.dense_network(i, _cast_net(head['inet'], dtype), activation)
    return olayers.dense_classifier(np.concatenate([u, i], axis=1), _cast_net(head['clf'], dtype), activation)


def attention_fuse(a, b, fw):
    """FusionLayer('attention') (fusion.py:54-68): project the narrower block, stack, per-feature two-way softmax
    of tanh(x . att_weight), weighted sum.  fw = {'att_weight': [D, D], optional 'proj_weight': [d_small, D]}."""
    if 'proj_weight' in fw:
        if a.shape[1] < b.shape[1]:
            a = a @ fw['proj_weight'].astype(a.dtype)
        else:
            b = b @ fw['proj_weight'].astype(a.dtype)
    x = np.stack([a, b], axis=1)                                   # [B, 2, D]
    att = np.tanh(x @ fw['att_weight'].astype(a.dtype))
    att = np.exp(att - att.max(axis=1, keepdims=True))
    att = att / att.sum(axis=1, keepdims=True)
    return (att * x).sum(axis=1)


def hybrid_cbrs(ug, ig, ub, ib, head, activation='relu', feature_based=True):
    """HybridCBRS.call (hybrid.py:72-89).  Fusion layers present in `head` ('fuse1a', 'fuse1b', 'fuse2') are attention
    fusions, absent ones concatenate; head['residual'] (a Dense stack whose last layer is linear) selects
    clf(activation(residual(x) + x1 + x2)) (hybrid.py:86-89)."""
    dtype = ug.dtype
    net = lambda name, x: olayers.dense_network(x, _cast_net(head[name], dtype), activation)
    fuse = lambda name, a, b: attention_fuse(a, b, head[name]) if name in head else np.concatenate([a, b], axis=1)
    ug, ig, ub, ib = net('dense1a', ug), net('dense1b', ig), net('dense2a', ub), net('dense2b', ib)
    if feature_based:
        x1, x2 = net('dense3a', fuse('fuse1a', ug, ig)), net('dense3b', fuse('fuse1b', ub, ib))
    else:
        x1, x2 = net('dense3a', fuse('fuse1a', ug, ub)), net('dense3b', fuse('fuse1b', ig, ib))
    x = fuse('fuse2', x1, x2)
    if 'residual' in head:
        res = _cast_net(head['residual'], dtype)
        r = olayers.dense_network(x, res[:-1], activation)
        r = olayers.dense(r, res[-1][0], res[-1][1], None)
        x = olayers._act(r + x1 + x2, activation)
    return olayers.dense_classifier(x, _cast_net(head['clf'], dtype), activation)


def basic_gnn_scores(adj, gnn, head, u_ids, i_ids, dtype=np.float32, batch=None):
    """BasicGNN.call (basic.py:61-75): E = gnn(None); E[u], E[i]; BasicRS.  Returns [P,1].

    `batch` re-runs the propagation per batch exactly like the reference ('faithful' mode); the
    result is identical because gnn(None) does not depend on the batch.
    """
    if batch is None:
        e = propagate(adj, gnn, dtype)
        return basic_rs(e[u_ids], e[i_ids], head)
    outs = []
    for lo in range(0, len(u_ids), batch):
        e = propagate(adj, gnn, dtype)
        outs.append(basic_rs(e[u_ids[lo:lo + batch]], e[i_ids[lo:lo + batch]], head))
    return np.concatenate(outs, axis=0)


def hybrid_gnn_scores(adj, gnn, head, u_ids, i_ids, bert, dtype=np.float32, feature_based=True):
    """HybridBertGNN.call (hybrid.py:126-140) with host-side BERT row gather (datasets.py:65-66)."""
    e = propagate(adj, gnn, dtype)
    bert = bert.astype(dtype)
    return hybrid_cbrs(e[u_ids], e[i_ids], bert[u_ids], bert[i_ids], head, feature_based=feature_based)


def top_k(u_idx, i_idx, scores, users, items, k):
    """metrics.py:11-34 with a deterministic tie rule.

    Rows (raw user, raw item, score) sorted by user ascending then score descending, first k per
    user among that user's test pairs.  The reference sorts with pandas' default (unstable)
    quicksort, so equal scores have no defined order there; here ties break on raw item id
    ascending.  Scores are compared in float64 like the reference's DataFrame column.
    """
    n_users = len(users)
    raw_u = users[np.asarray(u_idx, dtype=np.int64)]
    raw_i = items[np.asarray(i_idx, dtype=np.int64) - n_users]
    s = np.asarray(scores, dtype=np.float64).reshape(-1)
    order = np.lexsort((raw_i, -s, raw_u))
    raw_u, raw_i, s = raw_u[order], raw_i[order], s[order]
    start = np.searchsorted(raw_u, raw_u, side='left')
    keep = (np.arange(len(raw_u)) - start) < k
    return raw_u[keep], raw_i[keep], s[keep]


def count_params(*trees):
    """Trainable-parameter count (keras.py:10-22 semantics) of nested weight containers."""
    total = 0
    for t in trees:
        if isinstance(t, np.ndarray):
            total += t.size
        elif isinstance(t, dict):
            total += count_params(*[v for v in t.values() if not isinstance(v, str)])
        elif isinstance(t, (list, tuple)):
            total += count_params(*t)
    return total
